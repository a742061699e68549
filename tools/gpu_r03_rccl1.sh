#!/bin/bash
# r03: the N > 1 code path on ONE rank over RCCL (LVG_FORCE_DIST=1): default line and the train_lres workload; launch self-test on RCCL
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
timeout 200 python bench.py --gpus 1 --selftest-launch > gpurun_out/r03_launch_selftest_1gpu.log 2>&1; cat gpurun_out/r03_launch_selftest_1gpu.log | tail -1
LVG_FORCE_DIST=1 timeout 400 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r03_bench_rccl_1rank.log 2> gpurun_out/r03_bench_rccl_1rank.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_bench_rccl_1rank.log').read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','ms_per_step','n_gpus','grad_sync')})"
LVG_FORCE_DIST=1 timeout 600 python bench.py --workload train_lres --steps 2 --warmup 1 > gpurun_out/r03_train_lres_rccl_1rank.log 2> gpurun_out/r03_train_lres_rccl_1rank.err
tail -1 gpurun_out/r03_train_lres_rccl_1rank.log | cut -c1-400; tail -2 gpurun_out/r03_train_lres_rccl_1rank.err
