#!/bin/bash
# r03: weight / style side of the sres generator layers on a second stream: parity, then A/B of the sres leg (hipGraph) in one call
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 900 python -m pytest tests/test_sres_models.py tests/test_train_sres.py -m gpu -q 2>&1 | tail -5 ) > gpurun_out/r03_side_tests.log
tail -3 gpurun_out/r03_side_tests.log
{
for v in 1 0 1; do
  LVG_SRES_SIDE_STREAM_TERMS=$v LVG_BENCH_NO_TRAIN_LEGS=1 timeout 400 python bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r03_bench_side_$v.log 2>gpurun_out/r03_bench_side.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/r03_bench_side_$v.log').read().strip().splitlines()[-1])
s = d.get('sres', {})
print('SIDE=$v sres', s.get('value'), s.get('ms_per_step'), s.get('launch_mode'), s.get('error'), '| lres', d['ms_per_step'], '| sweep', d.get('batch_sweep'))
PY
done
} 2>&1 | tee gpurun_out/r03_side_ab.log
tail -2 gpurun_out/r03_bench_side.err
