#!/bin/bash
# r03 evidence in one gpurun call (re-run at the end of the round on the final tree): full GPU suite + smoke, default bench under rocprofv3 --stats, timed-window tables of the lres and sres
# steps, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, no tracing), filtered_lrelu per-layer timings + PMC of the shipped kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r03_gpu_tests.log; tail -3 gpurun_out/r03_gpu_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r03_smoke.log; cat gpurun_out/r03_smoke.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -o bench -- python bench.py > gpurun_out/r03_bench_default.log 2>gpurun_out/r03_bench_default.err
cp $(find gpurun_out/prof_default -name "*kernel_stats.csv" | head -1) gpurun_out/r03_bench_default_kernel_stats.csv; rm -rf gpurun_out/prof_default
grep -o '"value": [0-9.]*' gpurun_out/r03_bench_default.log | head -3
# timed windows
LVG_BENCH_NO_ROOFLINE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_window -o win -- python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r03_window.log 2>&1
python tools/trace_window.py $(find gpurun_out/prof_window -name "*kernel_trace.csv" | head -1) $(python -c "
import json
for l in open('gpurun_out/r03_window.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step']*d['steps'], d['steps'])") > gpurun_out/r03_bench_window_stats.csv 2>&1
rm -rf gpurun_out/prof_window
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sres_w -o w -- python tools/sres_step.py 3 > gpurun_out/r03_sres_step.log 2>&1
python tools/trace_window.py $(find gpurun_out/prof_sres_w -name "*kernel_trace.csv" | head -1) $(python -c "
import json
for l in open('gpurun_out/r03_sres_step.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['window_ms'], d['steps'])") > gpurun_out/r03_sres_step_window_stats.csv 2>&1
rm -rf gpurun_out/prof_sres_w
# HBM traffic
mkdir -p gpurun_out/traffic
LVG_BENCH_NO_ROOFLINE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic/lres_f -o p -- python bench.py --no-cpu-baseline --no-extra-legs --graph off --steps 2 --warmup 1 > gpurun_out/traffic/lres_f.log 2>&1
LVG_BENCH_NO_ROOFLINE=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic/lres_w -o p -- python bench.py --no-cpu-baseline --no-extra-legs --graph off --steps 2 --warmup 1 > gpurun_out/traffic/lres_w.log 2>&1
LVG_SRES_STEP_GRAPH=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic/sres_f -o p -- python tools/sres_step.py 1 > gpurun_out/traffic/sres_f.log 2>&1
LVG_SRES_STEP_GRAPH=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic/sres_w -o p -- python tools/sres_step.py 1 > gpurun_out/traffic/sres_w.log 2>&1
python tools/pmc_traffic.py $(find gpurun_out/traffic/lres_f -name "*counter_collection.csv") $(find gpurun_out/traffic/lres_w -name "*counter_collection.csv") gpurun_out/r03_traffic_lres.json | grep -v detail -A0 | grep '": [0-9]' | head -12
python tools/pmc_traffic.py $(find gpurun_out/traffic/sres_f -name "*counter_collection.csv") $(find gpurun_out/traffic/sres_w -name "*counter_collection.csv") gpurun_out/r03_traffic_sres.json | grep '": [0-9]' | head -12
rm -rf gpurun_out/traffic
python -c "import json; a=json.load(open('gpurun_out/r03_traffic_sres.json')); a.update(json.load(open('gpurun_out/r03_traffic_lres.json'))); json.dump(a, open('gpurun_out/r03_traffic_merged.json','w'), indent=1)"
# filtered_lrelu: per-layer timings and PMC of the shipped kernel
timeout 120 tools/bin/flrelu_check time > gpurun_out/r03_flrelu_check_time.log 2>&1; tail -14 gpurun_out/r03_flrelu_check_time.log | cut -c1-150
bash tools/gpu_pmc_flrelu.sh L8 1 1 2 r03_flrelu_pmc_L8_write > /dev/null 2>&1; cp gpurun_out/r03_flrelu_pmc_L8_write/summary.csv gpurun_out/r03_filtered_lrelu_mfma_pmc.csv; rm -rf gpurun_out/r03_flrelu_pmc_L8_write
head -30 gpurun_out/r03_filtered_lrelu_mfma_pmc.csv
du -sh gpurun_out | tail -1
