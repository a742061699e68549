# Round-end evidence in one gpurun call: rocprofv3 stats of the default bench, timed-window table,
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately, eager launches).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
rm -f gpurun_out/prof_bench/bench_kernel_trace.csv
export LVG_BENCH_NO_ROOFLINE=1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_window -o win -- python bench.py --no-cpu-baseline > gpurun_out/window.log 2>&1
python tools/trace_window.py gpurun_out/prof_window/win_kernel_trace.csv $(python -c "
import json
for l in open('gpurun_out/window.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step']*d['steps'], d['steps'])") > gpurun_out/window_stats.csv 2>&1
rm -f gpurun_out/prof_window/win_kernel_trace.csv
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --graph off --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --graph off --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -delete
du -sh gpurun_out/*; grep -o '"value": [0-9.]*' gpurun_out/bench.log gpurun_out/window.log
