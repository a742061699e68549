# Kernel trace of the timed region of the default bench (no roofline replays, no CPU leg) -> per-kernel table.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LVG_BENCH_NO_ROOFLINE=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_window -o win -- python bench.py --no-cpu-baseline --no-extra-legs ${BENCH_ARGS:-} > gpurun_out/window.log 2>&1
python tools/trace_window.py gpurun_out/prof_window/win_kernel_trace.csv $(python -c "
import json
for l in open('gpurun_out/window.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step']*d['steps'], d['steps'])") > gpurun_out/window_stats.csv 2>&1
rm -f gpurun_out/prof_window/win_kernel_trace.csv
head -${ROWS:-48} gpurun_out/window_stats.csv | cut -c1-${COLS:-175}
