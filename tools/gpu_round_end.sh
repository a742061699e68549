# One gpurun call: default bench, clean timed-window trace, sres probe + profile, full GPU test suite.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
LVG_BENCH_NO_ROOFLINE=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_window -o win -- python bench.py --no-cpu-baseline > gpurun_out/window.log 2>&1
python tools/trace_window.py gpurun_out/prof_window/win_kernel_trace.csv $(python -c "
import json
for l in open('gpurun_out/window.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step']*d['steps'], d['steps'])") > gpurun_out/window_stats.csv 2>&1
rm -f gpurun_out/prof_window/win_kernel_trace.csv
timeout 200 python tools/sres_probe.py --segments 4 > gpurun_out/sres_probe.log 2>&1
timeout 200 python tools/sres_probe.py --segments 4 --forward-only >> gpurun_out/sres_probe.log 2>&1
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sres -o sres -- python tools/sres_probe.py --segments 4 --steps 3 >> gpurun_out/sres_probe.log 2>&1
rm -f gpurun_out/prof_sres/sres_kernel_trace.csv
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/gpu_tests.log
mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
tail -3 gpurun_out/gpu_tests.log; grep sres gpurun_out/sres_probe.log; tail -2 gpurun_out/bench.log | cut -c1-600
