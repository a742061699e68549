#!/bin/bash
# r03: conv2d parity after the mixed 128 + 64 channel tiling, kernel timing, default bench (sres leg from a hipGraph), window profile of the sres step
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 600 python -m pytest tests/test_conv2d_frames.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r03_conv2d_tests.log
tail -4 gpurun_out/r03_conv2d_tests.log
( LVG_BENCH_LIB=0 timeout 600 python tools/conv2d_bench.py 5 2>&1 | tail -20 ) > gpurun_out/r03_conv2d_bench_split.log
cat gpurun_out/r03_conv2d_bench_split.log
( timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/r03_bench2.err | tail -1 ) > gpurun_out/r03_bench2.log
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r03_bench2.log').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'ms_per_step')})
    s = d.get('sres', {})
    print('sres', {k: s.get(k) for k in ('value', 'ms_per_step', 'launch_mode', 'error')}, s.get('conv2d'), s.get('mfma'))
    print('sres flrelu', s.get('roofline', {}).get('achieved'), s.get('roofline', {}).get('families'))
except Exception as e:
    print('bench parse failed', e)
PY
tail -3 gpurun_out/r03_bench2.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sres_w -o w -- python tools/sres_step.py 3 > gpurun_out/r03_sres_step.log 2>&1
f=$(find gpurun_out/prof_sres_w -name "*kernel_trace.csv" | head -1)
python tools/trace_window.py "$f" $(python -c "
import json
for l in open('gpurun_out/r03_sres_step.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['window_ms'], d['steps'])") > gpurun_out/r03_sres_step_window_stats.csv 2>&1
rm -rf gpurun_out/prof_sres_w
grep "^{" gpurun_out/r03_sres_step.log; head -48 gpurun_out/r03_sres_step_window_stats.csv | cut -c1-150
