#!/bin/bash
# r03 first GPU call: the new 2-D convolution kernels (parity + timing), the full GPU suite (recording the measured bf16 errors),
# the default bench line.
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 600 python -m pytest tests/test_conv2d_frames.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r03_conv2d_tests.log
tail -5 gpurun_out/r03_conv2d_tests.log
( timeout 600 python tools/conv2d_bench.py 5 2>&1 | tail -20 ) > gpurun_out/r03_conv2d_bench.log
cat gpurun_out/r03_conv2d_bench.log
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_conv2d_frames.py 2>&1 | tail -15 ) > gpurun_out/r03_gpu_tests.log
tail -6 gpurun_out/r03_gpu_tests.log
( timeout 600 python bench.py 2> gpurun_out/r03_bench.err | tail -1 ) > gpurun_out/r03_bench.log
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r03_bench.log').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'ms_per_step')}, 'sres', {k: d.get('sres', {}).get(k) for k in ('value', 'ms_per_step', 'error')})
    print('forward_only', d.get('forward_only', {}).get('value'), 'roofline', d['roofline']['achieved'], d['roofline']['frac'])
except Exception as e:
    print('bench parse failed', e)
PY
tail -3 gpurun_out/r03_bench.err
cat gpurun_out/parity_measured.json 2>/dev/null | head -40
