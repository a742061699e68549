#!/bin/bash
# r03: static-tap K loop of the time-major convolution: parity, per-layer timings and the default step, shipped vs generic-loop build vs
# the build with the static loop on the 64-channel tiles too (same call)
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 900 python -m pytest tests/test_conv3d_frames.py tests/test_lres_models.py -m gpu -q 2>&1 | tail -5 ) > gpurun_out/r03_static3d_tests.log
tail -3 gpurun_out/r03_static3d_tests.log
L=$PWD/long-video-gan_amd/lib
{
for v in shipped generic3d shipped static3d_bn64; do
  if [ $v = shipped ]; then unset LVG_HIP_LIB; else export LVG_HIP_LIB=$L/variant_$v.so; fi
  echo "== $v"; timeout 300 python tools/conv_bench.py 5 2>&1 | grep "hand\|total" | cut -c1-64
done
for v in shipped generic3d shipped static3d_bn64; do
  if [ $v = shipped ]; then unset LVG_HIP_LIB; else export LVG_HIP_LIB=$L/variant_$v.so; fi
  timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r03_bench_static3d_$v.log 2>/dev/null
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/r03_bench_static3d_$v.log | head -2 | tr '\n' ' ')"
done
} 2>&1 | tee gpurun_out/r03_static3d_ab.log
