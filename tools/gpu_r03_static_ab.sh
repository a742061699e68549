#!/bin/bash
# r03: static-tap K loop of the 2-D convolution: parity, then A/B against the generic-loop build in one call (new, old, new)
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 600 python -m pytest tests/test_conv2d_frames.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r03_conv2d_tests_static.log
tail -4 gpurun_out/r03_conv2d_tests_static.log
V=$PWD/long-video-gan_amd/lib/variant_conv2d_generic.so
{
for v in static generic static; do
  if [ $v = generic ]; then export LVG_HIP_LIB=$V; else unset LVG_HIP_LIB; fi
  echo "== $v"; LVG_BENCH_LIB=0 timeout 300 python tools/conv2d_bench.py 8 2>&1 | grep "fwd\|total" | cut -c1-140
done
unset LVG_HIP_LIB
echo "== static, 16 x 16 tiles on 8 waves"; LVG_CONV2D_BM=256 LVG_BENCH_LIB=0 timeout 300 python tools/conv2d_bench.py 8 2>&1 | grep "fwd\|total" | cut -c1-100
} 2>&1 | tee gpurun_out/r03_conv2d_static_ab.log
