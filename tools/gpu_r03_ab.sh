#!/bin/bash
# A/B of a variant library against the shipped one on the 2-D convolution bench (new, old, new):  bash tools/gpu_r03_ab.sh <variant name> [log tag]
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
TAG=${2:-$1}
( timeout 600 python -m pytest tests/test_conv2d_frames.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r03_conv2d_tests_$TAG.log
tail -3 gpurun_out/r03_conv2d_tests_$TAG.log
V=$PWD/long-video-gan_amd/lib/variant_$1.so
{
for v in shipped variant shipped; do
  if [ $v = variant ]; then export LVG_HIP_LIB=$V; else unset LVG_HIP_LIB; fi
  echo "== $v ($1)"; LVG_BENCH_LIB=0 timeout 300 python tools/conv2d_bench.py 8 2>&1 | grep "fwd\|total" | cut -c1-140
done
} 2>&1 | tee gpurun_out/r03_conv2d_ab_$TAG.log
