#!/bin/bash
# r03: XCD-contiguous workgroup order of the weight-gradient kernels: parity, per-layer A/B (shipped = remapped, variant = as before)
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 900 python -m pytest tests/test_conv2d_frames.py tests/test_conv3d_frames.py -m gpu -q -k "wgrad or adjoint or weight" 2>&1 | tail -4 ) > gpurun_out/r03_wgrad_xcd_tests.log; tail -2 gpurun_out/r03_wgrad_xcd_tests.log
V=$PWD/long-video-gan_amd/lib/variant_wgrad_noxcd.so
{
for v in shipped variant shipped; do
  if [ $v = variant ]; then export LVG_HIP_LIB=$V; else unset LVG_HIP_LIB; fi
  echo "== $v (2-D)"; LVG_BENCH_LIB=0 timeout 300 python tools/conv2d_bench.py 8 2>&1 | grep "total\|92x148 \|164x276" | cut -c85-150
  echo "== $v (3-D)"; timeout 300 python tools/wgrad_bench.py 2>&1 | tail -18 | cut -c1-110
done
} 2>&1 | tee gpurun_out/r03_wgrad_xcd_ab.log
