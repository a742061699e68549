#!/bin/bash
# r03: the transposing kernels of the 2-D modulated convolution: shipped library against variant libraries (long-video-gan_amd/lib/variant_*.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== shipped"; timeout 120 python tools/layout_bench.py 2>&1 | tail -7
for v in long-video-gan_amd/lib/variant_*.so; do [ -f $v ] || continue; echo "== $v"; LVG_HIP_LIB=$PWD/$v timeout 120 python tools/layout_bench.py 2>&1 | tail -7; done
echo "== shipped"; timeout 120 python tools/layout_bench.py 2>&1 | tail -7
} | tee gpurun_out/r03_layout_bench_$1.log
