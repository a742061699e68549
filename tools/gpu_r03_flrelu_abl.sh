#!/bin/bash
# r03: ablation builds of the shipped filtered_lrelu MFMA kernel (LVG_ABL bits: 1 no prefetch, 2 no y stores, 4 no activation math, 8 no stage D,
# 32 all x loads from plane 0; results are wrong by construction, timings only) on layer L8 (f16; modes 0 = forward, 1 = forward + mask, 2 = backward)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for a in 0 2 4 8 6 14 46 47 1 0; do
  echo "== LVG_ABL=$a"
  for m in 0 1 2; do LVG_LIB=$PWD/long-video-gan_amd/lib/variant_abl$a.so timeout 60 tools/bin/flrelu_check one L8 1 $m 2 10 2>&1 | grep "impl=MFMA" | cut -c1-100; done
done
} | tee gpurun_out/r03_flrelu_ablation.log
