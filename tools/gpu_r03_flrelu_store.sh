#!/bin/bash
# r03: stage D's output stores of the filtered_lrelu MFMA kernel: 0 = 8 bytes per lane, 1 = half-wave exchange -> 16 bytes per lane,
# 2 = through a wave-private LDS block (16 rows x 64 contiguous bytes per instruction). Parity of every variant against the oracle, then timings.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in 0 1 2 0 2; do
  echo "== LVG_FLRELU_STORE=$v"
  LVG_LIB=$PWD/long-video-gan_amd/lib/variant_st$v.so timeout 120 tools/bin/flrelu_check check 2>&1 | tail -2
  for L in L8 L10 L13; do for m in 0 1 2; do LVG_LIB=$PWD/long-video-gan_amd/lib/variant_st$v.so timeout 60 tools/bin/flrelu_check one $L 1 $m 2 10 2>&1 | grep "impl=MFMA" | cut -c1-100; done; done
done
} | tee gpurun_out/r03_flrelu_store_ab.log
