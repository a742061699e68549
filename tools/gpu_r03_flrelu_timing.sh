#!/bin/bash
# r03: where the filtered_lrelu MFMA kernel's time goes: s_memtime cycles per region of the tile loop (-DLVG_TIMING build), and the ablations
# 32 (all x loads hit plane 0: cache hits), 16 (all y stores hit plane 0 tile 0), 48 (both) on layer L8 (f16).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in tim a32 a16 a48; do
  echo "== variant $v"
  for L in L8 L10; do for m in 0 1 2; do LVG_LIB=$PWD/long-video-gan_amd/lib/variant_$v.so timeout 60 tools/bin/flrelu_check one $L 1 $m 2 10 2>&1 | grep -E "impl=MFMA|timing:" | cut -c1-330; done; done
done
} | tee gpurun_out/r03_flrelu_timing.log
