#!/bin/bash
# r03: filtered_lrelu MFMA kernel, old library (variant_old.so) against the working tree's (and its -DLVG_TIMING build): parity, then timings.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in old new tim old new; do
  lib=$PWD/long-video-gan_amd/lib/variant_$v.so; [ $v = new ] && lib=$PWD/long-video-gan_amd/lib/liblvg_hip.so
  echo "== $v"
  [ $v != tim ] && LVG_LIB=$lib timeout 120 tools/bin/flrelu_check check 2>&1 | tail -1
  for L in L8 L10 L13; do for m in 0 1 2; do LVG_LIB=$lib timeout 60 tools/bin/flrelu_check one $L 1 $m 2 10 2>&1 | grep -E "impl=MFMA|timing:" | cut -c1-330; done; done
done
} | tee gpurun_out/r03_flrelu_ab_$1.log
