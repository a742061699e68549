#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for wg in 1 2; do
for v in default mo_none memonly core; do
  echo "== WG/CU $wg variant $v"
  lib=$PWD/long-video-gan_amd/lib/variant_$v.so; [ $v = default ] && lib=$PWD/long-video-gan_amd/lib/liblvg_hip.so
  for L in L8; do for m in 0 1; do LVG_FLRELU_WG_PER_CU=$wg LVG_LIB=$lib timeout 60 tools/bin/flrelu_check one $L 1 $m 3 10 2>&1 | grep -E "impl=|timing" | cut -c1-330; done; done
done; done
} 2>&1 | tee gpurun_out/r04_flrelu_i.log
