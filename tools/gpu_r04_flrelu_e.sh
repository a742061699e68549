#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
LVG_FLRELU_DEBUG=1 timeout 60 tools/bin/flrelu_check one L8 1 1 3 1 2>&1 | grep -E "occupancy|impl=" | head -4
for v in default memonly w8; do
  echo "== time variant $v"
  lib=$PWD/long-video-gan_amd/lib/variant_$v.so; [ $v = default ] && lib=$PWD/long-video-gan_amd/lib/liblvg_hip.so
  for L in L8 L10; do for m in 0 1 2; do LVG_LIB=$lib timeout 60 tools/bin/flrelu_check one $L 1 $m 3 10 2>&1 | grep -E "impl=|timing" | cut -c1-330; done; done
done
} 2>&1 | tee gpurun_out/r04_flrelu_e.log
