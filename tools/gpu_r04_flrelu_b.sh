#!/bin/bash
# r04: timings of wave-kernel variant builds (impl 3) on L8 / L10 / L13, all three modes.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-b}; shift
{
for v in "$@"; do
  echo "== time variant $v"
  lib=$PWD/long-video-gan_amd/lib/variant_$v.so; [ $v = default ] && lib=$PWD/long-video-gan_amd/lib/liblvg_hip.so
  for L in ${LAYERS:-L8 L10 L13}; do for m in 0 1 2; do LVG_LIB=$lib timeout 60 tools/bin/flrelu_check one $L 1 $m 3 10 2>&1 | grep -E "impl=|timing" | cut -c1-330; done; done
done
} 2>&1 | tee gpurun_out/r04_flrelu_$tag.log
