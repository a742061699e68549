#!/bin/bash
# r06: ablation builds of the strip kernel (results WRONG, timings only): bash tools/gpu_r06_sabl.sh <tag> variant ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}; shift
{
LVG_FLRELU_DEBUG=1 timeout 60 tools/bin/flrelu_check one L8 1 1 5 2 2>&1 | grep "^filtered_lrelu" | head -2
for v in default "$@"; do
  lib=$PWD/long-video-gan_amd/lib/variant_$v.so; [ $v = default ] && lib=$PWD/long-video-gan_amd/lib/liblvg_hip.so
  for L in ${LAYERS:-L8}; do for m in ${MODES:-0 1 2}; do
    LVG_LIB=$lib timeout 60 tools/bin/flrelu_check one $L 1 $m 5 10 2>&1 | grep -E "impl=" | sed "s/$/ $v/" | cut -c1-130
  done; done
done
} 2>&1 | tee gpurun_out/r06_sabl_$tag.log
