#!/bin/bash
# r06: strip kernel (impl 5): parity against the oracle, then timings next to the band / wave kernels.
#   bash tools/gpu_r06_strip.sh <tag> [check|time|both] [variant ...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}; what=${2:-both}; shift; shift
{
if [ "$what" != time ]; then
  echo "== check (impl 5 = strip)"
  LVG_FLRELU_DEBUG=1 FLRELU_IMPLS=5 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -v "^filtered_lrelu" | tail -50
  echo "== check with clamp 256 (the no-clamp paths)"
  FLRELU_CLAMP=256 FLRELU_IMPLS=5 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -E "FAIL|failure"
  echo "== check, several items per wave"
  LVG_FLRELU_STRIP_MAXGRID=1 FLRELU_IMPLS=5 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -E "FAIL|failure"
fi
if [ "$what" != check ]; then
  echo "== time"
  FLRELU_IMPLS=${IMPLS:-5} timeout 200 tools/bin/flrelu_check time 2>&1 | grep -v bf16
  FLRELU_IMPLS=${IMPLS:-5} timeout 200 tools/bin/flrelu_check timecold 2>&1
  for v in "$@"; do
    echo "== time variant $v"
    for L in L8; do for m in 0 1 2; do LVG_LIB=$PWD/long-video-gan_amd/lib/variant_$v.so timeout 60 tools/bin/flrelu_check one $L 1 $m 5 10 2>&1 | grep -E "impl=" | sed "s/$/ $v/" | cut -c1-130; done; done
  done
fi
} 2>&1 | tee gpurun_out/r06_strip_$tag.log
