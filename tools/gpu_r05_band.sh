#!/bin/bash
# r05: row-band filtered_lrelu kernel (impl 4): parity against the oracle, then timings next to the wave kernel (impl 3).
#   bash tools/gpu_r05_band.sh <tag> [check|time|both] [variant ...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}; what=${2:-both}; shift; shift
{
if [ "$what" != time ]; then
  echo "== check (impl 4 = band; falls back to the wave kernel for what it does not take)"
  LVG_FLRELU_DEBUG=1 FLRELU_IMPLS=4 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -v "^filtered_lrelu_wave" | sort -u | tail -80
  echo "== check, several planes per workgroup"
  LVG_FLRELU_BAND_MAXGRID=2 FLRELU_IMPLS=4 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -E "many|L8_|L10|L13|FAIL|failure"
fi
if [ "$what" != check ]; then
  echo "== time"
  FLRELU_IMPLS=43 timeout 200 tools/bin/flrelu_check time 2>&1
  for v in "$@"; do
    echo "== time variant $v"
    for L in L8 L10 L13; do for m in 0 1 2; do LVG_LIB=$PWD/long-video-gan_amd/lib/variant_$v.so timeout 60 tools/bin/flrelu_check one $L 1 $m 4 10 2>&1 | grep -E "impl=" | cut -c1-200; done; done
  done
fi
} 2>&1 | tee gpurun_out/r05_band_$tag.log
