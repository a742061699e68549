#!/bin/bash
# r05: row-band filtered_lrelu kernel (impl 4): parity against the oracle, then timings next to the wave kernel (impl 3).
#   bash tools/gpu_r05_band.sh <tag> [check|time|both] [variant ...]      (LVG_FLRELU_BAND_WPS=2|3 selects the build)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}; what=${2:-both}; shift; shift
{
if [ "$what" != time ]; then
  echo "== check (impl 4 = band; falls back to the wave kernel for what it does not take)"
  LVG_FLRELU_DEBUG=1 FLRELU_IMPLS=4 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -v "^filtered_lrelu" | tail -50
  echo "== check with clamp 256 (the no-clamp paths)"
  FLRELU_CLAMP=256 FLRELU_IMPLS=4 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -E "FAIL|failure"
  echo "== check, several planes per workgroup"
  LVG_FLRELU_BAND_MAXGRID=2 FLRELU_IMPLS=4 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -E "many|L8_|L10|L13|FAIL|failure"
fi
if [ "$what" != check ]; then
  for wps in 2 3; do
    echo "== time, band kernel compiled for $wps waves per SIMD"
    LVG_FLRELU_BAND_WPS=$wps FLRELU_IMPLS=4 timeout 200 tools/bin/flrelu_check time 2>&1
  done
  for v in "$@"; do
    echo "== time variant $v"
    for wps in 2 3; do for L in L8; do for m in 0 1 2; do LVG_FLRELU_BAND_WPS=$wps LVG_LIB=$PWD/long-video-gan_amd/lib/variant_$v.so timeout 60 tools/bin/flrelu_check one $L 1 $m 4 10 2>&1 | grep -E "impl=" | sed "s/$/ wps $wps/" | cut -c1-120; done; done; done
  done
fi
} 2>&1 | tee gpurun_out/r05_band_$tag.log
