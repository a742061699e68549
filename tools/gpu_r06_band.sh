#!/bin/bash
# r06: software-pipelined row-band kernel: parity, then timings (warm and cold) next to the builds named as arguments (lib/variant_<name>.so).
#   bash tools/gpu_r06_band.sh <tag> [variant ...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}; shift
{
echo "== check (impl 4 = band)"
FLRELU_IMPLS=4 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -E "FAIL|failure"
echo "== check with clamp 256"
FLRELU_CLAMP=256 FLRELU_IMPLS=4 timeout 300 tools/bin/flrelu_check check 2>&1 | grep -E "FAIL|failure"
echo "== time default"
FLRELU_IMPLS=4 timeout 200 tools/bin/flrelu_check time 2>&1 | grep BAND
FLRELU_IMPLS=4 timeout 200 tools/bin/flrelu_check timecold 2>&1 | grep BAND
for v in "$@"; do
  echo "== time variant $v"
  LVG_LIB=$PWD/long-video-gan_amd/lib/variant_$v.so FLRELU_IMPLS=4 timeout 200 tools/bin/flrelu_check time 2>&1 | grep BAND
  LVG_LIB=$PWD/long-video-gan_amd/lib/variant_$v.so FLRELU_IMPLS=4 timeout 200 tools/bin/flrelu_check timecold 2>&1 | grep BAND
done
} 2>&1 | tee gpurun_out/r06_band_$tag.log
