#!/bin/bash
# r05: where the row-band filtered_lrelu kernel spends its time: ablation builds (tools/build_flrelu_variants.sh with SRC=filtered_lrelu_band),
# waves-per-SIMD builds, PMC instruction mix.  bash tools/gpu_r05_band_abl.sh <tag> <variants...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}; shift
{
for wps in 2 3; do
  echo "== default library, compiled for $wps waves per SIMD"
  for L in L8 L13; do for m in 0 1 2; do LVG_FLRELU_BAND_WPS=$wps timeout 60 tools/bin/flrelu_check one $L 1 $m 4 10 2>&1 | grep -E "impl=" | cut -c1-200; done; done
done
for v in "$@"; do
  echo "== variant $v (2 waves per SIMD)"
  for L in L8; do for m in 0 1 2; do LVG_FLRELU_BAND_WPS=2 LVG_LIB=$PWD/long-video-gan_amd/lib/variant_$v.so timeout 60 tools/bin/flrelu_check one $L 1 $m 4 10 2>&1 | grep -E "impl=" | cut -c1-200; done; done
done
echo "== PMC, L8 forward without mask, 2 waves per SIMD"
export LVG_FLRELU_BAND_WPS=2
bash tools/gpu_pmc_flrelu_short.sh L8 1 0 4 pmc_band_$tag
} 2>&1 | tee gpurun_out/r05_band_abl_$tag.log
