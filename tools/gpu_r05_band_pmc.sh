#!/bin/bash
# r05: PMC instruction mix of the row-band filtered_lrelu kernel.  bash tools/gpu_r05_band_pmc.sh <tag> "<case dtype mode>" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}; shift
{
for spec in "$@"; do
  set -- $spec
  bash tools/gpu_pmc_flrelu_short.sh $1 $2 $3 4 pmc_band_${tag}_$1_$3
done
} 2>&1 | tee gpurun_out/r05_band_pmc_$tag.log
