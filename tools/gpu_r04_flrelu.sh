#!/bin/bash
# r04: wave-per-tile filtered_lrelu kernel: unaligned-access probe, parity against the oracle (impl 3 = wave, 2 = round-2 MFMA),
# timings of the default library and of the variant builds named on the command line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}; shift
{
timeout 60 tools/bin/probe_unaligned
echo "== check"
FLRELU_IMPLS=3 timeout 300 tools/bin/flrelu_check check 2>&1 | tail -40
echo "== time default"
FLRELU_IMPLS=32 timeout 200 tools/bin/flrelu_check time 2>&1
for v in "$@"; do
  echo "== time variant $v"
  for L in L8 L10 L13; do for m in 0 1 2; do LVG_LIB=$PWD/long-video-gan_amd/lib/variant_$v.so timeout 60 tools/bin/flrelu_check one $L 1 $m 3 10 2>&1 | grep -E "impl=|timing" | cut -c1-330; done; done
done
} 2>&1 | tee gpurun_out/r04_flrelu_$tag.log
