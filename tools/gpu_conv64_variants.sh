# 64-channel convolution tiles: ring depth and tile height variants (same call).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in "NONE=0" "LVG_CONV_NB=3" "LVG_CONV_BM=128" "LVG_CONV_BM=128 LVG_CONV_NB=3" "LVG_CONV_BM=256 LVG_CONV_NB=3"; do
  echo "== $v"; env $v timeout 120 python tools/conv_bench.py 10 "->64@" 2>&1 | grep "hand" | cut -c1-64
done
} > gpurun_out/r04_conv64_variants.log 2>&1
cat gpurun_out/r04_conv64_variants.log
