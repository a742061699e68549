cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "LVG_CONV_BN=128 LVG_CONV_NB=2" "LVG_CONV_BN=64 LVG_CONV_NB=2" "LVG_CONV_BN=64 LVG_CONV_NB=3"; do
  echo "== $cfg"; env $cfg timeout 200 python tools/conv_bench.py 5 2>&1 | grep "hand\|total" | cut -c1-70
done 2>&1 | tee gpurun_out/r02_conv_bn64_nb3.log
