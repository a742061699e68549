cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "LVG_CONV_BM=128 LVG_CONV_NB=2" "LVG_CONV_BM=256 LVG_CONV_NB=3"; do
echo "== $cfg abl=0: $(env $cfg timeout 100 python tools/conv_bench.py 5 '80x512->512' 2>&1 | grep hand | cut -c30-60)"
for abl in $ABLS; do
  echo "== $cfg abl=$abl: $(env $cfg LVG_HIP_LIB=$PWD/long-video-gan_amd/lib/variant_conv_abl$abl.so timeout 100 python tools/conv_bench.py 5 '80x512->512' 2>&1 | grep hand | cut -c30-60)"
done; done > gpurun_out/r02_conv_abl.log 2>&1
cat gpurun_out/r02_conv_abl.log
