# Ablation builds (tools/build_conv_variants.sh 6 16 64 128 256) on the 64-output-channel layers, persistent form: what the tile time is made of.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$PWD/long-video-gan_amd/lib
{
echo "== persistent, shipped"; timeout 120 python tools/conv_bench.py 10 "->64@" 2>&1 | grep "hand" | cut -c1-64
for v in 6 16 64 128 256; do
  echo "== persistent, abl $v"; LVG_HIP_LIB=$L/variant_conv_abl$v.so timeout 120 python tools/conv_bench.py 10 "->64@" 2>&1 | grep "hand" | cut -c1-64
done
} > gpurun_out/r04_conv_persist_abl.log 2>&1
cat gpurun_out/r04_conv_persist_abl.log
