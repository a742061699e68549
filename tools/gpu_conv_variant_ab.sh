# A/B of a variant build of liblvg_hip.so (long-video-gan_amd/lib/variant_$1.so) against the shipped library: per-layer convolution
# timings, then the default bench line (new, old, new).   usage: bash tools/gpu_conv_variant_ab.sh <variant name>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$PWD/long-video-gan_amd/lib/variant_$1.so
{
echo "== shipped"; timeout 200 python tools/conv_bench.py 5 2>&1 | grep "hand\|total"
echo "== variant $1"; LVG_HIP_LIB=$V timeout 200 python tools/conv_bench.py 5 2>&1 | grep "hand\|total"
} 2>&1 | cut -c1-70 | tee gpurun_out/r02_conv_variant_$1.log
for v in variant shipped variant; do
  if [ $v = variant ]; then export LVG_HIP_LIB=$V; else unset LVG_HIP_LIB; fi
  timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r02_bench_variant_$v.log 2>&1
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/r02_bench_variant_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r02_conv_variant_$1.log
done
