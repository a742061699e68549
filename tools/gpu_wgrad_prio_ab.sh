cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$PWD/long-video-gan_amd/lib/variant_wgrad_prio1.so
{
echo "== shipped"; timeout 200 python tools/wgrad_bench.py 5 2>&1 | grep "hand\|total"
echo "== variant"; LVG_HIP_LIB=$V timeout 200 python tools/wgrad_bench.py 5 2>&1 | grep "hand\|total"
} 2>&1 | cut -c1-90 | tee gpurun_out/r02_wgrad_prio.log
for v in variant shipped variant; do
  if [ $v = variant ]; then export LVG_HIP_LIB=$V; else unset LVG_HIP_LIB; fi
  timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r02_bench_variant_$v.log 2>&1
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02_bench_variant_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r02_wgrad_prio.log
done
