# (build first: hipcc ... -DLVG_CONV_ABL=512 -c conv3d_igemm.hip + link with the other objects -> lib/variant_conv_abl512.so, see tools/build_conv_variants.sh)
# A/B of the convolution's LDS-staged 16-byte stores against the direct 8-byte stores (variant library built with -DLVG_CONV_ABL=512):
# parity tests on the new path, then per-layer timings and the default bench line with both libraries in the same call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$PWD/long-video-gan_amd/lib/variant_conv_abl512.so
timeout 600 python -m pytest tests/test_conv3d_frames.py tests/test_lres_models.py -m gpu -q --no-header -rf -x > gpurun_out/r02_conv_store_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02_conv_store_tests.log
tail -4 gpurun_out/r02_conv_store_tests.log
{
echo "== LDS-staged 16-byte stores"; timeout 200 python tools/conv_bench.py 5 2>&1 | grep "hand\|total"
echo "== direct 8-byte stores";      LVG_HIP_LIB=$V timeout 200 python tools/conv_bench.py 5 2>&1 | grep "hand\|total"
} 2>&1 | cut -c1-70 | tee gpurun_out/r02_conv_store_ab.log
for v in new old new; do
  if [ $v = old ]; then export LVG_HIP_LIB=$V; else unset LVG_HIP_LIB; fi
  timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r02_bench_store_$v.log 2>&1
  echo "stores=$v: $(grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/r02_bench_store_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r02_conv_store_ab.log
done
