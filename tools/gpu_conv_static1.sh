# Static-tap loop for the two-band 256 x 64 tiles (one workgroup per CU): parity tests, per-layer timings and the step, on / off in the same call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv3d_frames.py tests/test_lres_models.py -m gpu -q --no-header -rf -x > gpurun_out/r04_conv_static1_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04_conv_static1_tests.log
tail -4 gpurun_out/r04_conv_static1_tests.log
{
echo "== static-tap loop on the two-band 64-channel tiles"; timeout 200 python tools/conv_bench.py 10 2>&1 | grep "hand\|total" | cut -c1-64
echo "== generic loop (LVG_CONV_STATIC1=0)"; LVG_CONV_STATIC1=0 timeout 200 python tools/conv_bench.py 10 2>&1 | grep "hand\|total" | cut -c1-64
} > gpurun_out/r04_conv_static1_ab.log 2>&1
cat gpurun_out/r04_conv_static1_ab.log
for v in 1 0 1 0; do
  LVG_CONV_STATIC1=$v timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r04_bench_static1_$v.log 2>&1
  echo "static1=$v: $(grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/r04_bench_static1_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r04_conv_static1_ab.log
done
