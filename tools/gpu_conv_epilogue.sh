# Fast epilogue arithmetic of the convolution kernel: parity tests, then per-layer timings against the library built from the previous commit.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv3d_frames.py tests/test_lres_models.py -m gpu -q --no-header -rf -x > gpurun_out/r04_conv_epilogue_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04_conv_epilogue_tests.log
tail -6 gpurun_out/r04_conv_epilogue_tests.log
{
echo "== fast epilogue forms";  timeout 200 python tools/conv_bench.py 10 2>&1 | grep "hand\|total" | cut -c1-64
echo "== before";               LVG_HIP_LIB=$PWD/long-video-gan_amd/lib/variant_conv_before.so timeout 200 python tools/conv_bench.py 10 2>&1 | grep "hand\|total" | cut -c1-64
} > gpurun_out/r04_conv_epilogue_ab.log 2>&1
cat gpurun_out/r04_conv_epilogue_ab.log
for v in new before; do
  if [ $v = before ]; then export LVG_HIP_LIB=$PWD/long-video-gan_amd/lib/variant_conv_before.so; else unset LVG_HIP_LIB; fi
  timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r04_bench_epilogue_$v.log 2>&1
  echo "epilogue=$v: $(grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/r04_bench_epilogue_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r04_conv_epilogue_ab.log
done
