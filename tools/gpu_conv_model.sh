# lres model parity with the hand-written convolution on + A/B of the default bench line (no extra legs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv3d_frames.py tests/test_lres_models.py tests/test_tapconv_epilogue.py tests/test_trainer_gpu.py -m gpu -q --no-header -rf > gpurun_out/r02_conv_model_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02_conv_model_tests.log
tail -6 gpurun_out/r02_conv_model_tests.log
for hc in 1 0 1; do
  LVG_HAND_CONV=$hc timeout 600 python bench.py --no-extra-legs > gpurun_out/r02_bench_handconv$hc.log 2> gpurun_out/r02_bench_handconv$hc.err; echo "rc=$?" >> gpurun_out/r02_bench_handconv$hc.log
  echo "HAND_CONV=$hc: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r02_bench_handconv$hc.log | tr '\n' ' ')"
done
