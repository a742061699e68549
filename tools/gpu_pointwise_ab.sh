cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LVG_POINTWISE_HAND=1 timeout 600 python -m pytest tests/test_lres_models.py tests/test_conv3d_frames.py tests/test_tapconv_epilogue.py -m gpu -q --no-header -rf -x > gpurun_out/r02_pointwise_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02_pointwise_tests.log
tail -5 gpurun_out/r02_pointwise_tests.log
for v in 1 0 1 0; do
  LVG_POINTWISE_HAND=$v timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r02_bench_pointwise_$v.log 2>&1
  echo "POINTWISE_HAND=$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02_bench_pointwise_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r02_pointwise_ab.log
done
