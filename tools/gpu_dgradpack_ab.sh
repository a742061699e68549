cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_weight_prep.py tests/test_conv3d_frames.py tests/test_lres_models.py tests/test_trainer_gpu.py -m gpu -q --no-header -rf -x > gpurun_out/r02_dgradpack_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02_dgradpack_tests.log
tail -8 gpurun_out/r02_dgradpack_tests.log
for v in new; do
  timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r02_bench_dgradpack_$v.log 2>&1
  echo "dgradpack=$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02_bench_dgradpack_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r02_dgradpack_ab.log
done
