cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_bias_act_gpu.py tests/test_lres_models.py tests/test_trainer_gpu.py -m gpu -q --no-header -rf -x > gpurun_out/r02_biasgrad_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02_biasgrad_tests.log
tail -12 gpurun_out/r02_biasgrad_tests.log
for v in 1 0 1; do
  LVG_BIAS_GRAD_FUSED=$v timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r02_bench_biasgrad_$v.log 2>&1
  echo "BIAS_GRAD_FUSED=$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02_bench_biasgrad_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r02_biasgrad_ab.log
done
