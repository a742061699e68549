# style_prep parity + A/B of the default bench line with the fused style side on / off (same call, same box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_style_prep.py tests/test_lres_models.py tests/test_trainer_gpu.py -m gpu -q --no-header -rf -x > gpurun_out/r02_style_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02_style_tests.log
tail -12 gpurun_out/r02_style_tests.log
for v in 1 0 1; do
  LVG_STYLE_PREP=$v timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r02_bench_style_$v.log 2>&1
  echo "STYLE_PREP=$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02_bench_style_$v.log | tr '\n' ' ')" | tee -a gpurun_out/r02_style_ab.log
done
