# Round 2, first call: the new BASELINE-size parity tests on the round-1 kernels + default bench (tap-stack on).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_filtered_lrelu_gpu.py tests/test_lres_models.py tests/test_trainer_gpu.py -m gpu -q -x --no-header -rf > gpurun_out/r02_newtests.log 2>&1; echo "rc=$?" >> gpurun_out/r02_newtests.log
timeout 200 python bench.py > gpurun_out/r02_bench_tapstack.log 2>&1; echo "rc=$?" >> gpurun_out/r02_bench_tapstack.log
timeout 120 python bench.py --forward-only --no-cpu-baseline > gpurun_out/r02_bench_fwd.log 2>&1; echo "rc=$?" >> gpurun_out/r02_bench_fwd.log
tail -15 gpurun_out/r02_newtests.log; tail -2 gpurun_out/r02_bench_tapstack.log | cut -c1-600; tail -2 gpurun_out/r02_bench_fwd.log | cut -c1-400
