# Full GPU parity suite + smoke + default bench in one gpurun call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 170 python -m pytest tests/test_ada_augment.py tests/test_train_sres.py -m gpu -q > gpurun_out/new_tests.log 2>&1; echo "new tests rc=$?" >> gpurun_out/new_tests.log
timeout 400 python -m pytest tests -m gpu -q --deselect tests/test_ada_augment.py --deselect tests/test_train_sres.py > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/gpu_tests.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 150 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
tail -4 gpurun_out/new_tests.log; tail -3 gpurun_out/gpu_tests.log; tail -2 gpurun_out/smoke.log; grep -o '"value": [0-9.]*' gpurun_out/bench.log
