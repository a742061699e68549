# Full GPU parity suite + smoke + default bench in one gpurun call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/gpu_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 200 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
tail -3 gpurun_out/gpu_tests.log; tail -2 gpurun_out/smoke.log; grep -o '"value": [0-9.]*' gpurun_out/bench.log; grep -o '"ops".*' gpurun_out/bench.log | cut -c1-800
