cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_modconv_epilogue.py -m gpu -x -q > gpurun_out/epi_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/epi_tests.log
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/epi_tests.log; grep -o '"value": [0-9.]*' gpurun_out/bench.log; grep -o '"ops".*' gpurun_out/bench.log | cut -c1-700
