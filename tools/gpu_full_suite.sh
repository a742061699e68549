# Whole GPU suite + smoke + default bench (as the driver runs them); returns the MIOpen db the runs tuned. ROUND=rNN names the outputs.
R=${ROUND:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --no-header -rf > gpurun_out/${R}_gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/${R}_gpu_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${R}_smoke.log
timeout 900 python bench.py > gpurun_out/${R}_bench_default.log 2> gpurun_out/${R}_bench_default.err; echo "bench rc=$?" >> gpurun_out/${R}_bench_default.log
rm -rf gpurun_out/miopen_db; mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
tail -5 gpurun_out/${R}_gpu_tests.log; tail -2 gpurun_out/${R}_smoke.log; tail -c 3000 gpurun_out/${R}_bench_default.log
