# Round-end evidence in one gpurun call: the GPU suite, smoke, the default bench line as the driver runs it, a rocprofv3 --kernel-trace --stats
# summary of the main leg, and the HBM traffic passes (tools/gpu_traffic.sh).
bash tools/gpu_full_suite.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r04_bench_stats_run.log 2>&1
find gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_bench_default_kernel_stats.csv \;
rm -rf gpurun_out/prof_bench
bash tools/gpu_traffic.sh > gpurun_out/r04_traffic_run.log 2>&1
tail -3 gpurun_out/r04_traffic_run.log | cut -c1-300
