# r04 evidence, part 1: the default bench line as the driver runs it, and a rocprofv3 --kernel-trace --stats summary of the main leg.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r04_bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/r04_bench_default.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --no-extra-legs --no-cpu-baseline > gpurun_out/r04_bench_stats_run.log 2>&1
rm -f gpurun_out/prof_bench/*/bench_kernel_trace.csv gpurun_out/prof_bench/bench_kernel_trace.csv
find gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_bench_default_kernel_stats.csv \;
find gpurun_out/prof_bench -name "*.csv" ! -name "*kernel_stats.csv" -delete; find gpurun_out/prof_bench -name "*.db" -delete
tail -2 gpurun_out/r04_bench_default.log | cut -c1-600
