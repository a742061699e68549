cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/prof_leg
LVG_BENCH_NO_ROOFLINE=1 timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_leg -o leg -- python bench.py --forward-only --no-extra-legs --no-cpu-baseline --steps 20 --warmup 2 > gpurun_out/r05_window_forward_only.run.log 2>&1
find gpurun_out/prof_leg -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_window_forward_only.csv \;
rm -rf gpurun_out/prof_leg
tail -2 gpurun_out/r05_window_forward_only.run.log | cut -c1-300
