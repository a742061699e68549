cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LVG_BENCH_VERBOSE=1 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/verbose.log
grep "^\[op\]" gpurun_out/verbose.log | sort | uniq -c | sort -k1,1nr | head -80
