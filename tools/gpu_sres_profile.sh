cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sres -o sres -- python tools/sres_probe.py --steps 3 > gpurun_out/r02_sres_prof.log 2>&1
f=$(find gpurun_out/prof_sres -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-200 > gpurun_out/r02_sres_kernel_stats_head.csv
cp "$f" gpurun_out/r02_sres_kernel_stats.csv
find gpurun_out/prof_sres -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/r02_sres_prof.log; head -32 gpurun_out/r02_sres_kernel_stats_head.csv | cut -c1-160
