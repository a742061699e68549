# rocprofv3 evidence for the default bench line: kernel-trace stats of the same command + HBM traffic (two PMC passes, no tracing)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -o bench -- python bench.py > gpurun_out/r02_bench_under_rocprof.log 2>&1
cp $(find gpurun_out/prof_default -name "*kernel_stats.csv" | head -1) gpurun_out/r02_bench_default_kernel_stats.csv
rm -rf gpurun_out/prof_default
head -12 gpurun_out/r02_bench_default_kernel_stats.csv | cut -c1-150
mkdir -p gpurun_out/traffic
LVG_BENCH_NO_ROOFLINE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic/lres_f -o p -- python bench.py --no-cpu-baseline --no-extra-legs --graph off --steps 2 --warmup 1 > gpurun_out/traffic/lres_f.log 2>&1
LVG_BENCH_NO_ROOFLINE=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic/lres_w -o p -- python bench.py --no-cpu-baseline --no-extra-legs --graph off --steps 2 --warmup 1 > gpurun_out/traffic/lres_w.log 2>&1
python tools/pmc_traffic.py $(find gpurun_out/traffic/lres_f -name "*counter_collection.csv") $(find gpurun_out/traffic/lres_w -name "*counter_collection.csv") gpurun_out/traffic_lres.json | head -30
rm -rf gpurun_out/traffic/*_f gpurun_out/traffic/*_w
