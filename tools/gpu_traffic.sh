# HBM traffic per launch of the hand-written kernels in the two bench legs (two --pmc passes each; no tracing).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic
LVG_BENCH_NO_ROOFLINE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic/lres_f -o p -- python bench.py --no-cpu-baseline --no-extra-legs --graph off --steps 2 --warmup 1 > gpurun_out/traffic/lres_f.log 2>&1
LVG_BENCH_NO_ROOFLINE=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic/lres_w -o p -- python bench.py --no-cpu-baseline --no-extra-legs --graph off --steps 2 --warmup 1 > gpurun_out/traffic/lres_w.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic/sres_f -o p -- python tools/sres_step.py 1 > gpurun_out/traffic/sres_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic/sres_w -o p -- python tools/sres_step.py 1 > gpurun_out/traffic/sres_w.log 2>&1
python tools/pmc_traffic.py $(find gpurun_out/traffic/lres_f -name "*counter_collection.csv") $(find gpurun_out/traffic/lres_w -name "*counter_collection.csv") gpurun_out/${ROUND:-r05}_traffic_lres.json | head -40
python tools/pmc_traffic.py $(find gpurun_out/traffic/sres_f -name "*counter_collection.csv") $(find gpurun_out/traffic/sres_w -name "*counter_collection.csv") gpurun_out/${ROUND:-r05}_traffic_sres.json | head -40
rm -rf gpurun_out/traffic/*_f gpurun_out/traffic/*_w
