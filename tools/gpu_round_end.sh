cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
date +%s > gpurun_out/t0
timeout 420 python -m pytest tests/test_sres_models.py -m gpu -x -q > gpurun_out/sres.log 2>&1; echo "sres rc=$?" >> gpurun_out/sres.log
date +%s > gpurun_out/t1
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
date +%s > gpurun_out/t2
export LVG_BENCH_NO_ROOFLINE=1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --graph off --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --graph off --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
date +%s > gpurun_out/t3
mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
# keep the merge-back small: drop raw kernel traces of the pmc passes, keep counter files
find gpurun_out -name "*kernel_trace.csv" -path "*pmc*" -delete
du -sh gpurun_out/* 
tail -5 gpurun_out/sres.log; tail -2 gpurun_out/bench.log
