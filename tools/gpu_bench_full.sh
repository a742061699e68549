# Default bench (all legs); returns the updated MIOpen find-db / kernel cache so new shapes can be committed.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/r02_bench.log 2> gpurun_out/r02_bench.err; echo "rc=$?" >> gpurun_out/r02_bench.log
rm -rf gpurun_out/miopen_db; mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
tail -c 3000 gpurun_out/r02_bench.log
