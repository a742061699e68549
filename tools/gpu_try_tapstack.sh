cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 40 python -m pytest tests/test_tapconv_epilogue.py -m gpu -q -k "block" > gpurun_out/tap_block.log 2>&1; echo "rc=$?" >> gpurun_out/tap_block.log
LVG_TAP_STACK=1 timeout -s INT 80 python bench.py --no-cpu-baseline > gpurun_out/bench_tap.log 2>&1; echo "rc=$?" >> gpurun_out/bench_tap.log
mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
tail -3 gpurun_out/tap_block.log; tail -3 gpurun_out/bench_tap.log | cut -c1-400
