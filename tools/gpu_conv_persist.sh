# Persistent 64-channel convolution tiles: parity tests, then per-layer timings with the persistent form on / off.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv3d_frames.py -m gpu -q --no-header -rf -x > gpurun_out/r04_conv_persist_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04_conv_persist_tests.log
tail -6 gpurun_out/r04_conv_persist_tests.log
{
echo "== persistent 64-channel tiles"; timeout 200 python tools/conv_bench.py 10 2>&1 | grep "hand\|total" | cut -c1-64
echo "== one workgroup per tile";      LVG_CONV_PERSIST=0 timeout 200 python tools/conv_bench.py 10 2>&1 | grep "hand\|total" | cut -c1-64
echo "== persistent, 128-pixel tiles"; LVG_CONV_BM=128 timeout 200 python tools/conv_bench.py 10 "->64@" 2>&1 | grep "hand\|total" | cut -c1-64
} > gpurun_out/r04_conv_persist_ab.log 2>&1
cat gpurun_out/r04_conv_persist_ab.log
