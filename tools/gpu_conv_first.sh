cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_conv3d_frames.py -m gpu -q --no-header -rf -x > gpurun_out/r02_conv_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02_conv_pytest.log
tail -25 gpurun_out/r02_conv_pytest.log
timeout 300 python tools/conv_bench.py 5 > gpurun_out/r02_conv_bench.log 2>&1; tail -15 gpurun_out/r02_conv_bench.log
LVG_CONV_STAGE=reg timeout 200 python tools/conv_bench.py 5 > gpurun_out/r02_conv_bench_reg.log 2>&1; tail -13 gpurun_out/r02_conv_bench_reg.log
