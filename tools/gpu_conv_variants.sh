cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_conv3d_frames.py -m gpu -q --no-header -rf -x > gpurun_out/r02_conv_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02_conv_pytest.log
tail -5 gpurun_out/r02_conv_pytest.log
IFS=';'
for v in $VARIANTS; do
  echo "== $v"
  IFS=' ' env $v timeout 200 python tools/conv_bench.py 5 2>&1 | grep -v amdgpu.ids | cut -c1-60
done > gpurun_out/r02_conv_variants.log 2>&1
cat gpurun_out/r02_conv_variants.log
