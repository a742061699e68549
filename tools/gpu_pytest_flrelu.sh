cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_filtered_lrelu_gpu.py tests/test_sres_models.py tests/test_train_sres.py tests/test_persistence.py tests/test_cabi_exports.py -m gpu -q -x --no-header -rf > gpurun_out/r02_flrelu_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02_flrelu_pytest.log
tail -12 gpurun_out/r02_flrelu_pytest.log
timeout 200 python tools/sres_probe.py > gpurun_out/r02_sres_probe.log 2>&1; tail -5 gpurun_out/r02_sres_probe.log
