cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_modconv2d_layout.py tests/test_sres_models.py tests/test_train_sres.py -m gpu -q -x --no-header -rf > gpurun_out/r02_sres_cl_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02_sres_cl_pytest.log
tail -6 gpurun_out/r02_sres_cl_pytest.log
LVG_SRES_CHANNELS_LAST=0 timeout 300 python tools/sres_step.py 4 > gpurun_out/r02_sres_step_nchw.log 2>&1; grep "^{" gpurun_out/r02_sres_step_nchw.log
timeout 600 python tools/sres_step.py 4 > gpurun_out/r02_sres_step_cl.log 2>&1; grep "^{" gpurun_out/r02_sres_step_cl.log || tail -5 gpurun_out/r02_sres_step_cl.log
rm -rf gpurun_out/miopen_db; mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
