cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_tapconv_epilogue.py tests/test_ada_augment.py tests/test_modconv_epilogue.py -m gpu -q > gpurun_out/tap_tests.log 2>&1; echo "rc=$?" >> gpurun_out/tap_tests.log
mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
tail -12 gpurun_out/tap_tests.log
