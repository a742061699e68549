cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LVG_TAP_STACK=1 timeout -s INT 33 python -m pytest tests/test_lres_models.py -m gpu -q -x -k "match_reference_gpu or bf16" > gpurun_out/lres_tap.log 2>&1; echo "rc=$?" >> gpurun_out/lres_tap.log
mkdir -p gpurun_out/miopen_db && cp -r long-video-gan_amd/miopen_db/* gpurun_out/miopen_db/
tail -4 gpurun_out/lres_tap.log
