cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python tools/sres_profile.py 70 > gpurun_out/r02_sres_profile.log 2>&1; echo rc=$? >> gpurun_out/r02_sres_profile.log
tail -110 gpurun_out/r02_sres_profile.log | cut -c1-220
