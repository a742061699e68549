cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/glue_profile.py 90 > gpurun_out/r02_glue_profile.log 2>&1; echo rc=$? >> gpurun_out/r02_glue_profile.log
tail -190 gpurun_out/r02_glue_profile.log | cut -c1-250
