cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in 1 2 1 2; do echo "== LVG_CONV_STATIC1=$v"; LVG_CONV_STATIC1=$v timeout 120 python tools/conv_bench.py 10 "->64@" 2>&1 | grep "hand" | cut -c1-64; done
} > gpurun_out/r04_conv_static1_all.log 2>&1
cat gpurun_out/r04_conv_static1_all.log
