#!/bin/bash
# r03: NHWC x2 down-sampler through LDS (upfirdn2d_nhwc_down2_lds_kernel) against the streaming form (LVG_UPFIRDN_NO_LDS=1): parity tests, then timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_upfirdn2d_gpu.py -m gpu -q -x --no-header 2>&1 | tail -3
{
for v in 1 0 1 0; do echo "== LVG_UPFIRDN_NO_LDS=$v"; if [ $v = 1 ]; then export LVG_UPFIRDN_NO_LDS=1; else unset LVG_UPFIRDN_NO_LDS; fi; timeout 120 python tools/upfirdn_chunk_bench.py 2>&1 | grep "TB/s"; done
} | tee gpurun_out/r03_upfirdn_down2_lds_$1.log
