#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ada_augment.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/ada_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_ada_bench.log
