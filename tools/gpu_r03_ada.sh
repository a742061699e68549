#!/bin/bash
# r03: fused ADA stages: GPU parity tests (HIP vs oracle, adjoint identity, gradients vs the composition), train_sres step, pipeline timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ada_augment.py tests/test_train_sres.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -12
timeout 300 python tools/ada_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_ada_bench.log
