#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_weight_prep2d.py tests/test_modconv2d_layout.py tests/test_conv2d_frames.py tests/test_sres_models.py tests/test_train_sres.py -m gpu -q -x --no-header -rf 2>&1 | tail -15 | tee gpurun_out/r03_prep2d_pytest.log
for v in 0 1 0 1; do echo "LVG_SRES_WEIGHT_PREP=$v"; LVG_SRES_WEIGHT_PREP=$v timeout 300 python tools/sres_step.py 6 2>&1 | grep "^{"; done | tee gpurun_out/r03_sres_weight_prep_ab.log
