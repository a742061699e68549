#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do timeout 600 python -m pytest tests/test_lres_models.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -2; done
timeout 900 python -m pytest tests/test_conv3d_frames.py tests/test_conv2d_frames.py tests/test_trainer_gpu.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -2
python - <<'P'
import json
d=json.load(open('gpurun_out/parity_measured.json'))
for k,v in d.items():
    if 'kink' in k or 'f32_grads' in k: print(k, json.dumps(v))
P
timeout 900 python tools/diag_f32_backward_calls.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r03_f32_kink_flips.log; grep "sign flips" gpurun_out/r03_f32_kink_flips.log | head -3
