#!/bin/bash
# r03: float32 layers of the sres generator through split operands on the hand-written kernels: parity, sres model goldens, sres leg A/B
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 900 python -m pytest tests/test_conv2d_frames.py tests/test_conv3d_frames.py tests/test_sres_models.py tests/test_train_sres.py tests/test_modconv2d_layout.py -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r03_split_tests.log
tail -6 gpurun_out/r03_split_tests.log
{
for v in 1 0 1; do
  LVG_SRES_SPLIT_F32=$v timeout 300 python tools/sres_step.py 6 > gpurun_out/r03_sres_split_$v.log 2>&1
  echo "SPLIT_F32=$v: $(grep '^{' gpurun_out/r03_sres_split_$v.log)"
done
} 2>&1 | tee gpurun_out/r03_split_ab.log
grep "measured\|split" gpurun_out/parity_measured.json | head -0
python - <<'PY'
import json
d = json.load(open('gpurun_out/parity_measured.json'))
for k, v in d.items():
    if 'split' in k: print(k, v)
PY
