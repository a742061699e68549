#!/bin/bash
# r03: float32 lres tensors on the hand-written kernels (split operands): parity, float32 model goldens, then which convolution kernels a float32
# generator + discriminator pass launches (kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=long-video-gan_amd
( timeout 900 python -m pytest tests/test_conv3d_frames.py tests/test_lres_models.py tests/test_trainer_gpu.py tests/test_tapconv_epilogue.py tests/test_modconv_epilogue.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r03_split3d_tests.log; tail -4 gpurun_out/r03_split3d_tests.log
cat > /tmp/f32pass.py <<'PY'
import sys, os
sys.path.insert(0, 'long-video-gan_amd')
import torch, torch.nn.functional as F
from lvg.models.lres import VideoGenerator, VideoDiscriminator
torch.manual_seed(0)
G = VideoGenerator().cuda().requires_grad_(True); D = VideoDiscriminator(seq_length=32, max_edge=64).cuda().requires_grad_(False)
for _ in range(2):
    v = G(2, 32); F.softplus(-D(v)).mean().backward()
torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f32 -o f -- python /tmp/f32pass.py > gpurun_out/r03_f32pass.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_f32/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if any(k in r['Name'] for k in ('igemm', 'conv', 'Conv', 'Cijk', 'SubTensor', 'wgrad', 'gemm'))]
with open('gpurun_out/r03_f32_pass_conv_kernels.csv', 'w') as o:
    o.write('calls,total_us,avg_us,pct,kernel\n')
    for r in keep[:40]:
        o.write(f"{r['Calls']},{float(r['TotalDurationNs'])/1e3:.0f},{float(r['AverageNs'])/1e3:.1f},{r['Percentage']},\"{r['Name'][:120]}\"\n")
print(open('gpurun_out/r03_f32_pass_conv_kernels.csv').read()[:3000])
PY
rm -rf gpurun_out/prof_f32
