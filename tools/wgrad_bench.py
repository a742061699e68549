"""Timing of the weight gradients of the generator / discriminator convolutions at BASELINE.json configs[1] (8 clips,
bf16): hand-written kernel (lvg_conv3d_frames_wgrad + the range sum) against MIOpen's weight-gradient call on the
tap-stacked form the model used before. MEASUREMENT TOOL.   python tools/wgrad_bench.py [iters] [filter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
if os.path.isdir(_DB) and os.access(_DB, os.W_OK):
    os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_DB, 'db'))
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_DB, 'cache'))

import torch

from torch_utils.ops import conv3d_frames as cf
from lvg.models.lres import stack_taps

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ONLY = sys.argv[2] if len(sys.argv) > 2 else ''
dev, dt = 'cuda', torch.bfloat16
N = 8
SHAPES = [
    (32, 512, 512, 5, 8, 3), (48, 512, 512, 5, 8, 3), (80, 512, 512, 9, 16, 3), (80, 512, 256, 9, 16, 3),
    (144, 256, 256, 9, 16, 3), (128, 256, 256, 9, 16, 1), (128, 256, 128, 9, 16, 1), (128, 128, 128, 18, 32, 1), (128, 128, 64, 18, 32, 1),
    (128, 64, 64, 36, 64, 1),
    (128, 64, 64, 32, 32, 5), (128, 64, 128, 32, 32, 5), (64, 128, 128, 16, 16, 5), (64, 128, 256, 16, 16, 5), (32, 256, 256, 8, 8, 5), (32, 256, 512, 8, 8, 5),
]


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS


tot_h = tot_m = 0.0
for (t, ci, co, h, w, kt) in SHAPES:
    tag = f'{t}x{ci}->{co}@{h}x{w}k{kt}'
    if ONLY and ONLY not in tag:
        continue
    f = t * N
    x = torch.randn(f, ci, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    dz = torch.randn(f, kt * co, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    dy = dz[:, (kt // 2) * co:(kt // 2 + 1) * co]
    wt = torch.randn(co, ci, kt, 3, 3, device=dev, dtype=dt)
    wst = stack_taps(wt).contiguous(memory_format=torch.channels_last)
    flops = 2.0 * f * h * w * co * ci * kt * 9

    def hand():
        return cf.conv3d_frames_wgrad(x, dy, kt, 3, 3, N).to(dt)

    def miopen():
        return torch.ops.aten.convolution_backward(dz, x, wst, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]

    ok = cf.wgrad_supported(x, dy, kt, 3, 3)
    th = timeit(hand) if ok else float('nan')
    tm = timeit(miopen)
    tot_h += th
    tot_m += tm
    sp = cf.wgrad_splits(f, h, w, ci, co, kt, 3, 3)
    print(f'{tag:28s} hand {th*1e3:8.1f} us {flops/th/1e9:7.1f} TF (splits {sp}) | miopen stacked wgrad {tm*1e3:8.1f} us {flops/tm/1e9:7.1f} TF', flush=True)
print(f'total: hand {tot_h:.3f} ms, miopen {tot_m:.3f} ms')
