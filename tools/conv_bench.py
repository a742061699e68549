"""Timing of the generator's dense contractions at BASELINE.json configs[1] (8 clips, bf16): the hand-written
implicit-GEMM kernel with fused temporal taps + epilogue (lvg_conv3d_frames) against the MIOpen route it replaces
(one igemm convolution over tap-stacked output channels + lvg_tapconv_epilogue). MEASUREMENT TOOL.

    python tools/conv_bench.py [iters]        (LVG_CONV_STAGE=reg / LVG_CONV_BN=64|128 select kernel variants)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
if os.path.isdir(_DB) and os.access(_DB, os.W_OK):
    os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_DB, 'db'))
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_DB, 'cache'))

import torch
import torch.nn.functional as F

from torch_utils.ops import conv3d_frames as cf
from torch_utils.ops.modconv_epilogue import tap_gather_forward
from lvg.models.lres import stack_taps

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ONLY = sys.argv[2] if len(sys.argv) > 2 else ''
dev, dt = 'cuda', torch.bfloat16
N = 8
# (frames per clip, ci, co, h, w, kt)  -- the modulated convolutions of lres-G, 128-frame clips
SHAPES = [
    (24, 512, 512, 3, 4, 3), (32, 512, 512, 5, 8, 3), (48, 512, 512, 5, 8, 3), (80, 512, 512, 9, 16, 3), (80, 512, 256, 9, 16, 3),
    (144, 256, 256, 9, 16, 3), (128, 256, 256, 9, 16, 1), (128, 256, 128, 9, 16, 1), (128, 128, 128, 18, 32, 1), (128, 128, 64, 18, 32, 1),
    (128, 64, 64, 36, 64, 1),
    # discriminator (5 x 3 x 3)
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
    wt = (torch.randn(co, ci, kt, 3, 3, device=dev) / (ci * kt * 9) ** 0.5).to(dt)
    pre = torch.rand(f, co, device=dev) + 0.5
    post = torch.rand(f, co, device=dev) + 0.5
    b = torch.zeros(co, device=dev, dtype=dt)
    flops = 2.0 * f * h * w * co * ci * kt * 9
    wp = cf.pack_weight(wt)
    wst = stack_taps(wt).contiguous(memory_format=torch.channels_last)

    def hand():
        return cf.conv3d_frames_forward(x, wt, N, pre, b, None, post, act='lrelu', clamp=256.0, want_msq=True, packed=wp)

    def miopen():
        z = F.conv2d(x, wst, padding=1)
        return tap_gather_forward(z, pre, b, None, post, kt, N, act='lrelu', clamp=256.0, want_msq=True)

    def miopen_conv_only():
        return F.conv2d(x, wst, padding=1)

    ok = cf.supported(x, wt)
    th = timeit(hand) if ok else float('nan')
    tm = timeit(miopen)
    tc = timeit(miopen_conv_only)
    if ok:
        a, bsum = hand()[0].float(), miopen()[0].float()
        err = float((a - bsum).norm() / bsum.norm())
    else:
        err = float('nan')
    tot_h += th
    tot_m += tm
    print(f'{tag:28s} hand {th*1e3:8.1f} us {flops/th/1e9:7.1f} TF | miopen conv+gather {tm*1e3:8.1f} us {flops/tm/1e9:7.1f} TF '
          f'(conv alone {tc*1e3:8.1f} us {flops/tc/1e9:7.1f} TF) | rel diff {err:.2e} | wgs {cf.workgroups(f, h, w, ci, co, kt, 3, 3)}', flush=True)
print(f'total: hand {tot_h:.3f} ms, miopen route {tot_m:.3f} ms')
