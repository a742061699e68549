"""Timing of the super-resolution generator's dense contractions at BASELINE.json configs[3] (16 frames = 2 segments x 8,
f16): the hand-written 2-D implicit-GEMM kernels (lvg_conv2d_frames forward / data gradient, lvg_conv2d_frames_wgrad)
against the library convolution (channels-last F.conv2d and its backward). MEASUREMENT TOOL.

    python tools/conv2d_bench.py [iters] [filter]        (LVG_CONV2D_BM=256 / LVG_CONV2D_BN=64 select kernel variants)"""
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

from torch_utils.ops import conv2d_frames as c2

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ONLY = sys.argv[2] if len(sys.argv) > 2 else ''
LIB = os.environ.get('LVG_BENCH_LIB', '1') != '0'
dev, dt = 'cuda', torch.float16
N = 16
# (ci, co, h, w) of the 16-bit 3 x 3 layers L3 .. L13 (input planes h x w, padding 2): SURVEY.md Appendix A.3
SHAPES = [(539, 512, 29, 36), (539, 512, 38, 52), (539, 512, 38, 52), (539, 512, 56, 84), (539, 512, 56, 84), (539, 512, 92, 148), (539, 362, 92, 148),
          (389, 256, 92, 148), (283, 181, 164, 276), (208, 128, 164, 276), (155, 128, 164, 276)]


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


tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0, lib_fwd=0.0, lib_bwd=0.0)
for (ci, co, h, w) in SHAPES:
    tag = f'{ci}->{co}@{h}x{w}'
    if ONLY and ONLY not in tag:
        continue
    geo = c2.Geometry(h, w, 2)
    cip, cop = c2.round_up(ci, 64), c2.round_up(co, 64)
    xp = torch.zeros(N, geo.hx, geo.wx, cip, device=dev, dtype=dt)
    xp[:, 2:2 + h, 2:2 + w, :ci] = torch.randn(N, h, w, ci, device=dev, dtype=dt)
    weight = torch.randn(co, ci, 3, 3, device=dev) / (ci * 9) ** 0.5
    wp, wd_ = c2.pack_weight(weight, dt, cip, cop), c2.pack_weight_dgrad(weight, dt, cip, cop)
    dyp = torch.zeros(N, geo.hd, geo.wd, cop, device=dev, dtype=dt)
    dyp[:, :geo.ho, :geo.wo, :co] = torch.randn(N, geo.ho, geo.wo, co, device=dev, dtype=dt)
    flops = 2.0 * N * geo.ho * geo.wo * co * ci * 9              # algorithmic (unpadded channels, true output size)
    t_f = timeit(lambda: c2.conv2d_valid(xp, wp, geo.ho, geo.wo))
    t_d = timeit(lambda: c2.conv2d_valid(dyp, wd_, h, w))
    t_w = timeit(lambda: c2.conv2d_wgrad(xp, dyp))
    line = (f'{tag:22s} fwd {t_f*1e3:8.1f} us {flops/t_f/1e9:7.1f} TF | dgrad {t_d*1e3:8.1f} us {flops/t_d/1e9:7.1f} TF | '
            f'wgrad {t_w*1e3:8.1f} us {flops/t_w/1e9:7.1f} TF (splits {c2.wgrad_splits(N, geo.hx, geo.wx, geo.hd, geo.wd, cip, cop)})')
    tot['fwd'] += t_f; tot['dgrad'] += t_d; tot['wgrad'] += t_w
    if LIB:
        c8 = c2.round_up(ci, 8)
        xl = torch.randn(N, c8, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wl = (torch.randn(c2.round_up(co, 8), c8, 3, 3, device=dev) / (ci * 9) ** 0.5).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        t_lf = timeit(lambda: F.conv2d(xl, wl, padding=2))
        y = F.conv2d(xl, wl, padding=2)
        gy = torch.randn_like(y)
        t_lb = timeit(lambda: torch.autograd.grad(y, [xl, wl], gy, retain_graph=True))
        tot['lib_fwd'] += t_lf; tot['lib_bwd'] += t_lb
        line += f' | library fwd {t_lf*1e3:8.1f} us {flops/t_lf/1e9:6.1f} TF, dgrad + wgrad {t_lb*1e3:8.1f} us {2*flops/t_lb/1e9:6.1f} TF'
    print(line, flush=True)
print('total ms: ' + ', '.join(f'{k} {v:.3f}' for k, v in tot.items()))
