"""1x1 (skip) convolutions of the lres generator: hand-written kernel (ntap = 1 path of conv3d_igemm) vs MIOpen. MEASUREMENT TOOL."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
if os.path.isdir(_DB) and os.access(_DB, os.W_OK):
    os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_DB, 'db'))
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_DB, 'cache'))
import torch
import torch.nn.functional as F
from torch_utils.ops import conv3d_frames as cf
dev, dt, N, ITERS = 'cuda', torch.bfloat16, 8, 5
SHAPES = [(80, 512, 512, 9, 16), (144, 256, 256, 9, 16), (128, 256, 128, 9, 16), (128, 128, 128, 18, 32), (128, 128, 64, 18, 32), (128, 64, 64, 36, 64)]


def timeit(fn):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / ITERS


for (t, ci, co, h, w) in SHAPES:
    f = t * N
    x = torch.randn(f, ci, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 1, 1, 1, device=dev) / ci ** 0.5).to(dt)
    w2 = wt[:, :, 0].contiguous(memory_format=torch.channels_last)
    flops = 2.0 * f * h * w * co * ci
    byts = f * h * w * (ci + co) * 2
    ok = cf.supported(x, wt)
    th = timeit(lambda: cf.conv3d_frames_forward(x, wt, N, keep_sum=False)) if ok else float('nan')
    tm = timeit(lambda: F.conv2d(x, w2))
    err = float((cf.conv3d_frames_forward(x, wt, N, keep_sum=False)[0].float() - F.conv2d(x, w2).float()).norm() / F.conv2d(x, w2).float().norm()) if ok else -1
    print(f'{t}x{ci}->{co}@{h}x{w} 1x1: hand {th*1e3:7.1f} us ({byts/th/1e6:6.0f} GB/s) | miopen {tm*1e3:7.1f} us ({byts/tm/1e6:6.0f} GB/s) | rel diff {err:.1e}', flush=True)
