"""First timing of the super-resolution generator at BASELINE.json configs[3] size (8-frame 144x256
segments from 36x64, +-4 context frames): forward + backward of `segments` segments, eager launches."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_DB, 'db'))
os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_DB, 'cache'))

import torch  # noqa: E402

from lvg.models import sres  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--segments', type=int, default=4)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--dtype', default='fp16')
ap.add_argument('--forward-only', action='store_true')
args = ap.parse_args()
dtype = dict(fp16=torch.float16, bf16=torch.bfloat16)[args.dtype]

torch.manual_seed(0)
net = sres.VideoGenerator(hr_height=144, hr_width=256, lr_height=36, lr_width=64, compute_dtype=dtype).cuda().requires_grad_(True)
lr = torch.randn(args.segments, 3, 16, 36, 64, device='cuda').clamp(-1, 1)


def step():
    if args.forward_only:
        with torch.no_grad():
            return net(lr)
    for p in net.parameters():
        p.grad = None
    video = net(lr)
    video.square().mean().backward()
    return video


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
frames = args.segments * 8
print(f'[sres] {args.dtype} segments={args.segments} ({frames} frames 144x256) '
      f'{"forward" if args.forward_only else "forward+backward"}: {dt * 1e3:.2f} ms/step, {frames / dt:.1f} frames/s, '
      f'peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB')
