"""Composition of one sres generator update (bench.py's `sres` leg: 2 segments x 8 frames 144x256): device time by kernel and
by aten op + input shapes (torch.profiler). MEASUREMENT TOOL (GPU).   python tools/sres_profile.py [rows]"""
import os
import sys
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
if os.path.isdir(_DB) and os.access(_DB, os.W_OK):
    os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_DB, 'db'))
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_DB, 'cache'))

import torch
from torch.autograd import DeviceType
from torch.profiler import profile, ProfilerActivity
from lvg.train_sres import SuperResTrainer

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda', 0)
torch.manual_seed(0)
tr = SuperResTrainer(device=dev, compute_dtype=torch.float16, augment_real_sign_target=None, augment_p_init=0.0,
                     in_augment_strength=0.0, lr_cond_prob=1.0, overlap_grad_sync=False, with_ema=False)
lr = torch.rand(2, 3, tr.context_seq_length, 36, 64, device=dev) * 2 - 1
for _ in range(3):
    tr.update_G(lr)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.update_G(lr)
    torch.cuda.synchronize()
kern, ops = collections.defaultdict(lambda: [0.0, 0]), collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    t = getattr(ev, 'self_device_time_total', 0) or 0
    if t <= 0:
        continue
    if ev.device_type != DeviceType.CPU:
        a = kern[ev.name[:100]]
    else:
        a = ops[(ev.name[:34], str(ev.input_shapes)[:110] if ev.input_shapes else '')]
    a[0] += t
    a[1] += 1
print(f'kernels: {sum(v[0] for v in kern.values()) / 1e3:.2f} ms in {sum(v[1] for v in kern.values())} launches')
for name, (t, n) in sorted(kern.items(), key=lambda kv: -kv[1][0])[:ROWS // 2]:
    print(f'{t / 1e3:7.3f} ms {n:4d}x  {name}')
print('--- ops')
for (name, shapes), (t, n) in sorted(ops.items(), key=lambda kv: -kv[1][0])[:ROWS]:
    print(f'{t / 1e3:7.3f} ms {n:4d}x  {name:34s} {shapes}')
