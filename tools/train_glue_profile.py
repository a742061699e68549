"""Which source lines launch the PyTorch kernels of ONE LowResTrainer.train_step (update_G + update_D + EMA, no R1): one eager step under torch.profiler,
device time grouped by (aten op, input shapes, innermost frame inside this repo). MEASUREMENT TOOL (GPU).

    python tools/train_glue_profile.py [rows] [clips] [micro-batches]"""
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
import torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity

from lvg import ddp
from lvg.models import lres
from lvg.models.lres import VideoGenerator, VideoDiscriminator
from lvg.optim import FlatAdam

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 70
CLIPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ACCUM = int(sys.argv[3]) if len(sys.argv) > 3 else 1
from lvg.train_lres import LowResTrainer
dev, dtype = torch.device('cuda', 0), torch.bfloat16
torch.manual_seed(0)
tr = LowResTrainer(seq_length=128, device=dev, compute_dtype=dtype, G_grad_accum=ACCUM, D_grad_accum=ACCUM, overlap_grad_sync=False, with_ema=True)
real = torch.rand(CLIPS, 3, 128, 36, 64, device=dev) * 2 - 1
_n = [1]


def step():
    tr.train_step(_n[0], real, r1_interval=0)
    _n[0] += 1


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()

from torch.autograd import DeviceType
agg = collections.defaultdict(lambda: [0.0, 0])
kern = 0.0
for ev in prof.events():
    t = getattr(ev, 'self_device_time_total', 0) or 0
    if t <= 0:
        continue
    if ev.device_type != DeviceType.CPU:
        kern += t
        continue
    scope, par = [], ev.cpu_parent
    while par is not None:
        if not par.name.startswith('aten::'):
            scope.append(par.name.replace('autograd::engine::evaluate_function: ', 'bwd:')[:34])
        par = par.cpu_parent
    shapes = str(ev.input_shapes)[:64] if ev.input_shapes else ''
    a = agg[(ev.name[:30], shapes, ' < '.join(scope[:2]))]
    a[0] += t
    a[1] += 1
tot = sum(v[0] for v in agg.values())
print(f'device time of one eager step: {kern / 1e3:.2f} ms in kernels, {tot / 1e3:.2f} ms attributed to {sum(v[1] for v in agg.values())} op events')
glue = {k: v for k, v in agg.items() if k[0].startswith('aten::')}
print(f'aten ops: {sum(v[0] for v in glue.values()) / 1e3:.2f} ms in {sum(v[1] for v in glue.values())} events')
for (name, shapes, scope), (t, n) in sorted(glue.items(), key=lambda kv: -kv[1][0])[:ROWS]:
    print(f'{t / 1e3:7.3f} ms {n:4d}x  {name:30s} {shapes:64s} {scope}')
by_scope = collections.defaultdict(lambda: [0.0, 0])
for (name, shapes, scope), (t, n) in glue.items():
    by_scope[scope][0] += t
    by_scope[scope][1] += n
print('--- aten ops by enclosing scope')
for scope, (t, n) in sorted(by_scope.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'{t / 1e3:7.3f} ms {n:4d}x  {scope}')
