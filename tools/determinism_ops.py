"""First aten operation of the lres generator's forward whose INPUTS or outputs differ between two runs on the same inputs. MEASUREMENT TOOL (GPU): python tools/determinism_ops.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten
from lvg.models.lres import VideoGenerator

T = 40
torch.manual_seed(0)
G = VideoGenerator().cuda().train().requires_grad_(True)
emb = G.sample_temporal_emb(2, T, torch.Generator(device='cuda').manual_seed(3))


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        ins = [a for a in tree_flatten((args, kwargs or {}))[0] if torch.is_tensor(a) and a.is_cuda and a.numel() > 0 and a.dtype.is_floating_point]
        insum = [a.detach().double().abs().sum() for a in ins]          # BEFORE the op (it may be in place)
        out = func(*args, **(kwargs or {}))
        outs = [o for o in tree_flatten(out)[0] if torch.is_tensor(o) and o.is_cuda and o.numel() > 0 and o.dtype.is_floating_point]
        self.rows.append((str(func), [tuple(a.shape) for a in ins], insum, [o.detach().double().abs().sum() for o in outs]))
        return out


runs = []
for _ in range(3):
    with Log() as lg:
        video = G.forward_from_emb(emb, T, 1.0, torch.bfloat16)
    runs.append(lg.rows)
torch.cuda.synchronize()
a, b = runs[1], runs[2]
print(len(a), len(b), 'operations')
shown = 0
for i, (x, y) in enumerate(zip(a, b)):
    din = [float((p - q).abs()) for p, q in zip(x[2], y[2])]
    dout = [float((p - q).abs()) for p, q in zip(x[3], y[3])]
    if 'empty' in x[0] or 'new_empty' in x[0]:
        continue
    if any(d > 0 for d in din + dout):
        print(f'op {i}: {x[0]} inputs {x[1]}: input checksum diffs {din}, output diffs {dout}')
        if shown == 0:
            for j in range(max(0, i - 14), i):
                print(f'      before: op {j}: {a[j][0]} {a[j][1]}')
        shown += 1
        if shown >= 6:
            break
