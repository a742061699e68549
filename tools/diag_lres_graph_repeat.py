"""MEASUREMENT TOOL (round 5): repeat the eager / graph comparison of one LowResTrainer step (float32, noise pinned, temporal stretch on) and
name the parameter with the largest discriminator-gradient difference: looking for a race in the graph path."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'long-video-gan_amd'))
import torch
from lvg.train_lres import LowResTrainer

kw = dict(seq_length=8, height=36, width=64, device='cuda', compute_dtype=torch.float32, G_grad_accum=2, D_grad_accum=2,
          overlap_grad_sync=False, with_ema=True, temp_scale_augment=1.0, diffaug_policy='')
real = None

def one(use_graphs):
    global real
    torch.manual_seed(0)
    tr = LowResTrainer(use_graphs=use_graphs, **kw)
    if real is None:
        real = torch.rand(4, 3, 8, 36, 64, device='cuda') * 2 - 1
    draw, fixed = tr.G.sample_temporal_emb, {}
    def same_noise(batch, seq, generator=None, draw=draw, fixed=fixed):
        if (batch, seq) not in fixed:
            fixed[batch, seq] = draw(batch, seq, torch.Generator(device='cuda').manual_seed(100 * batch + seq))
        return fixed[batch, seq]
    tr.G.sample_temporal_emb = same_noise
    torch.manual_seed(5)
    tr.train_step(step=1, real_video=real, r1_interval=0)
    names = [n for n, _ in tr.D.named_parameters()]
    views = [v.clone() for v in tr.D_sync.views]
    return tr.G_sync.flat.clone(), tr.D_sync.flat.clone(), names, views

e = one(False)
for rep in range(6):
    g = one(True if rep % 2 == 0 else False)
    mG, mD = float(e[0].abs().max()), float(e[1].abs().max())
    worst = max(range(len(e[3])), key=lambda i: float((e[3][i] - g[3][i]).abs().max()))
    print(f'rep {rep} {"graph" if rep % 2 == 0 else "eager"}: G {float((e[0]-g[0]).abs().max())/mG:.2e}  D {float((e[1]-g[1]).abs().max())/mD:.2e}  worst D tensor {e[2][worst]} '
          f'{float((e[3][worst]-g[3][worst]).abs().max()):.2e} of {float(e[3][worst].abs().max()):.2e}', flush=True)
