"""MEASUREMENT TOOL (round 5): which replayed phase of SuperResTrainer departs from the eager one (float32, no random draws)?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'long-video-gan_amd'))
import torch
from lvg.train_sres import SuperResTrainer

def run(use_graphs, accum, nsteps=1, dtype=torch.float32):
    kw = dict(augment_real_sign_target=None, augment_p_init=0.0, in_augment_p=0.0, lr_cond_prob=1.0, G_grad_accum=accum, D_grad_accum=accum, overlap_grad_sync=False)
    torch.manual_seed(0)
    tr = SuperResTrainer(device='cuda', compute_dtype=dtype, use_graphs=use_graphs, **kw)
    g = torch.Generator(device='cuda').manual_seed(1)
    lr = torch.rand(2 * accum, 3, tr.context_seq_length, 36, 64, device='cuda', generator=g) * 2 - 1
    hr = torch.rand(2 * accum, 3, tr.seq_length, 144, 256, device='cuda', generator=g) * 2 - 1
    res = {}
    for s in range(nsteps):
        tr.update_G(lr)
        res[f'G{s}'] = tr.G_sync.flat.clone()
        tr.update_D(lr, lr, hr)
        res[f'D{s}'] = tr.D_sync.flat.clone()
        res[f'S{s}'] = torch.cat([b.float().flatten() for n, b in tr.G.named_buffers() if n.endswith('magnitude_ema') or n.endswith('w_avg')])
    return res

for accum in (1, 2):
    e, e2, g = run(False, accum), run(False, accum), run(True, accum)
    for k in e:
        m = float(e[k].abs().max())
        print(f'accum {accum} {k}: max {m:.3e} eager-eager2 {float((e[k]-e2[k]).abs().max())/m:.2e} eager-graph {float((e[k]-g[k]).abs().max())/m:.2e}', flush=True)
