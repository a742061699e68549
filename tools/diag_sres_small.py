"""MEASUREMENT TOOL (round 5): eager / eager / graph / segmented spread of one SuperResTrainer step at the SMALL test configuration, float32."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, 'long-video-gan_amd')); sys.path.insert(0, os.path.join(root, 'tests'))
import torch
from lvg.train_sres import SuperResTrainer
from helpers.ada_cfg import TRAIN_SRES_KW
SMALL = dict(seq_length=2, temporal_context=1, lr_height=9, lr_width=16, hr_height=36, hr_width=64,
             G_kwargs=dict(latent_z_dim=32, latent_w_dim=48, channel_base=1024, channel_max=24, num_fp16_res=2),
             D_kwargs=dict(channels_base=1024, channels_max=32, num_fp16_res=0), augment_kwargs=TRAIN_SRES_KW, overlap_grad_sync=False)
kw = dict(SMALL, augment_p_init=0.0, augment_real_sign_target=None, in_augment_strength=0.0, lr_cond_prob=1.0, G_grad_accum=2, D_grad_accum=2)
lr = hr = None
out = {}
for name, ug in (('eager', False), ('eager2', False), ('graph', True), ('segmented', 'segmented'), ('graph2', True)):
    torch.manual_seed(0)
    tr = SuperResTrainer(device='cuda', compute_dtype=torch.float32, use_graphs=ug, **kw)
    if lr is None:
        lr = torch.rand(4, 3, 4, 9, 16, device='cuda') * 2 - 1
        hr = torch.rand(4, 3, 2, 36, 64, device='cuda') * 2 - 1
    draw, fixed = tr.G.sample_latent_z, {}
    def same_z(batch_size, generator_z=None, draw=draw, fixed=fixed):
        if batch_size not in fixed:
            fixed[batch_size] = draw(batch_size, torch.Generator(device='cuda').manual_seed(7 + batch_size))
        return fixed[batch_size]
    tr.G.sample_latent_z = same_z
    torch.manual_seed(5)
    tr.update_G(lr)
    g = tr.G_sync.flat.clone()
    tr.update_D(lr, lr, hr)
    out[name] = (g, tr.D_sync.flat.clone(), [n for n, _ in tr.D.named_parameters()], [v.clone() for v in tr.D_sync.views])
e = out['eager']
for name in ('eager2', 'graph', 'segmented', 'graph2'):
    o = out[name]
    worst = max(range(len(e[3])), key=lambda i: float((e[3][i] - o[3][i]).abs().max()))
    print(f'{name:10s} G {float((e[0]-o[0]).abs().max())/float(e[0].abs().max()):.2e}  D {float((e[1]-o[1]).abs().max())/float(e[1].abs().max()):.2e}'
          f'  worst D tensor {e[2][worst]} {float((e[3][worst]-o[3][worst]).abs().max()):.2e} of {float(e[3][worst].abs().max()):.2e}', flush=True)
