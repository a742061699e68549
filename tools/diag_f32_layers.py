"""Per-parameter comparison of the two float32 GPU routes of the lres networks (library convolutions vs split operands on the hand-written
kernels): relative difference of every parameter gradient, in module order. MEASUREMENT TOOL (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F
from conftest import load_golden
from helpers.named_fill import fill_named
from lvg.models import lres
from lvg.models.lres import VideoGenerator, VideoDiscriminator
T = 16
g = load_golden('lres_models')


def run(split):
    lres.SPLIT_F32 = split
    G, D = VideoGenerator(), VideoDiscriminator(seq_length=T, max_edge=64)
    fill_named(G); fill_named(D)
    G, D = G.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
    noise = torch.tensor(g['noise'], device='cuda')
    ws = G.compute_latent_ws(G.temporal_emb.blur(noise), T)
    video = G.synthesize_video(G._temporal_input(ws), ws, T)
    F.softplus(-D(video)).mean().backward()
    out = {}
    for pre, net in (('G', G), ('D', D)):
        for name, p in net.named_parameters():
            if p.grad is not None:
                out[f'{pre}.{name}'] = p.grad.detach().double().cpu().numpy()
    return out


lib = run(False)
spl = run(True)
for k in lib:
    a, b = lib[k], spl[k]
    rel = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-300))
    flag = '  <<<' if rel > 3e-4 else ''
    print(f'{k:60s} {str(a.shape):24s} {rel:10.2e}{flag}')
