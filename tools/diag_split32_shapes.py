"""Every distinct call of conv3d_frames_split32 made by a float32 generator + discriminator pass (forward and data gradient), replayed on
random operands against the float64 convolution of the same operands. MEASUREMENT TOOL (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F
from conftest import load_golden
from helpers.named_fill import fill_named
from lvg.models import lres
from lvg.models.lres import VideoGenerator, VideoDiscriminator
from torch_utils.ops import conv3d_frames as c3
T = 16
g = load_golden('lres_models')
calls = {}
orig = c3.conv3d_frames_split32
def spy(x, weight, shift, *a, **k):
    calls.setdefault((tuple(x.shape), tuple(weight.shape), shift), 0)
    calls[(tuple(x.shape), tuple(weight.shape), shift)] += 1
    return orig(x, weight, shift, *a, **k)
c3.conv3d_frames_split32 = spy
G, D = VideoGenerator(), VideoDiscriminator(seq_length=T, max_edge=64)
fill_named(G); fill_named(D)
G, D = G.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
noise = torch.tensor(g['noise'], device='cuda')
ws = G.compute_latent_ws(G.temporal_emb.blur(noise), T)
video = G.synthesize_video(G._temporal_input(ws), ws, T)
F.softplus(-D(video)).mean().backward()
c3.conv3d_frames_split32 = orig
torch.manual_seed(0)
for (xs, wsh, shift), cnt in calls.items():
    f, ci, h, w = xs
    co, _, kt, kh, kw = wsh
    x = torch.randn(xs, device='cuda') * torch.rand(f, 1, 1, 1, device='cuda').mul(4).exp2()
    wt = torch.randn(wsh, device='cuda') * 0.05
    out = orig(x, wt, shift, keep_sum=False)[0]
    # float64 reference: frames are (t, n) time-major, n = shift; temporal taps reach +-shift frames
    n = shift if kt > 1 else f
    t = f // n
    x5 = x.double().reshape(t, n, ci, h, w).permute(1, 2, 0, 3, 4)
    ref = F.conv3d(x5, wt.double(), padding=(kt // 2, kh // 2, kw // 2)).permute(2, 0, 1, 3, 4).reshape(f, co, h, w)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print(f'x {str(xs):22s} w {str(wsh):24s} shift {shift:3d} calls {cnt:2d}  rel err {err:.2e}{"   <<<" if err > 1e-5 else ""}', flush=True)
