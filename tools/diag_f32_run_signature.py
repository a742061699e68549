"""One float32 split-route pass of the lres generator + discriminator: a signature (sum of |.|) of the incoming gradient and of the data
gradient of every _TapConvEpilogue backward call, plus the final error of g_spatial_input against the float64 golden. Run several times and
diff: the first call whose signature changes between runs is where the run-to-run difference enters. MEASUREMENT TOOL (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F
from conftest import load_golden
from helpers.named_fill import fill_named
from lvg.models import lres
from lvg.models.lres import VideoGenerator, VideoDiscriminator
T = 16
g = load_golden('lres_models'); g64 = load_golden('lres_models_f64')
sig = []
orig = lres._TapConvEpilogue._backward
def spy(ctx, x, weight, ysum, pre, b, res, post, dout, wt_packed, need):
    out = orig(ctx, x, weight, ysum, pre, b, res, post, dout, wt_packed, need)
    sig.append((tuple(x.shape), tuple(weight.shape), float(dout.double().abs().sum()), float(out[0].double().abs().sum()) if out[0] is not None else 0.0,
                float(out[1].double().abs().sum()) if out[1] is not None else 0.0))
    return out
lres._TapConvEpilogue._backward = staticmethod(spy)
G, D = VideoGenerator(), VideoDiscriminator(seq_length=T, max_edge=64)
fill_named(G); fill_named(D)
G, D = G.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
noise = torch.tensor(g['noise'], device='cuda')
ws = G.compute_latent_ws(G.temporal_emb.blur(noise), T)
video = G.synthesize_video(G._temporal_input(ws), ws, T)
logits = D(video)
F.softplus(-logits).mean().backward()
want = np.asarray(g64['g_spatial_input'], dtype=np.float64)
err = float(np.abs(G.spatial_input.grad.double().cpu().numpy() - want).max() / np.abs(want).max())
print(f'ERR {err:.2e} video {float(video.double().abs().sum()):.10e} logits {float(logits.double().abs().sum()):.10e}')
for i, s in enumerate(sig):
    print(f'{i:3d} x{s[0]} w{s[1]} dout {s[2]:.8e} gx {s[3]:.8e} gw {s[4]:.8e}')
