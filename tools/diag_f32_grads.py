"""Which float32 route is closer to float64? The seven parameter gradients of tests/test_lres_models.py (T = 16 generator + discriminator)
from (a) this repo's networks in float64 on the CPU (truth), (b) the reference's float32 CPU run (the golden), (c) GPU float32 with the
library convolutions, (d) GPU float32 with the split-operand route on the hand-written kernels. MEASUREMENT TOOL (GPU)."""
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
KEYS = ('g_spatial_input', 'g_to_rgb_weight', 'g_t0_bias_0', 'g_s3_weight_1', 'g_map_l1_bias', 'd_b0_conv_vid_weight', 'd_ep_linear_1_weight')


def run(device, dtype, split):
    lres.SPLIT_F32 = split
    G, D = VideoGenerator(), VideoDiscriminator(seq_length=T, max_edge=64)
    fill_named(G); fill_named(D)
    G, D = G.to(device=device, dtype=dtype).requires_grad_(True), D.to(device=device, dtype=dtype).requires_grad_(True)
    for net in (G, D):                                   # resampling taps cross the op boundary as float32 whatever the tensors are
        for mod in net.modules():
            for name, buf in list(mod._buffers.items()):
                if buf is not None and name in ('filter', '_downsample_filter') and buf.dtype != torch.float32:
                    mod._buffers[name] = buf.float()
    noise = torch.tensor(g['noise'], device=device, dtype=dtype)
    ws = G.compute_latent_ws(G.temporal_emb.blur(noise), T)
    video = G.synthesize_video(G._temporal_input(ws), ws, T, **({} if dtype == torch.float32 else dict(dtype=dtype)))
    logits = D(video, **({} if dtype == torch.float32 else dict(dtype=dtype)))
    F.softplus(-logits).mean().backward()
    pairs = dict(g_spatial_input=G.spatial_input, g_to_rgb_weight=G.to_rgb.weight, g_t0_bias_0=G.temporal_layers[0].bias_0,
                 g_s3_weight_1=G.spatial_layers[3].weight_1, g_map_l1_bias=G.latent_mapping.layer_1.bias,
                 d_b0_conv_vid_weight=D.blocks[0].conv_vid.weight, d_ep_linear_1_weight=D.epilogue.linear_1.weight)
    return {k: p.grad.detach().double().cpu().numpy() for k, p in pairs.items()}, video.detach().double().cpu().numpy()


torch.set_num_threads(16)
g64 = load_golden('lres_models_f64')                     # the reference run in float64 (tests/golden/make_golden_models_f64.py)
truth = {k: np.asarray(g64[k], dtype=np.float64) for k in KEYS}
vt = np.asarray(g64['video'], dtype=np.float64) if 'video' in g64 else np.asarray(g['video'], dtype=np.float64)
lib, vl = run('cuda', torch.float32, False)
spl, vs = run('cuda', torch.float32, True)
rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))
print(f'video max |err| vs float64: reference golden {np.abs(g["video"] - vt).max():.2e}  library {np.abs(vl - vt).max():.2e}  split {np.abs(vs - vt).max():.2e}')
print(f'{"gradient":24s} {"reference f32":>14s} {"GPU library":>12s} {"GPU split":>10s}   (max |err| / max |truth|)')
for k in KEYS:
    print(f'{k:24s} {rel(g[k], truth[k]):14.2e} {rel(lib[k], truth[k]):12.2e} {rel(spl[k], truth[k]):10.2e}')
