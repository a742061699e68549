"""Golden vectors for the low-resolution networks, produced by the REFERENCE model code on CPU
(its ops take the impl='ref' path there). Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_models.py [/root/reference]

Weights are NOT stored: both sides fill them with tests/helpers/named_fill.py."""

import os
import sys
import zlib

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from model import generator_lres, discriminator_lres  # noqa: E402
from helpers.named_fill import fill_named, analytic_buffers  # noqa: E402

assert os.path.realpath(generator_lres.__file__).startswith(os.path.realpath(REF))
torch.set_num_threads(8)
T = 16

G = generator_lres.VideoGenerator()
D = discriminator_lres.VideoDiscriminator(seq_length=T, max_edge=64)
fill_named(G)
fill_named(D)
G.requires_grad_(True)
D.requires_grad_(True)

in_len = G.compute_seq_lengths(T)[0]
emb_len = in_len * G.total_temporal_scale
noise = torch.randn(1, G.temporal_emb.noise_channels, emb_len + G.temporal_emb.kernel_size - 1, generator=torch.Generator().manual_seed(1))
emb = G.temporal_emb.blur(noise)
ws = G.compute_latent_ws(emb, T)
w0 = ws.pop(0)
temporal_input = G.w_to_temp_input(w0.permute(0, 2, 1).reshape(-1, w0.shape[1])).reshape(1, in_len, -1).permute(0, 2, 1)
feats = G.synthesize_video(temporal_input, ws, T, return_features=True)
video = feats[-1]
logits = D(video)
loss = F.softplus(-logits).mean()
loss.backward()

out = dict(
    noise=noise.numpy(), video=video.detach().numpy(), logits=logits.detach().numpy(), loss=np.array(float(loss)),
    g_spatial_input=G.spatial_input.grad.numpy(), g_to_rgb_weight=G.to_rgb.weight.grad.numpy(),
    g_t0_bias_0=G.temporal_layers[0].bias_0.grad.numpy(), g_s3_weight_1=G.spatial_layers[3].weight_1.grad.numpy(),
    g_map_l1_bias=G.latent_mapping.layer_1.bias.grad.numpy(),
    d_b0_conv_vid_weight=D.blocks[0].conv_vid.weight.grad.numpy(), d_ep_linear_1_weight=D.epilogue.linear_1.weight.grad.numpy(),
    feat_rms=np.array([float(f.detach().float().square().mean().sqrt()) for f in feats]),
)
# checksums of analytic buffers (filters designed by scipy.signal.firwin in both implementations)
for prefix, net in (('G', G), ('D', D)):
    for name, buf in analytic_buffers(net).items():
        out[f'buf_{prefix}_{name}'] = np.array([float(buf.double().sum()), float(buf.double().abs().sum()), float(buf.numel())])
np.savez_compressed(os.path.join(HERE, 'lres_models.npz'), **out)
print('video', video.shape, float(video.abs().mean()), 'logits', logits.flatten().tolist(), 'loss', float(loss))
print({k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items() if not k.startswith('buf_')})
