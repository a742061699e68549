"""float64 truth for the low-resolution networks' parameter gradients, produced by the REFERENCE model code on CPU in DOUBLE precision
(same weights / noise as make_golden_models.py): which float32 route -- the reference's own float32 run, the library convolutions on the
GPU, the split-operand route on the hand-written kernels -- is how far from the exact arithmetic. Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_models_f64.py [/root/reference]"""

import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from model import generator_lres, discriminator_lres  # noqa: E402
from helpers.named_fill import fill_named  # noqa: E402

assert os.path.realpath(generator_lres.__file__).startswith(os.path.realpath(REF))
torch.set_num_threads(8)
T = 16

G = generator_lres.VideoGenerator()
D = discriminator_lres.VideoDiscriminator(seq_length=T, max_edge=64)
fill_named(G)
fill_named(D)
G, D = G.double(), D.double()
for net in (G, D):                         # resampling taps cross the op boundary as float32 (upfirdn2d asserts it; it casts them itself)
    for mod in net.modules():
        for name, buf in list(mod._buffers.items()):
            if buf is not None and buf.dtype == torch.float64 and buf.ndim <= 2 and buf.numel() <= 64 and 'filter' in name:
                mod._buffers[name] = buf.float()
# the reference's discriminator epilogue casts its input to float32 explicitly (discriminator_lres.py:398): keep it double here
_type = torch.Tensor.type
torch.Tensor.type = lambda self, *a, **k: self if (a and a[0] is torch.float32) else _type(self, *a, **k)
G.requires_grad_(True)
D.requires_grad_(True)

in_len = G.compute_seq_lengths(T)[0]
emb_len = in_len * G.total_temporal_scale
noise = torch.randn(1, G.temporal_emb.noise_channels, emb_len + G.temporal_emb.kernel_size - 1, generator=torch.Generator().manual_seed(1)).double()
emb = G.temporal_emb.blur(noise)
ws = G.compute_latent_ws(emb, T)
w0 = ws.pop(0)
temporal_input = G.w_to_temp_input(w0.permute(0, 2, 1).reshape(-1, w0.shape[1])).reshape(1, in_len, -1).permute(0, 2, 1)
video = G.synthesize_video(temporal_input, ws, T)
logits = D(video)
loss = F.softplus(-logits).mean()
loss.backward()
out = dict(
    video=video.detach().numpy(), logits=logits.detach().numpy(), loss=np.array(float(loss)),
    g_spatial_input=G.spatial_input.grad.numpy(), g_to_rgb_weight=G.to_rgb.weight.grad.numpy(),
    g_t0_bias_0=G.temporal_layers[0].bias_0.grad.numpy(), g_s3_weight_1=G.spatial_layers[3].weight_1.grad.numpy(),
    g_map_l1_bias=G.latent_mapping.layer_1.bias.grad.numpy(),
    d_b0_conv_vid_weight=D.blocks[0].conv_vid.weight.grad.numpy(), d_ep_linear_1_weight=D.epilogue.linear_1.weight.grad.numpy())
assert all(v.dtype == np.float64 for v in out.values())
np.savez_compressed(os.path.join(HERE, 'lres_models_f64.npz'), **out)
print('video', video.shape, video.dtype, 'loss', float(loss))
f32 = np.load(os.path.join(HERE, 'lres_models.npz'))
for k in out:
    if k in f32.files and k not in ('loss',):
        print(f'{k:24s} reference float32 vs float64: max |err| / max |truth| = {np.abs(f32[k] - out[k]).max() / np.abs(out[k]).max():.2e}')
