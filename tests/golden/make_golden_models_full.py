"""Golden vectors at the BASELINE.json configs[1] size and for the R1 penalty, produced by the REFERENCE
low-resolution generator / discriminator on CPU (impl='ref' op path). Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_models_full.py [/root/reference]

* T = 128 generator forward (the frames/sec headline shape) + discriminator logits and loss; the video
  is stored as float16 (values in [-0.5, 0.5]: 1.2e-4 absolute storage error against the 1e-3 gate).
* R1: `video_gan_lres.py:184-194` on a T = 16 clip -- grad of logits.sum() w.r.t. the real video with
  create_graph, penalty = sum of squares, loss = penalty * gamma / 2, backward -> parameter gradients.

Weights are NOT stored: both sides fill them with tests/helpers/named_fill.py."""

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
out = {}

# ---- T = 128 forward ---------------------------------------------------------------------------------
T = 128
G = generator_lres.VideoGenerator()
D = discriminator_lres.VideoDiscriminator(seq_length=T, max_edge=64)
fill_named(G)
fill_named(D)
with torch.no_grad():
    in_len = G.compute_seq_lengths(T)[0]
    emb_len = in_len * G.total_temporal_scale
    noise = torch.randn(1, G.temporal_emb.noise_channels, emb_len + G.temporal_emb.kernel_size - 1, generator=torch.Generator().manual_seed(2))
    emb = G.temporal_emb.blur(noise)
    ws = G.compute_latent_ws(emb, T)
    w0 = ws.pop(0)
    temporal_input = G.w_to_temp_input(w0.permute(0, 2, 1).reshape(-1, w0.shape[1])).reshape(1, in_len, -1).permute(0, 2, 1)
    video = G.synthesize_video(temporal_input, ws, T)
    logits = D(video)
    loss = F.softplus(-logits).mean()
out.update(t128_noise_seed=np.array(2), t128_noise_shape=np.array(noise.shape), t128_noise_sum=np.array(float(noise.double().sum())),
           t128_video=video.numpy().astype(np.float16), t128_video_absmean=np.array(float(video.abs().mean())),
           t128_logits=logits.numpy(), t128_loss=np.array(float(loss)))
print('T=128 video', tuple(video.shape), 'range', float(video.min()), float(video.max()), 'logits', logits.flatten().tolist())

# ---- R1 penalty at T = 16 ------------------------------------------------------------------------------
T = 16
D = discriminator_lres.VideoDiscriminator(seq_length=T, max_edge=64)
fill_named(D)
D.requires_grad_(True)
real = (torch.rand(2, 3, T, 36, 64, generator=torch.Generator().manual_seed(5)) * 2 - 1).requires_grad_(True)
logits = D(real)
(r1_grad,) = torch.autograd.grad(outputs=[logits.sum()], inputs=[real], create_graph=True)
penalty = r1_grad.square().sum(dim=(1, 2, 3, 4))
gamma = 1.0
(penalty * (gamma / 2)).mean().backward()
named = dict(D.named_parameters())
keys = ['blocks.0.conv_vid.weight', 'blocks.0.conv_0.weight', 'blocks.1.conv_0.weight', 'blocks.1.conv_1.weight', 'blocks.2.conv_skip.weight',
        'blocks.3.conv_1._bias', 'epilogue.conv1d_0.weight', 'epilogue.linear_0.weight', 'epilogue.linear_1.weight']
out.update(r1_real_seed=np.array(5), r1_logits=logits.detach().numpy(), r1_input_grad=r1_grad.detach().numpy().astype(np.float32),
           r1_penalty=penalty.detach().numpy())
for k in keys:
    gk = named[k].grad
    assert gk is not None, k
    a = gk.numpy()
    # large tensors: keep a deterministic strided sample plus the norm
    flat = a.reshape(-1)
    out['r1_g_' + k.replace('.', '_') + '_norm'] = np.array(float(np.sqrt((flat.astype(np.float64) ** 2).sum())))
    out['r1_g_' + k.replace('.', '_') + '_sample'] = flat[:: max(1, flat.size // 4096)][:4096].copy()
none_grad = sorted(k for k, p in named.items() if p.grad is None)
out['r1_params_without_grad'] = np.array(','.join(none_grad))
print('R1 logits', logits.flatten().tolist(), 'penalty', penalty.tolist(), 'no grad:', none_grad)

np.savez_compressed(os.path.join(HERE, 'lres_models_full.npz'), **out)
print({k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items()})
