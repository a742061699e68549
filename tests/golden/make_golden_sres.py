"""Golden vectors for the super-resolution networks, produced by the REFERENCE model code on CPU
(its ops take the impl='ref' path there). Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sres.py [/root/reference]

Two fixtures in one file:
  * a reduced-width network pair (all 15 synthesis layers, 64x36 output) run forward + backward;
  * the layer schedule (sizes, rates, factors, paddings, filter checksums) of the full 256x144
    configuration BASELINE.json configs[3] names -- construction only, no forward.
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
from model import generator_sres, discriminator_sres  # noqa: E402
from helpers.named_fill import fill_named, analytic_buffers  # noqa: E402
from helpers.sres_cfg import SMALL_G, SMALL_D, FULL_G, small_inputs, video_ramp  # noqa: E402

assert os.path.realpath(generator_sres.__file__).startswith(os.path.realpath(REF))
torch.set_num_threads(8)

G = generator_sres.Generator(**SMALL_G)
D = discriminator_sres.VideoDiscriminator(**SMALL_D)
fill_named(G)
fill_named(D)
G.requires_grad_(True)
D.requires_grad_(True)

z, lr_video = small_inputs()
ctx = SMALL_G['cond_context']
conds = G.prep_cond(lr_video)
hr_video = G(z, lr_video)
logits = D(lr_video[:, :, ctx:-ctx], hr_video)
# second term: a direct, well-conditioned path into the generator (through D alone the early-layer
# gradients are ~1e-6 and dominated by float32 cancellation noise)
loss = F.softplus(-logits).mean() + (hr_video * video_ramp(hr_video)).sum()
loss.backward()

names = G.synthesis.layer_names
first, mid, last = (getattr(G.synthesis, names[i]) for i in (0, 7, -1))
out = dict(
    video=hr_video.detach().numpy(), logits=logits.detach().numpy(), loss=np.array(float(loss)),
    cond_rms=np.array([float(c.square().mean().sqrt()) for c in conds]),
    cond_shapes=np.array([list(c.shape) for c in conds]),
    cond_3=conds[3].numpy(), cond_12=conds[12].numpy(),
    g_first_weight=first.weight.grad.numpy(), g_mid_weight=mid.weight.grad.numpy(), g_mid_bias=mid.bias.grad.numpy(),
    g_mid_affine_weight=mid.affine.weight.grad.numpy(), g_last_weight=last.weight.grad.numpy(),
    g_map_fc0_weight=G.mapping.fc0.weight.grad.numpy(),
    d_b64_fromrgb_weight=D.b64.fromrgb.weight.grad.numpy(), d_b16_conv1_weight=D.b16.conv1.weight.grad.numpy(),
    d_b16_skip_weight=D.b16.skip.weight.grad.numpy(), d_b4_fc_bias=D.b4.fc.bias.grad.numpy(),
)
for prefix, net in (('G', G), ('D', D)):
    for name, buf in analytic_buffers(net).items():
        out[f'buf_{prefix}_{name}'] = np.array([float(buf.double().sum()), float(buf.double().abs().sum()), float(buf.numel())])


def schedule(net):
    rows = []
    for name in net.synthesis.layer_names:
        l = getattr(net.synthesis, name)
        rows.append([l.in_channels, l.out_channels, *map(int, l.in_size), *map(int, l.out_size), l.in_sampling_rate,
                     l.out_sampling_rate, l.up_factor, l.down_factor, l.up_taps, l.down_taps, *l.padding, int(l.use_fp16)])
    return np.array(rows, dtype=np.int64)


out['small_schedule'] = schedule(G)
full = generator_sres.Generator(**FULL_G)
out['full_schedule'] = schedule(full)
out['full_names'] = np.array(full.synthesis.layer_names)
out['full_keys'] = np.array(sorted(full.state_dict().keys()))
out['full_resample_scales'] = np.array([getattr(r, 'scale', 1) * (-1 if 'Down' in type(r).__name__ else 1) for r in full.resamples])
for name, buf in analytic_buffers(full).items():
    out[f'buf_F_{name}'] = np.array([float(buf.double().sum()), float(buf.double().abs().sum()), float(buf.numel())])
fullD = discriminator_sres.VideoDiscriminator(seq_length=8, lr_height=36, lr_width=64, hr_height=144, hr_width=256)
out['full_d_keys'] = np.array(sorted(fullD.state_dict().keys()))
out['full_d_shapes'] = np.array([int(np.prod(v.shape)) for k, v in sorted(fullD.state_dict().items())])
out['small_keys'] = np.array(sorted(G.state_dict().keys()))
out['small_d_keys'] = np.array(sorted(D.state_dict().keys()))

np.savez_compressed(os.path.join(HERE, 'sres_models.npz'), **out)
print('video', hr_video.shape, float(hr_video.abs().mean()), 'logits', logits.flatten().tolist(), 'loss', float(loss))
print(out['full_schedule'])
