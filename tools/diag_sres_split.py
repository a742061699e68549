"""How well do two float16 parts represent the operands of the sres generator's float32 layers in a real generator update (activations,
weights, and the gradients that arrive at those layers)? Relative L2 reconstruction error, share of the tensor's energy carried by elements
whose low part underflows, dynamic range. MEASUREMENT TOOL (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_DB, 'db')); os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_DB, 'cache'))
import torch, torch.nn.functional as F
from torch_utils.ops import modconv2d_layout as ml, conv2d_frames as c2
from torch_utils.ops.conv3d_frames import split_bf16x3
from lvg.train_sres import SuperResTrainer

torch.manual_seed(0)
tr = SuperResTrainer(device='cuda', compute_dtype=torch.float16, augment_real_sign_target=None, augment_p_init=0.0, in_augment_strength=0.0,
                     lr_cond_prob=1.0, overlap_grad_sync=False, with_ema=False)
lr = torch.rand(2, 3, tr.context_seq_length, 36, 64, device='cuda') * 2 - 1
orig = ml._split_parts
rows = []


def spy(t, mode):
    t64 = t.double()
    s = c2.pow2_scale(t)
    hi, lo = c2.split16(t.float() * s)
    rec = (hi.double() + lo.double()) / s.double()
    b = split_bf16x3(t)
    rec3 = b[0].double() + b[1].double() + b[2].double()
    a = t64.abs()
    small = (a * s.double()) < 0.125
    rows.append((tuple(t.shape), float((rec - t64).norm() / t64.norm()), float((rec3 - t64).norm() / t64.norm()),
                 float((t64[small] ** 2).sum() / (t64 ** 2).sum()), float(torch.log2(a.max() / a[a > 0].median()))))
    return orig(t, mode)


ml._split_parts = spy
tr.G.requires_grad_(True)
logits = tr.run_D(tr.crop_to_seq_length(lr), tr.G(lr))
F.softplus(-logits).mean().backward()
print(f'{"tensor":28s} {"f16x2 L2 err":>12s} {"bf16x3 L2 err":>13s} {"energy below 2^-13 max":>23s} {"log2(max/median)":>17s}')
for shape, e2, e3, frac, dr in rows:
    print(f'{str(shape):28s} {e2:12.1e} {e3:13.1e} {frac:23.1e} {dr:17.1f}')
