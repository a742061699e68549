"""Golden vectors for the trainer-side augmentation glue of the low-resolution GAN, produced by the REFERENCE code: the temporal stretch +
pad / crop inside VideoGAN.run_D (model/video_gan_lres.py:236-265) and DiffAugment (model/diff_augment.py), run on CPU with seeded generators.
The trainer object is NOT constructed (it builds the networks and the optimizers): `run_D` is called on a bare instance that carries the
attributes the method reads, with the discriminator replaced by the identity, so the returned "logits" are the augmented clip.
Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trainer_glue.py [/root/reference]"""

import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import importlib  # noqa: E402
import types  # noqa: E402

import torch  # noqa: E402

# the reference's utils.py imports I/O packages this image does not have (imageio, wandb, ...): empty stand-ins -- none of them is touched by run_D
for _ in range(16):
    try:
        video_gan_lres = importlib.import_module('model.video_gan_lres')
        break
    except ModuleNotFoundError as err:
        assert err.name and not err.name.startswith(('model', 'torch_utils', 'dnnlib', 'utils')), err
        sys.modules[err.name] = types.ModuleType(err.name)

assert os.path.realpath(video_gan_lres.__file__).startswith(os.path.realpath(REF))
cls = [c for c in vars(video_gan_lres).values() if isinstance(c, type) and hasattr(c, 'run_D') and c.__module__ == video_gan_lres.__name__][0]

out = {}
cases = [('a1_t16', 1.0, 16, 16), ('a05_t16', 0.5, 16, 16), ('a2_t24', 2.0, 24, 24)]
for name, amount, frames, seq in cases:
    gan = object.__new__(cls)
    gan.channels, gan.seq_length, gan.height, gan.width = 3, seq, 6, 7
    gan.diffaug_policy, gan.temp_scale_augment = '', amount
    gan.D = lambda v: v
    video = torch.randn(5, 3, frames, 6, 7, generator=torch.Generator().manual_seed(4))
    torch.manual_seed(11)
    got = cls.run_D(gan, video)
    out[name + '_out'] = got.numpy().astype(np.float32)
    out[name + '_next_rand'] = torch.rand(3).numpy()              # what the CPU generator yields next: pins the NUMBER of draws
    out[name + '_spec'] = np.asarray(repr(dict(amount=amount, frames=frames, seq_length=seq, video_seed=4, seed=11, shape=(5, 3, frames, 6, 7))))
# DiffAugment (model/diff_augment.py): each policy alone and the trainer's default chain, seeded CPU generator
diff_augment = importlib.import_module('model.diff_augment')
assert os.path.realpath(diff_augment.__file__).startswith(os.path.realpath(REF))
clip = torch.randn(4, 3, 5, 12, 20, generator=torch.Generator().manual_seed(8))
for policy in ('color', 'translation', 'cutout', 'color,translation,cutout'):
    torch.manual_seed(21)
    got = diff_augment.DiffAugment(clip, policy)
    key = 'diffaug_' + policy.replace(',', '_')
    out[key + '_out'] = got.numpy().astype(np.float32)
    out[key + '_next_rand'] = torch.rand(3).numpy()
out['diffaug_spec'] = np.asarray(repr(dict(clip_seed=8, seed=21, shape=(4, 3, 5, 12, 20))))
# generator EMA (video_gan_lres.py:207-214): the reference's update_G_ema on a bare instance whose G / G_ema are two float64 one-parameter-one-buffer
# modules; the weight it applied at a step is read back from what the lerp did to them
class _Tiny(torch.nn.Module):
    def __init__(self, v):
        super().__init__()
        self.w = torch.nn.Parameter(torch.full((3,), v, dtype=torch.float64))
        self.register_buffer('b', torch.full((2,), v * 2, dtype=torch.float64))


steps = [0, 1, 7, 100, 5000, 24999, 25000, 25001, 400000]
betas = []
for step in steps:
    gan = object.__new__(cls)
    gan.G_ema_beta, gan.G_ema_warmup_steps = 0.99985, 25000
    gan.G, gan.G_ema = _Tiny(1.0), _Tiny(0.0)
    cls.update_G_ema(gan, step)
    w = float(gan.G_ema.w[0])                         # = 1 - beta
    assert abs(float(gan.G_ema.b[0]) - 2 * w) < 1e-12  # buffers take the same step
    betas.append(1.0 - w)
out['ema_steps'] = np.asarray(steps, dtype=np.int64)
out['ema_betas'] = np.asarray(betas, dtype=np.float64)
# learning-rate warm-up (video_gan_lres.py:89-96) on stand-in optimizers
class _Opt:
    def __init__(self):
        self.param_groups = [dict(lr=None)]


lr_rows = []
for step in (0, 1, 49, 50, 51, 1000):
    gan = object.__new__(cls)
    gan.G_lrate, gan.D_lrate, gan.G_warmup_steps, gan.D_warmup_steps = 0.003, 0.002, 50, 0
    gan.G_opt, gan.D_opt = _Opt(), _Opt()
    cls.update_lrates(gan, step)
    lr_rows.append((step, gan.G_opt.param_groups[0]['lr'], gan.D_opt.param_groups[0]['lr']))
out['lrate_rows'] = np.asarray(lr_rows, dtype=np.float64)
# the step body itself (video_gan_lres.py:100-203): the reference's update_G / update_D / update_r1 / update_G_ema on the stand-in networks of
# tests/helpers/stub_nets.py (a one-rank gloo group serves utils.sync_grads), every random draw from the seeded default generator
sys.path.insert(0, os.path.dirname(HERE))
from helpers.stub_nets import StubG, StubD  # noqa: E402
import torch.distributed as dist  # noqa: E402

if not dist.is_initialized():
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29731', rank=0, world_size=1)
SEQ, BATCH = 8, 4
gan = object.__new__(cls)
gan.seq_length, gan.height, gan.width, gan.channels = SEQ, 6, 8, 3
gan.G_grad_accum, gan.D_grad_accum, gan.G_random_temp_translate, gan.G_magnitude_ema_beta = 2, 2, True, 0.999
gan.G_ema_beta, gan.G_ema_warmup_steps = 0.99985, 25000
gan.temp_scale_augment, gan.diffaug_policy, gan.r1_gamma = 1.0, 'color,translation,cutout', 10.0
gan.G, gan.G_ema, gan.D = StubG(), StubG(), StubD(SEQ)
for net in (gan.G, gan.G_ema, gan.D):
    net.requires_grad_(False)
gan.G_opt = torch.optim.Adam(gan.G.parameters(), lr=0.003, betas=(0.0, 0.99))
gan.D_opt = torch.optim.Adam(gan.D.parameters(), lr=0.002, betas=(0.0, 0.99))
real = torch.rand(BATCH, 3, SEQ, 6, 8, generator=torch.Generator().manual_seed(9)) * 2 - 1
torch.manual_seed(33)
for step in (0, 1):                                  # the loop of train_lres.py:216-230 with r1_interval = 2: R1 on step 0
    cls.update_G(gan, BATCH)
    cls.update_D(gan, real)
    if step % 2 == 0:
        cls.update_r1(gan, real, gain=2)
    cls.update_G_ema(gan, step)
for net_name in ('G', 'D', 'G_ema'):
    for n, t in list(getattr(gan, net_name).named_parameters()) + list(getattr(gan, net_name).named_buffers()):
        out[f'step_{net_name}_{n}'] = t.detach().numpy().astype(np.float64)
out['step_next_rand'] = torch.rand(3).numpy()
out['step_spec'] = np.asarray(repr(dict(seq_length=SEQ, batch=BATCH, real_seed=9, seed=33, steps=2, r1_interval=2)))

# the super-resolution step body (video_gan_sres.py:150-276): update_G / update_D / update_r1 / update_ada / update_G_ema of the reference's
# SuperResVideoGAN with the REFERENCE's AugmentPipe (discriminator-side ADA at p = 0.3 adapted every 2 steps, conditioning-side jitter) on the
# stand-in networks, the loop of train_sres.py:241-264 for three iterations (R1 on steps 0 and 2, ADA on steps 0 and 2)
from helpers.stub_nets import StubSresG, StubSresD  # noqa: E402
from helpers.ada_cfg import TRAIN_SRES_KW  # noqa: E402
video_gan_sres = importlib.import_module('model.video_gan_sres')
assert os.path.realpath(video_gan_sres.__file__).startswith(os.path.realpath(REF))
scls = video_gan_sres.SuperResVideoGAN
training_stats = importlib.import_module('torch_utils.training_stats')
importlib.import_module('torch_utils.ops.conv2d_gradfix').enabled = True          # as train_sres.py:81-82: R1 differentiates twice through
importlib.import_module('torch_utils.ops.grid_sample_gradfix').enabled = True     # the resampling convolutions and ADA's grid_sample
SSEQ, CTX, SB = 2, 1, 4
sg = object.__new__(scls)
sg.seq_length, sg.temporal_context, sg.context_seq_length, sg.channels = SSEQ, CTX, SSEQ + 2 * CTX, 3
sg.lr_height, sg.lr_width, sg.hr_height, sg.hr_width = 9, 16, 36, 64
sg.G_grad_accum, sg.D_grad_accum, sg.G_magnitude_ema_beta = 2, 2, 0.999
sg.G_ema_beta, sg.G_ema_warmup_steps, sg.r1_gamma, sg.lr_cond_prob = 0.99985, 25000, 1.0, 0.5
sg.augment_p_max, sg.augment_p_update_rate, sg.augment_real_sign_target = 0.5, 0.01, 0.6
sg.G, sg.G_ema, sg.D = StubSresG(CTX), StubSresG(CTX), StubSresD(SSEQ)
for net in (sg.G, sg.G_ema, sg.D):
    net.requires_grad_(False)
sg.G_opt = torch.optim.Adam(sg.G.parameters(), lr=0.003, betas=(0.0, 0.99))
sg.D_opt = torch.optim.Adam(sg.D.parameters(), lr=0.002, betas=(0.0, 0.99))
sg.augment = video_gan_sres.AugmentPipe(**TRAIN_SRES_KW).requires_grad_(False).train()
sg.augment.p.fill_(0.3)
sg.real_sign_collector = training_stats.Collector(regex='loss/D_sign_real')
k = 8.0
sg.in_augment = video_gan_sres.AugmentPipe(scale=1, scale_std=0.01 * k, rotate=1, rotate_max=0.002 * k, aniso=1, aniso_std=0.01 * k,
                                           xfrac=1, xfrac_std=0.002 * k, noise=1, noise_std=0.01 * k).requires_grad_(False).train()
sg.in_augment.p.fill_(0.5)
gen = torch.Generator().manual_seed(13)
lr = torch.rand(SB, 3, SSEQ + 2 * CTX, 9, 16, generator=gen) * 2 - 1
hr = torch.rand(SB, 3, SSEQ, 36, 64, generator=gen) * 2 - 1
torch.manual_seed(44)
for step in range(3):
    scls.update_G(sg, lr)
    scls.update_D(sg, lr, lr, hr)
    if step % 2 == 0:
        scls.update_r1(sg, scls.crop_to_seq_length(sg, lr), hr, gain=2)
    if step % 2 == 0:
        scls.update_ada(sg, gain=2)
    scls.update_G_ema(sg, step)
for net_name in ('G', 'D', 'G_ema'):
    for n, t in list(getattr(sg, net_name).named_parameters()) + list(getattr(sg, net_name).named_buffers()):
        out[f'sres_{net_name}_{n}'] = t.detach().numpy().astype(np.float64)
out['sres_augment_p'] = np.asarray(float(sg.augment.p))
out['sres_next_rand'] = torch.rand(3).numpy()
out['sres_spec'] = np.asarray(repr(dict(seq_length=SSEQ, temporal_context=CTX, batch=SB, data_seed=13, seed=44, steps=3, r1_interval=2, ada_interval=2,
                                        augment_p_init=0.3, augment_p_update_rate=0.01, lr_cond_prob=0.5)))
dist.destroy_process_group()
np.savez_compressed(os.path.join(HERE, 'trainer_glue.npz'), **out)
print({k: (v.shape if v.ndim else str(v)) for k, v in out.items()})
