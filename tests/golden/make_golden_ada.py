"""Golden vectors for the video ADA pipeline, produced by the REFERENCE AugmentPipe on CPU
(model/ada_augment.py; its ops take the impl='ref' path there). Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ada.py [/root/reference]

Deterministic cases use `debug_percentile` (every random draw replaced by a quantile); the seeded
cases pin the ORDER in which random numbers are consumed (torch CPU generator, manual_seed)."""

import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

import torch  # noqa: E402
from model import ada_augment  # noqa: E402
from helpers.ada_cfg import TRAIN_SRES_KW, IN_AUGMENT_KW, EXTRA_KW, sample_video  # noqa: E402

assert os.path.realpath(ada_augment.__file__).startswith(os.path.realpath(REF))
torch.set_num_threads(4)
out = {}
video = sample_video()
out['video'] = video.numpy()

for tag, kw in (('train', TRAIN_SRES_KW), ('in', IN_AUGMENT_KW), ('extra', EXTRA_KW)):
    pipe = ada_augment.AugmentPipe(**kw)
    out[f'{tag}_Hz_geom'] = pipe.Hz_geom.numpy()
    out[f'{tag}_Hz_fbank'] = pipe.Hz_fbank.numpy()
    for q in (0.2, 0.5, 0.85):
        torch.manual_seed(7)                       # additive noise stays random under debug_percentile
        out[f'{tag}_q{int(q * 100)}'] = pipe(video, debug_percentile=q).numpy()
    for p in (1.0, 0.4):
        pipe.p.fill_(p)
        torch.manual_seed(11)
        out[f'{tag}_seed11_p{int(p * 10)}'] = pipe(video).numpy()

# gradient w.r.t. the input through upfirdn2d / grid_sample (first order)
pipe = ada_augment.AugmentPipe(**TRAIN_SRES_KW)
v = video.clone().requires_grad_(True)
torch.manual_seed(7)
y = pipe(v, debug_percentile=0.7)
(y * torch.linspace(-1, 1, y.numel()).reshape(y.shape)).sum().backward()
out['train_q70'] = y.detach().numpy()
out['train_q70_grad'] = v.grad.numpy()

pipe.p.fill_(0.6)
torch.manual_seed(3)
long_video = sample_video(frames=20, height=6, width=8)
out['temporal_seed3'] = pipe.random_temporal_filter(long_video).numpy()

# image-space filter bank: the reference only type-checks for single-frame clips (T = 1)
still = sample_video(frames=1, height=48, width=48)[:2]
fpipe = ada_augment.AugmentPipe(imgfilter=1, imgfilter_bands=[1, 1, 0.5, 1])
out['filter_q80'] = fpipe(still, debug_percentile=0.8).numpy()
torch.manual_seed(5)
out['filter_seed5'] = fpipe(still).numpy()

np.savez_compressed(os.path.join(HERE, 'ada_augment.npz'), **out)
print({k: v.shape for k, v in out.items()})
