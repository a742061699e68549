"""Configurations shared by tests/golden/make_golden_ada.py (reference side) and tests/test_ada_augment.py."""

import torch

# train_sres.py:360-373 (discriminator-side ADA)
TRAIN_SRES_KW = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1,
                     brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)
# video_gan_sres.py:114-126 with in_augment_strength = 8 (conditioning-side augmentation)
IN_AUGMENT_KW = dict(scale=1, scale_std=0.08, rotate=1, rotate_max=0.016, aniso=1, aniso_std=0.08,
                     xfrac=1, xfrac_std=0.016, noise=1, noise_std=0.08)
# the transforms the shipped configurations leave off (noise / cutout only: imgfilter needs T = 1 in the reference)
EXTRA_KW = dict(xint=1, noise=1, cutout=1, cutout_size=0.4)


def sample_video(frames=4, height=18, width=32):
    g = torch.Generator().manual_seed(21)
    return torch.rand(3, 3, frames, height, width, generator=g) * 2 - 1
