"""Configurations shared by tests/golden/make_golden_sres.py (reference side) and
tests/test_sres_models.py (this repo's side)."""

import torch

# All 15 layers of the alias-free synthesis schedule at reduced width: 64x36 frames from 16x9.
SMALL_G = dict(z_dim=32, w_dim=48, img_width=64, img_height=36, img_channels=3, cond_width=16, cond_height=9,
               cond_context=1, channel_base=1024, channel_max=24, num_fp16_res=2)
SMALL_D = dict(seq_length=2, lr_height=9, lr_width=16, hr_height=36, hr_width=64, channels_base=1024, channels_max=32,
               num_fp16_res=0)   # the reference discriminator would run float16 even on CPU
# BASELINE.json configs[3]: 8-frame 144x256 segments from 36x64, temporal context 4.
FULL_G = dict(z_dim=512, w_dim=512, img_width=256, img_height=144, img_channels=3, cond_width=64, cond_height=36,
              cond_context=4, margin_size=10, fourfeats=False, num_fp16_res=4)


def small_inputs():
    g = torch.Generator().manual_seed(7)
    z = torch.randn(2, SMALL_G['z_dim'], generator=g)
    frames = SMALL_D['seq_length'] + 2 * SMALL_G['cond_context']
    lr_video = torch.randn(2, 3, frames, SMALL_G['cond_height'], SMALL_G['cond_width'], generator=g).clamp(-1, 1)
    return z, lr_video


def video_ramp(video: torch.Tensor) -> torch.Tensor:
    """Fixed linear functional on the generated frames (second loss term of the golden run)."""
    return torch.linspace(-1, 1, video.numel(), device=video.device, dtype=video.dtype).reshape(video.shape)
