"""Seeded inputs shared by tests/golden/make_golden_modconv2d.py (run against the reference in the build container)
and tests/test_conv2d_frames.py, so that only the reference's OUTPUTS (and gradients) are stored."""

import torch

CASES = dict(
    # name: (N, Ci, Co, H, W, k, padding)
    p2=(2, 11, 7, 9, 21, 3, 2),          # the generator's layers: kernel 3, padding = kernel - 1 (model/generator_sres.py:331)
    p1=(3, 8, 6, 6, 19, 3, 1),           # 'same'
    p0=(1, 5, 9, 8, 20, 3, 0),           # 'valid'
)


def inputs(name):
    """(x [N,Ci,H,W], weight [Co,Ci,k,k], style [N,Ci], input gain scalar, dy [N,Co,H',W']), float32 on CPU."""
    n, ci, co, h, w, k, pad = CASES[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)) + 17)
    x = torch.randn(n, ci, h, w, generator=g)
    weight = torch.randn(co, ci, k, k, generator=g)
    style = 1.0 + 0.5 * torch.randn(n, ci, generator=g)
    gain = torch.tensor(0.8)
    dy = torch.randn(n, co, h + 2 * pad - k + 1, w + 2 * pad - k + 1, generator=g)
    return x, weight, style, gain, dy
