"""Seeded inputs shared by tests/golden/make_golden_modconv3d.py (run against the reference in the build container)
and tests/test_conv3d_frames.py, so that only the reference's OUTPUTS are stored."""

import torch

CASES = dict(
    # name: (N, Ci, Co, T, H, W, kt, kh, kw)
    k333=(2, 64, 64, 5, 6, 7, 3, 3, 3),
    k133=(2, 64, 128, 3, 5, 9, 1, 3, 3),
    k311=(1, 128, 64, 6, 3, 4, 3, 1, 1),
)


def inputs(name):
    """(x [N,Ci,T,H,W], weight [Co,Ci,kt,kh,kw], style [N,Ci,T], bias [Co], input gain scalar), float32 on CPU."""
    n, ci, co, t, h, w, kt, kh, kw = CASES[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    x = torch.randn(n, ci, t, h, w, generator=g)
    weight = torch.randn(co, ci, kt, kh, kw, generator=g)
    style = 1.0 + 0.5 * torch.randn(n, ci, t, generator=g)
    bias = 0.3 * torch.randn(co, generator=g)
    gain = torch.tensor(0.75)
    return x, weight, style, bias, gain
