"""Deterministic, implementation-independent parameter fill keyed by state_dict NAME, so that the
reference networks (in the build container) and this repo's networks (anywhere) hold identical
weights without shipping 500 MB of tensors. Used by tests/golden/make_golden_models.py and by the
model parity tests."""

import zlib

import torch

_KEEP = ('filter', 'blur_filters', 'output_scale')   # analytic buffers: must already agree


def fill_named(module: torch.nn.Module) -> None:
    tensors = dict(module.named_parameters())
    tensors.update(dict(module.named_buffers()))
    with torch.no_grad():
        for name in sorted(tensors):
            t = tensors[name]
            leaf = name.split('.')[-1]
            if any(leaf.endswith(k) for k in _KEEP):
                continue
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if leaf == 'magnitude_ema':
                v = 0.5 + torch.rand(t.shape, generator=g)
            elif 'bias' in leaf:
                v = 0.1 * torch.randn(t.shape, generator=g)
                if '.affine' in '.' + name:
                    v = v + 1.0
            else:
                v = torch.randn(t.shape, generator=g)
            t.copy_(v.to(t.dtype))


def analytic_buffers(module: torch.nn.Module):
    return {n: b for n, b in module.named_buffers() if any(n.split('.')[-1].endswith(k) for k in _KEEP)}
