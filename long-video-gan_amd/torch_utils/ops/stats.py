"""Small statistics over activations that the reference spells as tensor expressions.

`mean_square(x)`: `x.detach().float().square().mean()` (model/generator_sres.py:278-286, model/generator_lres.py:298-312: the input-magnitude
statistic of every synthesis layer). As tensor expressions that is three passes over the activation (cast to float32, square, reduce: 18
bytes per 16-bit element, ~75 us per layer of the sres generator, 4.8 % of a train_sres iteration: profiles/r05_window_train_sres.csv);
`lvg_plane_sum_sq` reads the tensor once (2 bytes per element) and leaves per-chunk partial sums."""

import os

import torch

from . import _hip

_CHUNK = 8192           # elements per partial sum (one workgroup)
ENABLED = os.environ.get('LVG_MEAN_SQUARE_HIP', '1') == '1'


def mean_square(x: torch.Tensor) -> torch.Tensor:
    """float32 scalar tensor: mean of x**2 over all elements (detached). GPU tensors go through liblvg_hip.so (PluginUnavailable if it is missing)."""
    x = x.detach()
    dense = x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)) or (x.dim() == 5 and x.is_contiguous(memory_format=torch.channels_last_3d))
    if not (ENABLED and x.is_cuda and dense and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and x.numel() >= 4 * _CHUNK):
        return x.float().square().mean()
    n = x.numel()
    rows = n // _CHUNK
    part = torch.empty(rows, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _hip.lib().lvg_plane_sum_sq(x.data_ptr(), part.data_ptr(), rows, _CHUNK, _hip.dtype_code(x.dtype), _hip.stream(x.device))
    _hip.check(rc, 'plane_sum_sq')
    total = part.sum()
    if rows * _CHUNK < n:       # (the storage of a dense tensor is one run of numel elements: the tail through a flat view of it)
        tail = torch.as_strided(x, (n - rows * _CHUNK,), (1,), x.storage_offset() + rows * _CHUNK)
        total = total + tail.float().square().sum()
    return total / float(n)
