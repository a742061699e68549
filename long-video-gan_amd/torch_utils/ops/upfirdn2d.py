"""Pad / upsample / FIR-filter / downsample of 2-D planes (`torch_utils.ops.upfirdn2d`,
reference torch_utils/ops/upfirdn2d.py). GPU tensors run `lvg_upfirdn2d`
(csrc/upfirdn2d.hip): separable filters take ONE fused launch (rows then columns through LDS)
where the reference plugin is called twice (:241-245). Gradients of any order re-enter the
same op with up<->down, the filter flipped and the padding of :256-266."""

import numpy as np
import torch

from .. import custom_ops
from .. import misc
from . import conv2d_gradfix
from . import _hip

#----------------------------------------------------------------------------

_plugin = None

def _init():
    """Load liblvg_hip.so (the reference JIT-compiles its plugin here, upfirdn2d.py:23-33)."""
    global _plugin
    if _plugin is None:
        custom_ops.get_plugin(module_name='upfirdn2d_plugin')
        _plugin = _hip.lib()
    return True

def _xy(value, what='factor'):
    """int or [x, y] -> (x, y), both >= 1."""
    pair = (value, value) if isinstance(value, int) else tuple(value)
    if len(pair) != 2 or not all(isinstance(v, int) and v >= 1 for v in pair):
        raise AssertionError(f'{what} must be a positive int or an [x, y] pair of them, got {value!r}')
    return pair

def _pad4(padding):
    """int, [x, y] or [x0, x1, y0, y1] -> (x0, x1, y0, y1); entries may be negative (= crop)."""
    vals = [padding] * 2 if isinstance(padding, int) else list(padding)
    assert all(isinstance(v, int) for v in vals), f'padding entries must be ints, got {padding!r}'
    if len(vals) == 2:
        vals = [vals[0], vals[0], vals[1], vals[1]]
    assert len(vals) == 4, 'padding must have 1, 2 or 4 entries'
    return tuple(vals)

def _filter_wh(f):
    """(width, height) of a filter tensor; None is the 1x1 identity. A 1-D filter reports (taps, taps'),
    taps' = taps as stored in dim 0 -- callers that use it on both axes set height = width themselves."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and 1 <= f.ndim <= 2, 'filter must be a 1-D or 2-D tensor'
    with misc.suppress_tracer_warnings():
        width, height = int(f.shape[-1]), int(f.shape[0])
    assert width >= 1 and height >= 1
    return width, height

# names the reference module exposes (model code and pickles may reach for them)
_parse_scaling, _parse_padding, _get_filter_size = _xy, _pad4, _filter_wh

#----------------------------------------------------------------------------

def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    r"""Build the float32 filter tensor `upfirdn2d()` expects.

    `f` may be a tensor / array / list of shape [h, w] (non-separable), [taps] (separable),
    a scalar (impulse) or None (identity). 1-D inputs with fewer than 8 taps are expanded to
    their outer product unless `separable=True`. `normalize` scales to unit DC gain;
    `gain` is applied as gain**(ndim/2) so that a separable filter used on both axes yields it once."""
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)

#----------------------------------------------------------------------------

def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    r"""For each channel plane of `x` [N, C, H, W]:
    1. insert `up-1` zeros after every sample, 2. pad (negative = crop) by `padding` taken in
    up-sampled pixels, 3. convolve with `f` (correlate if `flip_filter`), valid region only,
    4. keep every `down`-th sample, times `gain`.

    `f`: float32 [fh, fw], [taps] (separable, used on both axes) or None. `up`/`down`: int or
    [x, y]. `padding`: int, [x, y] or [x0, x1, y0, y1]. Differentiable to any order in `x`."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _upfirdn2d_cuda(up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain).apply(x, f)
    return _upfirdn2d_ref(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)

#----------------------------------------------------------------------------

@misc.profiled_function
def _upfirdn2d_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """The op spelled out with stock PyTorch ops (CPU tensors and impl='ref'), one step per line of the
    definition in `upfirdn2d()`: zero-stuff, pad / crop, correlate each plane with the taps, decimate.
    Like the reference's definition, `gain` goes into the taps (as gain ** (ndim / 2) per 1-D pass) before
    they are cast to x's dtype."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    taps = torch.ones([1, 1], dtype=torch.float32, device=x.device) if f is None else f
    assert isinstance(taps, torch.Tensor) and 1 <= taps.ndim <= 2 and taps.dtype == torch.float32 and not taps.requires_grad
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = _xy(up, 'up'), _xy(down, 'down'), _pad4(padding)
    n, c, h, w = x.shape
    assert w * ux + px0 + px1 >= taps.shape[-1] and h * uy + py0 + py1 >= taps.shape[0], 'padded input smaller than the filter'

    # 1. one sample every (uy, ux) grid points, zeros in between
    if ux > 1 or uy > 1:
        grid = x.new_zeros([n, c, h * uy, w * ux])
        grid[:, :, ::uy, ::ux] = x
        x = grid
    # 2. constant-mode padding takes negative widths as cropping
    if px0 or px1 or py0 or py1:
        x = torch.nn.functional.pad(x, [px0, px1, py0, py1])
    # 3. per-plane correlation = grouped convolution with one (shared) kernel per channel; a true convolution
    #    unless flip_filter, hence the reversal
    taps = taps * (gain ** (taps.ndim / 2))
    taps = (taps if flip_filter else taps.flip(list(range(taps.ndim)))).to(x.dtype)
    if taps.ndim == 2:
        x = conv2d_gradfix.conv2d(input=x, weight=taps.expand(c, 1, -1, -1), groups=c)
    else:
        x = conv2d_gradfix.conv2d(input=x, weight=taps.reshape(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)      # along x
        x = conv2d_gradfix.conv2d(input=x, weight=taps.reshape(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)      # along y
    # 4. decimate
    return x[:, :, ::dy, ::dx]

#----------------------------------------------------------------------------
# HIP path.

def _launch(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain):
    """One lvg_upfirdn2d launch. `f`: None, 1-D (separable, both axes) or 2-D float32."""
    assert x.ndim == 4
    fw, fh = _get_filter_size(f)
    if f is not None:
        assert f.dtype == torch.float32 and f.device == x.device, 'f must be float32 on the same device as x'
        if f.ndim == 1:
            fh = fw
    n, c, ih, iw = x.shape
    assert x.numel() > 0, 'x has zero size'
    ow = (iw * upx + padx0 + padx1 - fw + downx) // downx
    oh = (ih * upy + pady0 + pady1 - fh + downy) // downy
    assert ow >= 1 and oh >= 1, 'output must be at least 1x1'
    channels_last = (c > 1 and x.stride(1) == 1)
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    f2d = fx = fy = None
    fsx = fsy = 0
    if f is None:
        pass
    elif f.ndim == 1:
        f = f.contiguous()
        fx = fy = f.data_ptr()
    elif fh == 1 or fw == 1:
        # [1, k] / [k, 1] filters (the models' time-axis resamplers) are separable by shape.
        flat = f.reshape(-1).contiguous()
        fx = flat.data_ptr() if fw > 1 else None
        fy = flat.data_ptr() if fh > 1 else None
        if fw == 1 and fh == 1:
            f2d, fsx, fsy = flat.data_ptr(), 1, 1
    else:
        f2d, fsx, fsy = f.data_ptr(), f.stride(1), f.stride(0)
    with torch.cuda.device(x.device):
        rc = _hip.lib().lvg_upfirdn2d(
            x.data_ptr(), y.data_ptr(), f2d, fx, fy,
            _hip.shape4(x), _hip.stride4(x), _hip.shape4(y), _hip.stride4(y),
            fw, fh, fsx, fsy, upx, upy, downx, downy, padx0, pady0,
            int(bool(flip_filter)), float(gain), _hip.dtype_code(x.dtype), _hip.stream(x.device))
    _hip.check(rc, 'upfirdn2d')
    return y

_upfirdn2d_cuda_cache = dict()

def _upfirdn2d_cuda(up=1, down=1, padding=0, flip_filter=False, gain=1):
    """autograd.Function class for one parameter combination, cached."""
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    key = (upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)
    cached = _upfirdn2d_cuda_cache.get(key)
    if cached is not None:
        return cached

    class Upfirdn2dCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, f): # pylint: disable=arguments-differ
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2])
            y = _launch(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)
            ctx.f = f
            ctx.x_shape = x.shape
            return y

        @staticmethod
        def backward(ctx, dy): # pylint: disable=arguments-differ
            # The adjoint of "zero-stuff by up, pad, correlate with f, decimate by down" is the same op with the
            # factors swapped and the filter reversed; the paddings below make its output exactly x-sized:
            # leading pad = taps - 1 - pad0 (the adjoint of a valid correlation is a full one, minus the forward
            # pad), trailing pad = whatever is left to reach in * up samples before the new decimation.
            assert not ctx.needs_input_grad[1], 'the filter is a constant'
            if not ctx.needs_input_grad[0]:
                return None, None
            f = ctx.f
            fw, fh = _filter_wh(f)
            if f is not None and f.ndim == 1:
                fh = fw
            ih, iw = ctx.x_shape[2:]
            oh, ow = dy.shape[2:]
            lead_x, lead_y = fw - 1 - padx0, fh - 1 - pady0
            trail_x = iw * upx - (ow * downx - padx0) - (upx - 1)
            trail_y = ih * upy - (oh * downy - pady0) - (upy - 1)
            adjoint = _upfirdn2d_cuda(up=[downx, downy], down=[upx, upy], padding=[lead_x, trail_x, lead_y, trail_y],
                                      flip_filter=(not flip_filter), gain=gain)
            return adjoint.apply(dy, f), None

    _upfirdn2d_cuda_cache[key] = Upfirdn2dCuda
    return Upfirdn2dCuda

#----------------------------------------------------------------------------

def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    r"""FIR-filter planes; zero padding chosen so the output has the input's size (plus the
    user `padding`, negative = crop)."""
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)

def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    r"""Upsample planes by `up` (int or [x, y]); padding chosen so the output is exactly
    `up` times the input; gain is multiplied by upx*upy to keep signal magnitude."""
    upx, upy = _parse_scaling(up)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2, pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)

def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    r"""Downsample planes by `down` (int or [x, y]); padding chosen so the output is
    exactly 1/`down` of the input."""
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2, pady0 + (fh - downy + 1) // 2, pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)

#----------------------------------------------------------------------------
