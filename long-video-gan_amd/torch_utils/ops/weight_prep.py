"""Weight side of a modulated convolution in one device pass per direction (csrc/weight_prep.hip):

    w = w / max|w| per output channel        (only when demodulating; reference model/generator_lres.py:98)
    w = w * scale                            (scale = 1 / sqrt(fan_in), :102-103; or the layer's weight gain)
    w2[co, ci] = sum over the taps of w^2    (the weight half of the demodulation einsum, :107)
    w16 = w.to(compute dtype)                (:119), stored tap-major [kt, kh, kw, Co, Ci] -- the layout the hand-written
                                             convolution consumes -- and returned as the [Co, Ci, kt, kh, kw] VIEW of it,
                                             so `pack_weight` downstream is free and every other consumer sees the usual shape

The PyTorch spelling is eight tensor passes over the weight forward and about twice that backward (per layer, per step);
`weight_prep` is one launch each way (the backward couples all elements of an output channel through the max
normalisation and handles ties like `torch.amax`). CPU tensors take the tensor expressions (= the definition tested against)."""

import math

import torch

from . import _hip
from .modconv_epilogue import _init


def _ref(weight, scale, normalize, dtype, want_w2):
    w = weight
    if normalize:
        w = w / w.abs().amax(dim=tuple(range(1, w.ndim)), keepdim=True)
    w = w * scale
    w2 = w.square().sum(dim=tuple(range(2, w.ndim))) if want_w2 else None
    return w.to(dtype), w2


def _tap_strides(g):
    """(co, ci, tap) element strides of a [Co, Ci, *taps] tensor whose tap dims collapse to one stride, else None."""
    if g.ndim == 2:
        return g.stride(0), g.stride(1), 0
    st, sz = g.stride()[2:], g.shape[2:]
    for i in range(len(sz) - 1):
        if sz[i] > 1 and st[i] != st[i + 1] * sz[i + 1]:
            return None
    return g.stride(0), g.stride(1), st[-1]


class _WeightPrep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, scale, normalize, dtype, want_w2):
        w = weight.contiguous()
        co, ci = w.shape[:2]
        taps_shape = tuple(w.shape[2:])
        taps = max(1, math.prod(taps_shape))
        wp = torch.empty(taps_shape + (co, ci), dtype=dtype, device=w.device)
        w2 = torch.empty((co, ci), dtype=torch.float32, device=w.device) if want_w2 else None
        amax = torch.empty(co, dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            rc = _hip.lib().lvg_weight_prep(w.data_ptr(), wp.data_ptr(), _hip.ptr(w2), amax.data_ptr(), co, ci, taps, scale, int(normalize),
                                            _hip.dtype_code(dtype), _hip.stream(w.device))
        _hip.check(rc, 'weight_prep')
        ctx.save_for_backward(w, amax)
        ctx.cfg = (scale, bool(normalize), dtype, taps)
        nd = len(taps_shape)
        view = wp.permute(nd, nd + 1, *range(nd))                      # [Co, Ci, *taps], memory stays tap-major
        if want_w2:
            return view, w2
        ctx.mark_non_differentiable()
        return view, None

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_w, g_w2):
        w, amax = ctx.saved_tensors
        scale, normalize, dtype, taps = ctx.cfg
        co, ci = w.shape[:2]
        if g_w is None:
            g_w = torch.zeros(w.shape, dtype=dtype, device=w.device)
        g_w = g_w.to(dtype)
        strides = _tap_strides(g_w)
        if strides is None:
            g_w = g_w.contiguous()
            strides = _tap_strides(g_w)
        g_w2 = g_w2.contiguous().float() if g_w2 is not None else None
        dw = torch.empty_like(w)
        arr = (_hip._i64 * 3)(*strides)
        with torch.cuda.device(w.device):
            rc = _hip.lib().lvg_weight_prep_backward(w.data_ptr(), amax.data_ptr(), g_w.data_ptr(), arr, _hip.ptr(g_w2), dw.data_ptr(),
                                                     co, ci, taps, scale, int(normalize), _hip.dtype_code(dtype), _hip.stream(w.device))
        _hip.check(rc, 'weight_prep_backward')
        return dw, None, None, None, None


def supported(weight, dtype):
    if weight.device.type != 'cuda' or weight.dtype != torch.float32 or dtype not in (torch.float16, torch.bfloat16) or weight.ndim < 2:
        return False
    ci, taps = weight.shape[1], max(1, math.prod(weight.shape[2:]))
    return ci <= 1024 and ci * taps * 8 <= 150 * 1024 and _init()


def dgrad_pack(w16):
    """Weight of the data-gradient convolution (taps mirrored, channel roles swapped) of a prepared weight, already in the layout the
    hand-written convolution consumes: [*taps, Ci, Co] contiguous. `w16` must be the tap-major view `weight_prep` returns."""
    co, ci = w16.shape[:2]
    taps_shape = tuple(w16.shape[2:])
    nd = len(taps_shape)
    wp = w16.permute(*range(2, 2 + nd), 0, 1)
    assert wp.is_contiguous(), 'dgrad_pack: tap-major weight expected'
    wt = torch.empty(taps_shape + (ci, co), dtype=w16.dtype, device=w16.device)
    with torch.cuda.device(w16.device):
        rc = _hip.lib().lvg_weight_dgrad_pack(wp.data_ptr(), wt.data_ptr(), max(1, math.prod(taps_shape)), co, ci, _hip.stream(w16.device))
    _hip.check(rc, 'weight_dgrad_pack')
    return wt


def weight_prep(weight, scale, normalize, dtype, want_w2=True):
    """-> (weight in `dtype` [Co, Ci, *taps] (tap-major memory on the GPU path), w2 [Co, Ci] float32 or None).

    weight [Co, Ci, *taps] float32; `normalize`: divide each output channel by its max |w| first.
    While gradients are recorded, the returned 5-D weight carries `_lvg_dgrad`: the same weight packed for the data-gradient
    convolution (one small launch here, next to the other weight-side work, instead of flip + transpose + copy in the backward pass)."""
    if supported(weight, dtype):
        w16, w2 = _WeightPrep.apply(weight, float(scale), bool(normalize), dtype, bool(want_w2))
        if torch.is_grad_enabled() and w16.ndim == 5 and w16.shape[0] % 64 == 0 and w16.shape[1] % 64 == 0:
            w16._lvg_dgrad = dgrad_pack(w16.detach())
        return w16, w2
    return _ref(weight, scale, normalize, dtype, want_w2)
