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


# ---- 2-D modulated convolution of the super-resolution generator (RMS normalisation; csrc/weight_prep.hip, lvg_weight_prep2d) ----------

def _ref2d(weight, scale):
    """(w', energy): w' = w * rsqrt(mean w^2 per output channel) * scale, energy[co, ci] = sum over the taps of w'^2
    (reference model/generator_sres.py:50-58, with the 1 / sqrt(fan_in) of the 16-bit layers folded in as `scale`)."""
    w = weight * weight.square().mean(dim=(1, 2, 3), keepdim=True).rsqrt()
    w = w * scale
    return w, w.square().sum(dim=(2, 3))


class Prepared2d:
    """Weight of one 3 x 3 modulated convolution prepared for the hand-written kernels: `weight` the float32 master [Co, Ci, 3, 3], `wp`
    [3, 3, co_pad, ci_pad] / `wt` [3, 3, ci_pad, co_pad] (or None) the normalised 16-bit weight for the forward / data-gradient
    convolution, `stat` [Co] the normalisation factor, `energy` [Co, Ci] float32 (differentiable w.r.t. `weight`), `scale`."""

    def __init__(self, weight, wp, wt, stat, energy, scale):
        self.weight, self.wp, self.wt, self.stat, self.energy, self.scale = weight, wp, wt, stat, energy, scale

    def tensors(self):
        return (self.wp, self.wt, self.stat, self.energy)

    def grad_from_conv(self, gw):
        """d weight through the convolution: gw = gradient of wp's elements, float32 [3, 3, co_pad, ci_pad] (lvg_conv2d_frames_wgrad)."""
        return _prep2d_backward(self.weight.detach(), self.stat, gw, None, self.scale, self.wp.shape[2], self.wp.shape[3])


def _prep2d_backward(w, stat, g, g_w2, scale, co_pad, ci_pad):
    co, ci = w.shape[:2]
    taps = w.shape[2] * w.shape[3]
    w = w.contiguous()
    dw = torch.empty_like(w)
    g = None if g is None else g.contiguous()
    g_w2 = None if g_w2 is None else g_w2.contiguous().float()
    assert g is None or (g.dtype == torch.float32 and g.shape == (w.shape[2], w.shape[3], co_pad, ci_pad))
    with torch.cuda.device(w.device):
        rc = _hip.lib().lvg_weight_prep2d_backward(w.data_ptr(), stat.data_ptr(), _hip.ptr(g), _hip.ptr(g_w2), dw.data_ptr(), co, ci, taps, co_pad, ci_pad,
                                                   scale, _hip.stream(w.device))
    _hip.check(rc, 'weight_prep2d_backward')
    return dw


class _WeightPrep2d(torch.autograd.Function):
    """weight -> energy (differentiable) with the packed 16-bit weights and the statistic as non-differentiable by-products of the same launch."""

    @staticmethod
    def forward(ctx, weight, scale, dtype, co_pad, ci_pad, want_dgrad):
        w = weight.contiguous()
        co, ci, kh, kw = w.shape
        wp = torch.empty([kh, kw, co_pad, ci_pad], dtype=dtype, device=w.device)
        wt = torch.empty([kh, kw, ci_pad, co_pad], dtype=dtype, device=w.device) if want_dgrad else None
        energy = torch.empty([co, ci], dtype=torch.float32, device=w.device)
        stat = torch.empty([co], dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            rc = _hip.lib().lvg_weight_prep2d(w.data_ptr(), wp.data_ptr(), _hip.ptr(wt), energy.data_ptr(), stat.data_ptr(), co, ci, kh * kw, co_pad, ci_pad,
                                              scale, _hip.dtype_code(dtype), _hip.stream(w.device))
        _hip.check(rc, 'weight_prep2d')
        ctx.save_for_backward(w, stat)
        ctx.cfg = (scale, co_pad, ci_pad)
        if wt is None:
            wt = torch.empty([0], dtype=dtype, device=w.device)
        ctx.mark_non_differentiable(wp, wt, stat)
        return energy, wp, wt, stat

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_energy, _g_wp, _g_wt, _g_stat):
        w, stat = ctx.saved_tensors
        scale, co_pad, ci_pad = ctx.cfg
        return _prep2d_backward(w, stat, None, g_energy, scale, co_pad, ci_pad), None, None, None, None, None


def supported2d(weight, dtype):
    if weight.device.type != 'cuda' or weight.dtype != torch.float32 or dtype not in (torch.float16, torch.bfloat16) or weight.ndim != 4:
        return False
    return weight.shape[1] * weight.shape[2] * weight.shape[3] * 8 <= 150 * 1024 and _init()


def prepare2d(weight, scale, dtype, pad=64):
    """-> Prepared2d of a float32 master weight [Co, Ci, kh, kw] on the GPU (one launch; plus the data-gradient packing while gradients
    are recorded). The channel counts of the packed weights are rounded up to multiples of `pad`."""
    co, ci = weight.shape[:2]
    co_pad, ci_pad = (co + pad - 1) // pad * pad, (ci + pad - 1) // pad * pad
    want_dgrad = torch.is_grad_enabled()
    energy, wp, wt, stat = _WeightPrep2d.apply(weight, float(scale), dtype, co_pad, ci_pad, want_dgrad)
    return Prepared2d(weight, wp, wt if want_dgrad else None, stat, energy, float(scale))
