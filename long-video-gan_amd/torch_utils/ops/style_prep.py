"""Style side of a modulated convolution in two launches forward and three backward (csrc/style_prep.hip):

    s = s / max|s| over (ci, t) per sample          (reference model/generator_lres.py:99, only when demodulating)
    demod = rsqrt(sum_ci w2[co, ci] * s^2 + 1e-8)   (:107-108; w2 = sum over the taps of the squared weight, `weight_prep`)

on styles in frames order [T, N, Ci] (row t * N + n, like the frames of the time-major activations). The PyTorch spelling is
about 8 small launches forward and 25 backward per layer (abs / amax / div / square / matmul / add / rsqrt and their autograd
nodes); the HIP path is one normalisation pass + one 64 x 64-tiled float32 product with the rsqrt fused on store forward, two
such products and one pass backward. CPU tensors take the tensor expressions (= the definition tested against)."""

import torch

from . import _hip
from .modconv_epilogue import _init


def _ref(style, w2):
    t, n, ci = style.shape
    s = style / style.abs().amax(dim=(0, 2), keepdim=True)
    demod = torch.matmul(s.square(), w2.t()).add(1e-8).rsqrt()
    return s.reshape(t * n, ci), demod.reshape(t * n, -1)


class _StylePrep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, style, w2):
        s, w2 = style.contiguous(), w2.contiguous()
        t, n, ci = s.shape
        co = w2.shape[0]
        mod = torch.empty((t * n, ci), dtype=torch.float32, device=s.device)
        demod = torch.empty((t * n, co), dtype=torch.float32, device=s.device)
        amax = torch.empty(n, dtype=torch.float32, device=s.device)
        with torch.cuda.device(s.device):
            rc = _hip.lib().lvg_style_prep(s.data_ptr(), w2.data_ptr(), mod.data_ptr(), demod.data_ptr(), amax.data_ptr(), t, n, ci, co,
                                           _hip.stream(s.device))
        _hip.check(rc, 'style_prep')
        ctx.save_for_backward(s, w2, mod, demod, amax)
        return mod, demod

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_mod, g_demod):
        s, w2, mod, demod, amax = ctx.saved_tensors
        t, n, ci = s.shape
        co = w2.shape[0]
        if g_demod is None:
            g_demod = torch.zeros_like(demod)
        g_demod = g_demod.contiguous().float()
        g_mod = g_mod.contiguous().float() if g_mod is not None else None
        gm = torch.empty_like(mod)
        ds = torch.empty_like(s)
        dw2 = torch.empty_like(w2)
        with torch.cuda.device(s.device):
            rc = _hip.lib().lvg_style_prep_backward(s.data_ptr(), amax.data_ptr(), w2.data_ptr(), mod.data_ptr(), demod.data_ptr(),
                                                    _hip.ptr(g_mod), g_demod.data_ptr(), gm.data_ptr(), ds.data_ptr(), dw2.data_ptr(),
                                                    t, n, ci, co, _hip.stream(s.device))
        _hip.check(rc, 'style_prep_backward')
        return ds, dw2


def supported(style, w2):
    if style.device.type != 'cuda' or style.dtype != torch.float32 or w2.dtype != torch.float32 or style.ndim != 3 or w2.ndim != 2:
        return False
    t, n, ci = style.shape
    return ci % 4 == 0 and w2.shape[0] % 4 == 0 and w2.shape[1] == ci and 0 < t * n < (1 << 24) and n <= 65535 and _init()


def style_prep(style, w2):
    """style [T, N, Ci] float32, w2 [Co, Ci] float32 -> (modulation [(T N), Ci], demodulation [(T N), Co]), both float32."""
    if supported(style, w2):
        return _StylePrep.apply(style, w2)
    return _ref(style, w2)
