"""2-D style-modulated convolution on channels-last MFMA kernels: prologue / epilogue ops.

`modulated_conv2d(x, cond, weight, mod, demod)` computes what the reference's `modulated_conv2d`
(model/generator_sres.py:24-67) computes for the super-resolution generator,

    conv2d(cat(x, cond) * mod[:, :, None, None], weight) * demod[:, :, None, None]

with two hand-written transposing kernels (csrc/modconv2d_layout.hip) around ONE dense channels-last
convolution: the prologue concatenates the previous layer's NCHW output with the conditioning frames,
applies the per-(sample, channel) modulation and writes NHWC with the channel count padded to a multiple
of 8; the epilogue applies the demodulation while going back to the NCHW planes that `filtered_lrelu`
tiles. Their backward passes are the opposite kernels plus a per-tile reduction for the gradient of the
modulation / demodulation (summed here; no atomics, run-to-run reproducible).

float16 / bfloat16 on the GPU only; other tensors take the plain-PyTorch definition `_ref`."""

import torch
import torch.nn.functional as F

from . import _hip

PAD = 8   # NHWC channel counts are padded to a multiple of this (16-byte vectors of 16-bit elements)


def _pad_to(c):
    return (c + PAD - 1) // PAD * PAD


def _ref(x, cond, weight, mod, demod, padding):
    """Plain-PyTorch definition (CPU tensors, float32): the same arithmetic, no layout change."""
    xin = cond if x is None else torch.cat((x, cond.to(x.dtype)), dim=1)
    y = F.conv2d(xin * mod.to(xin.dtype)[:, :, None, None], weight.to(xin.dtype), padding=padding)
    if demod is not None:
        y = y * demod.to(y.dtype)[:, :, None, None]
    return y


def _nchw_to_nhwc(src_a, src_b, scale, c_dst, oth=None):
    """-> (dst [N, c_dst, H, W] channels-last, partial or None)."""
    n, c_a, h, w = src_a.shape
    c_b = 0 if src_b is None else src_b.shape[1]
    dst = torch.empty([n, c_dst, h, w], dtype=src_a.dtype, device=src_a.device, memory_format=torch.channels_last)
    partial = None
    if oth is not None:
        partial = torch.empty([n, (h * w + 63) // 64, c_a + c_b], dtype=torch.float32, device=src_a.device)
    with torch.cuda.device(src_a.device):
        rc = _hip.lib().lvg_modconv2d_nchw_to_nhwc(
            src_a.data_ptr(), None if src_b is None else src_b.data_ptr(), None if scale is None else scale.data_ptr(),
            None if oth is None else oth.data_ptr(), dst.data_ptr(), None if partial is None else partial.data_ptr(),
            n, h * w, c_a, c_b, c_dst, 0 if oth is None else oth.shape[1], _hip.dtype_code(src_a.dtype), _hip.stream(src_a.device))
    _hip.check(rc, 'modconv2d_nchw_to_nhwc')
    return dst, partial


def _nhwc_to_nchw(src, scale, c_dst, oth_a=None, oth_b=None):
    """src [N, c_src, H, W] channels-last -> (dst [N, c_dst, H, W] contiguous, partial or None)."""
    n, c_src, h, w = src.shape
    dst = torch.empty([n, c_dst, h, w], dtype=src.dtype, device=src.device)
    partial = None
    c_a = 0 if oth_a is None else oth_a.shape[1]
    c_b = 0 if oth_b is None else oth_b.shape[1]
    if oth_a is not None:
        partial = torch.empty([n, (h * w + 63) // 64, c_a + c_b], dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        rc = _hip.lib().lvg_modconv2d_nhwc_to_nchw(
            src.data_ptr(), None if scale is None else scale.data_ptr(), None if oth_a is None else oth_a.data_ptr(),
            None if oth_b is None else oth_b.data_ptr(), dst.data_ptr(), None if partial is None else partial.data_ptr(),
            n, h * w, c_src, c_dst, c_a, c_b, _hip.dtype_code(src.dtype), _hip.stream(src.device))
    _hip.check(rc, 'modconv2d_nhwc_to_nchw')
    return dst, partial


def _cl(t):
    """Channels-last memory with the canonical strides (size-1 dims make `contiguous` a no-op otherwise)."""
    if t.is_contiguous(memory_format=torch.channels_last) and (t.shape[1] == 1 or t.stride(1) == 1):
        return t
    return t.contiguous(memory_format=torch.channels_last)


class _Prologue(torch.autograd.Function):
    """cat(x, cond) * mod -> NHWC with zero-padded channels."""

    @staticmethod
    def forward(ctx, x, cond, mod, c_pad):
        x, cond, mod = x.contiguous(), (None if cond is None else cond.contiguous()), mod.contiguous()
        out, _ = _nchw_to_nhwc(x, cond, mod, c_pad)
        ctx.save_for_backward(x, cond, mod)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        x, cond, mod = ctx.saved_tensors
        c_x = x.shape[1]
        d_out = _cl(d_out)
        need_x, need_mod = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        d_x, partial = _nhwc_to_nchw(d_out, mod[:, :c_x].contiguous(), c_x, oth_a=x if need_mod else None, oth_b=cond if need_mod else None)
        d_mod = partial.sum(dim=1) if need_mod else None
        return (d_x if need_x else None), None, d_mod, None


class _Epilogue(torch.autograd.Function):
    """NHWC conv output (padded channels) * demod -> NCHW with the true channel count."""

    @staticmethod
    def forward(ctx, y, demod, c_out):
        y = _cl(y)
        demod = None if demod is None else demod.contiguous()
        out, _ = _nhwc_to_nchw(y, demod, c_out)
        ctx.save_for_backward(y, demod)
        ctx.c_pad = y.shape[1]
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        y, demod = ctx.saved_tensors
        need_demod = demod is not None and ctx.needs_input_grad[1]
        d_y, partial = _nchw_to_nhwc(d_out.contiguous(), None, demod, ctx.c_pad, oth=y if need_demod else None)
        d_demod = partial.sum(dim=1) if need_demod else None
        return d_y, d_demod, None


def supported(x, cond):
    t = cond if x is None else x
    return t.device.type == 'cuda' and t.dtype in (torch.float16, torch.bfloat16)


def modulated_conv2d(x, cond, weight, mod, demod, padding=0):
    """x [N, C1, H, W] or None, cond [N, C2, H, W] (same dtype), weight [Co, C1 + C2, k, k] (any float dtype),
    mod float32 [N, C1 + C2], demod float32 [N, Co] or None. Returns NCHW [N, Co, H', W'] in x's dtype."""
    first, second = (cond, None) if x is None else (x, cond)
    if not supported(x, cond):
        return _ref(x, cond, weight, mod, demod, padding)
    dtype = first.dtype
    if second is not None and second.dtype != dtype:
        second = second.to(dtype)
    c_in, c_out = weight.shape[1], weight.shape[0]
    assert c_in == first.shape[1] + (0 if second is None else second.shape[1])
    ci_pad, co_pad = _pad_to(c_in), _pad_to(c_out)
    xin = _Prologue.apply(first, second, mod.float(), ci_pad)
    w = weight.to(dtype)
    if ci_pad != c_in or co_pad != c_out:
        w = F.pad(w, (0, 0, 0, 0, 0, ci_pad - c_in, 0, co_pad - c_out))
    y = F.conv2d(xin, _cl(w), padding=padding)
    return _Epilogue.apply(y, None if demod is None else demod.float(), c_out)
