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

import os

import torch
import torch.nn.functional as F

from . import _hip
from . import conv2d_frames
from . import weight_prep

HAND_CONV = os.environ.get('LVG_SRES_HAND_CONV', '1') != '0'     # 3 x 3 layers on csrc/conv2d_igemm.hip / conv2d_wgrad.hip (0: the library convolution)

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


def _nchw_to_nhwc_padded(src_a, src_b, scale, dst, offset, oth=None, zero_border=False, oth_planar=False):
    """cat(src_a, src_b) * scale -> the interior of the frames dst [N, Hd, Wd, C] at `offset`; -> partial or None
    (partial[n, tile, c] = sum over the 64 pixels of tile of src[n, c, p] * oth[n, p, c], oth [N, H, W, c_oth] dense frames -- or, with
    oth_planar, oth [N, c_oth, H, W] in the source's own layout).
    zero_border: the border pixels of dst are zero-filled by the call (dst may be torch.empty); else dst comes zero-filled."""
    n, c_a, h, w = src_a.shape
    c_b = 0 if src_b is None else src_b.shape[1]
    _, hd, wd, c_dst = dst.shape
    assert dst.is_contiguous() and hd >= h + offset[0] and wd >= w + offset[1] and c_dst >= c_a + c_b
    if src_a.device.type != 'cuda':
        src = src_a if src_b is None else torch.cat((src_a, src_b), dim=1)
        val = src.float() if scale is None else src.float() * scale[:, :, None, None]
        if zero_border:
            dst.zero_()
        dst[:, offset[0]:offset[0] + h, offset[1]:offset[1] + w, :c_a + c_b] = val.permute(0, 2, 3, 1).to(dst.dtype)
        if oth is None:
            return None
        if oth_planar:
            return (src[:, :oth.shape[1]].float() * oth.float()).sum(dim=(2, 3))[:, None, :]
        return (src.float().permute(0, 2, 3, 1) * oth[..., :c_a + c_b].float()).sum(dim=(1, 2))[:, None, :]
    partial = None
    if oth is not None:
        assert oth.is_contiguous() and (tuple(oth.shape) == (n, oth.shape[1], h, w) if oth_planar else oth.shape[:3] == (n, h, w))
        partial = torch.empty([n, (h * w + 63) // 64, c_a + c_b], dtype=torch.float32, device=src_a.device)
        if oth_planar and oth.shape[1] < c_a + c_b:
            partial.zero_()
    fn = _hip.lib().lvg_modconv2d_nchw_to_nhwc_padded_planar if (oth is not None and oth_planar) else _hip.lib().lvg_modconv2d_nchw_to_nhwc_padded
    with torch.cuda.device(src_a.device):
        rc = fn(src_a.data_ptr(), None if src_b is None else src_b.data_ptr(), None if scale is None else scale.data_ptr(),
                None if oth is None else oth.data_ptr(), dst.data_ptr(), None if partial is None else partial.data_ptr(),
                n, h, w, c_a, c_b, c_dst, 0 if oth is None else (oth.shape[1] if oth_planar else oth.shape[3]), hd, wd, offset[0], offset[1], 1 if zero_border else 0,
                _hip.dtype_code(src_a.dtype), _hip.stream(src_a.device))
    _hip.check(rc, 'modconv2d_nchw_to_nhwc_padded')
    return partial


def _frames_to_nchw(src, scale, c_dst, oth_a=None, oth_b=None):
    """src [N, H, W, c_src] dense frames -> (dst [N, c_dst, H, W] contiguous = src[..., :c_dst] * scale, partial or None)."""
    if src.device.type != 'cuda':
        v = src[..., :c_dst].float().permute(0, 3, 1, 2)
        dst = (v if scale is None else v * scale[:, :, None, None]).contiguous().to(src.dtype)
        partial = None
        if oth_a is not None:
            oth = oth_a if oth_b is None else torch.cat((oth_a, oth_b), dim=1)
            partial = (src[..., :oth.shape[1]].float().permute(0, 3, 1, 2) * oth.float()).sum(dim=(2, 3))[:, None, :]
        return dst, partial
    return _nhwc_to_nchw(src.permute(0, 3, 1, 2), scale, c_dst, oth_a=oth_a, oth_b=oth_b)


def _nhwc_to_nchw(src, scale, c_dst, oth_a=None, oth_b=None):
    """src [N, c_src, H, W] channels-last -> (dst [N, c_dst, H, W] contiguous, partial or None)."""
    n, c_src, h, w = src.shape
    dst = torch.empty([n, c_dst, h, w], dtype=src.dtype, device=src.device)
    partial = None
    c_a = 0 if oth_a is None else oth_a.shape[1]
    c_b = 0 if oth_b is None else oth_b.shape[1]
    if oth_a is not None:
        partial = torch.empty([n, (h * w + 63) // 64, c_a + c_b], dtype=torch.float32, device=src.device)
    assert ((c_src == 1 or src.stride(1) == 1) and (w == 1 or src.stride(3) == c_src) and (h == 1 or src.stride(2) == w * c_src)
            and (n == 1 or src.stride(0) == h * w * c_src)), 'dense channels-last frames expected'
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
        assert not ctx.needs_input_grad[1], 'modulated_conv2d: no gradient for the conditioning frames on the fused path'
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


class _ModConv2dHand(torch.autograd.Function):
    """The whole modulated 3 x 3 convolution on the hand-written kernels:
        prologue (cat, modulate, NCHW -> zero-padded channels-last frames)  ->  lvg_conv2d_frames  ->  epilogue (demodulate, -> NCHW)
    and in the backward pass the epilogue's opposite kernel (gradient frames, d demod), the data gradient (the same convolution
    kernel on the gradient frames), the weight gradient (lvg_conv2d_frames_wgrad) and the prologue's opposite kernel (d x, d mod).
    One autograd node: every intermediate frame has the geometry the kernels want (conv2d_frames.Geometry)."""

    @staticmethod
    def forward(ctx, first, second, weight, mod, demod, padding, prepared=None):
        """`weight` [Co, Ci, 3, 3]: the weight the convolution runs with (cast to the activations' dtype here) -- or, with `prepared`
        (weight_prep.Prepared2d: normalised, cast and packed by lvg_weight_prep2d), the float32 MASTER weight, whose gradient then
        includes the normalisation."""
        first, mod = first.contiguous(), mod.float().contiguous()
        second = None if second is None else second.contiguous()
        demod = None if demod is None else demod.float().contiguous()
        n, c_first, h, w = first.shape
        co, ci = weight.shape[:2]
        assert ci == c_first + (0 if second is None else second.shape[1]) and tuple(weight.shape[2:]) == (3, 3)
        geo = conv2d_frames.Geometry(h, w, padding)
        ci_pad, co_pad = conv2d_frames.round_up(ci, conv2d_frames.CH), conv2d_frames.round_up(co, conv2d_frames.CH)
        xp = torch.empty([n, geo.hx, geo.wx, ci_pad], dtype=first.dtype, device=first.device)
        _nchw_to_nhwc_padded(first, second, mod, xp, (2, 2), zero_border=True)
        alg = 2 * n * geo.ho * geo.wo * co * ci * 9                  # algorithmic work of each of the three contractions
        if prepared is not None:
            assert prepared.wp.shape == (3, 3, co_pad, ci_pad) and prepared.wp.dtype == first.dtype
            wp = prepared.wp
        else:
            wp = conv2d_frames.pack_weight(weight, first.dtype, ci_pad, co_pad)
        planes = PLANES_OUT and first.is_cuda and geo.wo % 2 == 0 and demod is not None
        if planes:
            # the convolution stores the demodulated NCHW planes itself (lvg_conv2d_frames_planes: no transposing pass over the result); the
            # backward pass gets d demod from these planes, sum d_out * out / demod
            out = conv2d_frames.conv2d_valid_planes(xp, wp, geo.ho, geo.wo, co, offset=(geo.q, geo.q), pre=demod, alg_flops=alg)
            y = out
        else:
            y = conv2d_frames.conv2d_valid(xp, wp, geo.ho, geo.wo, offset=(geo.q, geo.q), alg_flops=alg)
            out, _ = _frames_to_nchw(y, demod, co)
        ctx.save_for_backward(first, second, mod, demod, xp, y, weight)
        ctx.geo, ctx.alg, ctx.prepared, ctx.planes, ctx.co_pad = geo, alg, prepared, planes, co_pad
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        first, second, mod, demod, xp, y, weight = ctx.saved_tensors
        geo = ctx.geo
        assert not ctx.needs_input_grad[1], 'modulated_conv2d: no gradient for the conditioning frames on the fused path'
        n, c_first = first.shape[:2]
        co, ci = weight.shape[:2]
        ci_pad, co_pad = xp.shape[3], ctx.co_pad
        need_first, need_weight, need_mod, need_demod = ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3], demod is not None and ctx.needs_input_grad[4]
        # gradient frames: d_out * demod at (q, q) of the zero-filled patch-aligned frame; d demod = sum d_out * y from the same pass
        dyp = torch.empty([n, geo.hd, geo.wd, co_pad], dtype=first.dtype, device=first.device)
        partial = _nchw_to_nhwc_padded(d_out.contiguous(), None, demod, dyp, (geo.q, geo.q), oth=y if need_demod else None, zero_border=True, oth_planar=ctx.planes)
        d_demod = partial.sum(dim=1) if need_demod else None
        if need_demod and ctx.planes:
            d_demod = d_demod / demod                        # the saved planes are y * demod (demod = rsqrt(...) > 0)
        d_weight = None
        prepared = ctx.prepared
        if need_weight:
            gw = conv2d_frames.conv2d_wgrad(xp, dyp, alg_flops=ctx.alg)                # [3, 3, co_pad, ci_pad] float32
            d_weight = prepared.grad_from_conv(gw) if prepared is not None else gw[:, :, :co, :ci].permute(2, 3, 0, 1).to(weight.dtype)
        d_first = d_mod = None
        if need_first or need_mod:
            wt = prepared.wt if prepared is not None and prepared.wt is not None else \
                conv2d_frames.pack_weight_dgrad(weight if prepared is None else prepared.wp[:, :, :co, :ci].permute(2, 3, 0, 1), first.dtype, ci_pad, co_pad)
            if PLANES_OUT and first.is_cuda and geo.w % 2 == 0:
                # the data gradient stores d first = dx * mod as planes itself and forms d mod = sum dx * cat(first, second) on its accumulators
                res = conv2d_frames.conv2d_valid_planes(dyp, wt, geo.h, geo.w, c_first, pre=mod[:, :c_first].contiguous(), alg_flops=ctx.alg,
                                                        dot=(first, second) if need_mod else None)
                d_first, partial = res if need_mod else (res, None)
            else:
                dxp = conv2d_frames.conv2d_valid(dyp, wt, geo.h, geo.w, alg_flops=ctx.alg)
                d_first, partial = _frames_to_nchw(dxp, mod[:, :c_first].contiguous(), c_first, oth_a=first if need_mod else None, oth_b=second if need_mod else None)
            d_mod = partial.sum(dim=1) if need_mod else None
        return (d_first if need_first else None), None, d_weight, d_mod, d_demod, None, None


# Split forms of a float32 operand (conv2d_frames.split16 / conv3d_frames.split_bf16x3) and the partial products kept:
#   'f16x2'  : two float16 parts (11 + 11 bits) after an exact power-of-two scaling to a maximum of ~2^10; products hh, lh, hl: THREE
#              times the channels. Elements ~2^13 below the tensor's maximum lose their low part (float16's narrow exponent).
#   'bf16x3' : three bfloat16 parts (8 + 8 + 8 bits, float32's exponent range, no scaling); products 11, 12, 21, 13, 22, 31: SIX times
#              the channels. Exact to ~2^-24 whatever the dynamic range.
# Measured on a real generator update (tools/diag_sres_split.py, profiles/r03_sres_split_operands.log): the operands of these three layers
# -- activations, weights, arriving gradients with max / median up to 2^16 -- are represented by two float16 parts to 4.4e-8 in L2 (elements
# below 2^-13 of the maximum carry < 4e-6 of the energy), so the cheaper form is the default here; the 21-layer lres generator needs the
# three-part form (conv3d_frames.py).
SPLIT_MODE = os.environ.get('LVG_SRES_SPLIT_MODE', 'f16x2')
_SPLIT_FORMS = {'f16x2': (torch.float16, 2, (0, 1, 0), (0, 0, 1)), 'bf16x3': (torch.bfloat16, 3, (0, 0, 1, 0, 1, 2), (0, 1, 0, 2, 1, 0))}


def _split_parts(t, mode):
    """-> (parts, scale): t * scale = sum(parts) to float32 accuracy; scale is a 0-d tensor (1 for bfloat16 parts)."""
    if mode == 'f16x2':
        s = conv2d_frames.pow2_scale(t)
        return list(conv2d_frames.split16(t.float() * s)), s
    from .conv3d_frames import split_bf16x3
    return list(split_bf16x3(t)), torch.ones((), device=t.device)


class _ModConv2dSplit(torch.autograd.Function):
    """The modulated 3 x 3 convolution of the FLOAT32 layers on the hand-written 16-bit MFMA kernels with float32 accuracy: every
    operand is split into 16-bit parts (_SPLIT_FORMS) and the significant partial products run as ONE contraction over stacked
    channels, accumulated and stored in float32. The reference runs these layers in float32 with TF32 off (train_sres.py:304-306).
    Layout / modulation / splitting are plain tensor ops here (the float32 layers are the three smallest of the network: 38 x 31 pixels)."""

    @staticmethod
    def forward(ctx, first, second, weight, mod, demod, padding):
        c2 = conv2d_frames
        dt, nparts, xpat, wpat = _SPLIT_FORMS[SPLIT_MODE]
        n, c_first, h, w = first.shape
        co, ci = weight.shape[:2]
        geo = c2.Geometry(h, w, padding)
        cip, cop = c2.round_up(ci, c2.CH), c2.round_up(co, c2.CH)
        ws, sw = _split_parts(weight, SPLIT_MODE)
        k = len(xpat)
        xp = torch.zeros([n, geo.hx, geo.wx, k * cip], dtype=dt, device=first.device)      # the stacked parts per pixel
        fast = FUSED_SPLIT and SPLIT_MODE == 'f16x2' and second is None and first.is_cuda and first.dtype == torch.float32
        if fast:
            # scale, split and placement of (first * mod) in ONE pass over the float32 planes (lvg_split16_frames); the tensor expressions
            # below made ~15 passes over the 38 MB tensor of these layers
            sx = c2.split16_into_frame(first.contiguous(), mod.float().contiguous(), xp, 2, 2, cip, xpat)
            xs = None
        else:
            xin = (first if second is None else torch.cat((first, second), dim=1)).float() * mod.float()[:, :, None, None]
            xs, sx = _split_parts(xin.permute(0, 2, 3, 1), SPLIT_MODE)
            inner = xp[:, 2:2 + h, 2:2 + w]
            for j, pi in enumerate(xpat):
                inner[..., j * cip:j * cip + ci] = xs[pi]
        wp = torch.cat([c2.pack_weight(ws[pi], dt, cip, cop) for pi in wpat], dim=3)        # [3, 3, cop, k cip]
        alg = 2 * n * geo.ho * geo.wo * co * ci * 9                  # algorithmic work (the partial products are this implementation's cost, not the operation's)
        y = c2.conv2d_valid(xp, wp, geo.ho, geo.wo, offset=(geo.q, geo.q), out_dtype=torch.float32, alg_flops=alg)     # [n, ho, wo, cop] float32, scaled by sx sw
        if fast:
            # (the saved y stays UNSCALED on this path: the backward pass folds 1 / (sx sw) into d_demod)
            inv = (1.0 / (sx * sw)).reshape(1)
            out = c2.nhwc_f32_to_nchw(y, co, None if demod is None else demod.float().contiguous(), inv)
            y_scale = inv
        else:
            y = y * (1.0 / (sx * sw))
            yv = y[..., :co].permute(0, 3, 1, 2)
            out = (yv if demod is None else yv * demod.float()[:, :, None, None]).contiguous()
            y_scale = None
        # the weight gradient needs each part of x ONCE: they are the first occurrences in the stacked frame when the pattern starts 0, .. (f16x2:
        # blocks 0, 1; bf16x3: blocks 0, 2, 5 -- kept as a separate compact frame there)
        xw = xp if SPLIT_MODE == 'f16x2' else None
        if xw is None:
            xw = torch.zeros([n, geo.hx, geo.wx, nparts * cip], dtype=dt, device=first.device)
            innerw = xw[:, 2:2 + h, 2:2 + w]
            for pi in range(nparts):
                innerw[..., pi * cip:pi * cip + ci] = xs[pi]
        ctx.save_for_backward(first, second, mod, demod, xw, y, weight, sx, sw)
        ctx.geo, ctx.alg, ctx.mode, ctx.cip = geo, alg, SPLIT_MODE, cip
        ctx.y_scale = y_scale
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        c2 = conv2d_frames
        first, second, mod, demod, xw, y, weight, sx, sw = ctx.saved_tensors
        geo, mode, cip = ctx.geo, ctx.mode, ctx.cip
        dt, nparts, xpat, wpat = _SPLIT_FORMS[mode]
        assert not ctx.needs_input_grad[1], 'modulated_conv2d: no gradient for the conditioning frames on the fused path'
        n, c_first = first.shape[:2]
        co, ci = weight.shape[:2]
        cop = y.shape[3]
        d_out = d_out.float()
        d_demod = None
        if demod is not None and ctx.needs_input_grad[4]:
            d_demod = (d_out * y[..., :co].permute(0, 3, 1, 2)).sum(dim=(2, 3))
            if ctx.y_scale is not None:
                d_demod = d_demod * ctx.y_scale
        k = len(xpat)
        dyp = torch.zeros([n, geo.hd, geo.wd, k * cop], dtype=dt, device=first.device)       # stacked parts of the gradient per pixel at (q, q)
        fast = FUSED_SPLIT and mode == 'f16x2' and d_out.is_cuda
        if fast:
            s = c2.split16_into_frame(d_out.contiguous(), None if demod is None else demod.float().contiguous(), dyp, geo.q, geo.q, cop, xpat)
            gs = None
        else:
            g = d_out if demod is None else d_out * demod.float()[:, :, None, None]
            gs, s = _split_parts(g.permute(0, 2, 3, 1), mode)
            inner = dyp[:, geo.q:geo.q + geo.ho, geo.q:geo.q + geo.wo]
            for j, pi in enumerate(xpat):
                inner[..., j * cop:j * cop + co] = gs[pi]
        d_weight = None
        if ctx.needs_input_grad[2]:
            # each part of x against each part of the gradient: the blocks (i, j) of the [nparts cop, nparts cip] result with i + j < nparts
            # are the significant partial products (f16x2: all four are added, the low-low one is below the rounding)
            if mode == 'f16x2':
                gyp = dyp
            else:
                gyp = torch.zeros([n, geo.hd, geo.wd, nparts * cop], dtype=dt, device=first.device)
                innerg = gyp[:, geo.q:geo.q + geo.ho, geo.q:geo.q + geo.wo]
                for pi in range(nparts):
                    innerg[..., pi * cop:pi * cop + co] = gs[pi]
            gw = c2.conv2d_wgrad(xw, gyp, x_channels=nparts * cip, dy_channels=nparts * cop, alg_flops=ctx.alg)
            acc = None
            for total in range(nparts if mode != 'f16x2' else nparts + 1, -1, -1):                   # small terms first
                for i in range(nparts):
                    j = total - i
                    if 0 <= j < nparts and (mode == 'f16x2' or i + j < nparts):
                        blk = gw[:, :, i * cop:(i + 1) * cop, j * cip:(j + 1) * cip]
                        acc = blk if acc is None else acc + blk
            d_weight = (acc[:, :, :co, :ci].permute(2, 3, 0, 1) * (1.0 / (s * sx))).to(weight.dtype)
        d_first = d_mod = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
            ws, _ = _split_parts(weight, mode)
            wd = torch.cat([c2.pack_weight_dgrad(ws[pi], dt, cip, cop) for pi in wpat], dim=3)       # [3, 3, cip, k cop]
            dxf = c2.conv2d_valid(dyp, wd, geo.h, geo.w, out_dtype=torch.float32, alg_flops=ctx.alg)
            if fast:
                dx = c2.nhwc_f32_to_nchw(dxf, ci, None, (1.0 / (s * sw)).reshape(1))            # d (x * mod), contiguous NCHW planes in one pass
            else:
                dx = dxf[..., :ci].permute(0, 3, 1, 2) * (1.0 / (s * sw))                          # the same as a view
            if ctx.needs_input_grad[3]:
                xcat = (first if second is None else torch.cat((first, second), dim=1)).float()
                d_mod = (dx * xcat).sum(dim=(2, 3))
            if ctx.needs_input_grad[0]:
                d_first = (dx[:, :c_first] * mod.float()[:, :c_first, None, None]).to(first.dtype).contiguous()
        return d_first, None, d_weight, d_mod, d_demod, None


# 16-bit layers: the convolution writes demodulated NCHW planes itself (lvg_conv2d_frames_planes) instead of channels-last rows + lvg_modconv2d_nhwc_to_nchw (0: the two passes)
PLANES_OUT = os.environ.get('LVG_SRES_PLANES_OUT', '1') != '0'
FUSED_SPLIT = os.environ.get('LVG_SRES_FUSED_SPLIT', '1') != '0'     # operands of the float32 layers split in one pass (lvg_split16_frames; 0: tensor expressions)
SPLIT_F32 = os.environ.get('LVG_SRES_SPLIT_F32', '1') != '0'      # float32 3 x 3 layers on the hand-written kernels through split operands (0: the library convolution)


def split_conv_supported(first, weight):
    return (HAND_CONV and SPLIT_F32 and first.device.type == 'cuda' and first.dtype == torch.float32 and tuple(weight.shape[2:]) == (3, 3))


def hand_conv_supported(first, weight):
    return (HAND_CONV and first.device.type == 'cuda' and first.dtype in (torch.float16, torch.bfloat16) and tuple(weight.shape[2:]) == (3, 3))


def supported(x, cond):
    t = cond if x is None else x
    return t.device.type == 'cuda' and t.dtype in (torch.float16, torch.bfloat16)


def modulated_conv2d(x, cond, weight, mod, demod, padding=0):
    """x [N, C1, H, W] or None, cond [N, C2, H, W] (same dtype), weight [Co, C1 + C2, k, k] (any float dtype) or a
    weight_prep.Prepared2d (3 x 3 weights normalised / cast / packed by one launch: hand-written route only),
    mod float32 [N, C1 + C2], demod float32 [N, Co] or None. Returns NCHW [N, Co, H', W'] in x's dtype."""
    first, second = (cond, None) if x is None else (x, cond)
    if isinstance(weight, weight_prep.Prepared2d):
        assert supported(x, cond) and hand_conv_supported(first, weight.weight) and 0 <= padding <= 2, 'prepared weights: hand-written route only'
        if second is not None and second.dtype != first.dtype:
            second = second.to(first.dtype)
        return _ModConv2dHand.apply(first, second, weight.weight, mod, demod, padding, weight)
    if split_conv_supported(first, weight) and 0 <= padding <= 2:
        return _ModConv2dSplit.apply(first, None if second is None else second.to(first.dtype), weight, mod, demod, padding)
    if not supported(x, cond):
        return _ref(x, cond, weight, mod, demod, padding)
    dtype = first.dtype
    if second is not None and second.dtype != dtype:
        second = second.to(dtype)
    c_in, c_out = weight.shape[1], weight.shape[0]
    assert c_in == first.shape[1] + (0 if second is None else second.shape[1])
    if hand_conv_supported(first, weight) and 0 <= padding <= 2:
        return _ModConv2dHand.apply(first, second, weight, mod, demod, padding)
    ci_pad, co_pad = _pad_to(c_in), _pad_to(c_out)
    xin = _Prologue.apply(first, second, mod.float(), ci_pad)
    w = weight.to(dtype)
    if ci_pad != c_in or co_pad != c_out:
        w = F.pad(w, (0, 0, 0, 0, 0, ci_pad - c_in, 0, co_pad - c_out))
    y = F.conv2d(xin, _cl(w), padding=padding)
    return _Epilogue.apply(y, None if demod is None else demod.float(), c_out)
