"""Low-resolution video GAN networks ([B, C, T, H, W], 3-D convolutions) on the MI355X op stack.

Architecture and parameter/buffer NAMES follow the reference so that a reference state_dict
loads unchanged (reference model/generator_lres.py: VideoGenerator :649, Synthesis3dResBlock
:487, ToRGB :600, BlurredNoise :323, LatentMappingNetwork :444; model/discriminator_lres.py:
VideoDiscriminator :420, DiscriminatorBlock :264, DiscriminatorEpilogue :341). The code is a
re-implementation organised around the HIP custom ops:

  * every bias/activation/clamp goes through `bias_act` (one HBM pass, HIP), every resampling
    through the fused separable `upfirdn2d` (one pass instead of the reference's two);
  * `compute_dtype` (float32 | bfloat16 | float16) selects the dtype of the activations and of
    the dense contraction (MIOpen conv3d on MFMA); parameters stay float32 and all style /
    demodulation statistics are computed in float32;
  * the demodulation statistic is contracted as (sum_zyx w^2) @ s^2 instead of the reference's
    5-index einsum (:107) -- same value, k_t*k_h*k_w times fewer multiply-adds.
"""

import math
import os
from typing import List, Optional

import numpy as np
import scipy.signal
import torch
import torch.nn as nn
import torch.nn.functional as F

import torch_utils.distributed as dist_utils
import contextlib

from .. import ddp

from torch_utils.ops import bias_act, conv3d_frames, noise_bank, pointwise_thin, stats, style_prep, upfirdn2d, weight_prep
from torch_utils.ops.modconv_epilogue import dual_supported, modconv_epilogue, modconv_epilogue_dual, tap_gather_backward, tap_gather_forward

SQRT_HALF = math.sqrt(0.5)

# Frames-layout activations are kept channels-last ([(T N), H, W, C] in memory): MIOpen's implicit-GEMM
# kernels are NHWC-native, NCHW inputs cost two layout-transpose kernels plus copies per convolution
# (measured: 26 % of the step). The HIP ops take either layout.
CHANNELS_LAST = os.environ.get('LVG_CHANNELS_LAST', '1') != '0'


def _cl(x: torch.Tensor) -> torch.Tensor:
    return x.contiguous(memory_format=torch.channels_last) if CHANNELS_LAST else x


# --------------------------------------------------------------------------------------------------
# Small building blocks.

def crop_center(x: torch.Tensor, width: Optional[int] = None, height: Optional[int] = None, seq_length: Optional[int] = None) -> torch.Tensor:
    """Centered crop of a [N, C, T] or [N, C, T, H, W] tensor; returns a view."""
    if width is not None:
        x0 = (x.size(4) - width) // 2
        x = x[:, :, :, :, x0:x0 + width]
    if height is not None:
        y0 = (x.size(3) - height) // 2
        x = x[:, :, :, y0:y0 + height]
    if seq_length is not None:
        t0 = (x.size(2) - seq_length) // 2
        x = x[:, :, t0:t0 + seq_length]
    return x


def _linear_filter(scale: int) -> torch.Tensor:
    ramp = torch.linspace(0.5 / scale, 1 - 0.5 / scale, scale)
    taps = torch.cat((ramp, ramp.flip(0)))
    return taps / taps.sum()


class _FilterHolder(nn.Module):
    """Module whose only state is a `filter` buffer (keeps reference state_dict keys)."""

    def __init__(self, taps: torch.Tensor, scale: int = 2):
        super().__init__()
        self.scale = scale
        self.register_buffer('filter', taps)


class SpatialBilinearUpsample(_FilterHolder):
    def __init__(self, scale: int = 2):
        super().__init__(_linear_filter(scale), scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, c, t, h, w = x.shape
        y = upfirdn2d.upsample2d(x.reshape(n, c * t, h, w), self.filter, up=self.scale)
        return y.reshape(n, c, t, y.size(2), y.size(3))


class TemporalLinearUpsample(_FilterHolder):
    def __init__(self, scale: int = 2):
        super().__init__(_linear_filter(scale), scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, c, t, h, w = x.shape
        y = upfirdn2d.upsample2d(x.reshape(n, c, t, h * w), self.filter.unsqueeze(1), up=(1, self.scale))
        return y.reshape(n, c, y.size(2), h, w)


class TemporalLinearDownsample(_FilterHolder):
    def __init__(self, scale: int = 2):
        super().__init__(_linear_filter(scale), scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 3:
            y = upfirdn2d.downsample2d(x.unsqueeze(3), self.filter.unsqueeze(1), down=(1, self.scale))
            return y.squeeze(3)
        n, c, t, h, w = x.shape
        y = upfirdn2d.downsample2d(x.reshape(n, c, t, h * w), self.filter.unsqueeze(1), down=(1, self.scale))
        return y.reshape(n, c, y.size(2), h, w)


class TemporalKaiserDownsample(_FilterHolder):
    def __init__(self, scale: int = 2, filter_size: int = 6, cutoff: float = 1.0, width: float = 6.0, sampling_rate: float = 4.0):
        taps = scipy.signal.firwin(numtaps=scale * filter_size, cutoff=cutoff, width=width, fs=scale * sampling_rate)
        super().__init__(torch.tensor(taps, dtype=torch.float32), scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # [N, C, T] latents. The mapping network hands them out as a permuted view of [N, T, C]; present
        # that memory as planes [N, 1, T, C] so that the channel axis is the coalesced W axis of the kernel.
        if x.dim() == 3 and x.stride(1) == 1 and x.shape[1] > 1:
            rows = x.permute(0, 2, 1).unsqueeze(1)
            y = upfirdn2d.downsample2d(rows, self.filter.unsqueeze(1), down=(1, self.scale))
            return y.squeeze(1).permute(0, 2, 1)
        y = upfirdn2d.downsample2d(x.unsqueeze(3), self.filter.unsqueeze(1), down=(1, self.scale))
        return y.squeeze(3)


class MagnitudeEMA(nn.Module):
    """Tracks E[x^2] of a layer input; returns its reciprocal square root as a scalar gain.

    Across ranks the reference all-reduces the statistic inside every layer's forward
    (model/generator_lres.py:298-312: 21 one-float collectives per generator pass). Inside a
    `deferred_magnitude_sync()` scope (= lvg.ddp.deferred_stat_sync) the layer instead folds in its LOCAL
    mean and records it; one batched all-reduce afterwards (`finish_magnitude_sync`) corrects every buffer to
    the global-mean update, so buffers end up identical on all ranks and identical to the reference's. The
    only difference is that the gain used in THIS forward pass saw the local mean: a relative change of
    (1 - beta) * (local/global - 1) ~ 1e-5, and nothing at world size 1."""

    def __init__(self, dist_sync: bool = True):
        super().__init__()
        self.dist_sync = dist_sync
        self.register_buffer('magnitude_ema', torch.ones(()))

    def update(self, mean_square: torch.Tensor, beta: float) -> torch.Tensor:
        """Fold a freshly measured mean square (float32 scalar tensor) into the EMA; returns the gain."""
        mag = mean_square.detach()
        if self.dist_sync:
            ddp.ema_of_rank_mean(self.magnitude_ema, mag, beta, ddp.LERP_TOWARDS)
        else:
            self.magnitude_ema.lerp_(mag.to(self.magnitude_ema.dtype), 1.0 - beta)
        return self.magnitude_ema.rsqrt()

    def forward(self, x: torch.Tensor, beta: float = 1.0) -> torch.Tensor:
        if beta != 1:
            return self.update(stats.mean_square(x), beta)
        return self.magnitude_ema.rsqrt()


# the deferred exchange of the running statistics lives in lvg.ddp (shared with the super-resolution generator)
deferred_magnitude_sync = ddp.deferred_stat_sync
stack_pending = ddp.stack_pending
finish_magnitude_sync = ddp.finish_stat_sync


class FullyConnectedLayer(nn.Module):
    """y = act(x @ (w * gain)^T + b * lrate_mul); equalised learning rate parametrisation."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, activation: str = 'linear',
                 lrate_mul: float = 1.0, weight_std_init: float = 1.0, bias_init: float = 0.0):
        super().__init__()
        assert activation in bias_act.activation_funcs
        self.activation = activation
        self.weight = nn.Parameter(torch.randn(out_features, in_features) * (weight_std_init / lrate_mul))
        self.weight_gain = lrate_mul / math.sqrt(in_features)
        self.bias = nn.Parameter(torch.full((out_features,), bias_init / lrate_mul)) if bias else None
        self.bias_gain = lrate_mul

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w = (self.weight * self.weight_gain).t()
        b = self.bias
        if b is not None and self.bias_gain != 1:
            b = b * self.bias_gain
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w)
        return bias_act.bias_act(x.matmul(w), b, act=self.activation)


def _dense_conv3d(x: torch.Tensor, weight: torch.Tensor, padding) -> torch.Tensor:
    """conv3d on the path that maps best to MI355X: a 1x1x1 kernel is a plain (batched) GEMM over
    channels on the NCTHW tensor itself -- no im2col buffer; everything else goes to MIOpen."""
    if weight.shape[2:] == (1, 1, 1):
        n, c, t, h, w = x.shape
        y = torch.matmul(weight.reshape(weight.shape[0], c), x.reshape(n, c, t * h * w))
        return y.reshape(n, weight.shape[0], t, h, w)
    return F.conv3d(x, weight, padding=padding)


def modulated_conv3d(x: torch.Tensor, weight: torch.Tensor, style: torch.Tensor, input_gain: Optional[torch.Tensor],
                     padding, demodulate: bool, compute_dtype: torch.dtype) -> torch.Tensor:
    """Per-(sample, frame) style-modulated conv3d (reference generator_lres.py:83-125).

    x [N, Ci, T, H, W]; weight [Co, Ci, kt, kh, kw] float32; style [N, Ci, T] float32.
    Returns conv(x * gain * style) * demod in `compute_dtype`."""
    if demodulate:
        weight = weight / weight.abs().amax(dim=(1, 2, 3, 4), keepdim=True)
        style = style / style.abs().amax(dim=(1, 2), keepdim=True)
    weight = weight * (1.0 / math.sqrt(weight[0].numel()))
    mod = style if input_gain is None else style * input_gain
    x = x * mod.to(x.dtype)[:, :, :, None, None]
    y = _dense_conv3d(x.to(compute_dtype), weight.to(compute_dtype), padding)
    if demodulate:
        w2 = weight.square().sum(dim=(2, 3, 4))                      # [Co, Ci]
        demod = torch.matmul(w2, style.square()).add(1e-8).rsqrt()   # [N, Co, T]
        y = y * demod.to(y.dtype)[:, :, :, None, None]
    return y


# --------------------------------------------------------------------------------------------------
# TIME-MAJOR FRAMES layout: a video batch is stored as [(T N), C, H, W] (frame index f = t * N + n).
#
# Why (measured on MI355X, tools/conv_probe.py / conv2d_probe.py, bf16): MIOpen's NCDHW conv3d is an explicit
# Im3d2Col + GEMM (+ Col2Im3dU backward) at 35-230 TFLOP/s, its 2-D convs are implicit-GEMM MFMA kernels at
# 230-510 TFLOP/s. A kernel [Co, Ci, kt, kh, kw] is the sum over its kt temporal taps of 2-D convolutions of
# time-shifted frames, and with time OUTERMOST a time shift is a contiguous slice of the frame axis: no
# padding copy, no im2col buffer. Per-frame layers (styles, bias_act, spatial resampling) do not care about
# the frame order; resampling along time sees the tensor as [T, (N C H W)] rows.

def frames_from_video(video: torch.Tensor) -> torch.Tensor:
    """[N, C, T, H, W] -> [(T N), C, H, W]"""
    n, c, t, h, w = video.shape
    return video.permute(2, 0, 1, 3, 4).reshape(t * n, c, h, w)


def video_from_frames(frames: torch.Tensor, n: int) -> torch.Tensor:
    """[(T N), C, H, W] -> [N, C, T, H, W] (a permuted view)"""
    tn, c, h, w = frames.shape
    return frames.reshape(tn // n, n, c, h, w).permute(1, 2, 0, 3, 4)


def _tap_slices(k: int, pt: int, n: int, total: int):
    """(input frame slice, output frame slice) of temporal tap k, or None if the tap never overlaps."""
    shift = (k - pt) * n                               # output frame t reads input frame t + (k - pt)
    if abs(shift) >= total:
        return None
    if shift == 0:
        return slice(None), slice(None)
    if shift < 0:
        return slice(None, shift), slice(-shift, None)
    return slice(shift, None), slice(None, -shift)


class _TemporalConvFrames(torch.autograd.Function):
    """conv3d ('same' zero padding in time) as kt 2-D convolutions on time-major frames, with a
    hand-written backward: each tap's data/weight gradient is one MIOpen call accumulated straight into
    slices of the result (autograd's own slice backward would materialise a zero-padded full-size tensor
    per tap, in NCHW, and trigger layout conversions). Backward is built from differentiable ops, so
    double backward (R1) works."""

    @staticmethod
    def forward(ctx, x, weight, n, padding_hw):
        kt = weight.shape[2]
        pt = kt // 2
        total = x.shape[0]
        taps = [_cl(weight[:, :, k]) for k in range(kt)]
        y = F.conv2d(x, taps[pt], padding=padding_hw)
        for k in range(kt):
            sl = _tap_slices(k, pt, n, total)
            if k == pt or sl is None:
                continue
            y[sl[1]] += F.conv2d(x[sl[0]], taps[k], padding=padding_hw)
        ctx.save_for_backward(x, weight)
        ctx.n, ctx.padding_hw = n, padding_hw
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        n, pad = ctx.n, list(ctx.padding_hw)
        kt = weight.shape[2]
        pt = kt // 2
        total = x.shape[0]
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gy = _cl(gy)
        gx = None
        gws = []
        for k in [pt] + [k for k in range(kt) if k != pt]:
            sl = _tap_slices(k, pt, n, total)
            wk = _cl(weight[:, :, k])
            if sl is None:
                gws.append((k, torch.zeros_like(wk)))
                continue
            gxi, gwi, _ = torch.ops.aten.convolution_backward(
                gy[sl[1]], x[sl[0]], wk, None, [1, 1], pad, [1, 1], False, [0, 0], 1, [need_x, need_w, False])
            if need_x:
                if gx is None:
                    gx = gxi                       # centre tap first: full frame range
                else:
                    gx[sl[0]] = gx[sl[0]] + gxi
            if need_w:
                gws.append((k, gwi))
        gw = None
        if need_w:
            gws.sort(key=lambda kv: kv[0])
            gw = torch.stack([g for _, g in gws], dim=2)
        return gx, gw, None, None


# The contraction as a node that can be differentiated any number of times ON THE HAND-WRITTEN KERNELS (round 4; R1 differentiates the
# discriminator's input gradient again, reference model/video_gan_lres.py:180-197). A convolution is linear in each argument, so the
# three passes are closed under differentiation: with  C(x, w) = conv,  D(g, w) = data gradient,  W(x, g) = weight gradient,
#     C' : (dx, dw)  = (D(gy, w), W(x, gy))        D' : (dg, dw) = (C(ggx, w), W(ggx, g))        W' : (dx, dg) = (D(g, ggw), C(x, ggw))
# and every right-hand side is again one launch of conv3d_igemm / conv3d_wgrad at the layer's own shapes. Round 3 sent the whole pass to
# the library's kt-convolution form instead (`second_order()`), 2.0 s per R1 update in the driver line. LVG_HAND_SECOND_ORDER=0 restores that.
HAND_SECOND_ORDER = os.environ.get('LVG_HAND_SECOND_ORDER', '1') == '1'


def _flip_w(weight: torch.Tensor) -> torch.Tensor:
    """The weight of the data gradient: taps mirrored, channel roles swapped."""
    return weight.flip(2, 3, 4).transpose(0, 1)


class _HandConv(torch.autograd.Function):
    """C(x, w): conv3d ('same' in time and space) on time-major frames, channels-last."""

    @staticmethod
    def forward(ctx, x, weight, n):
        ctx.save_for_backward(x, weight)
        ctx.n = n
        return conv3d_frames.conv3d_frames_forward(_cl(x), weight, n, keep_sum=False)[0]

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = _HandDgrad.apply(gy, weight, ctx.n) if ctx.needs_input_grad[0] else None
        gw = _HandWgrad.apply(x, gy, tuple(weight.shape[2:]), ctx.n) if ctx.needs_input_grad[1] else None
        return gx, gw, None


class _HandDgrad(torch.autograd.Function):
    """D(g, w): the data gradient of C = the convolution of g with the mirrored, transposed weight."""

    @staticmethod
    def forward(ctx, g, weight, n):
        ctx.save_for_backward(g, weight)
        ctx.n = n
        return conv3d_frames.conv3d_frames_forward(_cl(g), _flip_w(weight), n, keep_sum=False)[0]

    @staticmethod
    def backward(ctx, ggx):
        g, weight = ctx.saved_tensors
        dg = _HandConv.apply(ggx, weight, ctx.n) if ctx.needs_input_grad[0] else None
        dw = _HandWgrad.apply(ggx, g, tuple(weight.shape[2:]), ctx.n) if ctx.needs_input_grad[1] else None
        return dg, dw, None


class _HandWgrad(torch.autograd.Function):
    """W(x, g): the weight gradient [Co, Ci, kt, kh, kw] of C, in x's dtype (float32 accumulation, one rounding)."""

    @staticmethod
    def forward(ctx, x, g, taps, n):
        ctx.save_for_backward(x, g)
        ctx.n = n
        kt, kh, kw = taps
        xc, gc = _cl(x), _cl(g)
        if taps == (1, 1, 1) and x.is_cuda:
            # no taps: the pixel index is the whole K dimension (csrc/pointwise_wgrad.hip; conv3d_wgrad is built for 3 x 3 spatial taps)
            return conv3d_frames.pointwise_wgrad(xc, gc).to(x.dtype).reshape(g.shape[1], x.shape[1], 1, 1, 1)
        return conv3d_frames.conv3d_frames_wgrad(xc, gc, kt, kh, kw, n).to(x.dtype)

    @staticmethod
    def backward(ctx, ggw):
        x, g = ctx.saved_tensors
        ggw = ggw.to(x.dtype)
        dx = _HandDgrad.apply(g, ggw, ctx.n) if ctx.needs_input_grad[0] else None
        dg = _HandConv.apply(x, ggw, ctx.n) if ctx.needs_input_grad[1] else None
        return dx, dg, None, None


def _hand_second_order_takes(x: torch.Tensor, weight: torch.Tensor, padding_hw) -> bool:
    """All three passes of this layer on the hand-written kernels (the derivatives reuse exactly these shapes)? CPU tensors take the
    plain definitions behind the same nodes (tests)."""
    if not HAND_SECOND_ORDER or weight.dim() != 5 or tuple(padding_hw) != (weight.shape[3] // 2, weight.shape[4] // 2):
        return False
    if not x.is_cuda:
        return True
    co, ci = weight.shape[:2]
    if x.dtype not in (torch.float16, torch.bfloat16) or ci % 64 or co % 64 or weight.dtype != x.dtype:
        return False
    xc = _cl(x)
    if tuple(weight.shape[2:]) == (1, 1, 1):
        f, _, h, w = x.shape
        wgrad_ok = POINTWISE_WGRAD_HAND and (f * h * w) % 8 == 0 and conv3d_frames.pointwise_wgrad_splits(f * h * w, ci, co) > 0
    else:
        wgrad_ok = _hand_wgrad_shape_ok(xc, co, weight, padding_hw)
    return (_hand_conv_takes(xc, weight, padding_hw) and _hand_conv_shape_ok(xc, co, ci, weight, padding_hw)
            and wgrad_ok and conv3d_frames.supported(xc, weight))


class _PairView(torch.autograd.Function):
    """Pixels -> pixel pairs (see _pairable) as a node: a re-labelling of the same memory, so its gradient is the inverse re-labelling of the
    incoming gradient (and that one's gradient is this node again): no copy in either direction, any order of differentiation."""

    @staticmethod
    def forward(ctx, t):
        return _pair_view(t)

    @staticmethod
    def backward(ctx, g):
        return _UnpairView.apply(g)


class _UnpairView(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return _unpair_view(t)

    @staticmethod
    def backward(ctx, g):
        return _PairView.apply(g)


def _second_order_conv(x: torch.Tensor, weight: torch.Tensor, n: int, padding_hw) -> Optional[torch.Tensor]:
    """The contraction of an R1 pass on the hand-written kernels, as nodes that can be differentiated again -- or None (library route).
    Round 6: also the layers round 5 moved to their own kernels for the first-order passes -- 3-channel side (pointwise_thin), 32-channel
    layers as pixel pairs, 1 x 1 convolutions (weight gradient on pointwise_wgrad). Measured on the 8-clip R1 update before: 46 of 67 ms of
    device time in the library's double-backward of exactly these layers (profiles/r06_launch_sites_r1_before.log)."""
    if not HAND_SECOND_ORDER:
        return None
    taps = tuple(weight.shape[2:])
    if taps == (1, 1, 1) and THIN_POINTWISE and pointwise_thin.supported(x, weight[:, :, 0, 0, 0]):
        return pointwise_thin.pointwise_thin(x, weight[:, :, 0, 0, 0], twice=True)
    if _pairable(x, weight) and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last):
        xp, wp = _PairView.apply(x), pair_weight(weight)                 # (pair_weight: tensor expressions, differentiable as they are)
        if _hand_second_order_takes(xp, wp, padding_hw):
            return _UnpairView.apply(_HandConv.apply(xp, wp, n))
        return None
    if _hand_second_order_takes(x, weight, padding_hw):
        return _HandConv.apply(x, weight, n)
    return None


def temporal_conv_frames(x: torch.Tensor, weight: torch.Tensor, n: int, padding_hw) -> torch.Tensor:
    """conv3d with 'same' zero padding in time, as kt 2-D convolutions. x [(T N), Ci, H, W] and
    weight [Co, Ci, kt, kh, kw] in the compute dtype."""
    if POINTWISE_GEMM and tuple(weight.shape[2:]) == (1, 1, 1):
        return pointwise_conv(x, weight[:, :, 0, 0, 0])
    if (POINTWISE_HAND_ALL and POINTWISE_HAND and TAP_STACK and not SECOND_ORDER and tuple(weight.shape[2:]) == (1, 1, 1) and tuple(padding_hw) == (0, 0)
            and x.dim() == 4 and x.shape[1] > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last) and _hand_conv_takes(x, weight, (0, 0))):
        # every 1 x 1 convolution the hand-written kernels can take (the discriminator's skip convolutions reach this function, the generator's call
        # pointwise_conv themselves): forward + data gradient on conv3d_igemm, weight gradient on pointwise_wgrad
        return pointwise_conv(x, weight[:, :, 0, 0, 0])
    if THIN_POINTWISE and not SECOND_ORDER and tuple(weight.shape[2:]) == (1, 1, 1) and pointwise_thin.supported(x, weight[:, :, 0, 0, 0]):
        # 3-channel side (ToRGB, the discriminator's first layer): streaming kernels for all three passes (csrc/pointwise_thin.hip)
        return pointwise_thin.pointwise_thin(x, weight[:, :, 0, 0, 0])
    if SECOND_ORDER:
        y = _second_order_conv(x, weight, n, padding_hw)
        if y is not None:
            return y
    return _TemporalConvFrames.apply(x, weight, n, tuple(padding_hw))


# Tap-stacked form of the temporal convolution: ONE 2-D convolution with kt*Co output channels, the temporal
# sum folded into the epilogue kernel (csrc/tapconv_epilogue.hip) instead of kt - 1 accumulate passes.
# Measured on MI355X (bench.py, batch 8): 78.0 -> 75.3 ms/step (13.2k -> 13.6k frames/s); parity-green on CPU
# and GPU. On by default since round 2 (LVG_TAP_STACK=0 restores the kt-convolution form).
TAP_STACK = os.environ.get('LVG_TAP_STACK', '1') == '1'
# The style / demodulation / weight-normalisation chains of all generator layers depend only on the latents, not on
# the activations: issue them on a second HIP stream (forked after the latents, joined per layer by an event) so their
# ~1500 sub-10-us launches -- forward and, since autograd replays a node on its forward stream, backward -- run beside
# the convolutions instead of between them. Inside a captured hipGraph this becomes a parallel branch.
SIDE_STREAM_TERMS = os.environ.get('LVG_SIDE_STREAM_TERMS', '1') == '1'

# The block-final bias + activation (reference generator_lres.py:575) and the NEXT layer's input modulation (:101) + magnitude
# statistic (:574) as ONE pass over the block output (modconv_epilogue_dual: reads h, writes the activated tensor for the skip
# connection and the modulated one for the convolution; ToRGB needs the modulated one only), and ONE backward pass for the
# gradients of both (instead of modulate-backward + gradient sum + bias_act-backward). Needs the next layer's terms ahead of time
# (SIDE_STREAM_TERMS). LVG_FUSE_BOUNDARY=0 restores the separate passes.
FUSE_BOUNDARY = os.environ.get('LVG_FUSE_BOUNDARY', '1') == '1'


class Modulated:
    """A layer output that already went through the next layer's input modulation: `plain` = the activated tensor (None when the
    consumer does not read it), `mod` = plain * next modulation, `msq` = mean square of plain (or None)."""
    __slots__ = ('plain', 'mod', 'msq')

    def __init__(self, plain, mod, msq):
        self.plain, self.mod, self.msq = plain, mod, msq


# The dense contraction of the generator's 16-bit modulated convolutions on the hand-written implicit-GEMM kernel
# (csrc/conv3d_igemm.hip: temporal taps inside the K loop, epilogue fused on store) instead of MIOpen's igemm over
# tap-stacked output channels + the tap-gather epilogue kernel. LVG_HAND_CONV=0 restores the MIOpen route; shapes the
# kernel does not cover (float32, Ci or Co not a multiple of 64) take it anyway.
HAND_CONV = os.environ.get('LVG_HAND_CONV', '1') == '1'
HAND_CONV_DGRAD = os.environ.get('LVG_HAND_CONV_DGRAD', '1') == '1'      # data gradients on the same kernel
HAND_CONV_WGRAD = os.environ.get('LVG_HAND_CONV_WGRAD', '1') == '1'      # weight gradients on csrc/conv3d_wgrad.hip
# (round 4: 128 -> 64. The 3 x 4-pixel layers of the 8-clip step are 72 tiles; on the library they were the first operation of the generator's
# forward whose output differs from run to run on identical inputs -- its split-K kernel adds partial sums with atomics,
# profiles/r04_determinism_first_op.log -- and the step time is the same either way: 41.28 / 41.39 ms, profiles/r04_min_tiles_ab.log)
HAND_CONV_MIN_TILES = int(os.environ.get('LVG_HAND_CONV_MIN_TILES', '64'))


# Channel counts that are multiples of 32 but not of 64 (the first discriminator block: 32 -> 32, 32 -> 64 at 64 x 64 pixels) reach the
# hand-written kernels as PIXEL PAIRS: in channels-last memory two horizontally adjacent pixels of C channels ARE one pixel of 2 C channels
# of a frame half as wide (a view, no copy), and a 3 x 3 (1 x 1) convolution of the pixels is a 3 x 3 (1 x 1) convolution of the pairs with
# a [2 Co, 2 Ci] weight per tap in which half of the blocks are zero: output pixel 2 j + a reads input pixel 2 j + a + dw - 1 = pair
# j + dj, half b. These layers are memory-bound (77 GFLOP on 0.5 GB of activations), so the wasted multiplications are free; what it buys is
# the kernel's streaming rate instead of the library's small-channel kernels (0.43 / 0.89 ms forward / backward per layer, r02 profile).
HAND_PAIR = os.environ.get('LVG_HAND_PAIR', '1') == '1'


def _pairable(x: torch.Tensor, weight: torch.Tensor) -> bool:
    if not (HAND_CONV and HAND_PAIR and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.dim() == 4 and weight.dim() == 5):
        return False
    co, ci, _, kh, kw = weight.shape
    if not ((ci % 64 or co % 64) and ci % 32 == 0 and co % 32 == 0) or (kh, kw) not in ((3, 3), (1, 1)) or x.shape[3] % 2 or x.shape[1] != ci:
        return False
    return True


def _pair_view(t: torch.Tensor) -> torch.Tensor:
    """[F, C, H, W] channels-last -> [F, 2 C, H, W / 2] channels-last over the same memory."""
    t = _cl(t)
    f, c, h, w = t.shape
    return t.as_strided((f, 2 * c, h, w // 2), (h * w * c, 1, w * c, 2 * c))


def _unpair_view(t: torch.Tensor) -> torch.Tensor:
    t = _cl(t)                                  # (a dense channels-last buffer is what the strides below describe; a no-op for the kernels' own outputs)
    f, c2, h, w2 = t.shape
    return t.as_strided((f, c2 // 2, h, 2 * w2), (h * w2 * c2, 1, w2 * c2, c2 // 2))


def pair_weight(w: torch.Tensor) -> torch.Tensor:
    """[Co, Ci, kt, kh, kw] -> [2 Co, 2 Ci, kt, kh, kw] acting on pixel pairs (output rows a * Co + co, input columns b * Ci + ci):
    tap dw of output half a is tap dj + 1, input half b with 2 dj + b = a + dw - 1."""
    co, ci, kt, kh, kw = w.shape
    w2 = w.new_zeros(2 * co, 2 * ci, kt, kh, kw)
    if kw == 1:
        w2[:co, :ci] = w
        w2[co:, ci:] = w
        return w2
    w2[:co, ci:, :, :, 0] = w[..., 0]      # a = 0: dw = 0 -> pair j - 1, half 1
    w2[:co, :ci, :, :, 1] = w[..., 1]      #        dw = 1 -> pair j, half 0
    w2[:co, ci:, :, :, 1] = w[..., 2]      #        dw = 2 -> pair j, half 1
    w2[co:, :ci, :, :, 1] = w[..., 0]      # a = 1: dw = 0 -> pair j, half 0
    w2[co:, ci:, :, :, 1] = w[..., 1]      #        dw = 1 -> pair j, half 1
    w2[co:, :ci, :, :, 2] = w[..., 2]      #        dw = 2 -> pair j + 1, half 0
    return w2


def unpair_weight_grad(g2: torch.Tensor, co: int, ci: int) -> torch.Tensor:
    """Gradient of `pair_weight`: every weight element appears twice."""
    kw = g2.shape[4]
    if kw == 1:
        return g2[:co, :ci] + g2[co:, ci:]
    return torch.stack((g2[:co, ci:, :, :, 0] + g2[co:, :ci, :, :, 1],
                        g2[:co, :ci, :, :, 1] + g2[co:, ci:, :, :, 1],
                        g2[:co, ci:, :, :, 1] + g2[co:, :ci, :, :, 2]), dim=-1)


def _dup(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Per-channel terms of a paired tensor: both halves carry the same channels."""
    return None if t is None else torch.cat((t, t), dim=-1)


# float32 tensors on the hand-written kernels through split operands (conv3d_frames.conv3d_frames_split32): float32 accuracy on the 16-bit
# matrix cores at six (weight gradient: nine) times the multiplications -- the route of the float32 parity tests and of default-precision
# training (the reference trains lres in float32 with TF32 off); LVG_SPLIT_F32=0 restores the library's float32 convolution.
SPLIT_F32 = os.environ.get('LVG_SPLIT_F32', '1') == '1'


def _split32_takes(x: torch.Tensor, weight: torch.Tensor, padding_hw) -> bool:
    if not (HAND_CONV and SPLIT_F32 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32):
        return False
    if tuple(padding_hw) != (weight.shape[3] // 2, weight.shape[4] // 2) or tuple(weight.shape[2:]) == (1, 1, 1):
        return False
    return conv3d_frames.split32_supported(x, weight)


def _hand_conv_shape_ok(x: torch.Tensor, co: int, ci: int, weight: torch.Tensor, padding_hw) -> bool:
    """Would the hand-written kernel take the DATA GRADIENT of conv(x, weight): a convolution of a [frames, Co, H, W] gradient
    with the mirrored [Ci, Co, ...] weight (decided on shapes: the gradient tensor does not exist yet)."""
    if x.is_cuda and x.dtype == torch.float32:
        kt, kh, kw = weight.shape[2:]
        return (HAND_CONV and SPLIT_F32 and tuple(padding_hw) == (kh // 2, kw // 2)
                and conv3d_frames.split32_shape_ok(x.shape[0], x.shape[2], x.shape[3], co, ci, kt, kh, kw))
    if not (HAND_CONV and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)) or tuple(padding_hw) != (weight.shape[3] // 2, weight.shape[4] // 2):
        return False
    f, _, h, w = x.shape
    kt, kh, kw = weight.shape[2:]
    if (f * h * w) * co * kt * 2 >= 2 ** 32:
        return False
    return conv3d_frames.workgroups(f, h, w, co, ci, kt, kh, kw) >= HAND_CONV_MIN_TILES


def _hand_wgrad_shape_ok(x: torch.Tensor, co: int, weight: torch.Tensor, padding_hw) -> bool:
    if x.is_cuda and x.dtype == torch.float32:
        kt, kh, kw = weight.shape[2:]
        return (HAND_CONV and SPLIT_F32 and tuple(padding_hw) == (kh // 2, kw // 2) and x.shape[1] % 64 == 0 and co % 64 == 0
                and conv3d_frames.wgrad_splits(x.shape[0], x.shape[2], x.shape[3], 3 * x.shape[1], 3 * co, kt, kh, kw) > 0)
    if not (HAND_CONV and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)) or tuple(padding_hw) != (weight.shape[3] // 2, weight.shape[4] // 2):
        return False
    f, ci, h, w = x.shape
    kt, kh, kw = weight.shape[2:]
    return conv3d_frames._pixel_stride(x) == ci and conv3d_frames.wgrad_splits(f, h, w, ci, co, kt, kh, kw) > 0


def _hand_conv_takes(x: torch.Tensor, weight: torch.Tensor, padding_hw, cl: bool = True) -> bool:
    if not (HAND_CONV and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)):
        return False
    if tuple(padding_hw) != (weight.shape[3] // 2, weight.shape[4] // 2):
        return False
    if _pairable(x, weight):
        co, ci, kt, kh, kw = weight.shape
        if weight.dtype != x.dtype or _cl(x).data_ptr() % 16:          # (what conv3d_frames.supported checks on the paired views)
            return False
        return conv3d_frames.workgroups(x.shape[0], x.shape[2], x.shape[3] // 2, 2 * ci, 2 * co, kt, kh, kw) >= HAND_CONV_MIN_TILES
    if not conv3d_frames.supported(_cl(x) if cl else x, weight):
        return False
    # tiny layers (fewer tiles than HAND_CONV_MIN_TILES) stay on MIOpen
    return conv3d_frames.workgroups(x.shape[0], x.shape[2], x.shape[3], x.shape[1], weight.shape[0], *weight.shape[2:]) >= HAND_CONV_MIN_TILES


SECOND_ORDER = False      # True inside `second_order()`: layers must build a graph that can be differentiated twice


@contextlib.contextmanager
def second_order():
    """Scope for passes whose gradient is differentiated again (R1): the tap-stacked convolution implements
    first-order gradients only, so layers fall back to the kt-convolution form inside this scope."""
    global SECOND_ORDER
    prev, SECOND_ORDER = SECOND_ORDER, True
    try:
        yield
    finally:
        SECOND_ORDER = prev


# 1x1 convolutions as plain GEMMs on the channels-last pixel matrix [(frames H W), Ci] (hipBLASLt) instead of MIOpen's
# implicit-GEMM + its split-K zero-fill / cast helpers. Opt-in (LVG_POINTWISE_GEMM=1) until measured on MI355X.
POINTWISE_GEMM = os.environ.get('LVG_POINTWISE_GEMM', '0') == '1'
# 1x1 (skip) convolutions through the hand-written kernel: forward + data gradient there (no zero-fill / cast helper launches of the
# library's split-K kernels), weight gradient on the library. Same run, LVG_POINTWISE_HAND=1/0: 43.45 / 43.7 ms per step.
POINTWISE_HAND = os.environ.get('LVG_POINTWISE_HAND', '1') == '1'
# their weight gradient as one GEMM over the pixel matrices: measured SLOWER (50.4 vs 44.3 ms per step, gpurun_out/r03_pair_ab.log: the
# library GEMM on a [Co, pixels] x [pixels, Ci] product with 10^5 .. 10^6 pixels); kept as a switch
POINTWISE_WGRAD_GEMM = os.environ.get('LVG_POINTWISE_WGRAD_GEMM', '0') == '1'
# weight gradient of the 1 x 1 convolutions that run forward + data gradient on the hand-written kernel: the K-loop of the weight-gradient kernel
# without taps (csrc/pointwise_wgrad.hip) instead of the library's split-K kernels and their helper launches
POINTWISE_WGRAD_HAND = os.environ.get('LVG_POINTWISE_WGRAD_HAND', '1') == '1'
POINTWISE_HAND_ALL = os.environ.get('LVG_POINTWISE_HAND_ALL', '1') == '1'      # ... also for the 1 x 1 convolutions that arrive through temporal_conv_frames
# 1 x 1 convolutions with a 3-channel side on the streaming kernels of csrc/pointwise_thin.hip instead of the library's implicit-GEMM kernels
THIN_POINTWISE = os.environ.get('LVG_THIN_POINTWISE', '1') == '1'


def pointwise_conv(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """conv2d with a [Co, Ci] / [Co, Ci, 1, 1] weight on frames [(T N), Ci, H, W]; result in x's memory format."""
    w2 = weight.reshape(weight.shape[0], weight.shape[1])
    cl = x.shape[1] > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last)
    if POINTWISE_HAND and cl and not SECOND_ORDER and TAP_STACK and _hand_conv_takes(x, w2[:, :, None, None, None], (0, 0)):
        # forward and data gradient on the hand-written kernel (no zero-fill / cast helper launches), weight gradient on the library
        return _TapConvEpilogue.apply(x, w2[:, :, None, None, None], None, None, None, None, 1, (0, 0), 'linear', None, False)[0]
    if (POINTWISE_GEMM or x.dtype == torch.float32) and cl:
        # views: [F, H, W, C] is the memory order. float32 tensors ALWAYS take this form: the library's float32 channels-last 1 x 1
        # convolution picks its backward-data algorithm by a timed search, and one of the candidates is 1e-2 off (measured: the gradient
        # of the network input against float64 flips between 2e-4 and 9e-3 from run to run, profiles/r03_f32_grad_flaky.log)
        return torch.matmul(x.permute(0, 2, 3, 1), w2.t()).permute(0, 3, 1, 2)
    return F.conv2d(x, _cl(w2[:, :, None, None]))


def stack_taps(weight: torch.Tensor) -> torch.Tensor:
    """[Co, Ci, kt, kh, kw] -> [kt*Co, Ci, kh, kw], tap-major along the output channels."""
    co, ci, kt, kh, kw = weight.shape
    return weight.permute(2, 0, 1, 3, 4).reshape(kt * co, ci, kh, kw)


class _TapConvEpilogue(torch.autograd.Function):
    """conv3d ('same' in time) + modulated-conv epilogue on time-major frames:
    out = clamp(act(conv(x, w) * pre + b + res) * gain) * post. First-order gradients only (generator path)."""

    @staticmethod
    def forward(ctx, x, weight, pre, b, res, post, n, padding_hw, act, clamp, want_msq):
        kt = weight.shape[2]
        # a bare temporal sum (no scale / bias / activation) IS its own saved sum: nothing extra is written
        plain = pre is None and b is None and res is None and post is None and act == 'linear' and clamp is None
        if not any(ctx.needs_input_grad):
            plain = True                                    # inference: no backward pass will read the saved sum -- do not write it
        pair = _pairable(x, weight) and _hand_conv_takes(x, weight, padding_hw)
        if pair:
            # pixel pairs (see _pairable): the same launch on views with twice the channels and half the width
            out, ysum, msq = conv3d_frames.conv3d_frames_forward(_pair_view(x), pair_weight(weight), n, _dup(pre), _dup(b),
                                                                 None if res is None else _pair_view(res), _dup(post), act=act, clamp=clamp,
                                                                 want_msq=want_msq, keep_sum=not plain)
            out = _unpair_view(out)
            ysum = None if ysum is None else _unpair_view(ysum)
        elif _split32_takes(x, weight, padding_hw):
            # float32 tensors: the same kernel on split 16-bit operands, float32 accumulators stored unrounded
            out, ysum, msq = conv3d_frames.conv3d_frames_split32(x, weight, n, pre, b, res, post, act=act, clamp=clamp,
                                                                 want_msq=want_msq, keep_sum=not plain)
        elif _hand_conv_takes(x, weight, padding_hw):
            # contraction, temporal sum and epilogue in ONE hand-written MFMA kernel (csrc/conv3d_igemm.hip)
            out, ysum, msq = conv3d_frames.conv3d_frames_forward(_cl(x), weight, n, pre, b, res, post, act=act, clamp=clamp,
                                                                 want_msq=want_msq, keep_sum=not plain)
        else:
            wst = _cl(stack_taps(weight))
            z = _cl(F.conv2d(_cl(x), wst, padding=padding_hw))
            out, ysum, msq = tap_gather_forward(z, pre, b, res, post, kt, n, act=act, clamp=clamp, want_msq=want_msq, keep_sum=not plain)
        ctx.save_for_backward(x, weight, out if plain else ysum, pre, b, res, post)
        ctx.cfg = (n, list(padding_hw), act, clamp)
        ctx.plain = pre is None and b is None and res is None and post is None and act == 'linear' and clamp is None
        ctx.wt = getattr(weight, '_lvg_dgrad', None)        # weight_prep's data-gradient packing of this weight, if it made one
        ctx.pair = pair
        if want_msq:
            ctx.mark_non_differentiable(msq)
        return out, msq

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout, _dmsq):
        x, weight, ysum, pre, b, res, post = ctx.saved_tensors
        n, pad, act, clamp = ctx.cfg
        need = ctx.needs_input_grad
        if ctx.pair:
            # the whole backward pass on the paired views; gradients of the duplicated per-channel terms fold their two halves
            co1, ci1 = weight.shape[:2]
            grads = _TapConvEpilogue._backward(ctx, _pair_view(x), pair_weight(weight), _pair_view(ysum), _dup(pre), _dup(b),
                                               None if res is None else _pair_view(res), _dup(post), _pair_view(dout), None, need)
            gx, gw, d_pre, d_b, d_res, d_post = grads
            fold = lambda t, c: None if t is None else t[..., :c] + t[..., c:]
            return ((None if gx is None else _unpair_view(gx)), (None if gw is None else unpair_weight_grad(gw, co1, ci1)), fold(d_pre, co1),
                    fold(d_b, co1), (None if d_res is None else dout), fold(d_post, co1), None, None, None, None, None)
        grads = _TapConvEpilogue._backward(ctx, x, weight, ysum, pre, b, res, post, dout, ctx.wt, need)
        return (*grads, None, None, None, None, None)

    @staticmethod
    def _backward(ctx, x, weight, ysum, pre, b, res, post, dout, wt_packed, need):
        n, pad, act, clamp = ctx.cfg
        co, ci, kt, kh, kw = weight.shape
        xc = _cl(x)
        hand_d = need[0] and HAND_CONV_DGRAD and _hand_conv_shape_ok(xc, co, ci, weight, pad)
        hand_w = need[1] and HAND_CONV_WGRAD and _hand_wgrad_shape_ok(xc, co, weight, pad)
        # frames of a few pixels (the two 3 x 4-pixel layers): the weight-gradient kernel cannot walk them (its K-step of 64 / W image rows allows two frame
        # changes), so the taps are unrolled into channels -- cheap at this size -- and the gradient is ONE pixel-index GEMM on pointwise_wgrad
        small_w = need[1] and not hand_w and _small_frame_wgrad_takes(xc, co, weight, pad)
        stacked = (need[0] and not hand_d) or (need[1] and not hand_w and not small_w)      # MIOpen needs the gradient scattered over the taps
        if ctx.plain and (kt == 1 or not stacked):
            dz, d_pre, d_post, d_sum = dout.contiguous(memory_format=torch.channels_last), None, None, None     # a bare convolution: nothing to undo
        else:
            dz, d_pre, d_post, d_sum = tap_gather_backward(dout, ysum, pre, b, res, post, kt if stacked else 1, n, act=act, clamp=clamp)
        dy = dz[:, (kt // 2) * co:(kt // 2 + 1) * co] if stacked else dz     # the centre tap of the scattered gradient IS the gradient of the sum
        gx = gw = None
        if hand_d:
            # data gradient on the hand-written kernel: the same convolution with the taps mirrored and the channel roles swapped
            wt = wt_packed
            if dy.dtype == torch.float32:
                gx = conv3d_frames.conv3d_frames_split32_dgrad(dy, weight, n)
            elif wt is not None and tuple(wt.shape) == (kt, kh, kw, ci, co) and wt.dtype == dy.dtype:
                gx = conv3d_frames.conv3d_frames_forward(dy, wt.permute(3, 4, 0, 1, 2), n, keep_sum=False, packed=wt)[0]
            else:
                gx = conv3d_frames.conv3d_frames_forward(dy, weight.flip(2, 3, 4).transpose(0, 1), n, keep_sum=False)[0]
        if hand_w and dy.dtype == torch.float32:
            gw = conv3d_frames.conv3d_frames_split32_wgrad(xc, dy, kt, kh, kw, n)
        elif hand_w:
            gw = conv3d_frames.conv3d_frames_wgrad(xc, dy, kt, kh, kw, n).to(weight.dtype)
        if small_w:
            gw = small_frame_wgrad(xc, dy, kt, kh, kw, n).to(weight.dtype)
            hand_w = True
        if need[1] and not hand_w and POINTWISE_WGRAD_HAND and kt == kh == kw == 1 and conv3d_frames.pointwise_wgrad_supported(xc, dy):
            # weight gradient of a 1 x 1 convolution on the matrix cores with the pixel index as K (csrc/pointwise_wgrad.hip): no zero-fill /
            # cast helper launches of the library's split-K kernels
            gw = conv3d_frames.pointwise_wgrad(xc, dy).to(weight.dtype).reshape(co, ci, 1, 1, 1)
            hand_w = True
            stacked = need[0] and not hand_d
        if stacked and POINTWISE_WGRAD_GEMM and kt == kh == kw == 1 and not (need[0] and not hand_d) and conv3d_frames._pixel_stride(xc) == ci:
            # weight gradient of a 1 x 1 convolution = one GEMM over the pixel matrices (views of the channels-last tensors): no zero-fill /
            # cast helper launches of the library's split-K convolution kernels; float32 accumulation, one rounding to the compute dtype
            gw = torch.matmul(_cl(dy).permute(0, 2, 3, 1).reshape(-1, co).t(), xc.permute(0, 2, 3, 1).reshape(-1, ci)).reshape(co, ci, 1, 1, 1)
            stacked = False
        if stacked:
            gxm, gwst, _ = torch.ops.aten.convolution_backward(
                _cl(dz), xc, _cl(stack_taps(weight)), None, [1, 1], pad, [1, 1], False, [0, 0], 1, [need[0] and not hand_d, need[1] and not hand_w, False])
            if need[0] and not hand_d:
                gx = gxm
            if need[1] and not hand_w:
                gw = gwst.reshape(kt, co, ci, kh, kw).permute(1, 2, 0, 3, 4)
        d_b = d_sum.sum(dim=0).to(b.dtype) if (b is not None and need[3]) else None
        d_res = None
        if res is not None and need[4]:
            assert act == 'linear' and clamp is None and post is None, 'residual gradient: linear epilogue only'
            d_res = dout
        return gx, gw, (d_pre if need[2] else None), d_b, d_res, (d_post if need[5] else None)


# Measured SLOWER than the library's tap-stacked weight gradient (40.33 against 40.13 ms per step, profiles/r05_small_frame_ab.log: 1 728 tiles of 36 K-steps
# with four MFMAs each) and left switched off; exact (tests/test_pointwise_thin.py).
SMALL_FRAME_WGRAD = os.environ.get('LVG_SMALL_FRAME_WGRAD', '0') == '1'
SMALL_FRAME_PIXELS = 16          # frames up to this many pixels take the unrolled form (27 x the activation bytes: 64 MB for the 3 x 4-pixel layers)


def _small_frame_wgrad_takes(x: torch.Tensor, co: int, weight: torch.Tensor, padding_hw) -> bool:
    if not (SMALL_FRAME_WGRAD and POINTWISE_WGRAD_HAND and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)):
        return False
    f, ci, h, w = x.shape
    kt, kh, kw = weight.shape[2:]
    if h * w > SMALL_FRAME_PIXELS or tuple(padding_hw) != (kh // 2, kw // 2) or kh % 2 == 0 or kw % 2 == 0 or kt % 2 == 0 or kt * kh * kw == 1:
        return False
    return ci % 64 == 0 and co % 64 == 0 and (f * h * w) % 8 == 0 and kt * kh * kw * ci <= 65536


def unroll_taps(x: torch.Tensor, kt: int, kh: int, kw: int, n: int) -> torch.Tensor:
    """x [(T N), Ci, H, W] -> [(T N), kt kh kw Ci, H, W] channels-last: channel block (dt, dh, dw) holds x shifted by that tap ('same' zero padding in
    rows and columns, zero frames outside the clip: a temporal tap is a shift by whole time steps = n frames)."""
    f, ci, h, w = x.shape
    pt = kt // 2
    xp = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2))
    blocks = []
    for dt in range(kt):
        s = (dt - pt) * n                                             # source frame = frame + s
        if abs(s) >= f:
            xs = torch.zeros_like(xp)
        elif s > 0:
            xs = torch.cat((xp[s:], torch.zeros_like(xp[:s])), dim=0)
        elif s < 0:
            xs = torch.cat((torch.zeros_like(xp[:-s]), xp[:f + s]), dim=0)
        else:
            xs = xp
        for dh in range(kh):
            for dw in range(kw):
                blocks.append(xs[:, :, dh:dh + h, dw:dw + w])
    return torch.cat(blocks, dim=1).contiguous(memory_format=torch.channels_last)


def small_frame_wgrad(x: torch.Tensor, dy: torch.Tensor, kt: int, kh: int, kw: int, n: int) -> torch.Tensor:
    """Weight gradient [Co, Ci, kt, kh, kw] (float32) of the frames convolution for frames of a few pixels: gw = dy^T . unroll_taps(x)."""
    ci, co = x.shape[1], dy.shape[1]
    gw = conv3d_frames.pointwise_wgrad(unroll_taps(_cl(x), kt, kh, kw, n), dy)          # [Co, kt kh kw Ci]
    return gw.reshape(co, kt, kh, kw, ci).permute(0, 4, 1, 2, 3)


def temporal_conv_epilogue(x: torch.Tensor, weight: torch.Tensor, n: int, padding_hw, pre: Optional[torch.Tensor] = None,
                           b: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, post: Optional[torch.Tensor] = None,
                           act: str = 'linear', clamp: Optional[float] = None, want_msq: bool = False):
    """conv3d with 'same' zero padding in time followed by the fused epilogue
    `clamp(act(y * pre + b + res) * gain) * post` (pre / post float32 [(T N), C], res like the output).
    Returns `out` or `(out, mean_square)`. With TAP_STACK the kt taps are one convolution and their sum is
    taken inside the epilogue kernel; otherwise kt convolutions are accumulated and the epilogue runs on the sum."""
    fused = weight.shape[2] > 1 or ((_hand_conv_takes(x, weight, padding_hw) or _split32_takes(x, weight, padding_hw)) and tuple(weight.shape[2:]) != (1, 1, 1))
    if TAP_STACK and not SECOND_ORDER and fused and (res is None or (act == 'linear' and clamp is None and post is None)):
        out, msq = _TapConvEpilogue.apply(x, weight, pre, b, res, post, n, tuple(padding_hw), act, clamp, bool(want_msq))
        return (out, msq) if want_msq else out
    y = temporal_conv_frames(x, weight, n, padding_hw)
    if res is not None:
        assert b is None and post is None and act == 'linear' and clamp is None and not want_msq
        return torch.addcmul(res, y, pre.to(y.dtype).reshape(y.shape[0], -1, 1, 1))
    if pre is None and post is None and not want_msq:
        if b is None and act == 'linear' and clamp is None:
            return y
        return bias_act.bias_act(y, b, act=act, clamp=clamp)          # differentiable to second order (discriminator, R1)
    return modconv_epilogue(y, pre=pre, b=b, post=post, act=act, clamp=clamp, want_msq=want_msq)


class _CropFrames(torch.autograd.Function):
    """Centre crop of frames / rows / columns whose backward writes the gradient into a zero tensor of the
    INPUT's memory format (autograd's slice backward always produces NCHW-contiguous zeros)."""

    @staticmethod
    def forward(ctx, x, f0, f1, y0, y1, x0, x1):
        ctx.shape, ctx.box = x.shape, (f0, f1, y0, y1, x0, x1)
        ctx.cl = x.dim() == 4 and x.shape[1] > 1 and x.stride(1) == 1
        out = x[f0:f1, :, y0:y1, x0:x1]
        return out.contiguous(memory_format=torch.channels_last if ctx.cl else torch.contiguous_format)

    @staticmethod
    def backward(ctx, g):
        f0, f1, y0, y1, x0, x1 = ctx.box
        full = torch.empty(ctx.shape, dtype=g.dtype, device=g.device,
                           memory_format=torch.channels_last if ctx.cl else torch.contiguous_format).zero_()
        full[f0:f1, :, y0:y1, x0:x1] = g
        return full, None, None, None, None, None, None


def crop_frames(x: torch.Tensor, n: int, seq_length: Optional[int] = None, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
    """Centred crop of [(T N), C, H, W] in time / height / width; no-op (same tensor) when nothing is cut."""
    tn, _, h, w = x.shape
    f0, f1, y0, y1, x0, x1 = 0, tn, 0, h, 0, w
    if seq_length is not None:
        t0 = (tn // n - seq_length) // 2
        f0, f1 = t0 * n, (t0 + seq_length) * n
    if height is not None:
        y0 = (h - height) // 2
        y1 = y0 + height
    if width is not None:
        x0 = (w - width) // 2
        x1 = x0 + width
    if (f0, f1, y0, y1, x0, x1) == (0, tn, 0, h, 0, w):
        return x
    return _CropFrames.apply(x, f0, f1, y0, y1, x0, x1)


def resample_time_frames(x: torch.Tensor, taps: torch.Tensor, n: int, up: int = 1, down: int = 1) -> torch.Tensor:
    """x2 up/down-sampling along time of [(T N), C, H, W]: the frame axis is H of a [1, 1, T, (N C H W)] view."""
    tn, c, h, w = x.shape
    nhwc = x.dim() == 4 and c > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last)
    rows = (x.permute(0, 2, 3, 1) if nhwc else x).reshape(1, 1, tn // n, n * c * h * w)
    if up > 1:
        y = upfirdn2d.upsample2d(rows, taps.unsqueeze(1), up=(1, up))
    else:
        y = upfirdn2d.downsample2d(rows, taps.unsqueeze(1), down=(1, down))
    if nhwc:
        return y.reshape(y.size(2) * n, h, w, c).permute(0, 3, 1, 2)
    return y.reshape(y.size(2) * n, c, h, w)


def modulated_conv_frames(x: torch.Tensor, weight: torch.Tensor, style: torch.Tensor, input_gain: Optional[torch.Tensor],
                          padding, demodulate: bool, compute_dtype: torch.dtype, return_demod: bool = False):
    """`modulated_conv3d` in time-major frames layout. x [(T N), Ci, H, W]; style [T, N, Ci] float32."""
    t, n, ci = style.shape
    if demodulate:
        weight = weight / weight.abs().amax(dim=(1, 2, 3, 4), keepdim=True)
        style = style / style.abs().amax(dim=(0, 2), keepdim=True)       # per sample, over channels and time
    weight = weight * (1.0 / math.sqrt(weight[0].numel()))
    mod = style if input_gain is None else style * input_gain
    x = x * mod.reshape(t * n, ci, 1, 1).to(x.dtype)
    y = temporal_conv_frames(x.to(compute_dtype), weight.to(compute_dtype), n, padding[1:])
    if demodulate:
        w2 = weight.square().sum(dim=(2, 3, 4))                               # [Co, Ci]
        demod = torch.matmul(style.square(), w2.t()).add(1e-8).rsqrt()        # [T, N, Co]
        if return_demod:
            return y, demod.reshape(t * n, -1, 1, 1)                          # caller fuses the multiply
        y = y * demod.reshape(t * n, -1, 1, 1).to(y.dtype)
    return y


# The weight side of `modulation_terms` (max normalisation, 1 / sqrt(fan_in), sum of squares over the taps, cast to the compute
# dtype in the hand-written convolution's layout) as ONE HIP pass per direction (torch_utils/ops/weight_prep.py) instead of eight
# tensor passes forward and about twice that backward per layer and step. LVG_WEIGHT_PREP=0 restores the tensor expressions.
WEIGHT_PREP = os.environ.get('LVG_WEIGHT_PREP', '1') == '1'
# The style side (per-sample max normalisation, demodulation rsqrt(w2 . s^2 + 1e-8)) likewise (torch_utils/ops/style_prep.py):
# ~8 launches forward and ~25 backward per layer become 2 + 3. LVG_STYLE_PREP=0 restores the tensor expressions.
STYLE_PREP = os.environ.get('LVG_STYLE_PREP', '1') == '1'
# The temporal noise filter bank on the float32 matrix cores (csrc/noise_bank.hip) instead of one dense product over materialised windows.
NOISE_BANK_HIP = os.environ.get('LVG_NOISE_BANK', '1') == '1'


def modulation_terms(weight: torch.Tensor, style: torch.Tensor, demodulate: bool, dtype: Optional[torch.dtype] = None):
    """Small-tensor side of a modulated convolution in frames layout.

    weight [Co, Ci, kt, kh, kw], style [T, N, Ci] float32 -> (scaled weight, per-frame modulation
    [(T N), Ci], per-frame demodulation [(T N), Co] or None); same normalisations as
    `modulated_conv_frames`, so conv(x * modulation, weight) * demodulation is the modulated conv.
    With a 16-bit `dtype` the weight comes back already in that dtype."""
    t, n, ci = style.shape
    scale = 1.0 / math.sqrt(weight[0].numel())
    if WEIGHT_PREP and dtype is not None and demodulate and weight_prep.supported(weight, dtype):
        weight, w2 = weight_prep.weight_prep(weight, scale, True, dtype, want_w2=True)
        if STYLE_PREP and style_prep.supported(style, w2):
            mod, demod = style_prep.style_prep(style, w2)          # max normalisation + demodulation: 2 launches, 3 backward
            return weight, mod, demod
        style = style / style.abs().amax(dim=(0, 2), keepdim=True)
    else:
        if demodulate:
            style = style / style.abs().amax(dim=(0, 2), keepdim=True)
            weight = weight / weight.abs().amax(dim=(1, 2, 3, 4), keepdim=True)
        weight = weight * scale
        w2 = weight.square().sum(dim=(2, 3, 4)) if demodulate else None
        if dtype is not None:
            weight = weight.to(dtype)
    demod = None
    if demodulate:
        demod = torch.matmul(style.square(), w2.t()).add(1e-8).rsqrt().reshape(t * n, -1)
    return weight, style.reshape(t * n, ci), demod


# --------------------------------------------------------------------------------------------------
# Generator.

class BlurredNoise(nn.Module):
    """Temporal embedding: white noise low-passed by a bank of Kaiser filters of geometrically
    spaced bandwidths (reference :323-391). Output [N, channels, T]."""

    def __init__(self, channels: int = 1024, min_sampling_rate: float = 250, max_sampling_rate: float = 10000,
                 blur_widths: int = 128, cutoff: float = 2.0, width: float = 12.0, sampling_rate_base: float = 2.0,
                 normalize_per_filter: float = 1.0):
        super().__init__()
        assert channels % blur_widths == 0
        self.channels, self.blur_widths = channels, blur_widths
        self.normalize_per_filter = normalize_per_filter
        self.noise_channels = channels // blur_widths
        self.kernel_size = int(np.ceil(max_sampling_rate / 2))
        if sampling_rate_base > 1:
            lo, hi = math.log(min_sampling_rate, sampling_rate_base), math.log(max_sampling_rate, sampling_rate_base)
            rates = np.clip(sampling_rate_base ** np.linspace(lo, hi, blur_widths), min_sampling_rate, max_sampling_rate)
        else:
            rates = np.linspace(min_sampling_rate, max_sampling_rate, blur_widths)
        bank = torch.zeros(blur_widths, self.kernel_size)
        for i, rate in enumerate(rates):
            taps = int(np.ceil(rate / 2))
            bank[i, -taps:] = torch.as_tensor(scipy.signal.firwin(numtaps=taps, cutoff=cutoff, width=width, fs=rate), dtype=torch.float32)
        if normalize_per_filter > 0:
            self.register_buffer('output_scale', (bank ** 2).sum(dim=1).rsqrt().reshape(1, -1, 1))
        self.register_buffer('blur_filters', bank.unsqueeze(1))
        self._packed_bank = None          # torch_utils.ops.noise_bank.PackedBank, built on first use (not part of the state)

    def forward(self, batch_size: int, seq_length: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(batch_size, self.noise_channels, seq_length + self.kernel_size - 1,
                            device=self.blur_filters.device, generator=generator)
        return self.blur(noise)

    def blur(self, noise: torch.Tensor) -> torch.Tensor:
        n, c, t = noise.shape
        assert c == self.noise_channels
        bank = self.blur_filters[:, 0, :]
        rows = noise.reshape(n * c, t)
        if NOISE_BANK_HIP and noise_bank.supported(rows, bank):
            # float32 GPU tensors: the staircase bank on the float32 matrix cores, sliding windows read out of the noise row (csrc/noise_bank.hip)
            if self._packed_bank is None:
                self._packed_bank = noise_bank.PackedBank()
            scale = None
            if self.normalize_per_filter > 0:
                scale = (1 + self.normalize_per_filter * (self.output_scale - 1)).reshape(-1)
            y = noise_bank.noise_filter_bank(rows, bank, self._packed_bank.get(bank), scale)
            return y.reshape(n, c * self.blur_widths, y.size(2))
        # The filter bank as one GEMM over sliding windows of the noise: [n c, T_out, K] @ [K, widths].
        # (The reference's grouped conv1d with 5000-tap kernels lands on MIOpen's naive direct kernel.)
        win = rows.unfold(1, self.kernel_size, 1)
        y = torch.matmul(win, self.blur_filters[:, 0, :].t()).transpose(1, 2)
        if self.normalize_per_filter > 0:
            y = y * (1 + self.normalize_per_filter * (self.output_scale - 1))
        return y.reshape(n, c * self.blur_widths, y.size(2))


class LatentMappingNetwork(nn.Module):
    def __init__(self, temporal_emb_dim: int = 1024, latent_w_dim: int = 1024, num_layers: int = 2,
                 activation: str = 'lrelu', lrate_mul: float = 0.01, normalize_input: bool = True):
        super().__init__()
        self.temporal_emb_dim, self.latent_w_dim = temporal_emb_dim, latent_w_dim
        self.normalize_input = normalize_input
        self.layer_names = []
        for i in range(num_layers):
            name = f'layer_{i}'
            setattr(self, name, FullyConnectedLayer(temporal_emb_dim if i == 0 else latent_w_dim, latent_w_dim,
                                                    activation=activation, lrate_mul=lrate_mul))
            self.layer_names.append(name)

    def forward(self, emb: torch.Tensor) -> torch.Tensor:
        n, c, t = emb.shape
        if self.normalize_input:
            emb = emb * emb.square().mean(dim=1, keepdim=True).add(1e-8).rsqrt()
        h = emb.permute(0, 2, 1).reshape(n * t, c)
        for name in self.layer_names:
            h = getattr(self, name)(h)
        return h.reshape(n, t, -1).permute(0, 2, 1)   # [N, C, T] view with unit channel stride


class Synthesis3dResBlock(nn.Module):
    """Two style-modulated conv3d + 1x1x1 skip, optional x2 temporal and/or spatial upsampling."""

    def __init__(self, latent_dim: int, in_channels: int, out_channels: Optional[int] = None,
                 out_width: Optional[int] = None, out_height: Optional[int] = None,
                 temporal_ksize: int = 1, spatial_ksize: int = 1, temporal_up: bool = False, spatial_up: bool = False,
                 activation: str = 'lrelu', activation_clamp: Optional[float] = 256.0, magnitude_ema: bool = True):
        super().__init__()
        self.latent_dim, self.in_channels = latent_dim, in_channels
        self.out_channels = out_channels or in_channels
        self.out_width, self.out_height = out_width, out_height
        self.temporal_up, self.spatial_up = temporal_up, spatial_up
        self.activation, self.activation_clamp = activation, activation_clamp
        self.magnitude_ema = magnitude_ema
        self.use_float16 = False
        kshape = (temporal_ksize, spatial_ksize, spatial_ksize)
        self.padding = tuple(k // 2 for k in kshape)
        self.affine_0 = FullyConnectedLayer(latent_dim, in_channels, bias_init=1.0)
        self.affine_1 = FullyConnectedLayer(latent_dim, in_channels, bias_init=1.0)
        self.weight_0 = nn.Parameter(torch.randn(in_channels, in_channels, *kshape))
        self.weight_1 = nn.Parameter(torch.randn(self.out_channels, in_channels, *kshape))
        self.weight_skip = nn.Parameter(torch.randn(self.out_channels, in_channels, 1, 1, 1))
        self.weight_skip_gain = 1 / math.sqrt(in_channels)
        self.bias_0 = nn.Parameter(torch.zeros(in_channels))
        self.bias_1 = nn.Parameter(torch.zeros(self.out_channels))
        if magnitude_ema:
            self.input_magnitude_ema_0 = MagnitudeEMA()
            self.input_magnitude_ema_1 = MagnitudeEMA()
        if temporal_up:
            self.temporal_upsample = TemporalLinearUpsample()
        if spatial_up:
            self.spatial_upsample = SpatialBilinearUpsample()

    def _styles(self, affine: FullyConnectedLayer, latent: torch.Tensor) -> torch.Tensor:
        n, c, t = latent.shape
        return affine(latent.permute(0, 2, 1).reshape(n * t, c)).reshape(n, t, -1).permute(0, 2, 1)

    def forward(self, x: torch.Tensor, latent: torch.Tensor, magnitude_ema_beta: float = 1.0,
                out_seq_length: Optional[int] = None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        if dtype is None:
            dtype = torch.float16 if (self.use_float16 and x.is_cuda) else torch.float32
        x = x.to(dtype)
        style_0 = self._styles(self.affine_0, latent)
        gain_0 = self.input_magnitude_ema_0(x, magnitude_ema_beta) if self.magnitude_ema else None
        if gain_0 is not None:
            x = x * gain_0.to(dtype)
        h = modulated_conv3d(x, self.weight_0, style_0, None, self.padding, True, dtype)
        h = bias_act.bias_act(h, self.bias_0.to(dtype), act=self.activation, clamp=self.activation_clamp)

        style_1 = self._styles(self.affine_1, latent)
        gain_1 = self.input_magnitude_ema_1(h, magnitude_ema_beta) if self.magnitude_ema else None
        h = modulated_conv3d(h, self.weight_1, style_1, gain_1, self.padding, True, dtype)

        skip = _dense_conv3d(x, (self.weight_skip * self.weight_skip_gain).to(dtype), (0, 0, 0))
        h = (skip + h) * SQRT_HALF

        if self.temporal_up:
            h = self.temporal_upsample(h)
        h = crop_center(h, seq_length=out_seq_length)
        if self.spatial_up:
            h = self.spatial_upsample(h)
        h = crop_center(h, width=self.out_width, height=self.out_height)
        return bias_act.bias_act(h, self.bias_1.to(dtype), act=self.activation, clamp=self.activation_clamp)


    def frame_terms(self, latent: torch.Tensor, dtype: torch.dtype):
        """Everything `forward_frames` needs that depends on the latent only: (weight, modulation, demodulation) of both
        convolutions, weights already in the compute dtype."""
        n, c, t = latent.shape
        lat = latent.permute(2, 0, 1).reshape(t * n, c)                         # rows ordered (t n), like the frames
        w0, mod_0, demod_0 = modulation_terms(self.weight_0, self.affine_0(lat).reshape(t, n, -1), True, dtype)
        w1, mod_1, demod_1 = modulation_terms(self.weight_1, self.affine_1(lat).reshape(t, n, -1), True, dtype)
        # parameter casts / constant scales: small launches that do not depend on the activations -- they belong here (second stream)
        w_skip = self.weight_skip.squeeze(2) * (self.weight_skip_gain * SQRT_HALF)
        return w0, mod_0, demod_0, w1, mod_1, demod_1, self.bias_0.to(dtype), self.bias_1.to(dtype), w_skip

    def forward_frames(self, x, latent: torch.Tensor, magnitude_ema_beta: float = 1.0,
                       out_seq_length: Optional[int] = None, dtype: Optional[torch.dtype] = None, terms=None, boundary=None):
        """Same layer in time-major frames layout: x [(T N), C, H, W] (or the `Modulated` output of the previous layer), latent
        [N, L, T]; `terms` = `frame_terms(latent, dtype)` when the caller computed them ahead (on another stream); `boundary` =
        (next layer's modulation [(T N), C], its dtype, whether it tracks the magnitude, whether it reads the plain tensor):
        the block output then comes back as `Modulated` (FUSE_BOUNDARY).

        Elementwise passes over the activations are the HBM-bound part of the block, so scalars are folded
        into small tensors instead of being applied to the activations: the input-magnitude gain goes into
        the style of conv 0 and into the skip weights (convolution is linear), sqrt(1/2) into the skip
        weights and the demodulation of conv 1, and (skip + conv1 * demod) is one addcmul."""
        if dtype is None:
            dtype = torch.float16 if (self.use_float16 and (x.mod if isinstance(x, Modulated) else x).is_cuda) else torch.float32
        n = latent.shape[0]
        track = self.magnitude_ema and magnitude_ema_beta != 1
        w0, mod_0, demod_0, w1, mod_1, demod_1, b0, b1, w_skip = terms if terms is not None else self.frame_terms(latent, dtype)

        # conv 0: modulate (one pass, which also measures E[x^2]) -> conv -> fused epilogue
        if isinstance(x, Modulated):                                            # done by the previous layer's last pass
            xm = (x.mod, x.msq) if track else x.mod
            x = x.plain
        else:
            x = x.to(dtype)
            xm = modconv_epilogue(x, post=mod_0, want_msq=track)
        gain_0 = None
        if self.magnitude_ema:
            gain_0 = self.input_magnitude_ema_0.update(xm[1], magnitude_ema_beta) if track else self.input_magnitude_ema_0(x)
        xm = xm[0] if track else xm
        # conv 0 and its epilogue: demodulation * input gain, bias, activation, clamp AND the modulation of conv 1
        hm = temporal_conv_epilogue(xm, w0, n, self.padding[1:], pre=(demod_0 if gain_0 is None else demod_0 * gain_0),
                                    b=b0, post=mod_1, act=self.activation, clamp=self.activation_clamp, want_msq=track)
        gain_1 = None
        if self.magnitude_ema:
            gain_1 = self.input_magnitude_ema_1.update(hm[1], magnitude_ema_beta) if track else self.input_magnitude_ema_1.magnitude_ema.rsqrt()
        hm = hm[0] if track else hm

        if gain_0 is not None:
            w_skip = w_skip * gain_0
        skip = pointwise_conv(x, w_skip.to(dtype))
        # conv 1: h = skip + conv * (demodulation * gain * sqrt(1/2)), one pass
        scale_1 = demod_1 * SQRT_HALF if gain_1 is None else demod_1 * (gain_1 * SQRT_HALF)
        h = temporal_conv_epilogue(hm, w1, n, self.padding[1:], pre=scale_1, res=skip)
        if self.temporal_up:
            h = resample_time_frames(h, self.temporal_upsample.filter, n, up=self.temporal_upsample.scale)
        h = crop_frames(h, n, seq_length=out_seq_length)
        if self.spatial_up:
            h = upfirdn2d.upsample2d(h, self.spatial_upsample.filter, up=self.spatial_upsample.scale)
        h = crop_frames(h, n, height=self.out_height, width=self.out_width)
        if boundary is not None and boundary[1] == h.dtype and (dual_supported(h) or not h.is_cuda):      # CPU tensors: the plain-PyTorch definition
            next_mod, _, next_track, next_plain = boundary
            if next_plain:
                res = modconv_epilogue_dual(h, b=b1, post=next_mod, act=self.activation, clamp=self.activation_clamp, want_msq=next_track)
                return Modulated(res[1], res[0], res[2] if next_track else None)
            res = modconv_epilogue(h, b=b1, post=next_mod, act=self.activation, clamp=self.activation_clamp, want_msq=next_track)
            return Modulated(None, res[0], res[1]) if next_track else Modulated(None, res, None)
        return bias_act.bias_act(h, b1, act=self.activation, clamp=self.activation_clamp)


class ToRGB(nn.Module):
    def __init__(self, latent_dim: int, in_channels: int, activation_clamp: Optional[float] = 256.0, magnitude_ema: bool = True):
        super().__init__()
        self.latent_dim, self.in_channels = latent_dim, in_channels
        self.activation_clamp, self.magnitude_ema = activation_clamp, magnitude_ema
        self.use_float16 = False
        self.affine = FullyConnectedLayer(latent_dim, in_channels, bias_init=1.0)
        self.weight = nn.Parameter(torch.randn(3, in_channels, 1, 1, 1))
        self.bias = nn.Parameter(torch.zeros(3))
        if magnitude_ema:
            self.input_magnitude_ema = MagnitudeEMA()

    def forward(self, x: torch.Tensor, latent: torch.Tensor, magnitude_ema_beta: float = 1.0, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        if dtype is None:
            dtype = torch.float16 if (self.use_float16 and x.is_cuda) else torch.float32
        n, c, t = latent.shape
        style = self.affine(latent.permute(0, 2, 1).reshape(n * t, c)).reshape(n, t, -1).permute(0, 2, 1)
        x = x.to(dtype)
        gain = self.input_magnitude_ema(x, magnitude_ema_beta) if self.magnitude_ema else None
        y = modulated_conv3d(x, self.weight, style, gain, (0, 0, 0), False, dtype)
        return bias_act.bias_act(y, self.bias.to(dtype), act='linear', clamp=self.activation_clamp)

    def frame_terms(self, latent: torch.Tensor, dtype: torch.dtype):
        n, c, t = latent.shape
        style = self.affine(latent.permute(2, 0, 1).reshape(t * n, c)).reshape(t, n, -1)
        weight, mod, _ = modulation_terms(self.weight, style, False)
        return weight, mod

    def forward_frames(self, x: torch.Tensor, latent: torch.Tensor, magnitude_ema_beta: float = 1.0, dtype: Optional[torch.dtype] = None,
                       terms=None) -> torch.Tensor:
        if dtype is None:
            dtype = torch.float16 if (self.use_float16 and (x.mod if isinstance(x, Modulated) else x).is_cuda) else torch.float32
        n = latent.shape[0]
        track = self.magnitude_ema and magnitude_ema_beta != 1
        weight, mod = terms if terms is not None else self.frame_terms(latent, dtype)
        if isinstance(x, Modulated):                                            # modulated by the last block's final pass
            xm = (x.mod, x.msq) if track else x.mod
            x = x.mod
        else:
            x = x.to(dtype)
            xm = modconv_epilogue(x, post=mod, want_msq=track)
        if self.magnitude_ema:                                                  # scalar gain folded into the 3 x Ci weight
            gain = self.input_magnitude_ema.update(xm[1], magnitude_ema_beta) if track else self.input_magnitude_ema(x)
            weight = weight * gain
        xm = xm[0] if track else xm
        y = temporal_conv_frames(xm, weight.to(dtype), n, (0, 0))
        return bias_act.bias_act(y, self.bias.to(dtype), act='linear', clamp=self.activation_clamp)


class VideoGenerator(nn.Module):
    """Low-resolution generator: temporal noise -> latents per level -> 6 temporal + 4 spatial
    residual blocks -> RGB video [N, 3, T, out_height, out_width] in float32."""

    def __init__(self, out_height: int = 36, out_width: int = 64, temporal_emb_dim: int = 1024, latent_w_dim: int = 1024,
                 temporal_ksize: int = 3, spatial_ksize: int = 3, temporal_padding: int = 8, spatial_padding: int = 0,
                 output_scale: float = 0.25, num_fp16_layers: int = 0, embedding_kwargs: Optional[dict] = None,
                 mapping_kwargs: Optional[dict] = None):
        super().__init__()
        self.out_height, self.out_width = out_height, out_width
        self.temporal_emb_dim, self.latent_w_dim = temporal_emb_dim, latent_w_dim
        self.temporal_padding, self.output_scale = temporal_padding, output_scale
        long_edge = max(out_height, out_width)
        scales = [max(1, long_edge // (2 ** (2 + i))) for i in range(5)]
        hs = [math.ceil(out_height / s) + 2 * spatial_padding for s in scales]
        ws = [math.ceil(out_width / s) + 2 * spatial_padding for s in scales]
        L = latent_w_dim
        tk = dict(spatial_ksize=spatial_ksize, temporal_ksize=temporal_ksize)
        sk = dict(spatial_ksize=spatial_ksize)
        self.temporal_layers = nn.ModuleList([
            Synthesis3dResBlock(L, 512, out_height=hs[0], out_width=ws[0], temporal_up=True, **tk),
            Synthesis3dResBlock(L, 512, out_height=hs[1], out_width=ws[1], temporal_up=True, spatial_up=True, **tk),
            Synthesis3dResBlock(L, 512, temporal_up=True, **tk),
            Synthesis3dResBlock(L, 512, 512, out_height=hs[2], out_width=ws[2], temporal_up=True, spatial_up=True, **tk),
            Synthesis3dResBlock(L, 512, 256, temporal_up=True, **tk),
            Synthesis3dResBlock(L, 256, **tk),
        ])
        self.spatial_layers = nn.ModuleList([
            Synthesis3dResBlock(L, 256, 128, out_height=hs[3], out_width=ws[3], spatial_up=True, **sk),
            Synthesis3dResBlock(L, 128, **sk),
            Synthesis3dResBlock(L, 128, 64, out_height=hs[4], out_width=ws[4], spatial_up=hs[4] != hs[3], **sk),
            Synthesis3dResBlock(L, 64, out_height=out_height, out_width=out_width, **sk),
        ])
        self.to_rgb = ToRGB(L, self.spatial_layers[-1].out_channels)
        self.num_layers = len(self.temporal_layers) + len(self.spatial_layers) + 1
        ordered = [self.to_rgb] + list(reversed(self.spatial_layers)) + list(reversed(self.temporal_layers))
        for layer in ordered[:num_fp16_layers]:
            layer.use_float16 = True
        self.total_temporal_scale = 2 ** sum(1 for l in self.temporal_layers if l.temporal_up)
        self.total_spatial_scale = 2 ** sum(1 for l in list(self.temporal_layers) + list(self.spatial_layers) if l.spatial_up)
        self.spatial_input = nn.Parameter(torch.randn(1, 512, 1, hs[0], ws[0]))
        self.temporal_emb = BlurredNoise(temporal_emb_dim, **(embedding_kwargs or {}))
        self.latent_mapping = LatentMappingNetwork(temporal_emb_dim, latent_w_dim, **(mapping_kwargs or {}))
        self.temporal_downsample_latent = TemporalKaiserDownsample()
        self.w_to_temp_input = FullyConnectedLayer(latent_w_dim, 512)

    # -- sequence-length bookkeeping ---------------------------------------------------------------

    def compute_seq_lengths(self, seq_length: int):
        """(input length, [output length of each temporal layer]) for a requested video length."""
        lengths = [seq_length]
        scale = 1
        for layer in reversed(self.temporal_layers):
            if layer.temporal_up:
                scale *= 2
            lengths.append(math.ceil(seq_length / scale) + 2 * self.temporal_padding)
        in_len = lengths.pop()
        lengths.reverse()
        return in_len, lengths

    def sample_temporal_emb(self, batch_size: int, seq_length: int, generator_emb: Optional[torch.Generator] = None) -> torch.Tensor:
        in_len = self.compute_seq_lengths(seq_length)[0]
        return self.temporal_emb(batch_size, in_len * self.total_temporal_scale, generator_emb)

    def compute_latent_ws(self, temporal_emb: torch.Tensor, seq_length: int) -> List[torch.Tensor]:
        """Latents for [temporal input, temporal layers..., spatial layers..., to_rgb]."""
        w = self.latent_mapping(temporal_emb)
        in_len, lengths = self.compute_seq_lengths(seq_length)
        full = crop_center(w, seq_length=lengths.pop())
        ws = [full.clone() for _ in range(len(self.spatial_layers) + 1)]
        lengths.reverse()
        lengths.append(in_len)
        for layer, length in zip(reversed(self.temporal_layers), lengths):
            if layer.temporal_up:
                w = self.temporal_downsample_latent(w)
            ws.insert(0, crop_center(w, seq_length=length))
        ws.insert(0, ws[0].clone())
        return ws

    # -- synthesis -----------------------------------------------------------------------------------

    def synthesize_video(self, temporal_input: torch.Tensor, latent_ws: List[torch.Tensor], seq_length: int,
                         magnitude_ema_beta: float = 1.0, dtype: Optional[torch.dtype] = None, return_features: bool = False):
        in_len, lengths = self.compute_seq_lengths(seq_length)
        assert temporal_input.shape[1:] == (512, in_len)
        n = temporal_input.shape[0]
        # time-major frames: [(T N), C, H, W]
        x = (temporal_input.permute(2, 0, 1)[:, :, :, None, None] + self.spatial_input[0, :, 0]) * SQRT_HALF
        x = _cl(x.reshape(in_len * n, 512, x.shape[3], x.shape[4]))
        feats = []
        layers = list(self.temporal_layers) + list(self.spatial_layers) + [self.to_rgb]
        lengths = list(lengths) + [None] * (len(layers) - len(lengths))
        terms = self._terms_ahead(layers, latent_ws, x, dtype)

        def layer_dtype(layer):
            return dtype if dtype is not None else (torch.float16 if (getattr(layer, 'use_float16', False) and x.is_cuda) else torch.float32)
        for wi, (layer, length) in enumerate(zip(layers, lengths)):
            extra = {} if layer is self.to_rgb else {'out_seq_length': length}
            if FUSE_BOUNDARY and layer is not self.to_rgb and not return_features:
                nxt = layers[wi + 1]
                nxt_terms = terms(wi + 1)
                if nxt_terms is not None:                # (modulation, dtype, tracks the magnitude, reads the plain tensor) of the next layer
                    extra['boundary'] = (nxt_terms[1], layer_dtype(nxt), bool(getattr(nxt, 'magnitude_ema', False)) and magnitude_ema_beta != 1,
                                         nxt is not self.to_rgb)
            x = layer.forward_frames(x, latent_ws[wi], magnitude_ema_beta, dtype=dtype, terms=terms(wi), **extra)
            if return_features and layer is not self.to_rgb:
                feats.append(video_from_frames(x, n))
        rgb = x
        video = video_from_frames(rgb.float() * self.output_scale, n).contiguous()
        if return_features:
            return feats + [video]
        return video

    def _terms_ahead(self, layers, latent_ws, x: torch.Tensor, dtype: Optional[torch.dtype]):
        """Issue every layer's `frame_terms` on a side stream (SIDE_STREAM_TERMS); returns `get(i)`, which makes the current
        stream wait for layer i's terms and hands them over (None: the layer computes them itself)."""
        if not (SIDE_STREAM_TERMS and x.is_cuda):
            return lambda i: None
        main = torch.cuda.current_stream(x.device)
        side = getattr(self, '_terms_stream', None)
        if side is None or side.device != x.device:
            side = self._terms_stream = torch.cuda.Stream(x.device)
        side.wait_stream(main)
        ready = []
        with torch.cuda.stream(side):
            for layer, latent in zip(layers, latent_ws):
                use16 = getattr(layer, 'use_float16', False)
                layer_dtype = dtype if dtype is not None else (torch.float16 if use16 else torch.float32)
                out = layer.frame_terms(latent, layer_dtype)
                ev = torch.cuda.Event()
                ev.record(side)
                ready.append((out, ev))

        def get(i):
            out, ev = ready[i]
            main.wait_event(ev)
            for tensor in out:
                tensor.record_stream(main)
            return out
        return get

    def _temporal_input(self, latent_ws: List[torch.Tensor]) -> torch.Tensor:
        w0 = latent_ws.pop(0)
        n, c, t = w0.shape
        return self.w_to_temp_input(w0.permute(0, 2, 1).reshape(n * t, c)).reshape(n, t, -1).permute(0, 2, 1)

    def forward_from_emb(self, temporal_emb: torch.Tensor, seq_length: int, magnitude_ema_beta: float = 1.0,
                         dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        ws = self.compute_latent_ws(temporal_emb, seq_length)
        return self.synthesize_video(self._temporal_input(ws), ws, seq_length, magnitude_ema_beta, dtype)

    def forward(self, batch_size: int, seq_length: int, magnitude_ema_beta: float = 1.0,
                generator_emb: Optional[torch.Generator] = None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        emb = self.sample_temporal_emb(batch_size, seq_length, generator_emb)
        return self.forward_from_emb(emb, seq_length, magnitude_ema_beta, dtype)

    def sample_temporal_input(self, batch_size: int, seq_length: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """Gaussian stand-in for the learned temporal input, [N, C0, T_in] (reference generator_lres.py:832)."""
        device = next(self.parameters()).device
        t_in = self.compute_seq_lengths(seq_length)[0]
        return torch.randn(batch_size, self.temporal_layers[0].in_channels, t_in, generator=generator, device=device)

    def sample_video_segments(self, batch_size: int, seq_length: int, segment_length: int = 8,
                              generator_emb: Optional[torch.Generator] = None, dtype: Optional[torch.dtype] = None):
        video = self.forward(batch_size, seq_length, generator_emb=generator_emb, dtype=dtype)
        yield from video.split(segment_length, dim=2)


# --------------------------------------------------------------------------------------------------
# Discriminator.

class Downsample3d(nn.Module):
    def __init__(self, spatial_down: bool = True, temporal_down: bool = True, downsample_filter=(1.0, 3.0, 3.0, 1.0)):
        super().__init__()
        self.spatial_down, self.temporal_down = spatial_down, temporal_down
        taps = torch.as_tensor(downsample_filter, dtype=torch.float32)
        self.register_buffer('_downsample_filter', taps / taps.sum())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, c, t, h, w = x.shape
        if self.spatial_down:
            y = upfirdn2d.downsample2d(x.reshape(n, c * t, h, w), self._downsample_filter, down=2)
            h, w = y.size(2), y.size(3)
            x = y.reshape(n, c, t, h, w)
        if self.temporal_down:
            y = upfirdn2d.downsample2d(x.reshape(n, c, t, h * w), self._downsample_filter.unsqueeze(1), down=[1, 2])
            x = y.reshape(n, c, y.size(2), h, w)
        return x


class Conv3dLayer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, spatial_ksize: int, temporal_ksize: int, bias: bool = True,
                 spatial_down: bool = False, temporal_down: bool = False, activation: str = 'linear', conv_clamp: Optional[float] = None):
        super().__init__()
        self.activation, self.conv_clamp = activation, conv_clamp
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, temporal_ksize, spatial_ksize, spatial_ksize))
        self.weight_gain = 1 / math.sqrt(in_channels * temporal_ksize * spatial_ksize * spatial_ksize)
        self.padding = (temporal_ksize // 2, spatial_ksize // 2, spatial_ksize // 2)
        self._bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.has_down = spatial_down or temporal_down
        if self.has_down:
            self.downsample = Downsample3d(spatial_down, temporal_down)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = _dense_conv3d(x, (self.weight * self.weight_gain).to(x.dtype), self.padding)
        if self.has_down:
            y = self.downsample(y)
        b = self._bias.to(x.dtype) if self._bias is not None else None
        return bias_act.bias_act(y, b, act=self.activation, clamp=self.conv_clamp)

    def forward_frames(self, x: torch.Tensor, n: int) -> torch.Tensor:
        """Time-major frames layout: x [(T N), C, H, W]."""
        if WEIGHT_PREP and not SECOND_ORDER and weight_prep.supported(self.weight, x.dtype):
            w = weight_prep.weight_prep(self.weight, self.weight_gain, False, x.dtype, want_w2=False)[0]     # scale + cast + layout: one pass
        else:
            w = (self.weight * self.weight_gain).to(x.dtype)
        b = self._bias.to(x.dtype) if self._bias is not None else None
        if not self.has_down:                   # conv -> bias / activation / clamp in one epilogue pass
            return temporal_conv_epilogue(x, w, n, self.padding[1:], b=b, act=self.activation, clamp=self.conv_clamp)
        y = temporal_conv_epilogue(x, w, n, self.padding[1:])          # the resamplers need the plain sum
        if self.downsample.spatial_down:
            y = upfirdn2d.downsample2d(y, self.downsample._downsample_filter, down=2)
        if self.downsample.temporal_down:
            y = resample_time_frames(y, self.downsample._downsample_filter, n, down=2)
        return bias_act.bias_act(y, b, act=self.activation, clamp=self.conv_clamp)


class Conv1dLayer(nn.Module):
    def __init__(self, in_channels: int, out_channels: Optional[int] = None, kernel_size: int = 1, bias: bool = True,
                 activation: str = 'linear', lr_multiplier: float = 1.0, weight_std_init: float = 1.0, bias_init: float = 0.0,
                 downsample: bool = False):
        super().__init__()
        out_channels = out_channels or in_channels
        self.activation, self.lr_multiplier = activation, lr_multiplier
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size) * (weight_std_init / lr_multiplier))
        self.weight_gain = lr_multiplier / math.sqrt(in_channels * kernel_size)
        self._bias = nn.Parameter(torch.full((out_channels,), bias_init / lr_multiplier)) if bias else None
        if downsample:
            self._downsample = TemporalLinearDownsample(scale=2)
        self.has_down = downsample

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b = None
        if self._bias is not None:
            b = (self._bias * self.lr_multiplier if self.lr_multiplier != 1 else self._bias).to(x.dtype)
        y = F.conv1d(x, (self.weight * self.weight_gain).to(x.dtype), b, padding=self.padding)
        if self.has_down:
            y = self._downsample(y)
        return bias_act.bias_act(y, act=self.activation)


class DiscriminatorBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, vid_channels: int = 0, spatial_ksize: int = 3, temporal_ksize: int = 5,
                 spatial_ksize_1: Optional[int] = None, temporal_ksize_1: Optional[int] = None, spatial_down: bool = True,
                 temporal_down: bool = True, conv_clamp: Optional[float] = 256, use_fp16: bool = False):
        super().__init__()
        self.in_channels, self.out_channels, self.vid_channels = in_channels, out_channels, vid_channels
        self.spatial_down, self.temporal_down, self.use_fp16 = spatial_down, temporal_down, use_fp16
        if vid_channels > 0:
            self.conv_vid = Conv3dLayer(vid_channels, in_channels, 1, 1, activation='lrelu', conv_clamp=conv_clamp)
        self.conv_0 = Conv3dLayer(in_channels, in_channels, spatial_ksize, temporal_ksize, activation='lrelu', conv_clamp=conv_clamp)
        self.conv_1 = Conv3dLayer(in_channels, out_channels, spatial_ksize_1 or spatial_ksize, temporal_ksize_1 or temporal_ksize,
                                  spatial_down=spatial_down, temporal_down=temporal_down, activation='lrelu', conv_clamp=conv_clamp)
        self.conv_skip = Conv3dLayer(in_channels, out_channels, 1, 1, bias=False, spatial_down=spatial_down,
                                     temporal_down=temporal_down, conv_clamp=conv_clamp)

    def forward(self, x: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        x = x.to(dtype if dtype is not None else (torch.float16 if self.use_fp16 else torch.float32))
        if self.vid_channels > 0:
            x = self.conv_vid(x)
        h = self.conv_0(x)
        skip = self.conv_skip(x)
        h = self.conv_1(h)
        return (h + skip) * SQRT_HALF

    def forward_frames(self, x: torch.Tensor, n: int, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        x = x.to(dtype if dtype is not None else (torch.float16 if self.use_fp16 else torch.float32))
        if self.vid_channels > 0:
            x = self.conv_vid.forward_frames(x, n)
        h = self.conv_0.forward_frames(x, n)
        skip = self.conv_skip.forward_frames(x, n)
        h = self.conv_1.forward_frames(h, n)
        return (h + skip) * SQRT_HALF


class DiscriminatorEpilogue(nn.Module):
    def __init__(self, in_res: int = 4, in_seq_length: int = 16, in_channels: int = 512, channels: int = 1024, temporal_ksize: int = 3,
                 num_conv1d_layers: int = 4, num_linear_layers: int = 2, conv_clamp: Optional[float] = 256, num_downsamples: int = 0):
        super().__init__()
        assert num_downsamples <= num_conv1d_layers and in_seq_length % (2 ** num_downsamples) == 0
        self.in_res, self.in_seq_length, self.in_channels = in_res, in_seq_length, in_channels
        self.conv1d_layer_names, self.linear_layer_names = [], []
        for i in range(num_conv1d_layers):
            name = f'conv1d_{i}'
            cin, k = ((in_res ** 2) * in_channels, 1) if i == 0 else (channels, temporal_ksize)
            setattr(self, name, Conv1dLayer(cin, channels, kernel_size=k, activation='lrelu', downsample=i < num_downsamples))
            self.conv1d_layer_names.append(name)
        for i in range(num_linear_layers):
            name = f'linear_{i}'
            cin = in_seq_length * channels // (2 ** num_downsamples) if i == 0 else channels
            last = i == num_linear_layers - 1
            setattr(self, name, FullyConnectedLayerD(cin, 1 if last else channels, activation='linear' if last else 'lrelu'))
            self.linear_layer_names.append(name)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, c, t, h, w = x.shape
        assert (c, t, h, w) == (self.in_channels, self.in_seq_length, self.in_res, self.in_res)
        f = x.float().permute(0, 1, 3, 4, 2).reshape(n, c * h * w, t)
        for name in self.conv1d_layer_names:
            f = getattr(self, name)(f)
        f = f.reshape(n, -1)
        for name in self.linear_layer_names:
            f = getattr(self, name)(f)
        return f


class FullyConnectedLayerD(FullyConnectedLayer):
    """Discriminator spelling of the constructor argument (`lr_multiplier`)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, activation: str = 'linear',
                 lr_multiplier: float = 1.0, weight_std_init: float = 1.0, bias_init: float = 0.0):
        super().__init__(in_features, out_features, bias, activation, lr_multiplier, weight_std_init, bias_init)


class VideoDiscriminator(nn.Module):
    """Residual 3-D conv discriminator over [N, 3, T, H, W] (zero-padded to max_edge^2) -> logits [N, 1]."""

    def __init__(self, seq_length: int, max_edge: int, channels: int = 3, channels_base: int = 2048, channels_max: int = 512,
                 spatial_ksize: int = 3, temporal_ksize: int = 5, spatial_ksize_1: Optional[int] = None,
                 temporal_ksize_1: Optional[int] = None, conv_clamp: Optional[float] = 256, num_fp16_res: int = 0,
                 epilogue_kwargs: Optional[dict] = None):
        super().__init__()
        self.seq_length, self.max_edge, self.channels = seq_length, max_edge, channels
        kw = dict(spatial_ksize=spatial_ksize, temporal_ksize=temporal_ksize, spatial_ksize_1=spatial_ksize_1,
                  temporal_ksize_1=temporal_ksize_1, conv_clamp=conv_clamp)
        self.blocks = nn.ModuleList([
            DiscriminatorBlock(32, 64, channels, spatial_ksize=spatial_ksize, temporal_ksize=1, temporal_down=False,
                               spatial_down=max_edge > 32, use_fp16=num_fp16_res > 0, conv_clamp=conv_clamp),
            DiscriminatorBlock(64, 128, use_fp16=num_fp16_res > 1, temporal_down=seq_length >= 4, **kw),
            DiscriminatorBlock(128, 256, use_fp16=num_fp16_res > 2, temporal_down=seq_length >= 8, **kw),
            DiscriminatorBlock(256, 512, use_fp16=num_fp16_res > 3, temporal_down=seq_length >= 16, **kw),
        ])
        sscale = 2 ** sum(1 for b in self.blocks if b.spatial_down)
        tscale = 2 ** sum(1 for b in self.blocks if b.temporal_down)
        self.epilogue = DiscriminatorEpilogue(max_edge // sscale, seq_length // tscale, self.blocks[-1].out_channels, **(epilogue_kwargs or {}))

    def forward(self, videos: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        assert videos.size(1) == self.channels and videos.size(2) == self.seq_length
        assert videos.size(3) == self.max_edge or videos.size(4) == self.max_edge
        px = (self.max_edge - videos.size(4)) // 2
        py = (self.max_edge - videos.size(3)) // 2
        n = videos.shape[0]
        f = _cl(frames_from_video(F.pad(videos, (px, px, py, py))))
        for block in self.blocks:
            f = block.forward_frames(f, n, dtype=dtype)
        f = video_from_frames(f, n)
        return self.epilogue(f)
