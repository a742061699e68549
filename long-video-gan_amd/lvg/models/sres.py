"""Super-resolution video GAN networks (per-frame alias-free 2-D generator conditioned on a
window of low-resolution frames; 2-D residual discriminator over channel-stacked frames) on the
MI355X op stack.

Architecture, hyper-parameter arithmetic and parameter/buffer NAMES follow the reference so that a
reference state_dict loads unchanged (reference model/generator_sres.py: modulated_conv2d :24,
SynthesisLayer :214, SynthesisNetwork :362, KaiserDownsample/Upsample :486/:500, Generator :515,
VideoGenerator :618; model/discriminator_sres.py: Conv2dLayer :147, DiscriminatorBlock :223,
DiscriminatorEpilogue :392, VideoDiscriminator :458). What is different, on purpose:

  * modulation is applied to the ACTIVATIONS (x * style) and demodulation to the conv OUTPUT, with
    ONE dense conv2d over shared weights. The reference materialises a per-sample weight tensor
    [N, Co, Ci, k, k] and runs a grouped convolution with groups = N (:43-58): for the 512-channel
    layers that is N x 9.4 MB of weights written and re-read per layer per step, and a grouped
    problem MIOpen's implicit-GEMM kernels do not cover. Same value: conv(x, w*s) == conv(x*s, w)
    and the demodulation norm sum_{i,k}(w s)^2 == (sum_k w^2) @ s^2.
  * every synthesis layer is conv2d -> `filtered_lrelu` (the fused HIP kernel: bias, 2x/4x FIR
    up-sampling, leaky ReLU, clamp, FIR down-sampling in one pass, csrc/filtered_lrelu.hip);
  * the conditioning pyramid is built once per distinct scale instead of once per layer (the
    reference re-runs the same Kaiser resampler for every layer that shares a sampling rate, :603);
  * `compute_dtype` picks the reduced-precision type of the high-resolution layers (float16 as in
    the reference, or bfloat16); statistics and styles stay float32.
"""

import math
import os
from typing import Iterator, List, Optional

import numpy as np
import scipy.signal
import scipy.special
import torch
import torch.nn as nn
import torch.nn.functional as F

import torch_utils.distributed as dist_utils
from torch_utils.ops import bias_act, conv2d_gradfix, conv2d_resample, filtered_lrelu, modconv2d_layout, stats, upfirdn2d, weight_prep

from .. import ddp
from .lres import FullyConnectedLayer, _linear_filter

# 16-bit layers of the generator: dense convolution on channels-last (MFMA implicit-GEMM) kernels between the fused
# prologue / epilogue of torch_utils.ops.modconv2d_layout. LVG_SRES_CHANNELS_LAST=0 keeps the NCHW convolution.
SIDE_STREAM_TERMS = os.environ.get('LVG_SRES_SIDE_STREAM_TERMS', '1') == '1'     # weight / style side of the generator layers on a second stream
CHANNELS_LAST = os.environ.get('LVG_SRES_CHANNELS_LAST', '1') == '1'
WEIGHT_PREP = os.environ.get('LVG_SRES_WEIGHT_PREP', '1') == '1'             # weight side of the 16-bit 3 x 3 layers in one launch each way (lvg_weight_prep2d)

SQRT_HALF = math.sqrt(0.5)


# --------------------------------------------------------------------------------------------------
# Filters.

def lowpass_taps(numtaps: int, cutoff: float, width: float, fs: float, radial: bool = False) -> Optional[torch.Tensor]:
    """Kaiser-windowed low-pass FIR (1-D, or a radially symmetric 2-D jinc); None for a 1-tap filter."""
    if numtaps == 1:
        return None
    if not radial:
        return torch.as_tensor(scipy.signal.firwin(numtaps=numtaps, cutoff=cutoff, width=width, fs=fs), dtype=torch.float32)
    pos = (np.arange(numtaps) - (numtaps - 1) / 2) / fs
    radius = np.hypot(*np.meshgrid(pos, pos))
    taps = scipy.special.j1(2 * cutoff * (np.pi * radius)) / (np.pi * radius)
    window = np.kaiser(numtaps, scipy.signal.kaiser_beta(scipy.signal.kaiser_atten(numtaps, width / (fs / 2))))
    taps = taps * np.outer(window, window)
    return torch.as_tensor(taps / taps.sum(), dtype=torch.float32)


class KaiserResample(nn.Module):
    """Holds the `filter` buffer of the conditioning resamplers (6 taps per phase, cutoff 1, fs 4*scale)."""

    def __init__(self, scale: int, filter_size: int = 6, cutoff: float = 1.0, width: float = 6.0, sampling_rate: float = 4.0):
        super().__init__()
        assert isinstance(scale, int) and scale > 1
        self.scale = scale
        taps = scipy.signal.firwin(numtaps=scale * filter_size, cutoff=cutoff, width=width, fs=scale * sampling_rate)
        self.register_buffer('filter', torch.tensor(taps, dtype=torch.float32))


class KaiserDownsample(KaiserResample):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        p = self.scale
        x = F.pad(x, (p, p, p, p), mode='replicate')
        return upfirdn2d.downsample2d(x, self.filter, down=self.scale, padding=-p)


class KaiserUpsample(KaiserResample):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = F.pad(x, (1, 1, 1, 1), mode='replicate')
        return upfirdn2d.upsample2d(x, self.filter, up=self.scale, padding=-self.scale)


# --------------------------------------------------------------------------------------------------
# Generator.

def modulation_terms2d(weight: torch.Tensor, style: torch.Tensor, demodulate: bool = True, input_gain: Optional[torch.Tensor] = None,
                       low_precision: bool = False):
    """(weight', mod [N, Ci], demod [N, Co] or None), all float32, such that
    conv2d(x * mod, weight') * demod  ==  the reference's modulated_conv2d(x, weight, style) (generator_sres.py:24-67).

    With `low_precision` the 1 / sqrt(fan_in) part of the demodulation is moved into the weight, so the un-demodulated
    convolution output stays O(1) in float16 (the reference's convolution output is demodulated through its weights,
    :50-58); demod is computed from the scaled weight and compensates exactly."""
    if demodulate:
        weight = weight * weight.square().mean(dim=(1, 2, 3), keepdim=True).rsqrt()
        style = style * style.square().mean().rsqrt()          # one statistic over the whole batch, as the reference
        if low_precision:
            weight = weight * (1.0 / math.sqrt(weight.shape[1] * weight.shape[2] * weight.shape[3]))
    mod = style if input_gain is None else style * input_gain
    demod = None
    if demodulate:
        energy = weight.square().sum(dim=(2, 3))                # [Co, Ci]
        demod = torch.matmul(style.square(), energy.t()).add(1e-8 * (1.0 / (weight.shape[1] * weight.shape[2] * weight.shape[3]) if low_precision else 1.0)).rsqrt()     # [N, Co]
    return weight, mod, demod


def modulated_conv2d(x: torch.Tensor, weight: torch.Tensor, style: torch.Tensor, demodulate: bool = True,
                     padding: int = 0, input_gain: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [N, Ci, H, W] (any float dtype); weight [Co, Ci, k, k] and style [N, Ci] float32.

    Returns conv2d(x, w * style_n) (demodulated per sample and output channel when asked), computed
    as conv2d(x * mod) * demod with the shared weight."""
    weight, mod, demod = modulation_terms2d(weight, style, demodulate, input_gain, low_precision=x.dtype != torch.float32)
    y = conv2d_gradfix.conv2d(input=x * mod.to(x.dtype)[:, :, None, None], weight=weight.to(x.dtype), padding=padding)
    if demod is not None:
        y = y * demod.to(y.dtype)[:, :, None, None]
    return y


class MappingNetwork(nn.Module):
    def __init__(self, z_dim: int, w_dim: int, num_ws: int, num_layers: int = 2, lr_multiplier: float = 0.01, w_avg_beta: float = 0.998):
        super().__init__()
        self.z_dim, self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = z_dim, w_dim, num_ws, num_layers, w_avg_beta
        widths = [z_dim] + [w_dim] * num_layers
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(widths[idx], widths[idx + 1], activation='lrelu', lrate_mul=lr_multiplier))
        self.register_buffer('w_avg', torch.zeros(w_dim))

    def forward(self, z: torch.Tensor, truncation_psi: float = 1, truncation_cutoff: Optional[int] = None, update_emas: bool = False) -> torch.Tensor:
        x = z.float()
        x = x * (x.square().mean(1, keepdim=True) + 1e-8).rsqrt()
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if update_emas:
            # (mean over ranks: inside lvg.ddp.deferred_stat_sync() the exchange is batched after the pass)
            ddp.ema_of_rank_mean(self.w_avg, x.detach().mean(dim=0), self.w_avg_beta, ddp.LERP_FROM)
        ws = x.unsqueeze(1).repeat(1, self.num_ws, 1)
        if truncation_psi != 1:
            cut = self.num_ws if truncation_cutoff is None else truncation_cutoff
            ws[:, :cut] = self.w_avg.lerp(ws[:, :cut], truncation_psi)
        return ws


class SynthesisInput(nn.Module):
    """Optional fixed Fourier-feature input (disabled in the shipped configs, `fourfeats=False`)."""

    def __init__(self, w_dim: int, channels: int, size, sampling_rate: float, bandwidth: float):
        super().__init__()
        self.channels = channels
        size = np.broadcast_to(np.asarray(size), [2])
        freqs = torch.randn(channels, 2)
        radii = freqs.square().sum(dim=1, keepdim=True).sqrt()
        freqs = freqs / (radii * radii.square().exp().pow(0.25)) * bandwidth
        phases = torch.rand(channels) - 0.5
        theta = torch.eye(2, 3)
        theta[0, 0] = 0.5 * size[0] / sampling_rate
        theta[1, 1] = 0.5 * size[1] / sampling_rate
        grid = F.affine_grid(theta.unsqueeze(0), [1, 1, int(size[1]), int(size[0])], align_corners=False)
        feats = torch.einsum('cd,nhwd->nchw', freqs, grid) + phases.reshape(1, -1, 1, 1)
        self.weight = nn.Parameter(torch.randn(channels, channels))
        self.register_buffer('features', torch.sin(feats * (2 * np.pi)))

    def forward(self, batch: int) -> torch.Tensor:
        feats = torch.einsum('nchw,kc->nkhw', self.features, self.weight / math.sqrt(self.channels))
        return feats.expand(batch, -1, -1, -1)


class SynthesisLayer(nn.Module):
    def __init__(self, w_dim: int, is_torgb: bool, is_critically_sampled: bool, use_fp16: bool,
                 in_channels: int, out_channels: int, in_size, out_size,
                 in_sampling_rate: int, out_sampling_rate: int, in_cutoff: float, out_cutoff: float,
                 in_half_width: float, out_half_width: float,
                 conv_kernel: int = 3, filter_size: int = 6, lrelu_upsampling: int = 2,
                 use_radial_filters: bool = False, conv_clamp: Optional[float] = 256, magnitude_ema_beta: float = 0.999):
        super().__init__()
        self.w_dim, self.is_torgb, self.is_critically_sampled, self.use_fp16 = w_dim, is_torgb, is_critically_sampled, use_fp16
        self.in_channels, self.out_channels = in_channels, out_channels
        self.in_size = np.broadcast_to(np.asarray(in_size), [2])
        self.out_size = np.broadcast_to(np.asarray(out_size), [2])
        self.in_sampling_rate, self.out_sampling_rate = in_sampling_rate, out_sampling_rate
        self.tmp_sampling_rate = max(in_sampling_rate, out_sampling_rate) * (1 if is_torgb else lrelu_upsampling)
        self.conv_kernel = 1 if is_torgb else conv_kernel
        self.conv_clamp = conv_clamp
        self.magnitude_ema_beta = magnitude_ema_beta
        self.compute_dtype = torch.float16

        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1.0)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, self.conv_kernel, self.conv_kernel))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.register_buffer('magnitude_ema', torch.ones(()))

        resampled = not is_torgb
        self.up_factor = int(np.rint(self.tmp_sampling_rate / in_sampling_rate))
        self.down_factor = int(np.rint(self.tmp_sampling_rate / out_sampling_rate))
        assert in_sampling_rate * self.up_factor == self.tmp_sampling_rate == out_sampling_rate * self.down_factor
        self.up_taps = filter_size * self.up_factor if (self.up_factor > 1 and resampled) else 1
        self.down_taps = filter_size * self.down_factor if (self.down_factor > 1 and resampled) else 1
        self.down_radial = use_radial_filters and not is_critically_sampled
        self.register_buffer('up_filter', lowpass_taps(self.up_taps, in_cutoff, in_half_width * 2, self.tmp_sampling_rate))
        self.register_buffer('down_filter', lowpass_taps(self.down_taps, out_cutoff, out_half_width * 2, self.tmp_sampling_rate, radial=self.down_radial))

        # Padding (in up-sampled pixels) that lands exactly `out_size` samples after the down-sampler,
        # with the sample grid centred (alias-free GAN paper, appendix C.3).
        total = (self.out_size - 1) * self.down_factor + 1
        total = total - (self.in_size + self.conv_kernel - 1) * self.up_factor
        total = total + self.up_taps + self.down_taps - 2
        lo = (total + self.up_factor) // 2
        hi = total - lo
        self.padding = [int(lo[0]), int(hi[0]), int(lo[1]), int(hi[1])]

    def hand_path(self, device: torch.device, force_fp32: bool = False) -> Optional[str]:
        """Which hand-written route this layer's modulated convolution takes on `device`: 'fused16' (16-bit layers: layout prologue /
        epilogue around the 2-D implicit-GEMM kernels), 'split32' (float32 3 x 3 layers: split operands on the same kernels), or None."""
        if device.type != 'cuda':
            return None
        low_precision = self.use_fp16 and not force_fp32
        if low_precision and CHANNELS_LAST and self.compute_dtype in (torch.float16, torch.bfloat16):
            return 'fused16'
        if not low_precision and self.conv_kernel == 3 and modconv2d_layout.HAND_CONV and modconv2d_layout.SPLIT_F32:
            return 'split32'
        return None

    def terms(self, w: torch.Tensor, low_precision: bool):
        """Everything of the modulated convolution that does not depend on the activations (weight normalisation, style, demodulation;
        modulation_terms2d without the input gain): SynthesisNetwork issues it for all layers ahead of time on a second stream."""
        style = self.affine(w)
        if self.is_torgb:
            style = style * (1 / math.sqrt(self.in_channels * self.conv_kernel ** 2))
        if (WEIGHT_PREP and low_precision and not self.is_torgb and self.conv_kernel == 3 and modconv2d_layout.HAND_CONV and CHANNELS_LAST
                and weight_prep.supported2d(self.weight, self.compute_dtype)):
            # the weight side (normalisation, 1 / sqrt(fan_in), energy, cast, packing for both convolutions) in one launch each way
            fan_in = self.in_channels * self.conv_kernel ** 2
            prepared = weight_prep.prepare2d(self.weight, 1.0 / math.sqrt(fan_in), self.compute_dtype)
            style = style * style.square().mean().rsqrt()
            demod = torch.matmul(style.square(), prepared.energy.t()).add(1e-8 / fan_in).rsqrt()
            return prepared, style, demod
        return modulation_terms2d(self.weight, style, demodulate=not self.is_torgb, input_gain=None, low_precision=low_precision)

    def forward(self, x: Optional[torch.Tensor], w: torch.Tensor, force_fp32: bool = False, update_emas: bool = False,
                cond: Optional[torch.Tensor] = None, terms=None) -> torch.Tensor:
        """x: the layer input [N, in_channels, H, W] as in the reference (previous output and conditioning frames
        concatenated), or -- with `cond` given -- the previous layer's output alone (None for the first layer): the
        concatenation then happens inside the fused prologue of the channels-last path. `terms`: this layer's `terms(...)`
        computed ahead of time (else computed here)."""
        low_precision = self.use_fp16 and not force_fp32 and (cond if x is None else x).device.type == 'cuda'
        dtype = self.compute_dtype if low_precision else torch.float32
        fused = cond is not None and low_precision and CHANNELS_LAST and dtype in (torch.float16, torch.bfloat16)
        if cond is not None and not fused:
            x = cond if x is None else torch.cat((x, cond.to(x.dtype)), dim=1)
            cond = None
        if fused:
            assert (0 if x is None else x.shape[1]) + cond.shape[1] == self.in_channels
        else:
            assert x.shape[1:] == (self.in_channels, int(self.in_size[1]), int(self.in_size[0])), x.shape
        if update_emas:
            if fused:       # mean square over the concatenated input
                # (stats.mean_square: one pass over the 16-bit activation instead of cast + square + reduce)
                counts = [float(t.numel()) for t in (x, cond) if t is not None]
                parts = [stats.mean_square(t) * c for t, c in zip([t for t in (x, cond) if t is not None], counts)]
                mag = sum(parts) / sum(counts)
            else:
                mag = stats.mean_square(x)
            ddp.ema_of_rank_mean(self.magnitude_ema, mag.detach(), self.magnitude_ema_beta, ddp.LERP_FROM)
        input_gain = self.magnitude_ema.rsqrt()

        if fused:
            weight, mod, demod = terms if terms is not None else self.terms(w, low_precision=True)
            x = modconv2d_layout.modulated_conv2d(None if x is None else x.to(dtype), cond.to(dtype), weight, mod * input_gain, demod, padding=self.conv_kernel - 1)
        elif dtype == torch.float32 and modconv2d_layout.split_conv_supported(x, self.weight):
            # float32 3 x 3 layers on the GPU: the contraction on the hand-written MFMA kernels with float32 accuracy (operands split into
            # float16 high / low parts, float32 accumulation and output: modconv2d_layout._ModConv2dSplit) instead of the library convolution
            weight, mod, demod = terms if terms is not None else self.terms(w, low_precision=False)
            x = modconv2d_layout.modulated_conv2d(x.float(), None, weight, mod * input_gain, demod, padding=self.conv_kernel - 1)
        else:
            style = self.affine(w)
            if self.is_torgb:
                style = style * (1 / math.sqrt(self.in_channels * self.conv_kernel ** 2))
            x = modulated_conv2d(x.to(dtype), self.weight, style, demodulate=not self.is_torgb,
                                 padding=self.conv_kernel - 1, input_gain=input_gain)
        if not x.is_contiguous():          # a 1x1 conv may hand back channels-last strides; the fused kernel tiles NCHW planes
            x = x.contiguous()
        x = filtered_lrelu.filtered_lrelu(
            x=x, fu=self.up_filter, fd=self.down_filter, b=self.bias.to(dtype), up=self.up_factor, down=self.down_factor,
            padding=self.padding, gain=(1 if self.is_torgb else math.sqrt(2)), slope=(1 if self.is_torgb else 0.2),
            clamp=self.conv_clamp)
        assert x.shape[1:] == (self.out_channels, int(self.out_size[1]), int(self.out_size[0])), x.shape
        return x


def synthesis_layer_table(img_width: int, img_height: int, img_channels: int, channel_base: int, channel_max: int,
                          num_layers: int, num_critical: int, first_cutoff: float, first_stopband: float,
                          last_stopband_rel: float, margin_size: int, num_fp16_res: int) -> List[dict]:
    """Per-layer (cutoff, stopband, sampling rate, canvas size, channels) schedule: geometric
    progression of cutoff/stopband from the first layer to the image Nyquist, sampling rate = next
    power of two above twice the stopband (reference SynthesisNetwork.__init__ :362-:413)."""
    res = max(img_width, img_height)
    last_cutoff = res / 2
    last_stopband = last_cutoff * last_stopband_rel
    expo = np.minimum(np.arange(num_layers + 1) / (num_layers - num_critical), 1)
    cutoffs = first_cutoff * (last_cutoff / first_cutoff) ** expo
    stopbands = first_stopband * (last_stopband / first_stopband) ** expo
    rates = np.exp2(np.ceil(np.log2(np.minimum(stopbands * 2, res))))
    half_widths = np.maximum(stopbands, rates / 2) - cutoffs
    size_x = np.ceil(rates * min(1, img_width / img_height)) + margin_size * 2
    size_y = np.ceil(rates * min(1, img_height / img_width)) + margin_size * 2
    size_x[-2:] = img_width
    size_y[-2:] = img_height
    channels = np.rint(np.minimum((channel_base / 2) / cutoffs, channel_max))
    channels[-1] = img_channels
    table = []
    for idx in range(num_layers + 1):
        table.append(dict(
            cutoff=float(cutoffs[idx]), half_width=float(half_widths[idx]), rate=int(rates[idx]),
            size=(int(size_x[idx]), int(size_y[idx])), channels=int(channels[idx]),
            is_torgb=(idx == num_layers), is_critical=(idx >= num_layers - num_critical),
            use_fp16=bool(rates[idx] * (2 ** num_fp16_res) > res)))
    return table


class SynthesisNetwork(nn.Module):
    def __init__(self, w_dim: int, img_width: int, img_height: int, img_channels: int, cond_channels: int,
                 channel_base: int = 32768, channel_max: int = 512, num_layers: int = 14, num_critical: int = 2,
                 first_cutoff: float = 2, first_stopband: float = 2 ** 2.1, last_stopband_rel: float = 2 ** 0.3,
                 margin_size: int = 10, fourfeats: bool = False, output_scale: float = 0.25, num_fp16_res: int = 4,
                 **layer_kwargs):
        super().__init__()
        self.w_dim, self.num_ws, self.num_layers = w_dim, num_layers + 1, num_layers
        self.img_width, self.img_height, self.img_channels = img_width, img_height, img_channels
        self.cond_channels, self.output_scale, self.fourfeats = cond_channels, output_scale, fourfeats
        table = synthesis_layer_table(img_width, img_height, img_channels, channel_base, channel_max, num_layers,
                                      num_critical, first_cutoff, first_stopband, last_stopband_rel, margin_size, num_fp16_res)
        if fourfeats:
            self.input = SynthesisInput(w_dim, table[0]['channels'], table[0]['size'], table[0]['rate'], table[0]['cutoff'])
        self.layer_names = []
        for idx, cur in enumerate(table):
            src = table[max(idx - 1, 0)]
            in_channels = cond_channels + (src['channels'] if (idx > 0 or fourfeats) else 0)
            layer = SynthesisLayer(
                w_dim=w_dim, is_torgb=cur['is_torgb'], is_critically_sampled=cur['is_critical'], use_fp16=cur['use_fp16'],
                in_channels=in_channels, out_channels=cur['channels'], in_size=src['size'], out_size=cur['size'],
                in_sampling_rate=src['rate'], out_sampling_rate=cur['rate'], in_cutoff=src['cutoff'], out_cutoff=cur['cutoff'],
                in_half_width=src['half_width'], out_half_width=cur['half_width'], **layer_kwargs)
            name = f'L{idx}_{layer.out_size[0]}_{layer.out_size[1]}_{layer.out_channels}'
            setattr(self, name, layer)
            self.layer_names.append(name)

    def layers(self) -> List[SynthesisLayer]:
        return [getattr(self, name) for name in self.layer_names]

    def _terms_ahead(self, ws, device: torch.device, force_fp32: bool):
        """Issue the weight / style side of every layer on a hand-written route (SynthesisLayer.terms: ~10 small launches per layer
        forward, twice that backward, on 10 MB weight tensors) on a SECOND stream ahead of the layers: it does not depend on the
        activations, so it runs beside the convolutions instead of between them (autograd replays the backward of these nodes on the
        same stream). Returns get(i): the terms of layer i with the current stream waiting for them, or None."""
        if not (SIDE_STREAM_TERMS and device.type == 'cuda'):
            return lambda i: None
        main = torch.cuda.current_stream(device)
        side = getattr(self, '_terms_stream', None)
        if side is None or side.device != device:
            side = self._terms_stream = torch.cuda.Stream(device)
        side.wait_stream(main)
        ready = []
        with torch.cuda.stream(side):
            for layer, w in zip(self.layers(), ws):
                path = layer.hand_path(device, force_fp32)
                if path is None:
                    ready.append(None)
                    continue
                out = layer.terms(w, low_precision=(path == 'fused16'))
                ev = torch.cuda.Event()
                ev.record(side)
                ready.append((out, ev))

        def get(i):
            if ready[i] is None:
                return None
            out, ev = ready[i]
            main.wait_event(ev)
            for item in out:
                for tensor in (item.tensors() if isinstance(item, weight_prep.Prepared2d) else (item,)):
                    if tensor is not None:
                        tensor.record_stream(main)
            return out
        return get

    def forward(self, ws: torch.Tensor, conds: List[torch.Tensor], **layer_kwargs) -> torch.Tensor:
        assert ws.shape[1:] == (self.num_ws, self.w_dim)
        ws = ws.float().unbind(dim=1)
        x = self.input(ws[0].size(0)) if self.fourfeats else None
        ahead = self._terms_ahead(ws, conds[0].device, bool(layer_kwargs.get('force_fp32', False)))
        for i, (layer, w, cond) in enumerate(zip(self.layers(), ws, conds)):
            x = layer(x, w, cond=cond, terms=ahead(i), **layer_kwargs)
        if self.output_scale != 1:
            x = x * self.output_scale
        return x.float()


class Generator(nn.Module):
    def __init__(self, z_dim: int, w_dim: int, img_width: int, img_height: int, img_channels: int,
                 cond_width: int, cond_height: int, cond_context: int, mapping_kwargs: dict = {},
                 margin_size: int = 10, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.w_dim = z_dim, w_dim
        self.img_width, self.img_height, self.img_channels = img_width, img_height, img_channels
        self.cond_width, self.cond_height, self.cond_context = cond_width, cond_height, cond_context
        self.cond_channels = img_channels * (2 * cond_context + 1)
        self.margin_size = margin_size
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_width=img_width, img_height=img_height, img_channels=img_channels,
                                          cond_channels=self.cond_channels, margin_size=margin_size, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)
        self.resamples = nn.ModuleList()
        cond_res = max(cond_width, cond_height)
        for layer in self.synthesis.layers():
            rel = layer.in_sampling_rate / cond_res
            if rel < 1:
                self.resamples.append(KaiserDownsample(scale=math.ceil(1 / rel)))
            elif rel > 1:
                self.resamples.append(KaiserUpsample(scale=math.ceil(rel)))
            else:
                self.resamples.append(nn.Identity())

    def prep_cond(self, cond: torch.Tensor) -> List[torch.Tensor]:
        """cond [N, 3, T + 2*context, h, w] -> per layer [(N T), 3*(2*context+1), in_h, in_w]:
        square canvas + margin (replicate), sliding temporal window folded into channels, then the
        layer's scale and canvas (centre crop or replicate pad)."""
        assert cond.shape[1] == self.img_channels and cond.shape[3:] == (self.cond_height, self.cond_width), cond.shape
        res = max(self.cond_width, self.cond_height)
        dx, dy = res - cond.size(4), res - cond.size(3)
        m = self.margin_size
        cond = F.pad(cond, (dx // 2 + m, (dx + 1) // 2 + m, dy // 2 + m, (dy + 1) // 2 + m, 0, 0), mode='replicate')
        win = cond.unfold(2, 2 * self.cond_context + 1, 1)                     # [N, C, T, H, W, S]
        n, c, t, h, w, s = win.shape
        cond = win.permute(0, 2, 1, 5, 3, 4).reshape(n * t, c * s, h, w)

        by_scale = {}
        out = []
        for layer, resample in zip(self.synthesis.layers(), self.resamples):
            key = (type(resample).__name__, getattr(resample, 'scale', 1))
            if key not in by_scale:
                by_scale[key] = resample(cond)
            y = by_scale[key]
            in_w, in_h = int(layer.in_size[0]), int(layer.in_size[1])
            x0 = max(0, (y.size(3) - in_w) // 2)
            y0 = max(0, (y.size(2) - in_h) // 2)
            y = y[:, :, y0:y0 + in_h, x0:x0 + in_w]
            gx, gy = in_w - y.size(3), in_h - y.size(2)
            if gx or gy:
                y = F.pad(y, (gx // 2, (gx + 1) // 2, gy // 2, (gy + 1) // 2), mode='replicate')
            out.append(y)
        return out

    def forward(self, z: torch.Tensor, cond: torch.Tensor, truncation_psi: float = 1, truncation_cutoff: Optional[int] = None,
                update_emas: bool = False, **synthesis_kwargs) -> torch.Tensor:
        t = cond.size(2) - 2 * self.cond_context
        assert t > 0 and cond.size(0) == z.size(0)
        conds = self.prep_cond(cond)
        ws = self.mapping(z.repeat_interleave(t, dim=0), truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff,
                          update_emas=update_emas)
        img = self.synthesis(ws, conds, update_emas=update_emas, **synthesis_kwargs)
        return img.reshape(z.size(0), t, *img.shape[1:]).transpose(1, 2)


class VideoGenerator(nn.Module):
    def __init__(self, hr_height: int = 256, hr_width: int = 256, lr_height: int = 32, lr_width: int = 32,
                 temporal_context: int = 4, latent_z_dim: int = 512, latent_w_dim: int = 512, margin_size: int = 10,
                 fourfeats: bool = False, num_fp16_res: int = 4, compute_dtype: torch.dtype = torch.float16,
                 **synthesis_kwargs):
        super().__init__()
        self.hr_height, self.hr_width, self.lr_height, self.lr_width = hr_height, hr_width, lr_height, lr_width
        self.temporal_context, self.latent_z_dim, self.latent_w_dim = temporal_context, latent_z_dim, latent_w_dim
        self.SG3 = Generator(z_dim=latent_z_dim, w_dim=latent_w_dim, img_width=hr_width, img_height=hr_height, img_channels=3,
                             cond_width=lr_width, cond_height=lr_height, cond_context=temporal_context,
                             margin_size=margin_size, fourfeats=fourfeats, num_fp16_res=num_fp16_res, **synthesis_kwargs)
        self.set_compute_dtype(compute_dtype)

    def set_compute_dtype(self, dtype: torch.dtype) -> None:
        for layer in self.SG3.synthesis.layers():
            layer.compute_dtype = dtype

    def sample_latent_z(self, batch_size: int, generator_z: Optional[torch.Generator] = None) -> torch.Tensor:
        device = next(self.parameters()).device
        return torch.randn(batch_size, self.latent_z_dim, generator=generator_z, device=device)

    def forward(self, lr_video: torch.Tensor, generator_z: Optional[torch.Generator] = None, magnitude_ema_beta: float = 1.0,
                latent_z: Optional[torch.Tensor] = None, **synthesis_kwargs) -> torch.Tensor:
        assert lr_video.size(2) - 2 * self.temporal_context > 0
        if latent_z is None:
            latent_z = self.sample_latent_z(lr_video.size(0), generator_z)
        return self.SG3(latent_z, lr_video, update_emas=(magnitude_ema_beta < 1), **synthesis_kwargs)

    def sample_video_segments(self, lr_video: torch.Tensor, segment_length: int = 8,
                              generator_z: Optional[torch.Generator] = None) -> Iterator[torch.Tensor]:
        """Long videos in `segment_length`-frame pieces that share one latent (each piece sees its
        own +-context low-resolution frames)."""
        ctx = self.temporal_context
        total = lr_video.size(2) - 2 * ctx
        assert total > 0 and total % segment_length == 0
        latent_z = self.sample_latent_z(lr_video.size(0), generator_z)
        for start in range(0, total, segment_length):
            yield self.SG3(latent_z, lr_video[:, :, start:start + segment_length + 2 * ctx])


# --------------------------------------------------------------------------------------------------
# Discriminator.

class SpatialBilinearUpsample(nn.Module):
    """[N, C, T, h, w] -> [N, C, T, h*scale, w*scale] with the 2*scale-tap triangle filter."""

    def __init__(self, scale: int = 2):
        super().__init__()
        assert isinstance(scale, int) and scale > 1
        self.scale = scale
        self.register_buffer('filter', _linear_filter(scale))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, c, t, h, w = x.shape
        y = upfirdn2d.upsample2d(x.reshape(n, c * t, h, w), self.filter, up=self.scale)
        return y.reshape(n, c, t, y.size(2), y.size(3))


class Conv2dLayer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, bias: bool = True, activation: str = 'linear',
                 up: int = 1, down: int = 1, resample_filter=(1, 3, 3, 1), conv_clamp: Optional[float] = None,
                 channels_last: bool = False, trainable: bool = True):
        super().__init__()
        self.activation, self.up, self.down, self.conv_clamp = activation, up, down, conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / math.sqrt(in_channels * kernel_size ** 2)
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        weight = torch.randn(out_channels, in_channels, kernel_size, kernel_size).to(memory_format=fmt)
        bias_t = torch.zeros(out_channels) if bias else None
        if trainable:
            self.weight = nn.Parameter(weight)
            self.bias = nn.Parameter(bias_t) if bias else None
        else:
            self.register_buffer('weight', weight)
            if bias:
                self.register_buffer('bias', bias_t)
            else:
                self.bias = None

    def forward(self, x: torch.Tensor, gain: float = 1) -> torch.Tensor:
        w = (self.weight * self.weight_gain).to(x.dtype)
        b = self.bias.to(x.dtype) if self.bias is not None else None
        x = conv2d_resample.conv2d_resample(x=x, w=w, f=self.resample_filter, up=self.up, down=self.down,
                                            padding=self.padding, flip_weight=(self.up == 1))
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return bias_act.bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=clamp)


class DiscriminatorBlock(nn.Module):
    def __init__(self, in_channels: int, tmp_channels: int, out_channels: int, resolution: int, img_channels: int,
                 first_layer_idx: int, architecture: str = 'resnet2', activation: str = 'lrelu', resample_filter=(1, 3, 3, 1),
                 conv_clamp: Optional[float] = None, use_fp16: bool = False, fp16_channels_last: bool = False, freeze_layers: int = 0):
        assert in_channels in (0, tmp_channels)
        assert architecture in ('orig', 'skip', 'resnet', 'resnet2')
        super().__init__()
        self.in_channels, self.out_channels, self.resolution, self.img_channels = in_channels, out_channels, resolution, img_channels
        self.first_layer_idx, self.architecture, self.use_fp16 = first_layer_idx, architecture, use_fp16
        self.channels_last = use_fp16 and fp16_channels_last
        self.compute_dtype = torch.float16
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
        self.num_layers = 0

        def next_trainable() -> bool:
            trainable = (self.first_layer_idx + self.num_layers) >= freeze_layers
            self.num_layers += 1
            return trainable

        common = dict(activation=activation, conv_clamp=conv_clamp, channels_last=self.channels_last)
        if in_channels == 0 or architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, trainable=next_trainable(), **common)
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, trainable=next_trainable(), **common)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, down=2, trainable=next_trainable(),
                                 resample_filter=resample_filter, **common)
        if architecture == 'resnet':
            self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=2, trainable=next_trainable(),
                                    resample_filter=resample_filter, channels_last=self.channels_last)

    def forward(self, x: Optional[torch.Tensor], img: Optional[torch.Tensor], force_fp32: bool = False):
        low = self.use_fp16 and not force_fp32
        dtype = self.compute_dtype if low else torch.float32
        fmt = torch.channels_last if (self.channels_last and not force_fp32) else torch.contiguous_format
        if x is not None:
            x = x.to(dtype=dtype, memory_format=fmt)
        if self.in_channels == 0 or self.architecture == 'skip':
            img = img.to(dtype=dtype, memory_format=fmt)
            y = self.fromrgb(img)
            x = y if x is None else x + y
            img = upfirdn2d.downsample2d(img, self.resample_filter) if self.architecture == 'skip' else None
        if self.architecture == 'resnet':
            y = self.skip(x)
            x = self.conv1(self.conv0(x))
            x = (x + y) * SQRT_HALF
        elif self.architecture == 'resnet2':
            y = upfirdn2d.downsample2d(x, self.resample_filter)
            y = torch.cat((y, y), dim=1)[:, :self.out_channels]
            x = self.conv1(self.conv0(x))
            x = (x + y) * SQRT_HALF
        else:
            x = self.conv1(self.conv0(x))
        return x, img


class MinibatchStdLayer(nn.Module):
    def __init__(self, group_size: Optional[int], num_channels: int = 1):
        super().__init__()
        self.group_size, self.num_channels = group_size, num_channels

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, c, h, w = x.shape
        g = n if self.group_size is None else min(self.group_size, n)
        f = self.num_channels
        y = x.reshape(g, -1, f, c // f, h, w)
        y = (y - y.mean(dim=0)).square().mean(dim=0).add(1e-8).sqrt()          # stddev over the group
        y = y.mean(dim=(2, 3, 4)).reshape(-1, f, 1, 1).repeat(g, 1, h, w)
        return torch.cat((x, y), dim=1)


class DiscriminatorEpilogue(nn.Module):
    def __init__(self, in_channels: int, height: int, width: int, mbstd_group_size: Optional[int] = 4, mbstd_num_channels: int = 1,
                 activation: str = 'lrelu', conv_clamp: Optional[float] = None, output_dim: int = 1, pool_mode: str = 'fully_connected'):
        assert pool_mode in ('fully_connected', 'average')
        super().__init__()
        self.in_channels, self.height, self.width, self.pool_mode = in_channels, height, width, pool_mode
        self.mbstd = MinibatchStdLayer(mbstd_group_size, mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * height * width, in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, output_dim)

    def forward(self, x: torch.Tensor, conditioning: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = x.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.conv(x)
        x = self.fc(x.flatten(1)) if self.pool_mode == 'fully_connected' else x.mean(dim=(2, 3))
        x = self.out(x)
        if conditioning is not None:
            assert conditioning.shape == x.shape
            x = (x * conditioning).sum(dim=1, keepdim=True) * (1 / math.sqrt(conditioning.size(1)))
        return x


class VideoDiscriminator(nn.Module):
    def __init__(self, channels: int = 3, seq_length: int = 8, lr_height: int = 32, lr_width: int = 32,
                 hr_height: int = 256, hr_width: int = 256, channels_base: int = 16384, channels_max: int = 512,
                 num_fp16_res: int = 4, conv_clamp: Optional[int] = 256, minibatch_std_group_size: int = 4,
                 minibatch_std_num_channels: int = 0, architecture: str = 'resnet', pool_mode: str = 'fully_connected',
                 compute_dtype: torch.dtype = torch.float16, fp16_channels_last: bool = False):
        super().__init__()
        self.channels, self.seq_length = channels, seq_length
        self.lr_height, self.lr_width, self.hr_height, self.hr_width = lr_height, lr_width, hr_height, hr_width
        res = max(hr_height, hr_width)
        res_log2 = int(np.log2(res))
        self.block_resolutions = [2 ** i for i in range(res_log2, 2, -1)]
        width = {r: min(channels_base // r, channels_max) for r in self.block_resolutions + [4]}
        fp16_res = max(2 ** (res_log2 + 1 - num_fp16_res), 8)
        img_channels = 2 * channels * seq_length                                # low-res and high-res frames, stacked
        layer_idx = 0
        for r in self.block_resolutions:
            block = DiscriminatorBlock(width[r] if r < res else 0, width[r], width[r // 2], resolution=r, img_channels=img_channels,
                                       first_layer_idx=layer_idx, use_fp16=(r >= fp16_res), conv_clamp=conv_clamp,
                                       architecture=architecture, fp16_channels_last=fp16_channels_last)
            block.compute_dtype = compute_dtype
            setattr(self, f'b{r}', block)
            layer_idx += block.num_layers
        self.b4 = DiscriminatorEpilogue(width[4], height=4, width=4, mbstd_group_size=minibatch_std_group_size,
                                        mbstd_num_channels=minibatch_std_num_channels, output_dim=1, conv_clamp=conv_clamp,
                                        pool_mode=pool_mode)
        self.upsample = SpatialBilinearUpsample(res // max(lr_height, lr_width))

    def forward(self, lr_video: torch.Tensor, hr_video: torch.Tensor) -> torch.Tensor:
        if lr_video.shape[3:] == (self.lr_height, self.lr_width):
            lr_video = self.upsample(lr_video)
        assert lr_video.shape[3:] == (self.hr_height, self.hr_width), lr_video.shape
        video = torch.cat((lr_video, hr_video), dim=1)                          # [N, 2C, T, H, W]
        p = (video.size(4) - video.size(3)) // 2
        video = F.pad(video, (0, 0, p, p))                                      # letter-box to a square
        img = video.flatten(1, 2)                                               # channels = (c t)
        x = None
        for r in self.block_resolutions:
            x, img = getattr(self, f'b{r}')(x, img)
        return self.b4(x)
