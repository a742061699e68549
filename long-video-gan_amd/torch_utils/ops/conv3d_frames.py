"""Dense contraction of the modulated convolutions of the low-resolution generator as ONE hand-written kernel:

    acc = conv3d(x, weight)            'same' zero padding in time / rows / columns, over time-major frames
    out = clamp(act(acc * pre[f, c] + b[c] + res) * gain) * post[f, c]       (+ mean(value ** 2) before `post`)

This is `F.conv3d` of the reference's `temporal_modulated_conv3d` (model/generator_lres.py:119; padding = k // 2,
:544-548) with the demodulation (:122), bias_act (:570), the next layer's modulation (:101) and the magnitude
statistic (:574) applied while the accumulators are still in registers -- `lvg_conv3d_frames`
(csrc/conv3d_igemm.hip: implicit GEMM on v_mfma_f32_32x32x16, LDS-DMA staging). The temporal taps are part of the
K loop, so nothing kt times the size of the output is ever written (the MIOpen route runs one 2-D convolution over
kt-stacked output channels and sums the taps in `lvg_tapconv_epilogue`).

Launch-level interface (used by lvg.models.lres); CPU tensors and unsupported shapes take the explicit PyTorch
composition below, which is also the definition the GPU tests compare against (next to the C oracle)."""

import torch
import torch.nn.functional as F

from . import _hip
from .modconv_epilogue import _init, _ref, _resolve


stats = {'flops': 0, 'launches': 0}     # algorithmic work handed to the hand-written kernel (read by bench.py's FLOP tally)


def pack_weight(weight):
    """[Co, Ci, kt, kh, kw] -> [kt, kh, kw, Co, Ci] contiguous (tap-major, input channel fastest)."""
    return weight.permute(2, 3, 4, 0, 1).contiguous()


def workgroups(frames, h, w, ci, co, kt, kh, kw):
    """Workgroups `lvg_conv3d_frames` launches for this shape; 0 when there is no kernel for it."""
    return int(_hip.lib().lvg_conv3d_frames_workgroups(frames, h, w, ci, co, kt, kh, kw))


def _pixel_stride(x):
    """Elements between consecutive pixels when x [(T N), C, H, W] is channels-last or a channel slice of a channels-last
    tensor (what the kernel can walk), else None."""
    if x.dim() != 4:
        return None
    f, c, h, w = x.shape
    s = x.stride(3) if w > 1 else (x.stride(2) if h > 1 else x.stride(0))
    if c > 1 and x.stride(1) != 1:
        return None
    if s < c or (w > 1 and x.stride(3) != s) or (h > 1 and x.stride(2) != w * s) or (f > 1 and x.stride(0) != h * w * s):
        return None
    return s


def supported(x, weight):
    """True when the hand-written kernel takes (x [(T N), Ci, H, W] channels-last -- or a channel slice of a
    channels-last tensor --, weight [Co, Ci, kt, kh, kw])."""
    if x.device.type != 'cuda' or x.dtype not in (torch.float16, torch.bfloat16) or weight.dtype != x.dtype:
        return False
    if x.dim() != 4 or weight.dim() != 5:
        return False
    s = _pixel_stride(x)
    if s is None or s % 8 != 0 or x.data_ptr() % 16 != 0 or x.shape[0] * x.shape[2] * x.shape[3] * s * 2 >= 2 ** 32:
        return False
    f, ci, h, w = x.shape
    co, ci2, kt, kh, kw = weight.shape
    if ci != ci2 or not (kt & 1 and kh & 1 and kw & 1):
        return False
    return _init() and workgroups(f, h, w, ci, co, kt, kh, kw) > 0


def _conv_ref(x, weight, shift):
    """float32 conv3d over time-major frames [(T N), Ci, H, W] -> [(T N), Co, H, W]."""
    tn, ci, h, w = x.shape
    kt, kh, kw = weight.shape[2:]
    v = x.float().reshape(tn // shift, shift, ci, h, w).permute(1, 2, 0, 3, 4)              # [N, Ci, T, H, W]
    y = F.conv3d(v, weight.float(), padding=(kt // 2, kh // 2, kw // 2))
    return y.permute(2, 0, 1, 3, 4).reshape(tn, weight.shape[0], h, w)


def conv3d_frames_forward(x, weight, shift, pre=None, b=None, res=None, post=None, act='linear', alpha=None, gain=None,
                          clamp=None, want_msq=False, keep_sum=True, packed=None):
    """-> (out, ysum | None, mean_square | None); out / ysum [(T N), Co, H, W] in x's dtype, channels-last.

    x [(T N), Ci, H, W] channels-last or a channel slice of a channels-last tensor (frame f = t * shift + n); weight [Co, Ci, kt, kh, kw] in x's dtype (`packed`:
    the same weight already through `pack_weight`); pre / post float32 [(T N), Co]; b [Co]; res like out."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    f, ci, h, w = x.shape
    co, _, kt, kh, kw = weight.shape
    if x.device.type == 'cuda' and _init():
        assert supported(x, weight), 'conv3d_frames: no hand-written kernel for this shape / dtype / layout'
        wp = pack_weight(weight) if packed is None else packed
        out = torch.empty((f, co, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ysum = torch.empty_like(out) if keep_sum else None
        part = torch.empty(workgroups(f, h, w, ci, co, kt, kh, kw), dtype=torch.float32, device=x.device) if want_msq else None
        if res is not None:
            res = res.contiguous(memory_format=torch.channels_last)
        pre = pre.contiguous() if pre is not None else None
        post = post.contiguous() if post is not None else None
        with torch.cuda.device(x.device):
            rc = _hip.lib().lvg_conv3d_frames(
                x.data_ptr(), wp.data_ptr(), _hip.ptr(pre), _hip.ptr(b), _hip.ptr(res), _hip.ptr(post),
                out.data_ptr(), _hip.ptr(ysum), _hip.ptr(part),
                f, h, w, ci, co, kt, kh, kw, shift, _pixel_stride(x), _hip.dtype_code(x.dtype), spec.cuda_idx, alpha, gain, clamp, _hip.stream(x.device))
        _hip.check(rc, 'conv3d_frames')
        stats['flops'] += 2 * f * h * w * co * ci * kt * kh * kw
        stats['launches'] += 1
        return out, ysum, (part.sum() / float(out.numel()) if want_msq else None)
    acc = _conv_ref(x, weight, shift)
    ysum = acc.to(x.dtype)
    out, msq = _ref(acc, pre, b, post, act, alpha, gain, clamp, want_msq, res=res)
    return out.to(x.dtype), (ysum if keep_sum else None), msq


# ----------------------------------------------------------------------------------------------------
# Weight gradient (csrc/conv3d_wgrad.hip).

_zero_pages = {}


def _zeros(device):
    z = _zero_pages.get(device)
    if z is None:
        z = _zero_pages[device] = torch.zeros(256, dtype=torch.uint8, device=device)
    return z


def wgrad_splits(frames, h, w, ci, co, kt, kh, kw):
    """Pixel ranges `lvg_conv3d_frames_wgrad` splits this shape into; 0 when there is no kernel for it."""
    return int(_hip.lib().lvg_conv3d_frames_wgrad_splits(frames, h, w, ci, co, kt, kh, kw))


def wgrad_supported(x, dy, kt, kh, kw):
    if x.device.type != 'cuda' or x.dtype not in (torch.float16, torch.bfloat16) or dy.dtype != x.dtype:
        return False
    sx, sd = _pixel_stride(x), _pixel_stride(dy)
    if sx is None or sd is None or sx % 8 or sd % 8 or x.data_ptr() % 16 or dy.data_ptr() % 16 or x.shape[2:] != dy.shape[2:] or x.shape[0] != dy.shape[0]:
        return False
    return _init() and wgrad_splits(x.shape[0], x.shape[2], x.shape[3], x.shape[1], dy.shape[1], kt, kh, kw) > 0


def _wgrad_ref(x, dy, kt, kh, kw, shift):
    """float32 weight gradient through autograd of the plain definition."""
    co, ci = dy.shape[1], x.shape[1]
    w = torch.zeros(co, ci, kt, kh, kw, dtype=torch.float32, device=x.device, requires_grad=True)
    with torch.enable_grad():
        y = _conv_ref(x, w, shift)
    return torch.autograd.grad(y, w, dy.float())[0]


def conv3d_frames_wgrad(x, dy, kt, kh, kw, shift):
    """Weight gradient [Co, Ci, kt, kh, kw] (float32) of `conv3d_frames_forward`'s contraction:
    x [(T N), Ci, H, W], dy [(T N), Co, H, W], both channels-last (or channel slices of channels-last tensors)."""
    f, ci, h, w = x.shape
    co = dy.shape[1]
    if x.device.type == 'cuda' and _init():
        assert wgrad_supported(x, dy, kt, kh, kw), 'conv3d_frames_wgrad: no hand-written kernel for this shape / dtype / layout'
        splits = wgrad_splits(f, h, w, ci, co, kt, kh, kw)
        part = torch.empty((splits, kt, kh * kw, co, ci), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _hip.lib().lvg_conv3d_frames_wgrad(
                x.data_ptr(), dy.data_ptr(), part.data_ptr(), _zeros(x.device).data_ptr(),
                f, h, w, ci, co, kt, kh, kw, shift, _pixel_stride(x), _pixel_stride(dy), splits, _hip.dtype_code(x.dtype), _hip.stream(x.device))
        _hip.check(rc, 'conv3d_frames_wgrad')
        stats['flops'] += 2 * f * h * w * co * ci * kt * kh * kw
        stats['launches'] += 1
        gw = part.sum(0) if splits > 1 else part[0]                  # fixed summation order: reproducible
        return gw.reshape(kt, kh, kw, co, ci).permute(3, 4, 0, 1, 2)
    return _wgrad_ref(x, dy, kt, kh, kw, shift)
