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

import os

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
# float32 tensors on the same kernels with float32 accuracy: operands split into THREE bfloat16 parts.
#   t = t1 + t2 + t3 exactly (8 + 8 + 8 mantissa bits, the exponent range of float32), and
#   x . w ~= x1 w1 + x1 w2 + x2 w1 + x1 w3 + x2 w2 + x3 w1            (the dropped terms are below 2^-24 of the product)
# runs as ONE contraction over six times the channels, [x1 | x1 | x2 | x1 | x2 | x3] against [w1 | w2 | w1 | w3 | w2 | w1], with float32
# accumulators stored unrounded (lvg_conv3d_frames_ex, out_dtype float32). The reference runs the lres networks in float32 with TF32
# off (train_lres.py:267-269): this is the route that keeps that precision on the 16-bit matrix cores. It costs six (weight gradient:
# nine) times the multiplications -- a parity / default-precision route, not the benchmark's. (Two float16 parts with a power-of-two
# scale are cheaper -- the sres generator's float32 layers use that -- but represent exactly only the elements within ~2^13 of a tensor's
# maximum; bfloat16 parts need no scale. Every call of the two networks replayed on random operands: 1e-6 .. 5e-6 of the float64 result.)

_X6, _W6 = (0, 0, 1, 0, 1, 2), (0, 1, 0, 2, 1, 0)


def split_bf16x3(t):
    """float32 -> three bfloat16 tensors whose sum is t to 24 bits."""
    t = t.float()
    b1 = t.bfloat16()
    r = t - b1.float()
    b2 = r.bfloat16()
    b3 = (r - b2.float()).bfloat16()
    return b1, b2, b3


SPLIT_STACK_HIP = os.environ.get('LVG_SPLIT_STACK_HIP', '1') == '1'


def split32_stack(x, parts):
    """x [(T N), C, H, W] float32 -> [(T N), len(parts) C, H, W] bfloat16 channels-last whose channel block k holds part parts[k] (0 .. 2) of
    `split_bf16x3(x)`: one pass (csrc/split32.hip) for channels-last GPU tensors with C % 8 == 0, else the tensor expressions. Bit-identical."""
    xc = x.contiguous(memory_format=torch.channels_last)
    f, c, h, w = xc.shape
    if SPLIT_STACK_HIP and xc.is_cuda and xc.dtype == torch.float32 and c % 8 == 0 and _pixel_stride(xc) == c and xc.data_ptr() % 16 == 0 and _init():
        out = torch.empty((f, len(parts) * c, h, w), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        pattern = sum(int(p) << (2 * k) for k, p in enumerate(parts))
        with torch.cuda.device(x.device):
            rc = _hip.lib().lvg_split32_stack(xc.data_ptr(), out.data_ptr(), f * h * w, c, c, len(parts), pattern, _hip.stream(x.device))
        _hip.check(rc, 'lvg_split32_stack')
        return out
    xs = split_bf16x3(xc)
    return torch.cat([xs[i] for i in parts], dim=1).contiguous(memory_format=torch.channels_last)


def split32_shape_ok(frames, h, w, ci, co, kt, kh, kw):
    """Is there a float32-output kernel for a convolution with ci input / co output channels (before the six-fold stacking)?"""
    if ci % 64 or co % 64 or not (kt & 1 and kh & 1 and kw & 1):
        return False
    if frames * h * w * 6 * ci * 2 >= 2 ** 32:
        return False
    return _init() and int(_hip.lib().lvg_conv3d_frames_workgroups_f32out(frames, h, w, 6 * ci, co, kt, kh, kw)) > 0


def split32_supported(x, weight):
    if x.device.type != 'cuda' or x.dtype != torch.float32 or x.dim() != 4 or weight.dim() != 5 or weight.shape[1] != x.shape[1]:
        return False
    f, ci, h, w = x.shape
    return split32_shape_ok(f, h, w, ci, weight.shape[0], *weight.shape[2:])


def conv3d_frames_split32(x, weight, shift, pre=None, b=None, res=None, post=None, act='linear', alpha=None, gain=None,
                          clamp=None, want_msq=False, keep_sum=True):
    """`conv3d_frames_forward` for float32 tensors: x [(T N), Ci, H, W] float32 (any layout), weight [Co, Ci, kt, kh, kw] float32 ->
    (out, ysum | None, mean_square | None), float32 channels-last."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    f, ci, h, w = x.shape
    co, _, kt, kh, kw = weight.shape
    assert split32_supported(x, weight), 'conv3d_frames_split32: no kernel for this shape'
    ws = split_bf16x3(weight)
    x6 = split32_stack(x, _X6)                                                                     # [f, 6 ci, h, w] bfloat16
    w6 = pack_weight(torch.cat([ws[i] for i in _W6], dim=1))                                       # [kt, kh, kw, co, 6 ci]
    out = torch.empty((f, co, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    ysum = torch.empty_like(out) if keep_sum else None
    part = torch.empty(int(_hip.lib().lvg_conv3d_frames_workgroups_f32out(f, h, w, 6 * ci, co, kt, kh, kw)), dtype=torch.float32, device=x.device) if want_msq else None
    if res is not None:
        res = res.float().contiguous(memory_format=torch.channels_last)
    pre = None if pre is None else pre.float().contiguous()
    b = None if b is None else b.float().contiguous()
    post = None if post is None else post.float().contiguous()
    with torch.cuda.device(x.device):
        rc = _hip.lib().lvg_conv3d_frames_ex(
            x6.data_ptr(), w6.data_ptr(), _hip.ptr(pre), _hip.ptr(b), _hip.ptr(res), _hip.ptr(post), out.data_ptr(), _hip.ptr(ysum), _hip.ptr(part),
            f, h, w, 6 * ci, co, kt, kh, kw, shift, 6 * ci, _hip.dtype_code(torch.bfloat16), _hip.dtype_code(torch.float32), spec.cuda_idx, alpha, gain, clamp,
            _hip.stream(x.device))
    _hip.check(rc, 'conv3d_frames_ex')
    stats['flops'] += 2 * f * h * w * co * ci * kt * kh * kw             # algorithmic (the six partial products are this route's cost)
    stats['launches'] += 1
    return out, ysum, (part.sum() / float(out.numel()) if want_msq else None)


def conv3d_frames_split32_dgrad(dy, weight, shift):
    """Data gradient of the convolution for float32 tensors: dy [(T N), Co, H, W], weight [Co, Ci, kt, kh, kw] -> [(T N), Ci, H, W] float32."""
    wt = weight.flip(2, 3, 4).transpose(0, 1)                            # taps mirrored, channel roles exchanged
    return conv3d_frames_split32(dy, wt, shift, keep_sum=False)[0]


def conv3d_frames_split32_wgrad(x, dy, kt, kh, kw, shift):
    """Weight gradient [Co, Ci, kt, kh, kw] float32 from float32 x [(T N), Ci, H, W], dy [(T N), Co, H, W]: [x1 | x2 | x3] against
    [g1 | g2 | g3] on the 16-bit weight-gradient kernel; of the nine blocks of its [3 Co, 3 Ci] result the six significant ones are added."""
    ci, co = x.shape[1], dy.shape[1]
    x3, g3 = split32_stack(x, (0, 1, 2)), split32_stack(dy, (0, 1, 2))
    flops0 = stats['flops']
    gw = conv3d_frames_wgrad(x3, g3, kt, kh, kw, shift)                  # [3 co, 3 ci, kt, kh, kw]
    stats['flops'] = flops0 + 2 * x.shape[0] * x.shape[2] * x.shape[3] * co * ci * kt * kh * kw
    blk = lambda i, j: gw[i * co:(i + 1) * co, j * ci:(j + 1) * ci]
    return ((blk(2, 0) + blk(1, 1) + blk(0, 2)) + (blk(1, 0) + blk(0, 1))) + blk(0, 0)      # small terms first


def split32_wgrad_supported(x, dy, kt, kh, kw):
    if x.device.type != 'cuda' or x.dtype != torch.float32 or dy.dtype != torch.float32 or x.shape[2:] != dy.shape[2:] or x.shape[0] != dy.shape[0]:
        return False
    return _init() and wgrad_splits(x.shape[0], x.shape[2], x.shape[3], 3 * x.shape[1], 3 * dy.shape[1], kt, kh, kw) > 0


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


def pointwise_wgrad_supported(x, dy):
    """x [(T N), Ci, H, W], dy [(T N), Co, H, W]: 16-bit channels-last GPU tensors (or channel slices), channels multiples of 64, pixels a multiple of 8."""
    if not (x.is_cuda and dy.is_cuda and x.dim() == 4 and dy.dim() == 4 and x.dtype == dy.dtype and x.dtype in (torch.float16, torch.bfloat16)):
        return False
    if x.shape[0] != dy.shape[0] or x.shape[2:] != dy.shape[2:] or not _init():
        return False
    sx, sy = _pixel_stride(x), _pixel_stride(dy)
    if sx is None or sy is None or sx % 8 or sy % 8 or x.data_ptr() % 16 or dy.data_ptr() % 16:
        return False
    pixels = x.shape[0] * x.shape[2] * x.shape[3]
    return int(_hip.lib().lvg_pointwise_wgrad_splits(pixels, x.shape[1], dy.shape[1])) > 0


def pointwise_wgrad_splits(pixels, ci, co):
    """> 0: lvg_pointwise_wgrad takes a [pixels, ci] x [pixels, co] pair of dense 16-bit pixel matrices (decided on shapes)."""
    if not _init() or ci % 64 or co % 64:
        return 0
    return int(_hip.lib().lvg_pointwise_wgrad_splits(pixels, ci, co))


def pointwise_wgrad(x, dy):
    """Weight gradient [Co, Ci] (float32) of a 1 x 1 convolution: sum over pixels dy[m][co] x[m][ci] (csrc/pointwise_wgrad.hip)."""
    assert pointwise_wgrad_supported(x, dy), 'pointwise_wgrad: no hand-written kernel for this shape / dtype / layout'
    f, ci, h, w = x.shape
    co = dy.shape[1]
    pixels = f * h * w
    splits = int(_hip.lib().lvg_pointwise_wgrad_splits(pixels, ci, co))
    part = torch.empty((splits, co, ci), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _hip.lib().lvg_pointwise_wgrad(x.data_ptr(), dy.data_ptr(), part.data_ptr(), _zeros(x.device).data_ptr(), pixels, ci, co,
                                            _pixel_stride(x), _pixel_stride(dy), splits, _hip.dtype_code(x.dtype), _hip.stream(x.device))
    _hip.check(rc, 'lvg_pointwise_wgrad')
    stats['flops'] += 2 * pixels * co * ci
    stats['launches'] += 1
    return part.sum(0) if splits > 1 else part[0]                    # fixed summation order: reproducible


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
