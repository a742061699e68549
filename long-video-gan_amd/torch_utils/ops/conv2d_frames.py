"""Dense 3 x 3 contraction of the super-resolution networks on the hand-written kernels: `lvg_conv2d_frames`
(csrc/conv2d_igemm.hip: implicit GEMM on v_mfma_f32_32x32x16 over 8 x 16 pixel tiles, forward and -- with the weight
mirrored and its channel roles exchanged -- data gradient) and `lvg_conv2d_frames_wgrad` (csrc/conv2d_wgrad.hip).

This is the `conv2d_gradfix.conv2d(x, w, padding = p)` inside the reference's `modulated_conv2d`
(model/generator_sres.py:63-66; F.conv2d on torch >= 1.11, torch_utils/ops/conv2d_gradfix.py:37-45) and what autograd
derives for it, restated on channels-last frames whose ZERO PADDING IS WRITTEN IN MEMORY: the layout prologue that
produces the frames (modconv2d_layout) writes the interior of a zero-filled, slightly larger frame, after which the
convolution is a 'valid' correlation and neither kernel has a mask or an edge case.

Coordinates (`Geometry`): a plane of h x w pixels convolved with 3 x 3 taps and padding p in {0, 1, 2}.
  input frame    hx x wx, the plane at (2, 2):   hx = hd + 2, wx = wd + 2
  valid positions (a, b) of the correlation over that frame: output pixel (oy, ox) of the padded convolution is position
                 (oy + q, ox + q) with q = 2 - p; positions outside the output carry no gradient
  gradient frame hd x wd = the valid positions rounded up to whole 4 x 16 pixel patches (the K-step of the weight
                 gradient): hd = ceil4(h + 2), wd = ceil16(w + 2); the output gradient sits at (q, q), zeros elsewhere
so that  gw[dh][dw] = sum dy_frame[a][b] * x_frame[a + dh][b + dw]  and  dx[y][x] = sum dy_frame[y + dh][x + dw] * w[2 - dh][2 - dw]
hold without offsets. Tensors here are plain contiguous [N, H, W, C] arrays; channel counts are multiples of 64.

CPU tensors take the explicit PyTorch composition below (the definition the GPU tests compare against, next to the C oracle)."""

import torch
import torch.nn.functional as F

from . import _hip
from .modconv_epilogue import _init

CH = 64            # channel counts of the kernels' operands are multiples of this
PATCH_H, PATCH_W = 4, 16

stats = {'flops': 0, 'launches': 0}     # algorithmic work handed to the hand-written kernels (read by bench.py's FLOP tally)


def round_up(v, m):
    return (v + m - 1) // m * m


class Geometry:
    def __init__(self, h, w, padding, k=3):
        assert k == 3 and 0 <= padding <= 2, 'hand-written 2-D contraction: 3 x 3 taps, padding 0..2'
        self.h, self.w, self.padding = h, w, padding
        self.ho, self.wo = h + 2 * padding - 2, w + 2 * padding - 2
        self.q = 2 - padding
        self.hd, self.wd = round_up(h + 2, PATCH_H), round_up(w + 2, PATCH_W)
        self.hx, self.wx = self.hd + 2, self.wd + 2


def pack_weight(weight, dtype, ci_pad, co_pad):
    """[Co, Ci, 3, 3] -> [3, 3, co_pad, ci_pad] in `dtype` (tap-major, input channel fastest), zero-padded channels."""
    co, ci = weight.shape[:2]
    wp = torch.zeros([3, 3, co_pad, ci_pad], dtype=dtype, device=weight.device)
    wp[:, :, :co, :ci] = weight.to(dtype).permute(2, 3, 0, 1)
    return wp


def pack_weight_dgrad(weight, dtype, ci_pad, co_pad):
    """[Co, Ci, 3, 3] -> [3, 3, ci_pad, co_pad]: taps mirrored, channel roles exchanged (the weight of the data gradient)."""
    co, ci = weight.shape[:2]
    wp = torch.zeros([3, 3, ci_pad, co_pad], dtype=dtype, device=weight.device)
    wp[:, :, :ci, :co] = weight.to(dtype).flip(2, 3).permute(2, 3, 1, 0)
    return wp


def supported(x, wp):
    """x [N, Hi, Wi, Ci] contiguous, wp [3, 3, Co, Ci] contiguous, 16-bit, on the GPU, channels multiples of 64."""
    if x.device.type != 'cuda' or x.dtype not in (torch.float16, torch.bfloat16) or wp.dtype != x.dtype:
        return False
    if x.dim() != 4 or wp.dim() != 4 or not x.is_contiguous() or not wp.is_contiguous():
        return False
    if wp.shape[0] != 3 or wp.shape[1] != 3 or wp.shape[3] != x.shape[3] or x.shape[3] % CH or wp.shape[2] % CH:
        return False
    return x.numel() * 2 < 2 ** 32 and _init()


def conv2d_valid(x, wp, ho, wo, offset=(0, 0), pre=None, out_dtype=None, alg_flops=None):
    """out[n, oy, ox, co] = pre[n, co] * sum x[n, oy + offset[0] + dh, ox + offset[1] + dw, ci] * wp[dh, dw, co, ci].

    x [N, Hi, Wi, Ci], wp [3, 3, Co, Ci] -> out [N, ho, wo, Co] (x's dtype, or float32 with out_dtype=torch.float32: the accumulators
    unrounded); ho <= Hi - offset[0] - 2, wo <= Wi - offset[1] - 2. `alg_flops`: the algorithmic work of the call (true channel counts:
    the tensors here are zero-padded to multiples of 64), for the measurement tally; default = the work as launched."""
    out_dtype = x.dtype if out_dtype is None else out_dtype
    n, hi, wi, ci = x.shape
    co = wp.shape[2]
    assert ho <= hi - offset[0] - 2 and wo <= wi - offset[1] - 2 and wp.shape == (3, 3, co, ci)
    if x.device.type == 'cuda' and _init():
        assert supported(x, wp), 'conv2d_frames: no hand-written kernel for this shape / dtype / layout'
        out = torch.empty([n, ho, wo, co], dtype=out_dtype, device=x.device)
        pre = None if pre is None else pre.float().contiguous()
        with torch.cuda.device(x.device):
            rc = _hip.lib().lvg_conv2d_frames(x.data_ptr(), wp.data_ptr(), _hip.ptr(pre), out.data_ptr(), n, hi, wi, ho, wo, ci, co, 3, 3,
                                              offset[0], offset[1], ci, co, _hip.dtype_code(x.dtype), _hip.dtype_code(out_dtype), _hip.stream(x.device))
        _hip.check(rc, 'conv2d_frames')
        stats['flops'] += 2 * n * ho * wo * co * ci * 9 if alg_flops is None else alg_flops
        stats['launches'] += 1
        return out
    v = x[:, offset[0]:offset[0] + ho + 2, offset[1]:offset[1] + wo + 2].permute(0, 3, 1, 2).float()
    y = F.conv2d(v, wp.permute(2, 3, 0, 1).float())
    if pre is not None:
        y = y * pre.float()[:, :, None, None]
    return y.permute(0, 2, 3, 1).contiguous().to(out_dtype)


def conv2d_valid_planes(x, wp, ho, wo, co_out, offset=(0, 0), pre=None, alg_flops=None, dot=None):
    """The contraction of `conv2d_valid` with the result as NCHW planes: out [N, co_out, ho, wo] = pre[n, co] * acc for the first co_out of wp's (padded)
    output channels, in x's dtype with ONE rounding (lvg_conv2d_frames_planes; wo even on the GPU). pre float32 [N, co_out] or None.
    dot = (a, b): NCHW tensors [N, ca, ho, wo], [N, cb, ho, wo] (b may be None) of x's dtype -> also returns partial [N, rows, ca + cb] float32 with
    partial.sum(1)[n, c] = sum over the pixels of acc[n, pixel, c] * cat(a, b)[n, c, pixel] (the accumulators BEFORE `pre`)."""
    n, hi, wi, ci = x.shape
    co = wp.shape[2]
    assert ho <= hi - offset[0] - 2 and wo <= wi - offset[1] - 2 and wp.shape == (3, 3, co, ci) and 1 <= co_out <= co
    da, db = dot if dot is not None else (None, None)
    ca, cb = (0 if da is None else da.shape[1]), (0 if db is None else db.shape[1])
    if x.device.type == 'cuda' and _init():
        assert supported(x, wp) and wo % 2 == 0, 'conv2d_frames_planes: no hand-written kernel for this shape / dtype / layout'
        out = torch.empty([n, co_out, ho, wo], dtype=x.dtype, device=x.device)
        pre = None if pre is None else pre.float().contiguous()
        assert pre is None or tuple(pre.shape) == (n, co_out)
        with torch.cuda.device(x.device):
            if da is None:
                partial = None
                rc = _hip.lib().lvg_conv2d_frames_planes(x.data_ptr(), wp.data_ptr(), _hip.ptr(pre), out.data_ptr(), n, hi, wi, ho, wo, ci, co, co_out, 3, 3,
                                                         offset[0], offset[1], ci, _hip.dtype_code(x.dtype), _hip.stream(x.device))
            else:
                assert da.is_contiguous() and tuple(da.shape) == (n, ca, ho, wo) and da.dtype == x.dtype and ca + cb <= co
                assert db is None or (db.is_contiguous() and tuple(db.shape) == (n, cb, ho, wo) and db.dtype == x.dtype)
                rows = int(_hip.lib().lvg_conv2d_frames_planes_dot_rows(ho, wo))
                partial = torch.empty([n, rows, ca + cb], dtype=torch.float32, device=x.device)
                rc = _hip.lib().lvg_conv2d_frames_planes_dot(x.data_ptr(), wp.data_ptr(), _hip.ptr(pre), out.data_ptr(), da.data_ptr(), _hip.ptr(db), partial.data_ptr(), ca, cb,
                                                             n, hi, wi, ho, wo, ci, co, co_out, 3, 3, offset[0], offset[1], ci, _hip.dtype_code(x.dtype), _hip.stream(x.device))
        _hip.check(rc, 'conv2d_frames_planes')
        stats['flops'] += 2 * n * ho * wo * co * ci * 9 if alg_flops is None else alg_flops
        stats['launches'] += 1
        return out if dot is None else (out, partial)
    v = x[:, offset[0]:offset[0] + ho + 2, offset[1]:offset[1] + wo + 2].permute(0, 3, 1, 2).float()
    acc = F.conv2d(v, wp.permute(2, 3, 0, 1).float())
    y = acc[:, :co_out]
    if pre is not None:
        y = y * pre.float()[:, :, None, None]
    y = y.contiguous().to(x.dtype)
    if dot is None:
        return y
    oth = da if db is None else torch.cat((da, db), dim=1)
    return y, (acc[:, :ca + cb] * oth.float()).sum(dim=(2, 3))[:, None, :]


def split16(t, dtype=torch.float16):
    """float32 tensor -> (high, low) parts in `dtype` with high + low = t to ~2^-22 relative (float16: 11 + 11 mantissa bits).
    The operand side of a float32-accurate contraction on the 16-bit matrix cores:
        x . w  ~=  xh . wh + xl . wh + xh . wl        (the dropped xl . wl term is ~2^-22 of the product)
    run as ONE contraction over three times the channels, [xh | xl | xh] against [wh | wh | wl], with float32 accumulation and
    float32 output (lvg_conv2d_frames, out_dtype float32). float16 keeps 11 bits per part but has a narrow exponent: the callers
    keep the operands O(1) (activations, normalised weights) or scale them by a power of two first (gradients)."""
    hi = t.to(dtype)
    lo = (t - hi.float()).to(dtype)
    return hi, lo


def pow2_scale(t, target=1024.0):
    """A power of two s (0-d float32 tensor, computed on the device without synchronising) that brings max |t| to [target / 2, target]."""
    amax = t.detach().abs().amax().clamp_min(1e-30).float()
    return torch.exp2(torch.floor(torch.log2(target / amax)))


def split16_into_frame(src, mul, frame, off_y, off_x, c_pad, pattern=(0, 1, 0), target=1024.0):
    """The float16x2 operand of a float32 tensor in one pass (csrc/modconv2d_layout.hip::split16_frames_kernel): src float32 NCHW [n, c, h, w]
    (contiguous, GPU), mul [n, c] float32 or None; frame [n, H, W, len(pattern) * c_pad] float16, zero-filled by the caller. Writes
    part pattern[blk] of (src * mul * s) into channel block blk of the frame's interior at (off_y, off_x) and returns the power-of-two scale s
    (0-d float32 tensor) -- bit for bit what  pow2_scale(src * mul), split16(src * mul * s)  and the strided copies produce."""
    n, c, h, w = src.shape
    assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and frame.dtype == torch.float16 and frame.is_contiguous()
    assert frame.shape[0] == n and frame.shape[3] == len(pattern) * c_pad and all(p_ in (0, 1) for p_ in pattern)
    planes = torch.empty([n, c], dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _hip.check(_hip.lib().lvg_plane_absmax(src.data_ptr(), planes.data_ptr(), n * c, h * w, _hip.dtype_code(src.dtype), _hip.stream(src.device)), 'plane_absmax')
        # max |src * mul| = max over (n, c) of |mul| * max over the plane of |src| (the float32 rounding of a product is monotonic)
        amax = (planes if mul is None else planes * mul.abs()).amax().clamp_min(1e-30)
        s = torch.exp2(torch.floor(torch.log2(target / amax)))
        bits = sum(int(p_) << i for i, p_ in enumerate(pattern))
        rc = _hip.lib().lvg_split16_frames(src.data_ptr(), _hip.ptr(mul), s.data_ptr(), frame.data_ptr(), n, c, h, w, frame.shape[1], frame.shape[2],
                                           off_y, off_x, c_pad, len(pattern), bits, _hip.stream(src.device))
    _hip.check(rc, 'split16_frames')
    return s


def nhwc_f32_to_nchw(y, c_dst, scale=None, factor=None):
    """y float32 [n, h, w, c_src] (contiguous, GPU) -> float32 NCHW [n, c_dst, h, w] = y[..., :c_dst] * scale[n, c] * factor (0-d tensor), one pass."""
    n, h, w, c_src = y.shape
    assert y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and c_dst <= c_src
    out = torch.empty([n, c_dst, h, w], dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        rc = _hip.lib().lvg_nhwc_f32_to_nchw(y.data_ptr(), _hip.ptr(scale), _hip.ptr(factor), out.data_ptr(), n, h * w, c_src, c_dst, _hip.stream(y.device))
    _hip.check(rc, 'nhwc_f32_to_nchw')
    return out


def wgrad_splits(n, hx, wx, hd, wd, ci, co):
    return int(_hip.lib().lvg_conv2d_frames_wgrad_splits(n, hx, wx, hd, wd, ci, co, 3, 3))


def conv2d_wgrad(x, dy, x_channels=None, dy_channels=None, alg_flops=None):
    """gw[dh, dw, co, ci] = sum_{n, a, b} dy[n, a, b, co] * x[n, a + dh, b + dw, ci]  (float32 [3, 3, Co, Ci]).

    x [N, Hx, Wx, Ci], dy [N, Hd, Wd, Co] contiguous with Hd % 4 == 0, Wd % 16 == 0, Hx >= Hd + 2, Wx >= Wd + 2.
    `x_channels` / `dy_channels`: use only the first so many channels of a pixel (the tensors' channel counts are the pixel strides)."""
    n, hx, wx, xs = x.shape
    n2, hd, wd, ds = dy.shape
    ci, co = (xs if x_channels is None else x_channels), (ds if dy_channels is None else dy_channels)
    assert n == n2 and hd % PATCH_H == 0 and wd % PATCH_W == 0 and hx >= hd + 2 and wx >= wd + 2 and ci <= xs and co <= ds
    if x.device.type == 'cuda' and _init():
        assert x.dtype in (torch.float16, torch.bfloat16) and dy.dtype == x.dtype and x.is_contiguous() and dy.is_contiguous() and ci % CH == 0 and co % CH == 0, \
            'conv2d_frames_wgrad: no hand-written kernel for this shape / dtype / layout'
        splits = wgrad_splits(n, hx, wx, hd, wd, ci, co)
        assert splits > 0, 'conv2d_frames_wgrad: no hand-written kernel for this shape'
        part = torch.empty([splits, 3, 3, co, ci], dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _hip.lib().lvg_conv2d_frames_wgrad(x.data_ptr(), dy.data_ptr(), part.data_ptr(), n, hx, wx, hd, wd, ci, co, 3, 3,
                                                    xs, ds, splits, _hip.dtype_code(x.dtype), _hip.stream(x.device))
        _hip.check(rc, 'conv2d_frames_wgrad')
        stats['flops'] += 2 * n * hd * wd * co * ci * 9 if alg_flops is None else alg_flops
        stats['launches'] += 1
        return part.sum(0) if splits > 1 else part[0]                 # fixed summation order: reproducible
    xv = x[:, :hd + 2, :wd + 2, :ci].permute(0, 3, 1, 2).float()
    w0 = torch.zeros(co, ci, 3, 3, dtype=torch.float32, device=x.device, requires_grad=True)
    with torch.enable_grad():
        y = F.conv2d(xv, w0)
    gw = torch.autograd.grad(y, w0, dy[..., :co].permute(0, 3, 1, 2).float())[0]
    return gw.permute(2, 3, 0, 1).contiguous()
