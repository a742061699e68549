"""Filtered leaky ReLU (`torch_utils.ops.filtered_lrelu`, reference
torch_utils/ops/filtered_lrelu.py:56): bias -> upsample FIR -> gain -> leaky ReLU -> clamp ->
downsample FIR as ONE kernel, `lvg_filtered_lrelu` (csrc/filtered_lrelu.hip). The forward
optionally emits a 2-bit-per-pixel sign/clamp mask; the backward is the same kernel with the
roles of the filters swapped, reading that mask (:239-268), so gradients of any order work.
Parameter combinations without a fused kernel take the generic path: upfirdn2d ->
`lvg_filtered_lrelu_act` (in place, same mask format) -> upfirdn2d (:223-229)."""

import collections
import os
import warnings

import numpy as np
import torch

from .. import custom_ops
from .. import misc
from . import upfirdn2d
from . import bias_act
from . import _hip

_Adjoint = collections.namedtuple('_Adjoint', 'padding gain mask_x mask_y')     # parameters of the backward launch (see _adjoint_call)

#----------------------------------------------------------------------------

_plugin = None

def _init():
    """Load liblvg_hip.so (the reference JIT-compiles three .cu files here, filtered_lrelu.py:22-32)."""
    global _plugin
    if _plugin is None:
        custom_ops.get_plugin(module_name='filtered_lrelu_plugin')
        _plugin = _hip.lib()
    return True

def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor)
    assert 1 <= f.ndim <= 2
    return f.shape[-1], f.shape[0] # width, height

def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple))
    assert all(isinstance(x, (int, np.integer)) for x in padding)
    padding = [int(x) for x in padding]
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1

#----------------------------------------------------------------------------

def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False, impl='cuda'):
    r"""Per channel plane of `x` [N, C, H, W]: add `b[c]`; upsample by `up` with FIR `fu`
    (zero insertion, `padding` in up-sampled pixels, signal gain up**2 restored); multiply by
    `gain`; leaky ReLU with negative `slope`; clamp to +-`clamp` if given; filter with `fd` and
    keep every `down`-th sample.

    `fu`/`fd`: float32 [taps] (separable), [h, w] or None (identity). `b`: [C] in x's dtype or
    None. `padding`: int, [x, y] or [x0, x1, y0, y1]. Differentiable to any order in x and b."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _filtered_lrelu_cuda(up=up, down=down, padding=padding, gain=gain, slope=slope, clamp=clamp, flip_filter=flip_filter).apply(x, fu, fd, b, None, 0, 0)
    return _filtered_lrelu_ref(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=padding, gain=gain, slope=slope, clamp=clamp, flip_filter=flip_filter)

#----------------------------------------------------------------------------

def _check_scalars(up, down, gain, slope, clamp):
    assert isinstance(up, int) and up >= 1
    assert isinstance(down, int) and down >= 1
    assert gain == float(gain) and gain > 0
    assert slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)

@misc.profiled_function
def _filtered_lrelu_ref(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False):
    """The op as its definition reads, one stock-PyTorch step per stage (CPU tensors and impl='ref'). The
    up-sampled intermediate is materialised, so this needs up**2 times the memory of the fused kernels, and a
    16-bit tensor is rounded after every stage instead of once."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    _check_scalars(up, down, gain, slope, clamp)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.dtype == x.dtype
        misc.assert_shape(b, [x.shape[1]])
    (fu_w, fu_h), (fd_w, fd_h) = _get_filter_size(fu), _get_filter_size(fd)
    px0, px1, py0, py1 = _parse_padding(padding)
    n, c, h, w = x.shape
    # sizes after "up-sample + pad + filter" (valid region) and after "filter + keep every down-th sample"
    mid_w, mid_h = w * up + px0 + px1 - (fu_w - 1), h * up + py0 + py1 - (fu_h - 1)
    want = [n, c, (mid_h - (fd_h - 1) + (down - 1)) // down, (mid_w - (fd_w - 1) + (down - 1)) // down]

    y = x if b is None else x + b.reshape(1, -1, 1, 1)
    y = upfirdn2d.upfirdn2d(x=y, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)     # zero-stuffing loses up**2 of the signal power
    y = torch.nn.functional.leaky_relu(y, slope)
    if gain != 1:
        y = y * gain
    if clamp is not None:
        y = y.clamp(-clamp, clamp)
    y = upfirdn2d.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip_filter)
    misc.assert_shape(y, want)
    assert y.dtype == x.dtype
    return y

#----------------------------------------------------------------------------
# HIP path.

def _taps_1d(f, scale_is_one):
    """(pointer-holder, taps) when `f` can be handed to the fused kernel as separable taps,
    else None. None -> identity (1 tap, NULL pointer)."""
    if f is None:
        return (None, 1)
    if f.ndim == 1:
        return (f.contiguous(), f.shape[0])
    return None  # dense 2-D filter: generic path

FORCE_GENERIC = False  # test hook: route every call through upfirdn2d -> act -> upfirdn2d

def _fused_supported(fu, fd, up, down, dtype):
    """True when `lvg_filtered_lrelu` has a fused kernel for this filter pair / factors / dtype."""
    tu, td = _taps_1d(fu, up == 1), _taps_1d(fd, down == 1)
    if tu is None or td is None or dtype not in (torch.float32, torch.float16, torch.bfloat16):
        return False
    return bool(_hip.lib().lvg_filtered_lrelu_supported(tu[1], td[1], up, down, _hip.dtype_code(dtype)))

def _fused(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filter, write_signs):
    """Try the fused kernel. Returns (y, so, return_code); return_code < 0 = no kernel."""
    tu, td = _taps_1d(fu, up == 1), _taps_1d(fd, down == 1)
    if FORCE_GENERIC or tu is None or td is None or x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        return None, None, -1
    lib = _hip.lib()
    (fu_t, fu_n), (fd_t, fd_n) = tu, td
    if not lib.lvg_filtered_lrelu_supported(fu_n, fd_n, up, down, _hip.dtype_code(x.dtype)):
        return None, None, -1
    b = b.contiguous()
    n, c, xh, xw = x.shape
    cw = xw * up + (px0 + px1) - (fu_n - 1)
    ch = xh * up + (py0 + py1) - (fu_n - 1)
    assert cw > fd_n - 1 and ch > fd_n - 1, 'upsampled buffer must be at least the size of downsampling filter'
    yw = (cw - (fd_n - 1) + (down - 1)) // down
    yh = (ch - (fd_n - 1) + (down - 1)) // down
    assert yw > 0 and yh > 0, 'output must be at least 1x1'
    y = torch.empty([n, c, yh, yw], dtype=x.dtype, device=x.device)
    so = None
    s = si
    mode = _hip.SIGNS_NONE
    sw_active = 0
    if write_signs:
        sw_active = yw * down - (down - 1) + (fd_n - 1)
        sh = yh * down - (down - 1) + (fd_n - 1)
        sw = (sw_active + 15) & ~15
        s = so = torch.empty([n, c, sh, sw >> 2], dtype=torch.uint8, device=x.device)
        mode = _hip.SIGNS_WRITE
    elif si is not None and si.numel():
        assert si.dtype == torch.uint8 and si.is_contiguous() and si.device == x.device and si.ndim == 4
        assert si.shape[0] == n and si.shape[1] == c, 'signs must have same batch & channels as x'
        sw_active = si.shape[3] << 2
        mode = _hip.SIGNS_READ
    sshape = _hip.pair(s.shape[3], s.shape[2]) if mode != _hip.SIGNS_NONE else _hip.pair(0, 0)
    with torch.cuda.device(x.device):
        rc = lib.lvg_filtered_lrelu(
            x.data_ptr(), y.data_ptr(), b.data_ptr(), s.data_ptr() if mode != _hip.SIGNS_NONE else None,
            None if fu_t is None else fu_t.data_ptr(), None if fd_t is None else fd_t.data_ptr(),
            _hip.shape4(x), _hip.stride4(x), _hip.shape4(y), _hip.stride4(y),
            fu_n, fd_n, up, down, px0, py0, sshape, sx, sy, sw_active,
            gain, slope, clamp, int(bool(flip_filter)), mode, _hip.dtype_code(x.dtype), _hip.stream(x.device))
    if rc == _hip.ERR_UNSUPPORTED:
        return None, None, -1
    _hip.check(rc, 'filtered_lrelu')
    return y, so, 0

def _act_inplace(y, si, sx, sy, gain, slope, clamp, write_signs):
    """`filtered_lrelu_act_` of the reference plugin: y = clamp(lrelu(y * gain)) in place,
    writing or reading the sign mask. Returns the written mask (or None)."""
    n, c, h, w = y.shape
    so = None
    s = si
    mode = _hip.SIGNS_NONE
    if write_signs:
        sw = (w + 15) & ~15
        s = so = torch.empty([n, c, h, sw >> 2], dtype=torch.uint8, device=y.device)
        mode = _hip.SIGNS_WRITE
    elif si is not None and si.numel():
        assert si.dtype == torch.uint8 and si.is_contiguous() and si.device == y.device and si.ndim == 4
        assert si.shape[0] == n and si.shape[1] == c, 'signs must have same batch & channels as x'
        mode = _hip.SIGNS_READ
    sshape = _hip.pair(s.shape[3], s.shape[2]) if mode != _hip.SIGNS_NONE else _hip.pair(0, 0)
    with torch.cuda.device(y.device):
        rc = _hip.lib().lvg_filtered_lrelu_act(
            y.data_ptr(), s.data_ptr() if mode != _hip.SIGNS_NONE else None,
            _hip.shape4(y), _hip.stride4(y), sshape, sx, sy, gain, slope, clamp, mode,
            _hip.dtype_code(y.dtype), _hip.stream(y.device))
    _hip.check(rc, 'filtered_lrelu_act_')
    return so

def _bias_grad(dx):
    """dx.sum([0, 2, 3]) (reference filtered_lrelu.py:254). Contiguous GPU tensors: one pass of plane sums (lvg_plane_sum, float32
    accumulation, fixed order) and the sum over the samples, instead of the generic strided tensor reduction."""
    # (the raw kernel call is invisible to autograd: when this backward pass is itself being recorded -- create_graph=True -- the
    # differentiable reduction below keeps the higher-order term)
    if dx.is_cuda and dx.ndim == 4 and dx.is_contiguous() and dx.dtype in (torch.float32, torch.float16, torch.bfloat16) and dx.numel() > 0 \
            and not (torch.is_grad_enabled() and dx.requires_grad) and os.environ.get('LVG_FLRELU_PLANE_SUM', '1') == '1':
        n, c, h, w = dx.shape
        part = torch.empty([n, c], dtype=torch.float32, device=dx.device)
        with torch.cuda.device(dx.device):
            rc = _hip.lib().lvg_plane_sum(dx.data_ptr(), part.data_ptr(), n * c, h * w, _hip.dtype_code(dx.dtype), _hip.stream(dx.device))
        _hip.check(rc, 'plane_sum')
        return part.sum(0).to(dx.dtype)
    return dx.sum([0, 2, 3])


_filtered_lrelu_cuda_cache = dict()

def _filtered_lrelu_cuda(up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False):
    """autograd.Function class for one parameter combination, cached."""
    _check_scalars(up, down, gain, slope, clamp)
    px0, px1, py0, py1 = _parse_padding(padding)
    gain, slope = float(gain), float(slope)
    clamp = float(clamp if clamp is not None else 'inf')
    key = (up, down, px0, px1, py0, py1, gain, slope, clamp, flip_filter)
    cached = _filtered_lrelu_cuda_cache.get(key)
    if cached is not None:
        return cached

    class FilteredLReluCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fu, fd, b, si, sx, sy): # pylint: disable=arguments-differ
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            assert fu is None or 1 <= fu.ndim <= 2
            assert fd is None or 1 <= fd.ndim <= 2
            for f in (fu, fd):
                assert f is None or (f.dtype == torch.float32 and f.device == x.device), 'fu and fd must be float32 on x\'s device'
            assert x.numel() > 0, 'x is empty'
            if b is None:
                b = torch.zeros([x.shape[1]], dtype=x.dtype, device=x.device)
            assert b.ndim == 1 and b.shape[0] == x.shape[1] and b.dtype == x.dtype, 'b must be a vector with the same number of channels and dtype as x'

            # The mask is only produced when a gradient can flow back through this call.
            have_si = si is not None and si.numel() > 0
            write_signs = (not have_si) and (x.requires_grad or b.requires_grad)

            strides = [x.stride(i) for i in range(x.ndim) if x.size(i) > 1]
            if any(a < b_ for a, b_ in zip(strides[:-1], strides[1:])):
                warnings.warn("low-performance memory layout detected in filtered_lrelu input", RuntimeWarning)

            y, so, return_code = _fused(x, fu, fd, b, si if have_si else None, up, down, px0, px1, py0, py1, sx, sy,
                                        gain, slope, clamp, flip_filter, write_signs)

            if return_code < 0:
                # Generic path: three launches and an up-sampled intermediate, but still only the
                # bit-packed mask is kept for the backward pass.
                warnings.warn("filtered_lrelu called with parameters that have no optimized HIP kernel, using generic fallback", RuntimeWarning)
                y = x.add(b.unsqueeze(-1).unsqueeze(-1))
                y = upfirdn2d.upfirdn2d(x=y, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up**2, flip_filter=flip_filter)
                if not y.is_contiguous():
                    y = y.contiguous()
                so = _act_inplace(y, si if have_si else None, sx, sy, gain, slope, clamp, write_signs)
                y = upfirdn2d.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip_filter)

            ctx.fu, ctx.fd = fu, fd
            ctx.save_for_backward(si if have_si else (so if so is not None else torch.empty([0])))
            ctx.x_shape = x.shape
            ctx.y_shape = y.shape
            ctx.s_ofs = sx, sy
            return y

        @staticmethod
        def backward(ctx, dy): # pylint: disable=arguments-differ
            # Only x and b are differentiable; filters, the mask and its offsets are constants of the call.
            want_dx, want_db = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
            assert not any(ctx.needs_input_grad[i] for i in (1, 2, 4, 5, 6))
            if not (want_dx or want_db):
                return (None,) * 7
            mask, = ctx.saved_tensors
            adj = _adjoint_call(ctx.x_shape, ctx.y_shape, ctx.fu, ctx.fd, ctx.s_ofs)
            # The adjoint of (pad, up-FIR, pointwise, down-FIR) is the same chain with the FIRs' roles exchanged and their taps mirrored;
            # the pointwise stage is replaced by its derivative, which the 2-bit mask encodes (clamp=None: nothing left to clamp).
            grad_op = _filtered_lrelu_cuda(up=down, down=up, padding=adj.padding, gain=adj.gain, slope=slope, clamp=None,
                                           flip_filter=not flip_filter)
            dx = grad_op.apply(dy, ctx.fd, ctx.fu, None, mask, adj.mask_x, adj.mask_y)
            db = _bias_grad(dx) if want_db else None        # (the bias enters before the up-FIR: its gradient is dx summed per channel)
            return dx, None, None, db, None, None, None

    def _adjoint_call(x_shape, y_shape, fu, fd, mask_ofs):
        """Geometry of the backward launch. Forward: U = upfirdn(x, fu, up, pad) has  in * up + pad0 + pad1 - (taps_u - 1)  samples per
        axis and y = U[::down] after the down-FIR. The adjoint feeds dy (out samples) through an up-by-`down` FIR of taps_d taps and a
        down-by-`up` FIR of taps_u taps and must return exactly `in` samples: solving  (out * down + q0 + q1 - (taps_d - 1) - (taps_u - 1)
        + (up - 1)) // up == in  with the leading pad fixed by causality (q0 = taps_u - 1 + taps_d - 1 - pad0) gives q1 below. The mask
        was written in U's coordinates; the adjoint's up-sampled plane starts (taps_u - 1 - pad0) samples earlier, hence the offsets."""
        taps_ux, taps_uy = _get_filter_size(fu)
        taps_dx, taps_dy = _get_filter_size(fd)
        if fu is not None and fu.ndim == 1:
            taps_uy = taps_ux                               # separable filter given as one vector
        if fd is not None and fd.ndim == 1:
            taps_dy = taps_dx
        in_h, in_w = x_shape[2], x_shape[3]
        out_h, out_w = y_shape[2], y_shape[3]

        def axis(n_in, n_out, taps_u, taps_d, lead):
            q0 = (taps_u - 1) + (taps_d - 1) - lead
            q1 = n_in * up - n_out * down + lead - (up - 1)
            return q0, q1

        qx0, qx1 = axis(in_w, out_w, taps_ux, taps_dx, px0)
        qy0, qy1 = axis(in_h, out_h, taps_uy, taps_dy, py0)
        return _Adjoint(padding=[qx0, qx1, qy0, qy1],
                        gain=gain * (up * up) / (down * down),   # d/dU of (gain * U) folded with the two FIR gains: up**2 forward, down**2 here
                        mask_x=mask_ofs[0] - (taps_ux - 1) + px0,
                        mask_y=mask_ofs[1] - (taps_uy - 1) + py0)

    _filtered_lrelu_cuda_cache[key] = FilteredLReluCuda
    return FilteredLReluCuda

#----------------------------------------------------------------------------
