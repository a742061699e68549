"""Fused bias + activation + gain + clamp (`torch_utils.ops.bias_act`, reference
torch_utils/ops/bias_act.py:52). GPU tensors run the HIP kernel `lvg_bias_act`
(csrc/bias_act.hip) with first- and second-order gradients implemented by the same kernel
(grad=1 / grad=2); CPU tensors and impl='ref' run the plain-PyTorch definition."""

import numpy as np
import os

import torch

import dnnlib

from .. import custom_ops
from .. import misc
from . import _hip

#----------------------------------------------------------------------------
# Activation table: same keys/fields the reference exposes (bias_act.py:21-31); models read
# `def_gain` from it. `cuda_idx` doubles as the LVG_ACT_* id of the C ABI.

def _act(func, def_alpha, def_gain, cuda_idx, ref, has_2nd_grad):
    return dnnlib.EasyDict(func=func, def_alpha=def_alpha, def_gain=def_gain, cuda_idx=cuda_idx, ref=ref, has_2nd_grad=has_2nd_grad)

activation_funcs = {
    'linear':   _act(lambda x, **_: x,                                             0,   1,          1, '',  False),
    'relu':     _act(lambda x, **_: torch.nn.functional.relu(x),                   0,   np.sqrt(2), 2, 'y', False),
    'lrelu':    _act(lambda x, alpha, **_: torch.nn.functional.leaky_relu(x, alpha), 0.2, np.sqrt(2), 3, 'y', False),
    'tanh':     _act(lambda x, **_: torch.tanh(x),                                 0,   1,          4, 'y', True),
    'sigmoid':  _act(lambda x, **_: torch.sigmoid(x),                              0,   1,          5, 'y', True),
    'elu':      _act(lambda x, **_: torch.nn.functional.elu(x),                    0,   1,          6, 'y', True),
    'selu':     _act(lambda x, **_: torch.nn.functional.selu(x),                   0,   1,          7, 'y', True),
    'softplus': _act(lambda x, **_: torch.nn.functional.softplus(x),               0,   1,          8, 'y', True),
    'swish':    _act(lambda x, **_: torch.sigmoid(x) * x,                          0,   np.sqrt(2), 9, 'x', True),
}

#----------------------------------------------------------------------------

_plugin = None

def _init():
    """Load the kernel library (the reference JIT-compiles here, bias_act.py:38-48).
    Raises if liblvg_hip.so is unavailable -- GPU tensors never fall back to PyTorch."""
    global _plugin
    if _plugin is None:
        custom_ops.get_plugin(module_name='bias_act_plugin')
        _plugin = _hip.lib()
    return True

#----------------------------------------------------------------------------

def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    r"""y = clamp(act(x + b) * gain).

    Args:
        x:      activation tensor of any shape.
        b:      1-D bias with `b.shape[0] == x.shape[dim]` and x's dtype, or None.
        dim:    dimension of `x` the bias runs along (ignored without `b`).
        act:    key of `activation_funcs` ('linear', 'relu', 'lrelu', 'tanh', 'sigmoid',
                'elu', 'selu', 'softplus', 'swish').
        alpha:  activation shape parameter (None = per-activation default).
        gain:   output scale (None = per-activation default).
        clamp:  clamp output to [-clamp, clamp]; None disables.
        impl:   'cuda' (HIP kernel when x is on the GPU) or 'ref' (plain PyTorch).

    Differentiable to second order (R1 needs the double backward)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _bias_act_cuda(dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp).apply(x, b)
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)

#----------------------------------------------------------------------------

def _resolve(act, alpha, gain, clamp):
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    return (spec,
            float(spec.def_alpha if alpha is None else alpha),
            float(spec.def_gain if gain is None else gain),
            float(-1 if clamp is None else clamp))

@misc.profiled_function
def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """Definition of the op in stock PyTorch ops (CPU path and impl='ref')."""
    assert isinstance(x, torch.Tensor)
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim
        assert b.shape[0] == x.shape[dim]
        bshape = [1] * x.ndim
        bshape[dim] = -1
        x = x + b.reshape(bshape)
    x = spec.func(x, alpha=alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x

#----------------------------------------------------------------------------
# HIP path.

def _dense_format(t):
    """Memory format under which `t` is handed to the kernel (dense, bias stride well defined):
    channels-last stays channels-last, everything else becomes contiguous."""
    if t.ndim == 4 and t.shape[1] > 1 and t.stride(1) == 1:
        return torch.channels_last
    if t.ndim == 5 and t.shape[1] > 1 and t.stride(1) == 1:
        return torch.channels_last_3d
    return torch.contiguous_format

def _launch(x, b, xref, yref, dy, grad, dim, act_id, alpha, gain, clamp):
    """One lvg_bias_act launch on x's device and the current stream."""
    y = torch.empty_like(x)
    assert y.stride() == x.stride()
    n = x.numel()
    if n == 0:
        return y
    for other in (xref, yref, dy):
        if other is not None and other.numel():
            assert other.shape == x.shape and other.dtype == x.dtype and other.device == x.device
            # same dense layout; strides of size-1 dims are arbitrary (a channels-last tensor with N == 1 or
            # C == 1 is also "contiguous" and .contiguous(memory_format=...) returns it unchanged)
            assert all(so == sx for so, sx, n_ in zip(other.stride(), x.stride(), x.shape) if n_ > 1), 'xref/yref/dy must share the layout of x'
    has_b = b is not None and b.numel() > 0
    if has_b:
        assert b.ndim == 1 and b.dtype == x.dtype and b.device == x.device and b.is_contiguous()
        assert 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
    with torch.cuda.device(x.device):
        rc = _hip.lib().lvg_bias_act(
            x.data_ptr(), _hip.ptr(b) if has_b else None, _hip.ptr(xref), _hip.ptr(yref), _hip.ptr(dy), y.data_ptr(),
            n, b.shape[0] if has_b else 0, x.stride(dim) if has_b else 1,
            _hip.dtype_code(x.dtype), grad, act_id, alpha, gain, clamp, _hip.stream(x.device))
    _hip.check(rc, 'bias_act')
    return y

# Bias gradient of a channels-last tensor in the SAME pass as dx (lvg_bias_act_grad_bias): the tensor reduction that the
# reference runs on the stored dx (bias_act.py:183) re-read 300 MB per block-final layer of the generator (10 x 72 us per step).
# First-order backward passes only (under create_graph the gradient must stay a differentiable function of dy).
# LVG_BIAS_GRAD_FUSED=0 restores dx.sum().
FUSED_BIAS_GRAD = os.environ.get('LVG_BIAS_GRAD_FUSED', '1') == '1'


def _grad_bias_slots(dy, dim):
    if not (FUSED_BIAS_GRAD and dy.is_cuda and dy.ndim == 4 and dim == 1 and dy.shape[1] > 1 and not torch.is_grad_enabled()):
        return 0
    if not dy.is_contiguous(memory_format=torch.channels_last):
        return 0
    return int(_hip.lib().lvg_bias_act_grad_bias_slots(dy.numel(), dy.shape[1], _hip.dtype_code(dy.dtype)))


def _launch_grad_bias(dy, xref, yref, slots, act_id, alpha, gain, clamp):
    """dx and the bias gradient (summed over the per-workgroup partial sums, float32 -> dy.dtype) from one launch."""
    dx = torch.empty_like(dy)
    part = torch.empty((slots, dy.shape[1]), dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        rc = _hip.lib().lvg_bias_act_grad_bias(dy.data_ptr(), _hip.ptr(xref), _hip.ptr(yref), dx.data_ptr(), part.data_ptr(), dy.numel(),
                                               dy.shape[1], _hip.dtype_code(dy.dtype), act_id, alpha, gain, clamp, _hip.stream(dy.device))
    _hip.check(rc, 'bias_act_grad_bias')
    return dx, part.sum(0).to(dy.dtype)


_bias_act_cuda_cache = dict()

def _bias_act_cuda(dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """autograd.Function class for one (dim, act, alpha, gain, clamp) combination, cached."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    key = (dim, act, alpha, gain, clamp)
    cached = _bias_act_cuda_cache.get(key)
    if cached is not None:
        return cached

    is_identity = (act == 'linear' and gain == 1 and clamp < 0)
    keep_x = ('x' in spec.ref) or spec.has_2nd_grad
    # 'linear' saves y only when clamped: the clamp mask of the backward needs the forward output.
    # (The reference plugin path passes no yref for 'linear' and lets gradients through clamped
    # elements, unlike its own ref path, bias_act.py:23,151-154; this follows the ref path.)
    keep_y = ('y' in spec.ref) or (act == 'linear' and clamp >= 0)
    empty = torch.empty([0])

    def bias_grad(dx):
        """dx summed over every dim but `dim`. Channels-last tensors with a handful of channels (ToRGB: 3) are summed as
        a [rows, 64 * C] matrix first: the direct strided reduction of [1024, 3, 36, 64] bf16 took 793 us per step."""
        c = dx.shape[dim] if dx.ndim > dim else 0
        if dx.ndim == 4 and dim == 1 and 1 < c < 32 and dx.is_cuda and dx.is_contiguous(memory_format=torch.channels_last) \
                and (dx.numel() // c) % 64 == 0:
            rows = dx.permute(0, 2, 3, 1).reshape(-1, 64 * c)          # a view: channels-last memory order
            return rows.sum(0, dtype=torch.float32).reshape(64, c).sum(0).to(dx.dtype)
        if dx.ndim == 4 and dim == 1 and dx.is_cuda and dx.is_contiguous() and dx.shape[2] * dx.shape[3] >= 64 and dx.numel() > 0 \
                and dx.dtype in (torch.float32, torch.float16, torch.bfloat16) and not (torch.is_grad_enabled() and dx.requires_grad):
            # NCHW planes (the sres discriminator's layers): one pass of per-plane sums (lvg_plane_sum: float32 accumulation, fixed order) and
            # the sum over the samples, instead of the strided tensor reduction (16-23 us per call against ~6, profiles/r06_launch_sites_train_sres.log)
            n_, c_, h_, w_ = dx.shape
            part = torch.empty([n_, c_], dtype=torch.float32, device=dx.device)
            with torch.cuda.device(dx.device):
                rc = _hip.lib().lvg_plane_sum(dx.data_ptr(), part.data_ptr(), n_ * c_, h_ * w_, _hip.dtype_code(dx.dtype), _hip.stream(dx.device))
            _hip.check(rc, 'plane_sum')
            return part.sum(0).to(dx.dtype)
        return dx.sum([i for i in range(dx.ndim) if i != dim])

    class BiasActCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b): # pylint: disable=arguments-differ
            ctx.memory_format = _dense_format(x)
            x = x.contiguous(memory_format=ctx.memory_format)
            b = b.contiguous() if b is not None else empty
            y = x
            if not is_identity or b is not empty:
                y = _launch(x, b, None, None, None, 0, dim, spec.cuda_idx, alpha, gain, clamp)
            ctx.save_for_backward(x if keep_x else empty, b if keep_x else empty, y if keep_y else empty)
            return y

        @staticmethod
        def backward(ctx, dy): # pylint: disable=arguments-differ
            dy = dy.contiguous(memory_format=ctx.memory_format)
            x, b, y = ctx.saved_tensors
            dx = db = None
            if ctx.needs_input_grad[1] and not is_identity and act != 'swish':
                slots = _grad_bias_slots(dy, dim)
                if slots > 0:                                           # dx and db from one pass over dy
                    return _launch_grad_bias(dy, x if x.numel() else None, y if y.numel() else None, slots, spec.cuda_idx, alpha, gain, clamp)
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dx = dy if is_identity else BiasActCudaGrad.apply(dy, x, b, y)
            if ctx.needs_input_grad[1]:
                db = bias_grad(dx)
            return dx, db

    class BiasActCudaGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y): # pylint: disable=arguments-differ
            ctx.memory_format = _dense_format(dy)
            dx = _launch(dy, b, x, y, None, 1, dim, spec.cuda_idx, alpha, gain, clamp)
            ctx.save_for_backward(dy if spec.has_2nd_grad else empty, x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx): # pylint: disable=arguments-differ
            d_dx = d_dx.contiguous(memory_format=ctx.memory_format)
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActCudaGrad.apply(d_dx, x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = _launch(d_dx, b, x, y, dy, 2, dim, spec.cuda_idx, alpha, gain, clamp)
            if spec.has_2nd_grad and ctx.needs_input_grad[2]:
                d_b = bias_grad(d_x)
            return d_dy, d_x, d_b, None

    _bias_act_cuda_cache[key] = BiasActCuda
    return BiasActCuda

#----------------------------------------------------------------------------
