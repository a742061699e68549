"""`torch_utils.ops.conv2d_gradfix` (reference torch_utils/ops/conv2d_gradfix.py).

In the reference this wraps cuDNN in a custom autograd.Function to make high-order gradients
cheaper, but the wrapper is a no-op there for every supported configuration: `enabled` is never
switched on by the train scripts and `_should_use_custom_op` returns False on torch >= 1.11
(:53-55). The dense contraction there always is `torch.nn.functional.conv2d`. Here that is the
default as well (on ROCm: MIOpen with native arbitrary-order autograd); with `enabled` (or
inside `closed_nodes()`, which the super-resolution trainer's R1 pass uses) a dense, ungrouped,
bias-free conv2d is built from three nodes that are closed under differentiation, so that every
pass of any order is one ordinary forward / backward-data / backward-weight call of the library
at the layer's own shapes."""

import contextlib

import torch

# pylint: disable=redefined-builtin

enabled = False                     # True: closed three-node graph for dense conv2d (see module docstring and closed_nodes())
weight_gradients_disabled = False   # honoured by the closed nodes (reference conv2d_gradfix.py:27-35)

@contextlib.contextmanager
def no_weight_gradients(disable=True):
    global weight_gradients_disabled
    old = weight_gradients_disabled
    if disable:
        weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old

@contextlib.contextmanager
def closed_nodes(on=True):
    """Scope in which dense conv2d calls build the three-node graph below (SuperResTrainer.update_r1)."""
    global enabled
    old, enabled = enabled, bool(on)
    try:
        yield
    finally:
        enabled = old


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(e) for e in v)


# A dense convolution as nodes that reproduce each other under differentiation (round 6; the library's own graph is differentiable twice,
# but its second-order node computes the weight term as a convolution with batch and channel roles exchanged -- 2 "channels", a 256 x 256
# "kernel" on the super-resolution discriminator: 3 ms per layer, 12.9 of the 20 ms of device time of an R1 micro-batch,
# profiles/r06_launch_sites_r1_sres.log). With  C(x, w) = conv,  D(g, w) = data gradient,  W(x, g) = weight gradient:
#     C' : (dx, dw) = (D(gy, w), W(x, gy))      D' : (dg, dw) = (C(ggx, w), W(ggx, g))      W' : (dx, dg) = (D(g, ggw), C(x, ggw))
# every right-hand side is a forward / backward-data / backward-weight call of the library at the layer's own shapes.

class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding)
        return torch.nn.functional.conv2d(x, w, None, stride, padding)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding = ctx.cfg
        gx = _ConvDgrad.apply(gy, w, tuple(x.shape), stride, padding) if ctx.needs_input_grad[0] else None
        gw = _ConvWgrad.apply(x, gy, tuple(w.shape), stride, padding) if ctx.needs_input_grad[1] and not weight_gradients_disabled else None
        return gx, gw, None, None


class _ConvDgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, w, x_shape, stride, padding):
        ctx.save_for_backward(g, w)
        ctx.cfg = (stride, padding)
        return torch.nn.grad.conv2d_input(x_shape, w, g, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, ggx):
        g, w = ctx.saved_tensors
        stride, padding = ctx.cfg
        dg = _Conv.apply(ggx, w, stride, padding) if ctx.needs_input_grad[0] else None
        dw = _ConvWgrad.apply(ggx, g, tuple(w.shape), stride, padding) if ctx.needs_input_grad[1] else None
        return dg, dw, None, None, None


class _ConvWgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, w_shape, stride, padding):
        ctx.save_for_backward(x, g)
        ctx.cfg = (stride, padding)
        return torch.nn.grad.conv2d_weight(x, w_shape, g, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, ggw):
        x, g = ctx.saved_tensors
        stride, padding = ctx.cfg
        ggw = ggw.to(x.dtype)
        dx = _ConvDgrad.apply(g, ggw, tuple(x.shape), stride, padding) if ctx.needs_input_grad[0] else None
        dg = _Conv.apply(x, ggw, stride, padding) if ctx.needs_input_grad[1] else None
        return dx, dg, None, None, None


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if enabled and bias is None and groups == 1 and _pair(dilation) == (1, 1) and input.dim() == 4 and weight.dtype == input.dtype:
        return _Conv.apply(input, weight, _pair(stride), _pair(padding))
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)

def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, output_padding=output_padding, groups=groups, dilation=dilation)
