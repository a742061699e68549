"""`torch_utils.ops.conv2d_gradfix` (reference torch_utils/ops/conv2d_gradfix.py).

In the reference this wraps cuDNN in a custom autograd.Function to make high-order gradients
cheaper, but the wrapper is a no-op there for every supported configuration: `enabled` is never
switched on by the train scripts and `_should_use_custom_op` returns False on torch >= 1.11
(:53-55). The dense contraction therefore always is `torch.nn.functional.conv2d`, which on ROCm
is MIOpen with native arbitrary-order autograd. Same here; the module-level switches are kept
so reference code that toggles them keeps running."""

import contextlib

import torch

# pylint: disable=redefined-builtin

enabled = False                     # kept for API parity; has no effect (see module docstring)
weight_gradients_disabled = False   # honoured by no_weight_gradients() bookkeeping only

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

def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)

def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, output_padding=output_padding, groups=groups, dilation=dilation)
