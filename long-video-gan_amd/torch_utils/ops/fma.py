"""`torch_utils.ops.fma`: out = a * b + c with a hand-written backward that sums gradients back
over broadcast dimensions (reference torch_utils/ops/fma.py:15-58). None of the networks calls it
 -- it exists for API parity. On ROCm `addcmul` is already a single fused
elementwise kernel, so there is no HIP kernel behind it."""

import torch


class _MulAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):  # pylint: disable=arguments-differ
        ctx.save_for_backward(a, b)
        ctx.shapes = (a.shape, b.shape, c.shape)
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, grad):  # pylint: disable=arguments-differ
        a, b = ctx.saved_tensors
        sa, sb, sc = ctx.shapes
        want = ctx.needs_input_grad
        return (_unbroadcast(grad * b, sa) if want[0] else None,
                _unbroadcast(grad * a, sb) if want[1] else None,
                _unbroadcast(grad, sc) if want[2] else None)


def fma(a, b, c):
    """a * b + c (operands broadcast against each other)."""
    return _MulAdd.apply(a, b, c)


def _unbroadcast(x, shape):
    """Reduce `x` to `shape`, undoing broadcasting: leading extra dimensions and dimensions where
    `shape` has size 1 are summed."""
    extra = x.ndim - len(shape)
    assert extra >= 0
    out = x.sum_to_size(shape) if tuple(x.shape) != tuple(shape) else x
    assert out.shape == shape
    return out
