"""a * b + c with hand-written, broadcast-aware gradients (`torch_utils.ops.fma`, reference
torch_utils/ops/fma.py:15-58). Unused by the models; API parity only.
addcmul is already one fused elementwise kernel on ROCm, so there is no HIP kernel for it."""

import torch

#----------------------------------------------------------------------------

def fma(a, b, c): # => a * b + c
    return _FusedMultiplyAdd.apply(a, b, c)

#----------------------------------------------------------------------------

class _FusedMultiplyAdd(torch.autograd.Function): # a * b + c
    @staticmethod
    def forward(ctx, a, b, c): # pylint: disable=arguments-differ
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, dout): # pylint: disable=arguments-differ
        a, b = ctx.saved_tensors
        da = _unbroadcast(dout * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _unbroadcast(dout * a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _unbroadcast(dout, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc

#----------------------------------------------------------------------------

def _unbroadcast(x, shape):
    """Sum `x` back down to `shape` (inverse of broadcasting)."""
    lead = x.ndim - len(shape)
    assert lead >= 0
    reduce_dims = [i for i in range(x.ndim) if x.shape[i] > 1 and (i < lead or shape[i - lead] == 1)]
    if reduce_dims:
        x = x.sum(dim=reduce_dims, keepdim=True)
    if lead:
        x = x.reshape(-1, *x.shape[lead + 1:])
    assert x.shape == shape
    return x

#----------------------------------------------------------------------------
