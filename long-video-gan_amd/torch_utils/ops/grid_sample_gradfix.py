"""`torch_utils.ops.grid_sample_gradfix` (module and switch named as in the reference's torch_utils/ops/grid_sample_gradfix.py, which its
train scripts toggle: train_lres.py:80): bilinear, zero-padded, align_corners=False sampling of an NCHW tensor on a grid, differentiable
TWICE with respect to the sampled tensor -- what R1 on an augmented real needs, and what `F.grid_sample` alone does not give because
ATen registers no derivative for its backward kernel.

Not on this repository's hot path: the ADA pipe warps through `lvg_ada_warp` (csrc/ada_augment.hip), whose own adjoint closes the second
order. This file keeps the reference's API for code that still imports it. Derivation used below: for a fixed grid the sampler is LINEAR in
the sampled tensor, y = S(grid) x. Hence dL/dx = S^T g, and the derivative of THAT with respect to g is S again -- the second-order node
is one more forward sampling of the incoming cotangent."""

import torch
import torch.nn.functional as F

# pylint: disable=redefined-builtin,arguments-differ

enabled = False     # off: plain F.grid_sample (first order only); the train scripts switch it on


def _sample(image: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    return F.grid_sample(input=image, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def grid_sample(input: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    return _LinearSampler.apply(input, grid) if enabled else _sample(input, grid)


class _LinearSampler(torch.autograd.Function):
    """y = S(grid) x with a backward pass that is itself a differentiable node."""

    @staticmethod
    def forward(ctx, image, grid):
        if image.ndim != 4 or grid.ndim != 4:
            raise ValueError('grid_sample: expects an NCHW tensor and an [N, H, W, 2] grid')
        ctx.save_for_backward(image, grid)
        return _sample(image, grid)

    @staticmethod
    def backward(ctx, cotangent):
        image, grid = ctx.saved_tensors
        d_image, d_grid = _SamplerAdjoint.apply(cotangent, image, grid)
        return d_image, d_grid


class _SamplerAdjoint(torch.autograd.Function):
    """(g, x, grid) -> (S^T g, dL/dgrid) through ATen's kernel; differentiable in g only (grid and x are constants of an R1 pass)."""

    @staticmethod
    def forward(ctx, cotangent, image, grid):
        ctx.save_for_backward(grid)
        wants = (ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        # interpolation 0 = bilinear, padding 0 = zeros, align_corners False
        return torch.ops.aten.grid_sampler_2d_backward(cotangent, image, grid, 0, 0, False, wants)

    @staticmethod
    def backward(ctx, dd_image, dd_grid):   # cotangents of (S^T g, dL/dgrid); the second one is not propagated
        grid, = ctx.saved_tensors
        if ctx.needs_input_grad[2]:
            raise NotImplementedError('grid_sample: second derivative with respect to the grid')
        through_g = _LinearSampler.apply(dd_image, grid) if ctx.needs_input_grad[0] else None      # d(S^T g)/dg applied to dd_image = S dd_image
        return through_g, None, None
