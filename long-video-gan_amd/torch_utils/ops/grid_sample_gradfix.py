"""`torch_utils.ops.grid_sample_gradfix` (reference torch_utils/ops/grid_sample_gradfix.py):
bilinear, zero-padded, align_corners=False grid_sample whose double backward w.r.t. the input
is supported (ADA's geometric augmentations under R1). Used by the augmentation pipe only
(SURVEY.md 8f "next"); dispatches to ATen's grid_sampler kernels."""

import torch

# pylint: disable=redefined-builtin,arguments-differ,protected-access

enabled = False  # the train scripts set this to True (train_lres.py:80)

def grid_sample(input, grid):
    if enabled:
        return _GridSample2dForward.apply(input, grid)
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)

class _GridSample2dForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid):
        assert input.ndim == 4 and grid.ndim == 4
        ctx.save_for_backward(input, grid)
        return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        return _GridSample2dBackward.apply(grad_output, input, grid)

class _GridSample2dBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, grid):
        mask = (ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        grad_input, grad_grid = torch.ops.aten.grid_sampler_2d_backward(grad_output, input, grid, 0, 0, False, mask)
        ctx.save_for_backward(grid)
        return grad_input, grad_grid

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        grid, = ctx.saved_tensors
        assert not ctx.needs_input_grad[2]
        grad2_grad_output = _GridSample2dForward.apply(grad2_grad_input, grid) if ctx.needs_input_grad[0] else None
        return grad2_grad_output, None, None
