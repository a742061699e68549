"""Fused stages of the ADA augmentation pipeline on the GPU (csrc/ada_augment.hip; reference model/ada_augment.py):

`ada_warp`   -- the geometric stage (:271-304) in ONE launch: reflect padding, x2 up-sampling with the 12-tap low-pass, the bilinear
                resampling through the inverse affine map and the x2 down-sampling, with the padding margins read from a device tensor
                (the reference reads them back to the host, :286). The stage is linear in the clip; its backward (the generator trains
                THROUGH the augmentation of its fakes) is the adjoint kernel lvg_ada_warp_adjoint, and the two are each other's
                backward to any order.
`ada_colour` -- colour matrix, additive noise and cutout (:376-381, :407-427) in one pass over the pixels, forward and backward.

CPU tensors and anything the kernels do not take use the compositions in lvg/ada_augment.py (the definition tested against)."""

import torch

from . import _hip
from .modconv_epilogue import _init


def warp_supported(x, taps):
    """(the size bound is the ADJOINT kernel's workspace, 4 (h + 6) (w + 6) floats per plane, indexed with 32 bits: a clip the forward
    kernel would take but whose backward could not run goes to the composed path as a whole)"""
    return (x.device.type == 'cuda' and x.dtype == torch.float32 and x.dim() == 4 and taps.dim() == 1 and taps.shape[0] == 12
            and x.shape[2] >= 2 and x.shape[3] >= 2 and x.numel() < 2 ** 31
            and x.shape[0] * x.shape[1] * (x.shape[2] + 6) * (x.shape[3] + 6) * 4 < 2 ** 31 and _init())


def _warp_launch(x, g, m, taps, transposed):
    x = x.contiguous()
    n, k, h, w = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        if transposed:
            work = torch.empty([n, k, (h + 6) * 2, (w + 6) * 2], dtype=torch.float32, device=x.device)     # gradient of the intermediate grid
            rc = _hip.lib().lvg_ada_warp_adjoint(x.data_ptr(), g.data_ptr(), m.data_ptr(), taps.data_ptr(), work.data_ptr(), y.data_ptr(),
                                                 n, k, h, w, _hip.stream(x.device))
        else:
            rc = _hip.lib().lvg_ada_warp(x.data_ptr(), g.data_ptr(), m.data_ptr(), taps.data_ptr(), y.data_ptr(), n, k, h, w, _hip.stream(x.device))
    _hip.check(rc, 'ada_warp_adjoint' if transposed else 'ada_warp')
    return y


class _AdaWarpLinear(torch.autograd.Function):
    """The stage is linear in the clip: y = A x (lvg_ada_warp) and d x = A^T d y (lvg_ada_warp_adjoint) are each the backward of the
    other, to any order (R1 differentiates the discriminator's input gradient a second time, through the augmentation of the reals)."""

    @staticmethod
    def forward(ctx, x, g, m, taps, transposed):
        ctx.save_for_backward(g, m, taps)
        ctx.transposed = transposed
        return _warp_launch(x, g, m, taps, transposed)

    @staticmethod
    def backward(ctx, gy):
        g, m, taps = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        return _AdaWarpLinear.apply(gy, g, m, taps, not ctx.transposed), None, None, None, None


def ada_warp(x, g_inv, margins, taps):
    """x [N, K, H, W] float32 on the GPU; g_inv [N, 3, 3] (pixel units, centred); margins int32 [4] = (mx0, my0, mx1, my1) on the device;
    taps [12] the normalised low-pass. No host read, forward or backward."""
    return _AdaWarpLinear.apply(x, g_inv.detach().float().contiguous(), margins.to(torch.int32).contiguous(), taps.detach().float().contiguous(), False)


def colour_supported(x):
    return x.device.type == 'cuda' and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] == 3 and x.numel() < 2 ** 31 and _init()


def _colour_launch(x, cm, nz, sg, ct, mode):
    x = x.contiguous()
    n, c, t, h, w = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _hip.lib().lvg_ada_colour(x.data_ptr(), _hip.ptr(cm), _hip.ptr(nz), _hip.ptr(sg), _hip.ptr(ct), y.data_ptr(), n, t, h, w, mode, _hip.stream(x.device))
    _hip.check(rc, 'ada_colour')
    return y


class _AdaColourLinear(torch.autograd.Function):
    """The linear part of the pass: mask(M x) or, transposed, M^T mask(x) -- each the backward of the other, to any order (R1
    differentiates the input gradient of the discriminator a second time, through the augmentation)."""

    @staticmethod
    def forward(ctx, x, cm, ct, transposed):
        ctx.save_for_backward(cm, ct)
        ctx.transposed = transposed
        return _colour_launch(x, cm, None, None, ct, 1 if transposed else 2)

    @staticmethod
    def backward(ctx, g):
        cm, ct = ctx.saved_tensors
        return _AdaColourLinear.apply(g, cm, ct, not ctx.transposed), None, None, None


class _AdaColour(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cmat, noise, sigma, cut):
        n = x.shape[0]
        cm = None if cmat is None else cmat.detach().float().contiguous()
        nz = None if noise is None else noise.float().contiguous()
        sg = None if sigma is None else sigma.float().reshape(n).contiguous()
        ct = None if cut is None else cut.float().reshape(n, 4).contiguous()
        ctx.save_for_backward(cm, ct)
        return _colour_launch(x, cm, nz, sg, ct, 0)

    @staticmethod
    def backward(ctx, dy):
        cm, ct = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        return _AdaColourLinear.apply(dy, cm, ct, True), None, None, None, None


def ada_colour(x, cmat=None, noise=None, sigma=None, cut=None):
    """x [N, 3, T, H, W] float32 on the GPU -> C[:3, :3] . x + C[:3, 3] (cmat [N, 4, 4] or None), + noise * sigma[n] (or None), zero inside
    the cutout rectangle (cut [N, 4] = cx, cy, sx, sy in image fractions, or None)."""
    return _AdaColour.apply(x, cmat, noise, sigma, cut)
