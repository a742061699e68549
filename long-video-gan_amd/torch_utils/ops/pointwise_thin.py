"""1 x 1 convolutions with a thin side (<= 4 channels) on channels-last 16-bit frames (csrc/pointwise_thin.hip): the generator's ToRGB
(reference model/generator_lres.py:600-640) and the discriminator's first layer (model/discriminator_lres.py:169 with kernel size 1), all
three passes as HBM streams over the wide tensor. `pointwise_thin(..., twice=True)` builds the same three passes as nodes that are closed
under differentiation (R1, reference model/video_gan_lres.py:180-197, differentiates the discriminator's input gradient again)."""

import torch

from . import _hip

_WIDE = (8, 16, 32, 64, 128)


def supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """x [F, Ci, H, W] channels-last float16 / bfloat16 on the GPU, weight [Co, Ci] of the same dtype, one side 1 .. 4 channels, the other 8 .. 128."""
    if not (x.is_cuda and x.dim() == 4 and weight.dim() == 2 and x.dtype in (torch.float16, torch.bfloat16) and weight.dtype == x.dtype):
        return False
    co, ci = weight.shape
    if x.shape[1] != ci or not ((ci in _WIDE and 1 <= co <= 4) or (co in _WIDE and 1 <= ci <= 4)):
        return False
    f, _, h, w = x.shape
    dense_cl = x.stride() == (h * w * ci, 1, w * ci, ci)
    return dense_cl and x.data_ptr() % 16 == 0 and f * h * w > 0


def _pixels(t: torch.Tensor) -> int:
    return t.shape[0] * t.shape[2] * t.shape[3]


def _dense_cl(t: torch.Tensor) -> torch.Tensor:
    f, c, h, w = t.shape
    if not t.is_cuda or (t.stride() == (h * w * c, 1, w * c, c) and t.data_ptr() % 16 == 0):
        return t
    out = torch.empty_strided((f, c, h, w), (h * w * c, 1, w * c, c), dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


def _apply(x: torch.Tensor, w_tw: torch.Tensor, wide_in: bool, co: int) -> torch.Tensor:
    """One launch: w_tw float32 [thin, wide]; wide_in: x carries the wide side (thin_out kernel), else the thin side (thin_in kernel)."""
    f, c, h, w = x.shape
    if not x.is_cuda:
        # CPU tensors: the definition (the nodes' algebra is tested there; GPU tensors never come here)
        wm = w_tw if wide_in else w_tw.t()                                       # [co, ci]
        return torch.matmul(x.permute(0, 2, 3, 1).float(), wm.t()).to(x.dtype).permute(0, 3, 1, 2)
    y = torch.empty_strided((f, co, h, w), (h * w * co, 1, w * co, co), dtype=x.dtype, device=x.device)
    thin, wide = w_tw.shape
    fn = _hip.lib().lvg_pointwise_thin_out if wide_in else _hip.lib().lvg_pointwise_thin_in
    with torch.cuda.device(x.device):                                        # (the launch goes to the CURRENT HIP device: make it the tensor's)
        _hip.check(fn(x.data_ptr(), w_tw.data_ptr(), y.data_ptr(), _pixels(x), wide, thin, _hip.dtype_code(x.dtype), _hip.stream(x.device)),
                   'lvg_pointwise_thin_out' if wide_in else 'lvg_pointwise_thin_in')
    return y


def _wgrad(wide_t: torch.Tensor, thin_t: torch.Tensor) -> torch.Tensor:
    """sum over pixels thin[m][t] * wide[m][c] -> float32 [thin, wide]."""
    wide, thin = wide_t.shape[1], thin_t.shape[1]
    if not wide_t.is_cuda:
        return torch.matmul(thin_t.permute(1, 0, 2, 3).reshape(thin, -1).float(), wide_t.permute(0, 2, 3, 1).reshape(-1, wide).float())
    pixels = _pixels(wide_t)
    blocks = int(_hip.lib().lvg_pointwise_thin_wgrad_blocks(pixels, wide))
    part = torch.empty((blocks, thin, wide), dtype=torch.float32, device=wide_t.device)
    with torch.cuda.device(wide_t.device):
        _hip.check(_hip.lib().lvg_pointwise_thin_wgrad(wide_t.data_ptr(), thin_t.data_ptr(), part.data_ptr(), pixels, wide, thin, _hip.dtype_code(wide_t.dtype),
                                                       blocks, _hip.stream(wide_t.device)), 'lvg_pointwise_thin_wgrad')
    return part.sum(dim=0)


class _PointwiseThin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        co, ci = weight.shape
        ctx.wide_in = ci > co
        w32 = weight.float()
        y = _apply(x, (w32 if ctx.wide_in else w32.t()).contiguous(), ctx.wide_in, co)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        co, ci = weight.shape
        gy = _dense_cl(gy)
        gx = gw = None
        w32 = weight.float()
        if ctx.needs_input_grad[0]:
            # dx[m][ci] = sum_co gy[m][co] w[co][ci]: the opposite kernel, the weight again as [thin, wide]
            gx = _apply(gy, (w32 if ctx.wide_in else w32.t()).contiguous(), not ctx.wide_in, ci)
        if ctx.needs_input_grad[1]:
            if ctx.wide_in:
                gw = _wgrad(x, gy)                     # [thin = co, wide = ci]
            else:
                gw = _wgrad(gy, x).t()                 # [thin = ci, wide = co] -> [co, ci]
            gw = gw.to(weight.dtype)
        return gx, gw


# The same passes for graphs that are differentiated again. A 1 x 1 convolution is y = x W^T on the pixel matrix: its data gradient is the
# 1 x 1 convolution of the incoming gradient with W^T as the weight, its weight gradient G(x, g) = g^T x is bilinear with
#     dG/dx applied to ggw = g ggw  = the convolution of g with ggw^T,       dG/dg applied to ggw = x ggw^T = the convolution of x with ggw,
# so two node types -- the convolution (thin_out / thin_in kernel by the direction) and the weight gradient -- reproduce each other.

class _ThinConv(torch.autograd.Function):
    """y[m][co] = sum_ci x[m][ci] w[co][ci]; weight [Co, Ci] in x's dtype."""

    @staticmethod
    def forward(ctx, x, weight):
        co, ci = weight.shape
        wide_in = ci > co
        w32 = weight.float()
        ctx.save_for_backward(x, weight)
        return _apply(_dense_cl(x), (w32 if wide_in else w32.t()).contiguous(), wide_in, co)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = _ThinConv.apply(gy, weight.t()) if ctx.needs_input_grad[0] else None
        gw = _ThinWgrad.apply(x, gy) if ctx.needs_input_grad[1] else None
        return gx, gw


class _ThinWgrad(torch.autograd.Function):
    """G[co][ci] = sum_m g[m][co] x[m][ci], in x's dtype (float32 accumulation, one rounding)."""

    @staticmethod
    def forward(ctx, x, g):
        ctx.save_for_backward(x, g)
        xd, gd = _dense_cl(x), _dense_cl(g)
        if x.shape[1] > g.shape[1]:
            gw = _wgrad(xd, gd)                    # [thin = co, wide = ci]
        else:
            gw = _wgrad(gd, xd).t()                # [thin = ci, wide = co] -> [co, ci]
        return gw.to(x.dtype)

    @staticmethod
    def backward(ctx, ggw):
        x, g = ctx.saved_tensors
        ggw = ggw.to(x.dtype)
        dx = _ThinConv.apply(g, ggw.t()) if ctx.needs_input_grad[0] else None
        dg = _ThinConv.apply(x, ggw) if ctx.needs_input_grad[1] else None
        return dx, dg


def pointwise_thin(x: torch.Tensor, weight: torch.Tensor, twice: bool = False) -> torch.Tensor:
    """conv2d(x, weight[:, :, None, None]) for `supported(x, weight)` tensors; result channels-last in x's dtype. twice: nodes whose
    gradients can be differentiated again."""
    assert supported(x, weight)
    return _ThinConv.apply(x, weight) if twice else _PointwiseThin.apply(x, weight)
