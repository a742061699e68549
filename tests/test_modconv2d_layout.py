"""2-D modulated-convolution prologue / epilogue (csrc/modconv2d_layout.hip) and the channels-last
`modulated_conv2d` built on them: HIP vs the C oracle (forward), vs float64 autograd of the plain
definition (backward), and vs the reference's own `modulated_conv2d` golden (tests/golden/sres_models.npz
covers the networks; here the op alone)."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from torch_utils.ops import modconv2d_layout as ml

TOL = {torch.float16: dict(rtol=2e-3, atol=2e-3), torch.bfloat16: dict(rtol=1.6e-2, atol=1.6e-2)}


def test_plain_definition_matches_reference_algebra_cpu():
    """conv(cat(x, cond) * mod, w) * demod == the reference's per-sample-weight grouped convolution."""
    torch.manual_seed(0)
    n, c1, c2, co, h, w = 3, 5, 2, 4, 7, 6
    x, cond = torch.randn(n, c1, h, w, dtype=torch.float64), torch.randn(n, c2, h, w, dtype=torch.float64)
    weight = torch.randn(co, c1 + c2, 3, 3, dtype=torch.float64)
    style = torch.randn(n, c1 + c2, dtype=torch.float64)
    # reference algebra (generator_sres.py:43-62): w[n] = weight * style[n], demodulated per output channel, groups = n
    wn = weight[None] * style[:, None, :, None, None]
    d = (wn.square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt()
    wn = wn * d[:, :, None, None, None]
    xin = torch.cat((x, cond), dim=1)
    want = F.conv2d(xin.reshape(1, -1, h, w), wn.reshape(-1, c1 + c2, 3, 3), padding=2, groups=n).reshape(n, co, h + 2, w + 2)
    demod = (torch.matmul(style.square(), weight.square().sum(dim=(2, 3)).t()) + 1e-8).rsqrt()
    got = ml._ref(x, cond, weight, style, demod, padding=2)
    torch.testing.assert_close(got, want, rtol=1e-10, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 37, 27, 19, 23), (1, 512, 27, 12, 16), (3, 5, 0, 9, 7), (2, 64, 3, 8, 8)],
                         ids=['odd_hw', 'wide', 'no_cond', 'vec8'])
def test_prologue_epilogue_vs_oracle(shape, dtype, oracle):
    n, c1, c2, h, w = shape
    rs = np.random.RandomState(4)
    x = torch.tensor(rs.randn(n, c1, h, w), dtype=dtype, device='cuda')
    cond = torch.tensor(rs.randn(n, c2, h, w), dtype=dtype, device='cuda') if c2 else None
    mod = torch.tensor(0.5 + rs.rand(n, c1 + c2), dtype=torch.float32, device='cuda')
    c_pad = ml._pad_to(c1 + c2)
    out, _ = ml._nchw_to_nhwc(x, cond, mod, c_pad)
    assert out.shape == (n, c_pad, h, w) and out.is_contiguous(memory_format=torch.channels_last) and (c_pad == 1 or out.stride(1) == 1)
    if cond is None:
        want = oracle.modconv2d_prologue(None, x.double().cpu().numpy(), mod.cpu().numpy(), c_pad)
    else:
        want = oracle.modconv2d_prologue(x.double().cpu().numpy(), cond.double().cpu().numpy(), mod.cpu().numpy(), c_pad)
    np.testing.assert_allclose(out.double().cpu().numpy(), want, **TOL[dtype])
    assert float(out[:, c1 + c2:].abs().max()) == 0.0 if c_pad > c1 + c2 else True
    # epilogue: NHWC with padded channels -> NCHW, first c_out channels
    c_out = c1 + c2 - 1 if c1 + c2 > 1 else 1
    demod = torch.tensor(0.5 + rs.rand(n, c_out), dtype=torch.float32, device='cuda')
    back, _ = ml._nhwc_to_nchw(out, demod, c_out)
    want2 = oracle.modconv2d_epilogue(out.double().cpu().numpy(), demod.cpu().numpy(), c_out)
    assert back.is_contiguous()
    np.testing.assert_allclose(back.double().cpu().numpy(), want2, **TOL[dtype])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_modulated_conv2d_forward_backward_vs_float64_autograd(dtype):
    """The whole op (prologue -> channels-last conv -> epilogue) and all its gradients vs the plain definition
    evaluated in float64 on the CPU from the same (16-bit rounded) inputs."""
    torch.manual_seed(1)
    n, c1, c2, co, h, w = 2, 21, 6, 13, 18, 22
    x = torch.randn(n, c1, h, w, device='cuda').to(dtype).requires_grad_(True)
    cond = torch.randn(n, c2, h, w, device='cuda').to(dtype)
    weight = (torch.randn(co, c1 + c2, 3, 3, device='cuda') / np.sqrt((c1 + c2) * 9)).requires_grad_(True)
    mod = (0.5 + torch.rand(n, c1 + c2, device='cuda')).requires_grad_(True)
    demod = (0.5 + torch.rand(n, co, device='cuda')).requires_grad_(True)
    y = ml.modulated_conv2d(x, cond, weight, mod, demod, padding=2)
    assert y.dtype == dtype and y.is_contiguous() and y.shape == (n, co, h + 2, w + 2)
    dy = torch.randn_like(y)
    gx, gw, gm, gd = torch.autograd.grad(y, [x, weight, mod, demod], dy)

    x64 = x.detach().double().cpu().requires_grad_(True)
    w64 = weight.detach().to(dtype).double().cpu().requires_grad_(True)        # the op casts the weight to the compute dtype
    m64, d64 = mod.detach().double().cpu().requires_grad_(True), demod.detach().double().cpu().requires_grad_(True)
    y64 = ml._ref(x64, cond.double().cpu(), w64, m64, d64, padding=2)
    g64 = torch.autograd.grad(y64, [x64, w64, m64, d64], dy.double().cpu())
    tol = TOL[dtype]
    scale = float(y64.abs().max())
    assert float((y.double().cpu() - y64).abs().max()) <= 4 * tol['atol'] * scale
    for got, want, name in zip((gx, gw, gm, gd), g64, ('x', 'weight', 'mod', 'demod')):
        err = float((got.double().cpu() - want).abs().max())
        assert err <= 6 * tol['atol'] * float(want.abs().max()), (name, err, float(want.abs().max()))


@pytest.mark.gpu
def test_reductions_are_reproducible():
    """No atomics: two runs of the backward give bit-identical d mod / d demod."""
    torch.manual_seed(2)
    x = torch.randn(2, 40, 33, 29, device='cuda', dtype=torch.float16, requires_grad=True)
    cond = torch.randn(2, 3, 33, 29, device='cuda', dtype=torch.float16)
    weight = torch.randn(16, 43, 3, 3, device='cuda') * 0.05
    mod = torch.rand(2, 43, device='cuda', requires_grad=True)
    demod = torch.rand(2, 16, device='cuda', requires_grad=True)
    outs = []
    for _ in range(2):
        y = ml.modulated_conv2d(x, cond, weight, mod, demod, padding=2)
        outs.append(torch.autograd.grad(y.float().square().sum(), [mod, demod]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
