"""Style side of the modulated convolution (csrc/style_prep.hip, torch_utils/ops/style_prep.py): per-sample max normalisation of the
styles and the demodulation term of the reference (model/generator_lres.py:99, :107-108) against the oracle (orc_style_prep, pinned to
the reference's temporal_modulated_conv3d in tests/test_conv3d_frames.py) and, for the gradients, the float64 autograd of the tensor
expressions."""

import numpy as np
import pytest
import torch

from torch_utils.ops import style_prep as sp


def _inputs(seed, t, n, ci, co, ties=False):
    g = torch.Generator().manual_seed(seed)
    style = 1.0 + 0.5 * torch.randn(t, n, ci, generator=g)
    if ties:                                       # two equal maxima (one negative) in every sample
        style[0, :, 1] = 9.0
        style[t - 1, :, ci - 2] = -9.0
    w2 = torch.rand(co, ci, generator=g) / ci
    return style, w2


def test_definition_matches_oracle_cpu(oracle):
    style, w2 = _inputs(0, 5, 3, 8, 12)
    mod, demod = sp.style_prep(style, w2)                       # CPU tensors: the tensor expressions
    omod, odemod = oracle.style_prep(style.numpy(), w2.numpy())
    np.testing.assert_allclose(mod.numpy(), omod, rtol=1e-6)
    np.testing.assert_allclose(demod.numpy(), odemod, rtol=1e-5)


# (T, N, Ci, Co): generator layers at BASELINE.json configs[1] (8 clips) and ragged sizes (tiles of 64 rows / 64 columns / 16 in K)
SHAPES = [(24, 8, 512, 512), (144, 8, 256, 256), (128, 8, 128, 64), (128, 8, 64, 64), (5, 3, 36, 20), (1, 1, 4, 4), (7, 2, 68, 132)]


@pytest.mark.gpu
@pytest.mark.parametrize('shape', SHAPES)
def test_hip_forward_matches_oracle_gpu(oracle, shape):
    t, n, ci, co = shape
    style, w2 = _inputs(1, t, n, ci, co)
    assert sp.supported(style.cuda(), w2.cuda())
    mod, demod = sp.style_prep(style.cuda(), w2.cuda())
    omod, odemod = oracle.style_prep(style.numpy(), w2.numpy())
    assert mod.shape == (t * n, ci) and demod.shape == (t * n, co)
    np.testing.assert_allclose(mod.cpu().numpy(), omod, rtol=2e-7, atol=0)              # one IEEE division
    np.testing.assert_allclose(demod.cpu().numpy(), odemod, rtol=3e-6, atol=0)          # float32 sums of <= 512 positive terms + rsqrt
    # bit-identical to the tensor expression for the modulation
    assert torch.equal(mod, sp._ref(style.cuda(), w2.cuda())[0])


@pytest.mark.gpu
@pytest.mark.parametrize('shape,ties', [((24, 8, 512, 512), False), ((144, 8, 256, 256), True), ((128, 8, 64, 64), False), ((5, 3, 36, 20), True),
                                         ((7, 2, 68, 132), False)])
def test_hip_backward_matches_float64_autograd_gpu(shape, ties):
    t, n, ci, co = shape
    style, w2 = _inputs(2, t, n, ci, co, ties)
    g = torch.Generator().manual_seed(3)
    g_mod, g_demod = torch.randn(t * n, ci, generator=g), torch.randn(t * n, co, generator=g)
    sa, wa = style.cuda().requires_grad_(True), w2.cuda().requires_grad_(True)
    mod, demod = sp.style_prep(sa, wa)
    ds, dw2 = torch.autograd.grad([mod, demod], [sa, wa], [g_mod.cuda(), g_demod.cuda()], retain_graph=True)
    sb, wb = style.double().requires_grad_(True), w2.double().requires_grad_(True)
    rmod, rdemod = sp._ref(sb, wb)
    rs, rw2 = torch.autograd.grad([rmod, rdemod], [sb, wb], [g_mod.double(), g_demod.double()])
    for got, ref in ((ds, rs), (dw2, rw2)):
        err = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, err
    # reproducible, and each output alone (the other gradient absent)
    again = torch.autograd.grad([mod, demod], [sa, wa], [g_mod.cuda(), g_demod.cuda()], retain_graph=True)
    assert torch.equal(again[0], ds) and torch.equal(again[1], dw2)
    only_demod, = torch.autograd.grad([demod], [sa], [g_demod.cuda()], retain_graph=True)
    ref_only, = torch.autograd.grad([sp._ref(sb, wb)[1]], [sb], [g_demod.double()])
    assert float((only_demod.cpu().double() - ref_only).abs().max() / ref_only.abs().max()) < 2e-5


@pytest.mark.gpu
def test_modulation_terms_route_gpu(monkeypatch):
    """`modulation_terms` with the fused style side against the tensor expressions (forward values and all gradients)."""
    from lvg.models import lres
    g = torch.Generator().manual_seed(4)
    weight = torch.randn(128, 64, 3, 3, 3, generator=g).cuda()
    style = (1.0 + 0.5 * torch.randn(16, 2, 64, generator=g)).cuda()
    outs = []
    for flag in (True, False):
        monkeypatch.setattr(lres, 'STYLE_PREP', flag)
        wq, sq = weight.clone().requires_grad_(True), style.clone().requires_grad_(True)
        w16, mod, demod = lres.modulation_terms(wq, sq, True, torch.bfloat16)
        loss = (mod * torch.linspace(-1, 1, mod.numel(), device='cuda').view_as(mod)).sum() + (demod * torch.linspace(1, 2, demod.numel(), device='cuda').view_as(demod)).sum() \
            + w16.float().square().sum()
        gw, gs = torch.autograd.grad(loss, [wq, sq])
        outs.append((w16, mod, demod, gw, gs))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for x, y, tol in ((a[2], b[2], 3e-6), (a[3], b[3], 3e-5), (a[4], b[4], 3e-5)):
        assert float((x - y).abs().max() / y.abs().max()) < tol
