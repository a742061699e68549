"""Hand-written 2-D implicit-GEMM convolution of the super-resolution networks (csrc/conv2d_igemm.hip, csrc/conv2d_wgrad.hip,
torch_utils/ops/conv2d_frames.py) and the fused modulated convolution around it (modconv2d_layout._ModConv2dHand).

CPU: the oracle's conv2d restatement (+ the modulation algebra of model/generator_sres.py:28-67) against vectors produced by
the REFERENCE's `modulated_conv2d` incl. its gradients (tests/golden/make_golden_modconv2d.py); the frame geometry and the
plumbing of the fused autograd node (plain-PyTorch composition of the same steps) against the same vectors.
GPU: HIP vs oracle on seeded ragged cases (f16 / bf16; tiles overhanging the frame, several channel chunks, padded channel
counts, paddings 0 / 1 / 2), the fused node vs the reference golden, and at the BASELINE.json configs[3] sizes the
size-independent properties "a one-hot kernel is a shift" (bit-exact) and "the weight gradient is the adjoint of the forward"."""

import math

import numpy as np
import pytest
import torch

from conftest import load_golden, record_measured
from helpers.modconv2d_inputs import CASES, inputs
from torch_utils.ops import conv2d_frames as c2
from torch_utils.ops import modconv2d_layout as ml


def _modulation(weight, style, gain):
    """numpy restatement of generator_sres.py:43-62 -> (normalised weight, modulation [N,Ci], demodulation [N,Co])."""
    w = weight / np.sqrt((weight ** 2).mean(axis=(1, 2, 3), keepdims=True))
    s = style / np.sqrt((style ** 2).mean())
    demod = 1.0 / np.sqrt(np.einsum('oiyx,ni->no', w ** 2, s ** 2) + 1e-8)
    return w, s * gain, demod


@pytest.mark.parametrize('name', list(CASES))
def test_oracle_matches_reference_modulated_conv2d(oracle, name):
    """conv2d(x * mod, w') * demod with the oracle's conv2d == the reference's per-sample-weight grouped convolution; the
    oracle's data / weight gradient restatements == what autograd derives through the reference (for fixed mod / demod)."""
    g = load_golden('modconv2d')
    x, weight, style, gain, dy = [t.double().numpy() for t in inputs(name)]
    pad = CASES[name][6]
    w, mod, demod = _modulation(weight, style, float(gain))
    y = oracle.conv2d(x * mod[:, :, None, None], w, padding=pad) * demod[:, :, None, None]
    np.testing.assert_allclose(y, g[name + '_y'], rtol=2e-4, atol=2e-5)
    # d/dx with mod, demod as the reference computes them (they do not depend on x)
    gx = oracle.conv2d_dgrad(dy * demod[:, :, None, None], w, x.shape[2], x.shape[3], padding=pad) * mod[:, :, None, None]
    np.testing.assert_allclose(gx, g[name + '_gx'], rtol=2e-4, atol=2e-5)


def _torch_modulation(weight, style, gain):
    w = weight * weight.square().mean(dim=(1, 2, 3), keepdim=True).rsqrt()
    s = style * style.square().mean().rsqrt()
    demod = (torch.matmul(s.square(), w.square().sum(dim=(2, 3)).t()) + 1e-8).rsqrt()
    return w, s * gain, demod


@pytest.mark.parametrize('name', list(CASES))
def test_fused_node_plumbing_matches_reference_cpu(name):
    """The fused autograd node on CPU tensors (every step through its plain-PyTorch composition: padded frames, offsets,
    weight packing, gradient frames) reproduces the reference's output and all three gradients."""
    g = load_golden('modconv2d')
    x, weight, style, gain, dy = inputs(name)
    n, ci, co, h, w_, k, pad = CASES[name]
    c_first = ci // 2
    x, weight, style = x.requires_grad_(True), weight.requires_grad_(True), style.requires_grad_(True)
    first, second = x[:, :c_first], x[:, c_first:].detach()
    wn, mod, demod = _torch_modulation(weight, style, gain)
    y = ml._ModConv2dHand.apply(first, second, wn, mod, demod, pad)
    np.testing.assert_allclose(y.detach().numpy(), g[name + '_y'], rtol=2e-4, atol=2e-5)
    gx, gw, gs = torch.autograd.grad(y, [x, weight, style], dy)
    np.testing.assert_allclose(gx[:, :c_first].numpy(), g[name + '_gx'][:, :c_first], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gw.numpy(), g[name + '_gw'], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(gs.numpy(), g[name + '_gs'], rtol=5e-4, atol=5e-5)


def test_plain_definitions_match_oracle_cpu(oracle):
    gen = torch.Generator().manual_seed(3)
    geo = c2.Geometry(7, 21, 2)
    assert (geo.ho, geo.wo, geo.q, geo.hd, geo.wd, geo.hx, geo.wx) == (9, 23, 0, 12, 32, 14, 34)
    x = torch.randn(2, 7, 21, 5, generator=gen)
    weight = torch.randn(6, 5, 3, 3, generator=gen)
    xp = torch.zeros(2, geo.hx, geo.wx, 64)
    xp[:, 2:9, 2:23, :5] = x
    out = c2.conv2d_valid(xp, c2.pack_weight(weight, torch.float32, 64, 64), geo.ho, geo.wo, offset=(geo.q, geo.q))
    ref = oracle.conv2d(x.permute(0, 3, 1, 2).numpy(), weight.numpy(), padding=2)
    np.testing.assert_allclose(out[..., :6].permute(0, 3, 1, 2).numpy(), ref, rtol=1e-4, atol=1e-5)
    assert float(out[..., 6:].abs().max()) == 0.0
    dy = torch.randn(2, geo.ho, geo.wo, 6, generator=gen)
    dyp = torch.zeros(2, geo.hd, geo.wd, 64)
    dyp[:, :geo.ho, :geo.wo, :6] = dy
    gw = c2.conv2d_wgrad(xp, dyp)
    refg = oracle.conv2d_wgrad(x.permute(0, 3, 1, 2).numpy(), dy.permute(0, 3, 1, 2).numpy(), 3, 3, padding=2)
    np.testing.assert_allclose(gw[:, :, :6, :5].permute(2, 3, 0, 1).numpy(), refg, rtol=1e-4, atol=1e-4)
    dx = c2.conv2d_valid(dyp, c2.pack_weight_dgrad(weight, torch.float32, 64, 64), 7, 21)
    refd = oracle.conv2d_dgrad(dy.permute(0, 3, 1, 2).numpy(), weight.numpy(), 7, 21, padding=2)
    np.testing.assert_allclose(dx[..., :5].permute(0, 3, 1, 2).numpy(), refd, rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------------------------------
# GPU: kernels vs the oracle

GPU_CASES = [
    # n, ci, co, hi, wi, ho, wo, off            (ho x wo ragged against the 8 x 16 tile on purpose)
    (2, 64, 64, 12, 21, 10, 19, 0),             # one chunk, tiles overhanging right and bottom
    (1, 128, 128, 19, 40, 15, 35, 1),           # two chunks (band double buffering), offset 1, BN = 128
    (3, 192, 64, 10, 18, 8, 16, 0),             # exactly one tile per frame, three chunks
    (2, 64, 192, 34, 50, 32, 48, 0),            # four tile rows, Co = 3 x 64
    (1, 256, 128, 9, 70, 5, 66, 2),             # one ragged tile row, five tiles across, four chunks
]


def _np(t):
    return t.detach().double().cpu().numpy()


def _nchw(t):
    return _np(t).transpose(0, 3, 1, 2)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', GPU_CASES)
def test_hip_forward_matches_oracle_gpu(oracle, case, dtype):
    n, ci, co, hi, wi, ho, wo, off = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, hi, wi, ci, generator=g).to(dtype).cuda()
    wp = (torch.randn(3, 3, co, ci, generator=g) / math.sqrt(9 * ci)).to(dtype).cuda()
    pre = (0.5 + torch.rand(n, co, generator=g)).cuda()
    assert c2.supported(x, wp)
    out = c2.conv2d_valid(x, wp, ho, wo, offset=(off, off), pre=pre)
    ref = oracle.conv2d(_nchw(x)[:, :, off:off + ho + 2, off:off + wo + 2], _np(wp).transpose(2, 3, 0, 1)) * _np(pre)[:, :, None, None]
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11          # one output rounding; accumulation is float32
    np.testing.assert_allclose(_nchw(out), ref, rtol=1.5 * eps, atol=eps * 0.05)


WGRAD_CASES = [
    # n, ci, co, hd, wd      (input frames hd + 2 x wd + 2)
    (2, 64, 64, 8, 16),                          # 2 patches per frame
    (1, 128, 64, 12, 48),                        # 9 patches, two ci tiles
    (3, 64, 192, 4, 32),                         # patches straddling frames in the K range, three co tiles
    (2, 128, 128, 20, 64),                       # 40 patches: several K ranges
]


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', WGRAD_CASES)
def test_hip_wgrad_matches_oracle_gpu(oracle, case, dtype):
    n, ci, co, hd, wd = case
    g = torch.Generator().manual_seed(5)
    # operands with few mantissa bits: every product is exact in float32, the sum is compared tightly
    x = (torch.randint(-4, 5, (n, hd + 2, wd + 2, ci), generator=g).float() / 4).to(dtype).cuda()
    dy = (torch.randint(-4, 5, (n, hd, wd, co), generator=g).float() / 4).to(dtype).cuda()
    gw = c2.conv2d_wgrad(x, dy)
    gw2 = c2.conv2d_wgrad(x, dy)
    assert torch.equal(gw, gw2)                                          # fixed-order range sums: reproducible
    ref = oracle.conv2d_wgrad(_nchw(x), _nchw(dy), 3, 3, padding=0)      # [co, ci, 3, 3]
    np.testing.assert_allclose(_np(gw).transpose(2, 3, 0, 1), ref, rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_fused_node_matches_reference_gpu(name, dtype):
    """The modulated convolution as the generator runs it (prologue -> lvg_conv2d_frames -> epilogue, and the four backward
    launches) vs the reference's modulated_conv2d and its gradients: 16-bit operands, float32 accumulation."""
    g = load_golden('modconv2d')
    x, weight, style, gain, dy = inputs(name)
    n, ci, co, h, w_, k, pad = CASES[name]
    c_first = ci // 2
    first = x[:, :c_first].to(dtype).cuda().requires_grad_(True)
    second = x[:, c_first:].to(dtype).cuda()
    weight, style = weight.cuda().requires_grad_(True), style.cuda().requires_grad_(True)
    wn, mod, demod = _torch_modulation(weight, style, gain.cuda())
    y = ml.modulated_conv2d(first, second, wn, mod, demod, padding=pad)
    assert c2.stats['launches'] > 0
    tol = 4e-2 if dtype == torch.bfloat16 else 6e-3                      # values up to ~3; inputs, weights and outputs rounded to 16 bits
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), g[name + '_y'], rtol=0, atol=tol)
    gx, gw, gs = torch.autograd.grad(y, [first, weight, style], dy.to(dtype).cuda())
    scale = lambda a: float(np.abs(a).max())
    np.testing.assert_allclose(gx.float().cpu().numpy(), g[name + '_gx'][:, :c_first], rtol=0, atol=tol * scale(g[name + '_gx']))
    np.testing.assert_allclose(gw.cpu().numpy(), g[name + '_gw'], rtol=0, atol=tol * scale(g[name + '_gw']))
    np.testing.assert_allclose(gs.cpu().numpy(), g[name + '_gs'], rtol=0, atol=tol * scale(g[name + '_gs']))


@pytest.mark.gpu
def test_fused_node_vs_library_route_full_layer_gpu(monkeypatch):
    """One generator layer shape of BASELINE.json configs[3] (L8: 539 -> 512 channels, 148 x 92 planes, padding 2; 2 frames):
    the hand-written route against the library-convolution route of the same op (same 16-bit operands), output and gradients."""
    torch.manual_seed(0)
    n, ci, co, h, w = 2, 539, 512, 92, 148
    first = torch.randn(n, 512, h, w, device='cuda').half().requires_grad_(True)
    second = torch.randn(n, 27, h, w, device='cuda').half()
    weight = (torch.randn(co, ci, 3, 3, device='cuda') / math.sqrt(9 * ci)).requires_grad_(True)
    mod = (1 + 0.3 * torch.randn(n, ci, device='cuda')).requires_grad_(True)
    demod = (0.5 + torch.rand(n, co, device='cuda')).requires_grad_(True)
    dy = torch.randn(n, co, h + 2, w + 2, device='cuda').half()

    def run(flag):
        monkeypatch.setattr(ml, 'HAND_CONV', flag)
        y = ml.modulated_conv2d(first, second, weight, mod, demod, padding=2)
        return [y.detach().float()] + [t.float() for t in torch.autograd.grad(y, [first, weight, mod, demod], dy)]
    hand, lib = run(True), run(False)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))
    errs = {k: rel(a, b) for k, a, b in zip(('y', 'd_x', 'd_weight', 'd_mod', 'd_demod'), hand, lib)}
    record_measured('sres_L8_hand_vs_library_rel_l2_f16', **errs)
    assert max(errs.values()) < 2e-3, errs                               # both routes round the same operands to f16 and accumulate in float32


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(16, 576, 512, 94, 150), (16, 192, 128, 166, 278)])
def test_one_hot_kernel_is_a_shift_full_size_gpu(shape):
    """Size-independent property at the configs[3] sizes (L8 and L13 as padded: 16 frames): with w[dh, dw] = a one-hot matrix
    the convolution copies channel ci_sel of the input, shifted by (dh, dw), into channel co_sel -- bit-exact, every tile."""
    n, ci, co, ho, wo = shape
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, ho + 2, wo + 2, ci, generator=g).half().cuda()
    for (dh, dw, co_sel, ci_sel) in [(0, 0, 0, ci - 1), (1, 2, co - 1, 0), (2, 1, 65, 64)]:
        wp = torch.zeros(3, 3, co, ci, dtype=torch.float16, device='cuda')
        wp[dh, dw, co_sel, ci_sel] = 1.0
        out = c2.conv2d_valid(x, wp, ho, wo)
        assert torch.equal(out[..., co_sel], x[:, dh:dh + ho, dw:dw + wo, ci_sel])
        out[..., co_sel] = 0
        assert float(out.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(16, 576, 512, 96, 160), (16, 192, 128, 168, 288)])
def test_weight_gradient_is_the_adjoint_of_the_forward_full_size_gpu(shape):
    """<dy, conv(x, w)> = <wgrad(x, dy), w> for a random w at the configs[3] sizes (gradient frames of whole patches): checks the
    split-K ranges and the patch walk of the weight-gradient kernel against the forward kernel (itself checked bit-exactly above)."""
    n, ci, co, hd, wd = shape
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, hd + 2, wd + 2, ci, generator=g).half().cuda()
    dy = torch.randn(n, hd, wd, co, generator=g).half().cuda()
    wp = (torch.randn(3, 3, co, ci, generator=g) / math.sqrt(9 * ci)).half().cuda()
    y = c2.conv2d_valid(x, wp, hd, wd)
    terms = y.double() * dy.double()
    lhs = float(terms.sum())
    gw = c2.conv2d_wgrad(x, dy)
    rhs = float((gw.double() * wp.double()).sum())
    # the two sides differ by the ONE rounding of y to f16 (relative 2^-12 rms per element, independent signs): the expected
    # difference is 2^-12 * sqrt(sum terms^2) (the sum itself is ~1e3 by cancellation of 1e8 terms); gate at 5 sigma
    sigma = 2.0 ** -12 * float(terms.square().sum().sqrt())
    record_measured(f'sres_wgrad_adjoint_{ci}x{co}', lhs=lhs, rhs=rhs, diff=abs(lhs - rhs), sigma=sigma)
    assert abs(lhs - rhs) <= 5 * sigma, (lhs, rhs, sigma)


# ---------------------------------------------------------------------------------------------------------------------
# float32-accurate contraction from split 16-bit operands

def test_split16_keeps_22_bits_cpu():
    g = torch.Generator().manual_seed(1)
    t = torch.randn(4096, generator=g) * torch.logspace(-2, 2, 4096)
    hi, lo = c2.split16(t)
    err = ((hi.double() + lo.double()) - t.double()).abs()
    # 11 + 11 mantissa bits (two roundings) while the low part is a normal float16, i.e. |t| >= 2^-3; an absolute 2^-25 below that
    assert bool((err <= torch.maximum(t.double().abs() * 2.0 ** -21, torch.tensor(2.0 ** -24, dtype=torch.float64))).all())
    s = c2.pow2_scale(torch.tensor([3e-7, -1.5e-6]))
    assert float(torch.log2(s)) == float(torch.floor(torch.log2(s))) and 512 <= 1.5e-6 * float(s) <= 1024


@pytest.mark.parametrize('name', list(CASES))
def test_split_node_plumbing_matches_reference_cpu(name):
    """The float32 node (split operands, stacked channels) on CPU tensors through the plain-PyTorch composition: output and the three
    gradients of the reference's modulated_conv2d."""
    g = load_golden('modconv2d')
    x, weight, style, gain, dy = inputs(name)
    n, ci, co, h, w_, k, pad = CASES[name]
    c_first = ci // 2
    x, weight, style = x.requires_grad_(True), weight.requires_grad_(True), style.requires_grad_(True)
    first, second = x[:, :c_first], x[:, c_first:].detach()
    wn, mod, demod = _torch_modulation(weight, style, gain)
    y = ml._ModConv2dSplit.apply(first, second, wn, mod, demod, pad)
    np.testing.assert_allclose(y.detach().numpy(), g[name + '_y'], rtol=2e-4, atol=2e-5)
    gx, gw, gs = torch.autograd.grad(y, [x, weight, style], dy)
    np.testing.assert_allclose(gx[:, :c_first].numpy(), g[name + '_gx'][:, :c_first], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gw.numpy(), g[name + '_gw'], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(gs.numpy(), g[name + '_gs'], rtol=5e-4, atol=5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_split_node_is_float32_accurate_gpu(oracle, name):
    """float32 inputs through the split-operand contraction on the 16-bit matrix cores vs the float64 oracle: errors at the float32
    level (the north-star tolerance is 1e-3; the gate here is 2e-5 of the tensor's scale), forward and all gradients."""
    x, weight, style, gain, dy = inputs(name)
    n, ci, co, h, w_, k, pad = CASES[name]
    xd, wd, sd, dyd = [t.double().numpy() for t in (x, weight, style, dy)]
    wn, mod, demod = _modulation(wd, sd, float(gain))
    y_ref = oracle.conv2d(xd * mod[:, :, None, None], wn, padding=pad) * demod[:, :, None, None]
    gx_ref = oracle.conv2d_dgrad(dyd * demod[:, :, None, None], wn, h, w_, padding=pad) * mod[:, :, None, None]
    gwn_ref = oracle.conv2d_wgrad(xd * mod[:, :, None, None], dyd * demod[:, :, None, None], 3, 3, padding=pad)
    first = x.cuda().requires_grad_(True)
    wt = torch.tensor(wn, dtype=torch.float32, device='cuda').requires_grad_(True)
    y = ml.modulated_conv2d(first, None, wt, torch.tensor(mod, dtype=torch.float32, device='cuda'), torch.tensor(demod, dtype=torch.float32, device='cuda'), padding=pad)
    before = c2.stats['launches']
    gx, gw = torch.autograd.grad(y, [first, wt], dy.cuda() * 1e-4)          # small gradients: exercises the power-of-two scaling
    assert c2.stats['launches'] - before == 2
    for got, ref, scale in ((y, y_ref, 1.0), (gx, gx_ref, 1e-4), (gw, gwn_ref, 1e-4)):
        err = np.abs(got.detach().double().cpu().numpy() - ref * scale).max() / (np.abs(ref).max() * scale)
        record_measured(f'sres_split_f32_{name}_{tuple(ref.shape)}', rel_max=err)
        assert err < 2e-5, err


@pytest.mark.gpu
def test_split_node_vs_library_float32_layer_gpu(monkeypatch):
    """L1 of the generator (539 -> 512 channels, 36 x 29 planes, float32): the split-operand route and the library's float32
    convolution against float64 on the same operands -- the hand-written route is not further from float64 than the library."""
    torch.manual_seed(0)
    n, ci, co, h, w = 2, 539, 512, 29, 36
    first = torch.randn(n, 512, h, w, device='cuda').requires_grad_(True)
    second = torch.randn(n, 27, h, w, device='cuda')
    weight = torch.randn(co, ci, 3, 3, device='cuda').requires_grad_(True)
    mod = (1 + 0.3 * torch.randn(n, ci, device='cuda')).requires_grad_(True)
    demod = (0.02 * (0.5 + torch.rand(n, co, device='cuda'))).requires_grad_(True)
    dy = torch.randn(n, co, h + 2, w + 2, device='cuda') * 1e-3

    def run(flag, dt):
        monkeypatch.setattr(ml, 'SPLIT_F32', flag)
        args = [t.detach().to(dt).requires_grad_(True) for t in (first, weight, mod, demod)]
        y = ml.modulated_conv2d(args[0], second.to(dt), args[1], args[2], args[3], padding=2)
        return [y.detach().double()] + [t.double() for t in torch.autograd.grad(y, args, dy.to(dt))]
    truth = run(False, torch.float64)
    hand, lib = run(True, torch.float32), run(False, torch.float32)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    errs = {}
    for k_, h_, l_, t_ in zip(('y', 'd_x', 'd_weight', 'd_mod', 'd_demod'), hand, lib, truth):
        errs[k_ + '_hand'], errs[k_ + '_lib'] = rel(h_, t_), rel(l_, t_)
        assert errs[k_ + '_hand'] < 2e-5 and errs[k_ + '_hand'] <= 4 * errs[k_ + '_lib'] + 2e-6, (k_, errs)
    record_measured('sres_L1_split_f32_vs_f64_rel_max', **errs)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 512, 512, 31, 38), (1, 37, 21, 9, 13), (3, 70, 130, 17, 5)])
def test_fused_operand_split_equals_tensor_expressions_gpu(shape, monkeypatch):
    """lvg_split16_frames / lvg_plane_absmax (one pass: scale, float16 x 2 split, placement in the padded frame) against the tensor
    expressions they replace (pow2_scale, split16, strided copies), and lvg_nhwc_f32_to_nchw (scaled float32 result back to NCHW planes) against mul / permute / contiguous, through the whole float32
    node: outputs and gradients bit for bit (d_mod: to the reduction order of a float32 sum)."""
    n, ci, co, h, w = shape
    torch.manual_seed(11)
    first = torch.randn(n, ci, h, w, device='cuda') * 3
    weight = torch.randn(co, ci, 3, 3, device='cuda') * 0.05
    mod = 1 + 0.3 * torch.randn(n, ci, device='cuda')
    demod = 0.02 * (0.5 + torch.rand(n, co, device='cuda'))
    dy = torch.randn(n, co, h + 2, w + 2, device='cuda') * 1e-3

    def run(flag):
        monkeypatch.setattr(ml, 'FUSED_SPLIT', flag)
        args = [t.detach().clone().requires_grad_(True) for t in (first, weight, mod, demod)]
        y = ml._ModConv2dSplit.apply(args[0], None, args[1], args[2], args[3], 2)
        return [y.detach()] + list(torch.autograd.grad(y, args, dy))
    fused, plain = run(True), run(False)
    for name, a, b in zip(('y', 'd_x', 'd_weight', 'd_mod', 'd_demod'), fused, plain):
        if name == 'd_mod':
            # the same products summed over a contiguous tensor instead of a permuted view: the library's reduction order differs
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), name
            continue
        assert torch.equal(a, b), (name, float((a - b).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 64, 128, 20, 36), (1, 128, 362, 21, 54), (3, 192, 64, 9, 150), (2, 64, 181, 33, 22), (1, 64, 3, 8, 16), (1, 64, 128, 11, 38)],
                         ids=['128ch', '362_of_384_w54', 'wide_64_w150', '181_of_192_w22', 'three_of_64', 'w38'])
def test_plane_output_equals_channels_last_output_gpu(shape, dtype):
    """lvg_conv2d_frames_planes (round 6: the convolution stores NCHW planes itself -- exchanged MFMA operands, [channel][pixel] staging) against
    lvg_conv2d_frames followed by the tensor transposition: the same accumulators, so with pre = None the two agree BIT FOR BIT (one rounding
    each); with the demodulation scale the plane kernel rounds once where the two-pass route rounds twice: compared with the float32
    product. Shapes: ragged tile edges in both directions (widths of 16 k + 4 and 16 k + 6 = the network's 38 / 54 / 86 / 150 / 278, heights not multiples of 8), channel counts that are not
    multiples of 64 (the padded channels are computed and dropped), the last-64-channels launch."""
    from torch_utils.ops import conv2d_frames as c2
    n, ci, co, ho, wo = shape
    co_pad = c2.round_up(co, c2.CH)
    torch.manual_seed(3)
    x = torch.randn(n, ho + 4, wo + 5, ci, device='cuda').to(dtype)
    w = (torch.randn(3, 3, co_pad, ci, device='cuda') / (3 * ci ** 0.5)).to(dtype)
    ref = c2.conv2d_valid(x, w, ho, wo, offset=(1, 2))                                   # [n, ho, wo, co_pad]
    got = c2.conv2d_valid_planes(x, w, ho, wo, co, offset=(1, 2))
    assert got.shape == (n, co, ho, wo) and got.dtype == dtype and got.is_contiguous()
    assert torch.equal(got, ref[..., :co].permute(0, 3, 1, 2))
    pre = 0.5 + torch.rand(n, co, device='cuda')
    conv_acc = c2.conv2d_valid(x, w, ho, wo, offset=(1, 2), out_dtype=torch.float32)[..., :co].permute(0, 3, 1, 2)
    acc = conv_acc * pre[:, :, None, None]
    got = c2.conv2d_valid_planes(x, w, ho, wo, co, offset=(1, 2), pre=pre)
    # one rounding of the float32 accumulator times the scale: of the float32 product (bfloat16: multiply, then convert) or of the exact product
    # (float16: hipcc contracts multiply + conversion into v_fma_mixlo_f16)
    exact = conv_acc.double() * pre.double()[:, :, None, None]
    assert bool(((got.double() - exact).abs() <= (acc.to(dtype).double() - exact).abs()).all())       # at least as close as the float32-rounded product


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 70, 27, 100, 18, 22), (1, 128, 0, 181, 33, 52)], ids=['cond_100_w24', 'no_cond_181_w54'])
def test_plane_output_route_of_the_modulated_convolution_gpu(shape, dtype, monkeypatch):
    """The fused 16-bit node with the convolution storing demodulated planes itself (round 6) against the two-pass form it replaces (channels-last
    result + transposing pass): output within one 16-bit rounding (one rounding instead of two), every gradient within the 16-bit tolerance of
    the test above -- d demod comes from the saved planes (sum d_out * out / demod) instead of the channels-last y."""
    n, c1, c2, co, h, w = shape
    torch.manual_seed(7)
    x = torch.randn(n, c1, h, w, device='cuda').to(dtype)
    cond = torch.randn(n, c2, h, w, device='cuda').to(dtype) if c2 else None
    weight = torch.randn(co, c1 + c2, 3, 3, device='cuda') / ((c1 + c2) * 9) ** 0.5
    mod = 0.5 + torch.rand(n, c1 + c2, device='cuda')
    demod = 0.5 + torch.rand(n, co, device='cuda')
    dy = torch.randn(n, co, h + 2, w + 2, device='cuda').to(dtype)

    def run(planes):
        monkeypatch.setattr(ml, 'PLANES_OUT', planes)
        args = [t.detach().clone().requires_grad_(True) for t in (x, weight, mod, demod)]
        y = ml.modulated_conv2d(args[0], cond, args[1], args[2], args[3], padding=2)
        return [y.detach()] + list(torch.autograd.grad(y, args, dy))
    new, old = run(True), run(False)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    for name, a, b in zip(('y', 'd_x', 'd_weight', 'd_mod', 'd_demod'), new, old):
        a, b = a.float(), b.float()
        if name == 'y':
            assert float(((a - b).abs() / b.abs().clamp_min(1e-3)).max()) <= 1.01 * eps, name
        else:
            assert float((a - b).abs().max()) <= 4 * eps * float(b.abs().max()), (name, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 128, 100, 27, 20, 38), (1, 64, 181, 0, 33, 22), (2, 192, 60, 3, 9, 150)], ids=['cond_27', 'single', 'wide_150'])
def test_plane_output_dot_products_gpu(shape, dtype):
    """lvg_conv2d_frames_planes_dot: next to the planes, per-half-tile sums of accumulator * partner plane (the data gradient's d styles): the rows added
    in order against the float64 sum of float32-accumulator * partner; the planes themselves unchanged by the extra output; reproducible."""
    from torch_utils.ops import conv2d_frames as c2
    n, ci, ca, cb, ho, wo = shape
    co_pad = c2.round_up(ca + cb, c2.CH)
    torch.manual_seed(5)
    x = torch.randn(n, ho + 2, wo + 2, ci, device='cuda').to(dtype)
    w = (torch.randn(3, 3, co_pad, ci, device='cuda') / (3 * ci ** 0.5)).to(dtype)
    a = torch.randn(n, ca, ho, wo, device='cuda').to(dtype)
    b = torch.randn(n, cb, ho, wo, device='cuda').to(dtype) if cb else None
    pre = 0.5 + torch.rand(n, ca, device='cuda')
    planes, partial = c2.conv2d_valid_planes(x, w, ho, wo, ca, pre=pre, dot=(a, b))
    assert torch.equal(planes, c2.conv2d_valid_planes(x, w, ho, wo, ca, pre=pre))
    acc = c2.conv2d_valid(x, w, ho, wo, out_dtype=torch.float32)[..., :ca + cb].permute(0, 3, 1, 2).double()
    oth = (a if b is None else torch.cat((a, b), dim=1)).double()
    want = (acc * oth).sum(dim=(2, 3))
    got = partial.double().sum(dim=1)
    scale = (acc.abs() * oth.abs()).sum(dim=(2, 3))
    assert partial.shape[2] == ca + cb and bool(((got - want).abs() <= 2e-6 * scale + 1e-6).all()), float(((got - want).abs() / scale).max())
    _, again = c2.conv2d_valid_planes(x, w, ho, wo, ca, pre=pre, dot=(a, b))
    assert torch.equal(partial, again)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(12))
def test_plane_output_random_shapes_gpu(seed):
    """Seeded random shapes for the plane-storing convolution (+ dot products): frame counts 1..3, 64..256 input channels, output channel counts that are or are
    not multiples of 64 / 128 (both launches: tiles of 128 channels and the last 64), heights 3..40, even widths 4..70 (every residue of the 16-pixel tile and of the
    8-pixel store segment), random input offsets, with and without a second partner tensor; float16 and bfloat16 alternate."""
    from torch_utils.ops import conv2d_frames as c2
    rs = np.random.RandomState(400 + seed)
    dtype = torch.float16 if seed % 2 == 0 else torch.bfloat16
    n, ci = int(rs.randint(1, 4)), 64 * int(rs.randint(1, 5))
    co = int(rs.randint(1, 300))
    ho, wo = int(rs.randint(3, 41)), 2 * int(rs.randint(2, 36))
    oy, ox = int(rs.randint(0, 3)), int(rs.randint(0, 4))
    co_pad = c2.round_up(co, c2.CH)
    ca = int(rs.randint(1, co + 1))
    cb = int(rs.randint(0, co - ca + 1)) if seed % 3 else 0
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(n, ho + 2 + oy + int(rs.randint(0, 3)), wo + 2 + ox + int(rs.randint(0, 5)), ci, device='cuda', generator=g).to(dtype)
    w = (torch.randn(3, 3, co_pad, ci, device='cuda', generator=g) / (3 * ci ** 0.5)).to(dtype)
    a = torch.randn(n, ca, ho, wo, device='cuda', generator=g).to(dtype)
    b = torch.randn(n, cb, ho, wo, device='cuda', generator=g).to(dtype) if cb else None
    ref = c2.conv2d_valid(x, w, ho, wo, offset=(oy, ox))
    got, partial = c2.conv2d_valid_planes(x, w, ho, wo, co, offset=(oy, ox), dot=(a, b))
    assert torch.equal(got, ref[..., :co].permute(0, 3, 1, 2)), (n, ci, co, ho, wo, oy, ox)
    acc = c2.conv2d_valid(x, w, ho, wo, offset=(oy, ox), out_dtype=torch.float32)[..., :ca + cb].permute(0, 3, 1, 2).double()
    oth = (a if b is None else torch.cat((a, b), dim=1)).double()
    want, scale = (acc * oth).sum(dim=(2, 3)), (acc.abs() * oth.abs()).sum(dim=(2, 3))
    assert bool(((partial.double().sum(dim=1) - want).abs() <= 2e-6 * scale + 1e-6).all()), (n, ci, co, ca, cb, ho, wo)
