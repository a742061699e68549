"""Hand-written implicit-GEMM convolution with fused temporal taps + epilogue (csrc/conv3d_igemm.hip,
torch_utils/ops/conv3d_frames.py).

CPU: the oracle's conv3d restatement (+ the modulation algebra) against vectors produced by the REFERENCE's
`temporal_modulated_conv3d` + bias_act (tests/golden/make_golden_modconv3d.py), and the shipped plain-PyTorch
definition against the oracle.
GPU: HIP vs oracle on seeded ragged cases (f16 / bf16; tile seams, frame / clip borders, every epilogue input),
HIP vs the reference golden, and at BASELINE.json configs[1] sizes the size-independent property
"a one-hot kernel is a zero-padded shift" (bit-exact) plus agreement with the MIOpen route inside the generator block."""

import math

import numpy as np
import pytest
import torch

from conftest import load_golden, record_measured
from helpers.modconv3d_inputs import inputs
from torch_utils.ops import conv3d_frames as cf


def _modulation(weight, style, gain):
    """numpy restatement of generator_lres.py:97-112 -> (scaled weight, modulation [N,Ci,T], demodulation [N,Co,T])."""
    w = weight / np.abs(weight).max(axis=(1, 2, 3, 4), keepdims=True)
    s = style / np.abs(style).max(axis=(1, 2), keepdims=True)
    w = w / math.sqrt(np.prod(w.shape[1:]))
    demod = 1.0 / np.sqrt(np.einsum('oizyx,nit->not', w ** 2, s ** 2) + 1e-8)
    return w, s * gain, demod


def _frames(v):
    """[N, C, T, H, W] -> time-major frames [(T N), C, H, W]"""
    n, c, t, h, w = v.shape
    return np.ascontiguousarray(v.transpose(2, 0, 1, 3, 4)).reshape(t * n, c, h, w)


def _video(fr, n):
    tn, c, h, w = fr.shape
    return fr.reshape(tn // n, n, c, h, w).transpose(1, 2, 0, 3, 4)


@pytest.mark.parametrize('name', ['k333', 'k133', 'k311'])
def test_oracle_matches_reference_modulated_conv3d(oracle, name):
    g = load_golden('modconv3d')
    x, weight, style, bias, gain = [t.double().numpy() for t in inputs(name)]
    n = x.shape[0]
    w, mod, demod = _modulation(weight, style, float(gain))
    # the oracle's style side (orc_style_prep, frames order) is the same function: pinned to the reference through this test
    omod, odemod = oracle.style_prep(style.transpose(2, 0, 1), (w ** 2).sum(axis=(2, 3, 4)))
    t_, ci_, co_ = style.shape[2], style.shape[1], w.shape[0]
    np.testing.assert_allclose(omod.reshape(t_, n, ci_).transpose(1, 2, 0) * float(gain), mod, rtol=1e-12, atol=0)
    np.testing.assert_allclose(odemod.reshape(t_, n, co_).transpose(1, 2, 0), demod, rtol=1e-10, atol=0)
    mod, demod = omod.reshape(t_, n, ci_).transpose(1, 2, 0) * float(gain), odemod.reshape(t_, n, co_).transpose(1, 2, 0)
    xm = x * mod[:, :, :, None, None]
    y = oracle.conv3d_frames(_frames(xm), w, shift=n)
    y = _video(y, n) * demod[:, :, :, None, None]
    np.testing.assert_allclose(y, g[name + '_conv'], rtol=2e-4, atol=2e-5)
    z = oracle.bias_act(y, bias, dim=1, act='lrelu', clamp=2.0)
    np.testing.assert_allclose(z, g[name + '_act'], rtol=2e-4, atol=2e-5)


def _case(seed, t, n, ci, co, h, w, kt, kh, kw, dtype, device, with_res=False):
    g = torch.Generator().manual_seed(seed)
    f = t * n
    x = torch.randn(f, ci, h, w, generator=g).to(dtype).to(device).contiguous(memory_format=torch.channels_last)
    weight = (torch.randn(co, ci, kt, kh, kw, generator=g) / math.sqrt(ci * kt * kh * kw)).to(dtype).to(device)
    pre = (0.5 + torch.rand(f, co, generator=g)).to(device)
    post = None if with_res else torch.randn(f, co, generator=g).to(device)
    b = None if with_res else (0.3 * torch.randn(co, generator=g)).to(dtype).to(device)
    res = torch.randn(f, co, h, w, generator=g).to(dtype).to(device).contiguous(memory_format=torch.channels_last) if with_res else None
    return x, weight, pre, b, res, post


def _np(t):
    return None if t is None else t.detach().double().cpu().numpy()


def _oracle_all(oracle, x, weight, n, pre, b, res, post, act, clamp):
    acc = oracle.conv3d_frames(_np(x), _np(weight), shift=n)
    out, ysum, msq = oracle.modconv_epilogue(acc, _np(pre), _np(b), _np(res), _np(post), taps=1, shift=n, act=act, clamp=clamp)
    return out, acc, msq


@pytest.mark.parametrize('kt,kh,kw,with_res', [(3, 3, 3, False), (1, 3, 3, True), (3, 1, 1, False)])
def test_plain_definition_matches_oracle_cpu(oracle, kt, kh, kw, with_res):
    x, weight, pre, b, res, post = _case(1, 4, 2, 8, 16, 3, 5, kt, kh, kw, torch.float32, 'cpu', with_res)
    act, clamp = ('linear', None) if with_res else ('lrelu', 1.5)
    out, ysum, msq = cf.conv3d_frames_forward(x, weight, 2, pre, b, res, post, act=act, clamp=clamp, want_msq=True)
    o_out, o_acc, o_msq = _oracle_all(oracle, x, weight, 2, pre, b, res, post, act, clamp)
    np.testing.assert_allclose(out.numpy(), o_out, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ysum.numpy(), o_acc, rtol=1e-4, atol=1e-5)
    assert abs(float(msq) - o_msq) <= 1e-4 * o_msq


GPU_CASES = [
    # t, n, ci, co,  h,  w, kt, kh, kw, with_res     (pixels = t*n*h*w: ragged against the 128-pixel tile on purpose)
    (5, 2, 64, 64, 6, 7, 3, 3, 3, False),            # several clips per tile row, clip borders inside tiles
    (3, 2, 64, 128, 5, 9, 1, 3, 3, True),            # spatial only + residual, BN = 128
    (6, 1, 128, 64, 3, 4, 3, 1, 1, False),           # temporal only, two channel chunks
    (4, 3, 128, 192, 9, 16, 3, 3, 3, False),         # two chunks x three taps: band double buffering, Co = 3 x 64
    (2, 2, 64, 64, 18, 32, 1, 3, 3, False),          # W = 32: band of 194 rows
    (1, 2, 64, 64, 36, 64, 1, 3, 3, True),           # W = 64: band of 258 rows, single band
    (7, 1, 256, 128, 5, 8, 3, 3, 3, False),          # four chunks
    (6, 2, 64, 128, 8, 8, 5, 3, 3, False),           # discriminator kernel: 5 temporal taps (45 taps in all)
    (3, 2, 64, 64, 7, 9, 1, 5, 5, False),            # 5 x 5 spatial taps: reach of 2 rows + 2 pixels (25 mask bits)
    (8, 1, 64, 64, 4, 5, 7, 1, 3, True),             # 7 temporal taps (the mask's temporal bits), kh != kw, frames shorter than the reach
]


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', GPU_CASES)
def test_hip_matches_oracle_gpu(oracle, case, dtype):
    t, n, ci, co, h, w, kt, kh, kw, with_res = case
    x, weight, pre, b, res, post = _case(7, t, n, ci, co, h, w, kt, kh, kw, dtype, 'cuda', with_res)
    assert cf.supported(x, weight)
    act, clamp = ('linear', None) if with_res else ('lrelu', 1.5)
    out, ysum, msq = cf.conv3d_frames_forward(x, weight, n, pre, b, res, post, act=act, clamp=clamp, want_msq=True)
    o_out, o_acc, o_msq = _oracle_all(oracle, x, weight, n, pre, b, res, post, act, clamp)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11          # one output rounding; accumulation is float32
    np.testing.assert_allclose(_np(ysum), o_acc, rtol=1.5 * eps, atol=eps * 0.05)
    np.testing.assert_allclose(_np(out), o_out, rtol=1.5 * eps, atol=eps * 0.05)
    assert abs(float(msq) - o_msq) <= 1e-3 * o_msq
    # the no-statistic / no-saved-sum launch writes the same `out`
    out2, ysum2, msq2 = cf.conv3d_frames_forward(x, weight, n, pre, b, res, post, act=act, clamp=clamp, want_msq=False, keep_sum=False)
    assert ysum2 is None and msq2 is None and torch.equal(out, out2)


@pytest.mark.gpu
def test_channel_slice_input_and_mirrored_weight_is_the_data_gradient_gpu():
    """x may be a channel slice of a wider channels-last tensor (what the backward pass hands over: the centre tap of the
    tap-stacked gradient), and conv(dy, mirrored + transposed weight) is the data gradient of the convolution."""
    t, n, ci, co, h, w, kt = 4, 2, 64, 128, 9, 16, 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(t * n, ci, h, w, generator=g).cuda().requires_grad_(True)
    weight = (torch.randn(co, ci, kt, 3, 3, generator=g) / math.sqrt(ci * kt * 9)).cuda()
    dz = torch.randn(t * n, kt * co, h, w, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    dy = dz[:, co:2 * co]
    assert not dy.is_contiguous(memory_format=torch.channels_last) and cf.supported(dy, weight.transpose(0, 1).to(torch.bfloat16))
    wt = weight.to(torch.bfloat16).flip(2, 3, 4).transpose(0, 1)
    gx = cf.conv3d_frames_forward(dy, wt, n, keep_sum=False)[0]
    gx_copy = cf.conv3d_frames_forward(dy.contiguous(memory_format=torch.channels_last), wt, n, keep_sum=False)[0]
    assert torch.equal(gx, gx_copy)
    y = cf._conv_ref(x, weight.to(torch.bfloat16).float(), n)
    ref, = torch.autograd.grad(y, x, dy.float())
    np.testing.assert_allclose(_np(gx), _np(ref), rtol=1.5 * 2.0 ** -8, atol=2e-3)


def test_wgrad_definition_matches_oracle_cpu(oracle):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(6, 3, 4, 8, generator=g)
    dy = torch.randn(6, 5, 4, 8, generator=g)
    gw = cf.conv3d_frames_wgrad(x, dy, 3, 3, 3, 2)
    np.testing.assert_allclose(gw.numpy(), oracle.conv3d_frames_wgrad(x.numpy(), dy.numpy(), 3, 3, 3, shift=2), rtol=1e-4, atol=1e-5)


WGRAD_CASES = [
    # t, n, ci, co,  h,  w, kt     (K-step = 64 pixels = 64 / w image rows: frame changes inside a step for h % (64 / w) != 0)
    (4, 2, 64, 64, 9, 16, 3),        # 4 rows per step, 9-row frames: steps straddle frames
    (3, 2, 64, 128, 5, 8, 3),        # 8 rows per step, 5-row frames: up to two frame changes per step
    (2, 2, 128, 64, 18, 32, 1),      # 2 rows per step
    (1, 3, 64, 64, 36, 64, 1),       # 1 row per step, single band slot triple
    (6, 2, 64, 64, 8, 8, 5),         # discriminator kernel, whole frame per step
    (5, 1, 128, 192, 3, 16, 3),      # ragged: 15 rows -> partial last step
]


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', WGRAD_CASES)
def test_hip_wgrad_matches_oracle_gpu(oracle, case, dtype, monkeypatch):
    t, n, ci, co, h, w, kt = case
    g = torch.Generator().manual_seed(9)
    x = torch.randn(t * n, ci, h, w, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    dz = torch.randn(t * n, 2 * co, h, w, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    dy = dz[:, co:]                                                     # a channel slice: pixel stride 2 * co
    assert cf.wgrad_supported(x, dy, kt, 3, 3)
    ref = oracle.conv3d_frames_wgrad(_np(x), _np(dy), kt, 3, 3, shift=n)
    scale = np.abs(ref).max()
    for splits in (None, 1, 3):
        if splits is not None:
            monkeypatch.setenv('LVG_WGRAD_SPLITS', str(splits))
        gw = cf.conv3d_frames_wgrad(x, dy, kt, 3, 3, n)
        assert gw.shape == (co, ci, kt, 3, 3) and gw.dtype == torch.float32
        np.testing.assert_allclose(_np(gw), ref, rtol=1e-4, atol=2e-5 * scale)      # exact products, float32 accumulation
    again = cf.conv3d_frames_wgrad(x, dy, kt, 3, 3, n)
    assert torch.equal(gw, again)                                                   # no atomics: reproducible


@pytest.mark.gpu
def test_tile_variants_agree_bitwise_gpu():
    """Every tile shape / weight-ring depth runs the same arithmetic in the same order."""
    from torch_utils.ops import _hip
    x, weight, pre, b, res, post = _case(3, 4, 2, 128, 128, 9, 16, 3, 3, 3, torch.bfloat16, 'cuda')
    a = cf.conv3d_frames_forward(x, weight, 2, pre, b, res, post, act='lrelu', clamp=2.0)[0]
    try:
        for bm, bn, nb in [(128, 64, 2), (256, 128, 3), (256, 64, 2), (128, 128, 3), (128, 128, 2)]:
            assert _hip.lib().lvg_conv3d_frames_set_plan(bm, bn, nb, 0) == 0
            r = cf.conv3d_frames_forward(x, weight, 2, pre, b, res, post, act='lrelu', clamp=2.0)[0]
            assert torch.equal(a, r), (bm, bn, nb)
    finally:
        _hip.lib().lvg_conv3d_frames_set_plan(0, 0, 0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(44, 4, 64, 64, 36, 64, 1), (52, 4, 64, 64, 32, 32, 5), (88, 4, 128, 64, 18, 32, 3), (44, 4, 64, 64, 36, 63, 1)])
def test_persistent_workgroups_compute_the_same_bits_gpu(case):
    """The persistent form of the 64-channel tiles (a workgroup walks many tiles: next band prefetched during the K-steps, weight ring running
    on, output staged in the finished band buffer) against one workgroup per tile: every output -- activation, saved sum, the per-tile
    partial sums of squares -- bit for bit, with residual, bias and both scales; ragged last tile and odd width included."""
    from torch_utils.ops import _hip
    t, n, ci, co, h, w, kt = case
    x, weight, pre, b, res, post = _case(5, t, n, ci, co, h, w, kt, 3, 3, torch.bfloat16, 'cuda', with_res=True)
    lib = _hip.lib()
    outs = {}
    try:
        for bm in (256, 128):
            for persist in (0, 1):
                assert lib.lvg_conv3d_frames_set_plan(bm, 0, 0, persist) == 0
                outs[bm, persist] = cf.conv3d_frames_forward(x, weight, n, pre, b, res, post, act='lrelu', clamp=2.0, want_msq=True)
    finally:
        lib.lvg_conv3d_frames_set_plan(0, 0, 0, 0)
    torch.cuda.synchronize()
    for bm in (256, 128):
        for got, want, name in zip(outs[bm, 1], outs[bm, 0], ['out', 'ysum', 'msq']):
            assert torch.isfinite(want.float()).all()
            assert torch.equal(got, want), (bm, name, float((got.float() - want.float()).abs().max()))
    assert torch.equal(outs[256, 1][0], outs[128, 1][0])


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['k333', 'k133', 'k311'])
def test_hip_matches_reference_golden_gpu(name):
    """Reference temporal_modulated_conv3d + bias_act (float32, CPU) vs the fused kernel on bf16 frames."""
    g = load_golden('modconv3d')
    x, weight, style, bias, gain = inputs(name)
    n = x.shape[0]
    w, mod, demod = _modulation(weight.double().numpy(), style.double().numpy(), float(gain))
    dev = 'cuda'
    fr = lambda a: torch.tensor(np.ascontiguousarray(a.transpose(2, 0, 1)).reshape(-1, a.shape[1]), dtype=torch.float32, device=dev)
    xm = torch.tensor(_frames(x.double().numpy() * mod[:, :, :, None, None]), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    # weights are O(1/sqrt(fan_in)): scale up into the well-resolved bf16 range and fold the factor into `pre`
    scale = float(np.abs(w).max())
    wb = torch.tensor(w / scale, device=dev).to(torch.bfloat16)
    out, ysum, _ = cf.conv3d_frames_forward(xm, wb, n, pre=fr(demod) * scale, b=bias.to(dev).to(torch.bfloat16), act='lrelu', clamp=2.0)
    got = _video(_np(out), n)
    np.testing.assert_allclose(got, g[name + '_act'], rtol=0, atol=3e-2)       # bf16 inputs, weights and output (values up to 2)
    assert np.abs(got - g[name + '_act']).mean() < 4e-3


FULL_SHAPES = [
    # BASELINE.json configs[1] (8 clips): frames, ci, co, h, w, kt, kh, kw, clips
    (80 * 8, 512, 512, 9, 16, 3, 3, 3, 8),
    (128 * 8, 128, 128, 18, 32, 1, 3, 3, 8),
    (128 * 8, 64, 64, 36, 64, 1, 3, 3, 8),
    (128 * 8, 64, 64, 32, 32, 5, 3, 3, 8),           # discriminator block at 32 x 32
]


@pytest.mark.gpu
@pytest.mark.parametrize('shape', FULL_SHAPES)
def test_one_hot_kernel_is_a_zero_padded_shift_full_size_gpu(shape):
    """Size-independent property at the full configs[1] sizes: with weight[o, c, dt, dh, dw] = [o == c] for ONE tap the
    convolution is a shift with zero fill -- bit-exact, which checks every tile seam, band offset and border mask."""
    f, ci, co, h, w, kt, kh, kw, n = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(f, ci, h, w, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    for (dt, dh, dw) in {(0, 0, 0), (kt - 1, kh - 1, kw - 1), (kt // 2, 0, kw - 1), (kt - 1, kh // 2, 0)}:
        weight = torch.zeros(co, ci, kt, kh, kw, dtype=torch.bfloat16, device='cuda')
        idx = torch.arange(min(ci, co), device='cuda')
        weight[idx, idx, dt, dh, dw] = 1
        out, _, _ = cf.conv3d_frames_forward(x, weight, n, keep_sum=False)
        st, sh, sw = dt - kt // 2, dh - kh // 2, dw - kw // 2
        v = x.reshape(f // n, n, ci, h, w)
        ref = torch.zeros_like(v)
        t = f // n
        ts, td = slice(max(st, 0), t + min(st, 0)), slice(max(-st, 0), t + min(-st, 0))
        hs, hd = slice(max(sh, 0), h + min(sh, 0)), slice(max(-sh, 0), h + min(-sh, 0))
        ws, wd = slice(max(sw, 0), w + min(sw, 0)), slice(max(-sw, 0), w + min(-sw, 0))
        ref[td, :, :, hd, wd] = v[ts, :, :, hs, ws]
        assert torch.equal(out, ref.reshape(f, ci, h, w)), (dt, dh, dw)


@pytest.mark.gpu
def test_generator_block_hand_conv_vs_miopen_route_gpu(monkeypatch):
    """One temporal residual block (forward + all gradients) in bf16 on the hand-written kernel and on the MIOpen +
    tap-gather route, both against the same block in float32: the hand-written route may not be further from the
    float32 result than the route it replaces (beyond noise), and the two 16-bit routes agree within bf16 bands."""
    from lvg.models import lres
    torch.manual_seed(0)
    blk = lres.Synthesis3dResBlock(64, 128, 128, temporal_ksize=3, spatial_ksize=3).cuda()
    n, t = 2, 6
    x0 = torch.randn(t * n, 128, 9, 16, device='cuda').contiguous(memory_format=torch.channels_last)
    lat = torch.randn(n, 64, t, device='cuda')

    monkeypatch.setattr(lres, 'HAND_CONV_MIN_TILES', 1)                # the test block is far smaller than the policy threshold

    def run(flag, dtype):
        monkeypatch.setattr(lres, 'HAND_CONV', flag)
        x = x0.clone().requires_grad_(True)
        for p in blk.parameters():
            p.grad = None
        y = blk.forward_frames(x, lat, magnitude_ema_beta=0.999, dtype=dtype)
        (y.float() * torch.linspace(-1, 1, y.numel(), device='cuda').reshape(y.shape)).sum().backward()
        return [y.detach().float(), x.grad.float()] + [p.grad.float().clone() for p in blk.parameters()]

    truth = run(False, torch.float32)
    hand = run(True, torch.bfloat16)
    miopen = run(False, torch.bfloat16)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))
    worst = dict(hand=0.0, miopen=0.0, hand_vs_miopen=0.0)
    for i, (h, m, tr) in enumerate(zip(hand, miopen, truth)):
        eh, em = rel(h, tr), rel(m, tr)
        worst = dict(hand=max(worst['hand'], eh), miopen=max(worst['miopen'], em), hand_vs_miopen=max(worst['hand_vs_miopen'], rel(h, m)))
        # the gate that matters: the hand-written route is not further from float32 than the library route it replaces
        assert eh <= 1.25 * em + 2e-3, (i, eh, em)
        # bf16 end to end through two modulated convolutions (K up to 3456 terms) and their backward: relative L2 error of
        # every tensor (output, input gradient, 10 parameter gradients) below 6e-2; the measured worst case is recorded
        assert eh < 6e-2 and rel(h, m) < 6e-2, (i, eh, rel(h, m))
    record_measured('lres_block_bf16_rel_l2_vs_f32', **worst)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(80 * 8, 512, 512, 9, 16, 3, 8), (128 * 8, 64, 64, 36, 64, 1, 8), (128 * 8, 64, 128, 32, 32, 5, 8)])
def test_weight_gradient_is_the_adjoint_of_the_forward_full_size_gpu(shape):
    """Size-independent property at the BASELINE.json configs[1] sizes: <dy, conv(x, w)> = <wgrad(x, dy), w> for every w --
    checks the split-K ranges, frame changes inside K-steps and clip borders of the weight-gradient kernel against the
    forward kernel (itself checked bit-exactly at these sizes by the one-hot test)."""
    f, ci, co, h, w, kt, n = shape
    g = torch.Generator().manual_seed(13)
    x = torch.randn(f, ci, h, w, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(f, co, h, w, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    weight = (torch.randn(co, ci, kt, 3, 3, generator=g) / math.sqrt(ci * kt * 9)).to(torch.bfloat16).cuda()
    y = cf.conv3d_frames_forward(x, weight, n, keep_sum=False)[0]                 # one bf16 rounding per output element
    gw = cf.conv3d_frames_wgrad(x, dy, kt, 3, 3, n)
    lhs = float((y.double() * dy.double()).sum())
    rhs = float((gw.double() * weight.double()).sum())
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(lhs - rhs) < 2e-4 * scale, (lhs, rhs, scale)
    # and against the library's weight gradient on the same tensors (bf16 result)
    ref = torch.ops.aten.convolution_backward(
        dy, x, weight[:, :, kt // 2].contiguous(memory_format=torch.channels_last), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    centre = gw[:, :, kt // 2]
    assert float((centre - ref.float()).norm() / ref.float().norm()) < 1e-2       # the centre temporal tap is a plain 2-D weight gradient


# ---------------------------------------------------------------------------------------------------------------------
# Pixel pairs: channel counts that are multiples of 32 but not of 64 on the hand-written kernels (lres._pairable)

@pytest.mark.parametrize('k', [3, 1])
def test_pixel_pair_weight_is_the_same_convolution_cpu(k):
    """conv(x, w) == unpair(conv(pair(x), pair_weight(w))) with 'same' zero padding, and unpair_weight_grad is the adjoint of pair_weight."""
    import torch.nn.functional as F
    from lvg.models import lres
    g = torch.Generator().manual_seed(21)
    f, ci, co, h, w = 3, 32, 96, 5, 12
    x = torch.randn(f, ci, h, w, generator=g).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(co, ci, 1, k, k, generator=g)
    ref = F.conv2d(x, wt[:, :, 0], padding=k // 2)
    w2 = lres.pair_weight(wt)
    y2 = F.conv2d(lres._pair_view(x), w2[:, :, 0], padding=k // 2).contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(lres._unpair_view(y2).numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    assert float((w2 == 0).float().mean()) >= 0.5 - 1e-6                    # half of the blocks of every tap are structurally zero
    g2 = torch.randn(w2.shape, generator=g)
    lhs = float((w2.double() * g2.double()).sum())
    rhs = float((wt.double() * lres.unpair_weight_grad(g2, co, ci).double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))                      # (the fold adds in float32)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(32, 32, 3), (32, 64, 3), (32, 64, 1)])
def test_pixel_pair_layers_match_library_route_gpu(monkeypatch, shape):
    """The first discriminator block's layers (32 -> 32, 32 -> 64 channels; bf16) through the pixel-pair views on the hand-written
    kernels against the library route of the same op: forward with the fused epilogue, input / weight / bias gradients."""
    from lvg.models import lres
    ci, co, k = shape
    g = torch.Generator().manual_seed(4)
    n, t, h, w = 2, 3, 16, 64
    x0 = torch.randn(t * n, ci, h, w, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    wt0 = (torch.randn(co, ci, 1, k, k, generator=g) / math.sqrt(ci * k * k)).to(torch.bfloat16).cuda()
    b0 = (0.3 * torch.randn(co, generator=g)).to(torch.bfloat16).cuda()
    dy = torch.randn(t * n, co, h, w, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    monkeypatch.setattr(lres, 'HAND_CONV_MIN_TILES', 1)

    def run(flag, dtype=torch.bfloat16):
        monkeypatch.setattr(lres, 'HAND_PAIR', flag)
        x, wt, b = x0.to(dtype).requires_grad_(True), wt0.to(dtype).requires_grad_(True), b0.to(dtype).requires_grad_(True)
        before = cf.stats['launches']
        if k == 1:
            y = lres.pointwise_conv(x, wt)
        else:
            y = lres.temporal_conv_epilogue(x, wt, n, (1, 1), b=b, act='lrelu', clamp=256.0)
        grads = torch.autograd.grad(y, [x, wt] + ([b] if k == 3 else []), dy.to(dtype))
        return cf.stats['launches'] - before, [y.detach().float()] + [t_.float() for t_ in grads]
    launches, hand = run(True)
    assert launches >= 2                                                    # forward + data gradient ran on the hand-written kernel
    launches0, lib = run(False)
    assert launches0 == 0
    _, truth = run(False, torch.float32)                                    # the same bf16-rounded operands in float32 arithmetic
    rel = lambda a, b_: float((a - b_).norm() / b_.norm().clamp_min(1e-12))
    # Both 16-bit routes differ from float32 mostly through leaky-ReLU sign flips of pre-activations within one bf16 rounding of zero
    # (~0.2 % of the elements, a factor 5 in the gradient each): the hand-written route may not be further from float32 than the library
    # route (beyond noise), and both stay inside the bf16 band.
    for i, (h_, l_, t_) in enumerate(zip(hand, lib, truth)):
        eh, el = rel(h_, t_), rel(l_, t_)
        assert eh <= 1.25 * el + 2e-3 and eh < 3e-2, (i, eh, el)


# ---------------------------------------------------------------------------------------------------------------------
# float32 tensors through split 16-bit operands (conv3d_frames.conv3d_frames_split32)

@pytest.mark.gpu
@pytest.mark.parametrize('case', [(5, 2, 64, 64, 6, 7, 3, 3, 3, False), (3, 2, 64, 128, 5, 16, 1, 3, 3, True), (4, 1, 128, 64, 8, 8, 5, 3, 3, False)])
def test_split32_matches_oracle_at_float32_accuracy_gpu(oracle, case):
    """float32 x / weight / bias / residual through the 16-bit kernel on split operands vs the float64 oracle: forward with the fused
    epilogue, saved sum, magnitude statistic; data and weight gradients (gradients scaled to 1e-4 to exercise the power-of-two
    scaling). Gate: 2e-5 of the tensor scale (measured ~1e-6; the north-star tolerance is 1e-3)."""
    t, n, ci, co, h, w, kt, kh, kw, with_res = case
    x, weight, pre, b, res, post = _case(7, t, n, ci, co, h, w, kt, kh, kw, torch.float32, 'cuda', with_res)
    assert cf.split32_supported(x, weight)
    act, clamp = ('linear', None) if with_res else ('lrelu', 1.5)
    out, ysum, msq = cf.conv3d_frames_split32(x, weight, n, pre, b, res, post, act=act, clamp=clamp, want_msq=True)
    o_out, o_acc, o_msq = _oracle_all(oracle, x, weight, n, pre, b, res, post, act, clamp)
    errs = {}
    for name, got, ref in (('out', out, o_out), ('ysum', ysum, o_acc)):
        errs[name] = float(np.abs(_np(got) - ref).max() / np.abs(ref).max())
        assert errs[name] < 2e-5, (name, errs)
    assert abs(float(msq) - o_msq) <= 1e-5 * o_msq
    g = torch.Generator().manual_seed(3)
    dy = (torch.randn(t * n, co, h, w, generator=g) * 1e-4).cuda().contiguous(memory_format=torch.channels_last)
    gx = cf.conv3d_frames_split32_dgrad(dy, weight, n)
    gx_ref = oracle.conv3d_frames(_np(dy), _np(weight.flip(2, 3, 4).transpose(0, 1)), shift=n)
    errs['gx'] = float(np.abs(_np(gx) - gx_ref).max() / np.abs(gx_ref).max())
    if w in (8, 16, 32, 64):
        gw = cf.conv3d_frames_split32_wgrad(x, dy, kt, kh, kw, n)
        gw_ref = oracle.conv3d_frames_wgrad(_np(x), _np(dy), kt, kh, kw, shift=n)
        errs['gw'] = float(np.abs(_np(gw) - gw_ref).max() / np.abs(gw_ref).max())
    record_measured(f'lres_split32_{kt}x{kh}x{kw}_{ci}to{co}', **errs)
    assert max(errs.values()) < 2e-5, errs


def test_second_order_nodes_are_closed_under_differentiation_cpu():
    """lres._HandConv / _HandDgrad / _HandWgrad (the twice-differentiable contraction of the R1 pass; on CPU tensors the same nodes run
    the plain definitions): value, input gradient and the gradients of an R1-style penalty on that gradient against autograd of F.conv3d."""
    import torch.nn.functional as F
    from lvg.models import lres
    torch.manual_seed(0)
    T, N, ci, co, h, w = 6, 2, 4, 5, 5, 6
    x = torch.randn(T * N, ci, h, w, requires_grad=True)
    wt = (torch.randn(co, ci, 3, 3, 3) * 0.3).requires_grad_(True)

    def ref(x, wt):
        v = x.reshape(T, N, ci, h, w).permute(1, 2, 0, 3, 4)
        return F.conv3d(v, wt, padding=1).permute(2, 0, 1, 3, 4).reshape(T * N, co, h, w)

    def r1(fn):
        y = fn(x, wt)
        (g,) = torch.autograd.grad(y.tanh().sum(), [x], create_graph=True)
        gw, gx = torch.autograd.grad(g.square().sum(), [wt, x])
        return y.detach(), g.detach(), gw, gx

    with lres.second_order():
        got = r1(lambda x, wt: lres.temporal_conv_frames(x, wt, N, (1, 1)))
    for a, b, name in zip(got, r1(ref), ['y', 'dx', 'd penalty / dw', 'd penalty / dx']):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6, name


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_second_order_on_hand_kernels_matches_library_route_gpu(dtype, monkeypatch):
    """The R1 pattern through one discriminator-sized layer (64 -> 128 channels, 5 x 3 x 3, 32 x 32 pixels): penalty on the input gradient,
    differentiated with respect to weight and input, on the hand-written kernels vs the library's kt-convolution form (float32 as the
    yardstick: both 16-bit routes must sit within 16-bit rounding of it). Every pass must have run on conv3d_igemm / conv3d_wgrad."""
    from lvg.models import lres
    from torch_utils.ops import conv3d_frames
    torch.manual_seed(1)
    T, N, ci, co, h, w = 8, 2, 64, 128, 32, 32
    x32 = torch.randn(T * N, ci, h, w, device='cuda').contiguous(memory_format=torch.channels_last)
    w32 = torch.randn(co, ci, 5, 3, 3, device='cuda') / (ci * 45) ** 0.5

    def r1(dt, hand):
        monkeypatch.setattr(lres, 'HAND_SECOND_ORDER', hand)
        x = x32.to(dt).requires_grad_(True)
        wt = w32.to(dt).requires_grad_(True)
        with lres.second_order():
            y = lres.temporal_conv_frames(x, wt, N, (1, 1))
        (g,) = torch.autograd.grad(y.float().tanh().sum(), [x], create_graph=True)
        gw, gx = torch.autograd.grad(g.float().square().sum(), [wt, x])
        return [t.detach().float() for t in (y, g, gw, gx)]

    ref = r1(torch.float32, False)
    before = conv3d_frames.stats['launches']
    hand = r1(dtype, True)
    assert conv3d_frames.stats['launches'] - before >= 5            # C, D, then C / W (from D') and D / W (from C')
    lib = r1(dtype, False)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    for a, b, c, name in zip(hand, lib, ref, ['y', 'dx', 'd penalty / dw', 'd penalty / dx']):
        scale = float(c.abs().max())
        assert float((a - c).abs().max()) <= tol * scale, (name, 'hand route', float((a - c).abs().max()) / scale)
        assert float((b - c).abs().max()) <= tol * scale, (name, 'library route', float((b - c).abs().max()) / scale)


def test_split32_stack_definition_cpu():
    """The stacked operand of the float32 route: three bfloat16 parts whose sum is the tensor to 24 bits, blocks in the requested order (CPU: the tensor expressions)."""
    from torch_utils.ops import conv3d_frames as cf
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(3, 16, 2, 5, generator=g) * torch.logspace(-6, 6, 16).reshape(1, 16, 1, 1)).contiguous(memory_format=torch.channels_last)
    out = cf.split32_stack(x, (0, 0, 1, 0, 1, 2))
    assert out.shape == (3, 96, 2, 5) and out.dtype == torch.bfloat16 and out.is_contiguous(memory_format=torch.channels_last)
    p1, p2, p3 = out[:, 0:16], out[:, 32:48], out[:, 80:96]
    assert torch.equal(out[:, 16:32], p1) and torch.equal(out[:, 48:64], p1) and torch.equal(out[:, 64:80], p2)
    err = (p1.double() + p2.double() + p3.double() - x.double()).abs() / x.double().abs().clamp_min(1e-30)
    assert float(err.max()) < 2.0 ** -22


@pytest.mark.gpu
@pytest.mark.parametrize('parts', [(0, 0, 1, 0, 1, 2), (0, 1, 2)])
def test_split32_stack_kernel_is_bit_identical_to_the_tensor_expressions_gpu(parts):
    from torch_utils.ops import conv3d_frames as cf
    g = torch.Generator().manual_seed(1)
    for shape in ((5, 64, 3, 4), (2, 8, 1, 1), (7, 136, 9, 16)):
        x = torch.randn(*shape, generator=g) * torch.logspace(-20, 20, shape[1]).reshape(1, -1, 1, 1)
        x.view(-1)[::97] = 0.0
        x.view(-1)[5] = float('inf')
        xd = x.cuda().contiguous(memory_format=torch.channels_last)
        got = cf.split32_stack(xd, parts)
        cf.SPLIT_STACK_HIP = False
        try:
            want = cf.split32_stack(xd, parts)
        finally:
            cf.SPLIT_STACK_HIP = True
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        got, want = got.permute(0, 2, 3, 1).contiguous(), want.permute(0, 2, 3, 1).contiguous()      # (1 x 1 frames: channels-last strides are not unique)
        nan = got.isnan() & want.isnan()                                          # inf - inf: NaN in both (its sign / payload bits are not defined)
        diff = (got.view(torch.int16) != want.view(torch.int16)) & ~nan
        assert int(diff.sum()) == 0, (shape, int(diff.sum()), got[diff][:4], want[diff][:4])
        assert int(nan.sum()) > 0


def _r1_pattern(fn, x, wt):
    """value, input gradient, and the gradients of an R1-style penalty on that input gradient"""
    y = fn(x, wt)
    (g,) = torch.autograd.grad(y.float().tanh().sum(), [x], create_graph=True)
    gw, gx = torch.autograd.grad(g.float().square().sum(), [wt, x])
    return [t.detach().float() for t in (y, g, gw, gx)]


@pytest.mark.parametrize('co,ci', [(16, 3), (3, 16)])
def test_thin_pointwise_nodes_are_closed_under_differentiation_cpu(co, ci):
    """pointwise_thin._ThinConv / _ThinWgrad (round 6: the 3-channel layers of an R1 pass on their own kernels; on CPU tensors the nodes run
    the plain definitions): the R1 pattern against autograd of F.conv2d."""
    import torch.nn.functional as F
    from torch_utils.ops import pointwise_thin
    torch.manual_seed(2)
    x = torch.randn(5, ci, 4, 6).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(co, ci) * 0.4).requires_grad_(True)
    got = _r1_pattern(lambda x, wt: pointwise_thin._ThinConv.apply(x, wt), x, wt)
    want = _r1_pattern(lambda x, wt: F.conv2d(x, wt[:, :, None, None]), x, wt)
    for a, b, name in zip(got, want, ['y', 'dx', 'd penalty / dw', 'd penalty / dx']):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6, name


@pytest.mark.parametrize('taps', [(3, 3, 3), (1, 3, 3), (1, 1, 1)])
def test_pixel_pair_nodes_are_closed_under_differentiation_cpu(taps):
    """lres._PairView / _UnpairView around lres._HandConv with lres.pair_weight (the 32-channel layers of an R1 pass as pixel pairs): the R1
    pattern against autograd of F.conv3d on the unpaired tensors."""
    import torch.nn.functional as F
    from lvg.models import lres
    torch.manual_seed(3)
    T, N, ci, co, h, w = 4, 2, 3, 5, 4, 6
    kt, kh, kw = taps
    x = torch.randn(T * N, ci, h, w).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(co, ci, kt, kh, kw) * 0.3).requires_grad_(True)

    def ref(x, wt):
        v = x.reshape(T, N, ci, h, w).permute(1, 2, 0, 3, 4)
        return F.conv3d(v, wt, padding=(kt // 2, kh // 2, kw // 2)).permute(2, 0, 1, 3, 4).reshape(T * N, co, h, w)

    def paired(x, wt):
        return lres._UnpairView.apply(lres._HandConv.apply(lres._PairView.apply(x), lres.pair_weight(wt), N))

    for a, b, name in zip(_r1_pattern(paired, x, wt), _r1_pattern(ref, x, wt), ['y', 'dx', 'd penalty / dw', 'd penalty / dx']):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6, name


@pytest.mark.gpu
@pytest.mark.parametrize('case', [('thin_3_to_32', 3, 32, (1, 1, 1), 64), ('pair_32_to_32', 32, 32, (1, 3, 3), 64), ('pair_32_to_64', 32, 64, (1, 3, 3), 64),
                                  ('pair_skip_32_to_64', 32, 64, (1, 1, 1), 64), ('skip_64_to_128', 64, 128, (1, 1, 1), 32)], ids=lambda c: c[0])
def test_second_order_routes_of_the_first_discriminator_block_gpu(case, monkeypatch):
    """The layers of the low-resolution discriminator that round 5 left on the library inside an R1 pass (3-channel input layer, 32-channel
    layers, 1 x 1 skip convolutions): the R1 pattern on the hand-written kernels against the library's kt-convolution form, float32 as the
    yardstick (both 16-bit routes within 16-bit rounding of it) -- and no library convolution may have run on the hand route."""
    from lvg.models import lres
    from torch_utils.ops import conv3d_frames
    name, ci, co, taps, size = case
    torch.manual_seed(4)
    T, N = 8, 2
    x32 = torch.randn(T * N, ci, size, size, device='cuda').contiguous(memory_format=torch.channels_last)
    fan = ci * taps[0] * taps[1] * taps[2]
    w32 = torch.randn(co, ci, *taps, device='cuda') / fan ** 0.5
    pad = (taps[1] // 2, taps[2] // 2)
    dt = torch.bfloat16

    def r1(dtype, hand):
        monkeypatch.setattr(lres, 'HAND_SECOND_ORDER', hand)
        x = x32.to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wt = w32.to(dtype).requires_grad_(True)
        with lres.second_order():
            return _r1_pattern(lambda x, wt: lres.temporal_conv_frames(x, wt, N, pad), x, wt)

    ref = r1(torch.float32, False)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        hand = r1(dt, True)
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    library = ('igemm_fwd', 'igemm_bwd', 'igemm_wrw', 'miopen', 'Cijk', 'ck::', 'xdlops', 'SubTensorOp', 'batched_transpose')
    assert not [k for k in names if any(tag in k for tag in library)], names
    lib = r1(dt, False)
    measured = {}
    for a, b, c, what in zip(hand, lib, ref, ['y', 'dx', 'd penalty / dw', 'd penalty / dx']):
        scale = float(c.abs().max())
        e_hand, e_lib = float((a - c).abs().max()) / scale, float((b - c).abs().max()) / scale
        measured[what] = [e_hand, e_lib]
        # bfloat16 rounding of y under tanh'' (|y| up to ~5 with 3 .. 288 unit-variance terms) costs up to a few per cent of the largest
        # element on EITHER 16-bit route: the gate is 2e-2, or twice what the library's route shows on the same tensors
        assert e_hand <= max(2e-2, 2 * e_lib), (name, what, 'hand route', e_hand, 'library route', e_lib)
    record_measured(f'lres_second_order_{name}', **{k.replace(' ', '_').replace('/', 'by'): v for k, v in measured.items()})
