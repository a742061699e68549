"""lvg.ada_augment.AugmentPipe vs golden vectors produced by the REFERENCE pipeline
(tests/golden/make_golden_ada.py): every transform at fixed quantiles (debug_percentile), the
seeded runs (random numbers must be consumed in the reference's order), the analytic filters, the
input gradient through upfirdn2d / grid_sample, and the random temporal filter. CPU run = plain
PyTorch op definitions; GPU run = HIP upfirdn2d (12-tap sym6 up / down)."""

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers.ada_cfg import TRAIN_SRES_KW, IN_AUGMENT_KW, EXTRA_KW, sample_video

from lvg.ada_augment import AugmentPipe

CASES = (('train', TRAIN_SRES_KW), ('in', IN_AUGMENT_KW), ('extra', EXTRA_KW))


def _run_fixed(device, atol):
    g = load_golden('ada_augment')
    video = torch.tensor(g['video'], device=device)
    for tag, kw in CASES:
        if device != 'cpu' and kw.get('noise', 0) > 0:
            continue                      # additive noise comes from the device generator: only the CPU stream is pinned
        pipe = AugmentPipe(**kw).to(device)
        np.testing.assert_allclose(pipe.Hz_geom.cpu().numpy(), g[f'{tag}_Hz_geom'], rtol=1e-6)
        np.testing.assert_allclose(pipe.Hz_fbank.cpu().numpy(), g[f'{tag}_Hz_fbank'], rtol=1e-6, atol=1e-8)
        for q in (20, 50, 85):
            torch.manual_seed(7)
            got = pipe(video, debug_percentile=q / 100).cpu().numpy()
            np.testing.assert_allclose(got, g[f'{tag}_q{q}'], rtol=0, atol=atol, err_msg=f'{tag} q{q}')
    pipe = AugmentPipe(**TRAIN_SRES_KW).to(device)
    v = video.clone().requires_grad_(True)
    torch.manual_seed(7)
    y = pipe(v, debug_percentile=0.7)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g['train_q70'], rtol=0, atol=atol)
    (y * torch.linspace(-1, 1, y.numel(), device=device).reshape(y.shape)).sum().backward()
    np.testing.assert_allclose(v.grad.cpu().numpy(), g['train_q70_grad'], rtol=0, atol=10 * atol)


def test_fixed_quantile_transforms_match_reference_cpu():
    torch.set_num_threads(4)
    _run_fixed('cpu', 2e-5)


def test_seeded_runs_consume_random_numbers_like_the_reference_cpu():
    torch.set_num_threads(4)
    g = load_golden('ada_augment')
    video = torch.tensor(g['video'])
    for tag, kw in CASES:
        pipe = AugmentPipe(**kw)
        for p in (1.0, 0.4):
            pipe.p.fill_(p)
            torch.manual_seed(11)
            got = pipe(video).numpy()
            np.testing.assert_allclose(got, g[f'{tag}_seed11_p{int(p * 10)}'], rtol=0, atol=2e-5, err_msg=f'{tag} p={p}')
    pipe = AugmentPipe(**TRAIN_SRES_KW)
    pipe.p.fill_(0.6)
    torch.manual_seed(3)
    got = pipe.random_temporal_filter(sample_video(frames=20, height=6, width=8)).numpy()
    np.testing.assert_allclose(got, g['temporal_seed3'], rtol=0, atol=2e-6)
    still = sample_video(frames=1, height=48, width=48)[:2]
    fpipe = AugmentPipe(imgfilter=1, imgfilter_bands=[1, 1, 0.5, 1])
    np.testing.assert_allclose(fpipe(still, debug_percentile=0.8).numpy(), g['filter_q80'], rtol=0, atol=2e-5)
    torch.manual_seed(5)
    np.testing.assert_allclose(fpipe(still).numpy(), g['filter_seed5'], rtol=0, atol=2e-5)


def test_identity_when_everything_is_off_and_image_filter_runs_cpu():
    video = sample_video()
    assert torch.equal(AugmentPipe()(video), video)
    pipe = AugmentPipe(**TRAIN_SRES_KW)
    pipe.p.fill_(0.0)
    torch.testing.assert_close(pipe(video), video, rtol=1e-5, atol=1e-5)      # gates closed: resampled through identity
    big = sample_video(frames=2, height=48, width=48)                          # 43-tap bank needs > 21 px of reflect padding
    out = AugmentPipe(imgfilter=1)(big, debug_percentile=0.8)                  # T > 1: per-frame depthwise taps
    assert out.shape == big.shape and torch.isfinite(out).all() and not torch.allclose(out, big)
    one = AugmentPipe(imgfilter=1)(big[:, :, :1], debug_percentile=0.8)
    torch.testing.assert_close(out[:, :, :1], one, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_fixed_quantile_transforms_match_reference_gpu():
    _run_fixed('cuda', 1e-4)


@pytest.mark.gpu
def test_r1_style_double_backward_through_augment_gpu():
    """R1 differentiates the discriminator input gradient again: second-order through the HIP
    upfirdn2d up/down pair and grid_sample must exist and be finite."""
    from torch_utils.ops import grid_sample_gradfix
    grid_sample_gradfix.enabled = True                                       # as the train scripts do
    pipe = AugmentPipe(**TRAIN_SRES_KW).cuda()
    v = sample_video().cuda().requires_grad_(True)
    w = torch.randn(1, 3, 1, 1, 1, device='cuda', requires_grad=True)
    y = (pipe(v, debug_percentile=0.3) * w).tanh().sum()
    (gv,) = torch.autograd.grad(y, v, create_graph=True)
    gv.square().sum().backward()
    assert torch.isfinite(w.grad).all() and float(w.grad.abs().sum()) > 0


# ---- fused stages (torch_utils/ops/ada_ops.py, csrc/ada_augment.hip): oracle vs the composition of reference ops, HIP vs oracle -------

def _random_maps(n, width, height, seed):
    """Inverse affine maps [N, 3, 3] in centred pixel units: any rotation, zoom 0.6 .. 1.6 per axis, shifts up to a fifth of the image."""
    from lvg import ada_augment as aa
    g = torch.Generator().manual_seed(seed)
    th = (torch.rand(n, generator=g) * 2 - 1) * np.pi
    sx, sy = torch.exp2(torch.rand(n, generator=g) * 1.4 - 0.7), torch.exp2(torch.rand(n, generator=g) * 1.4 - 0.7)
    tx, ty = (torch.rand(n, generator=g) - 0.5) * 0.4 * width, (torch.rand(n, generator=g) - 0.5) * 0.4 * height
    return aa.shift2(tx, ty) @ aa.turn2(th) @ aa.zoom2(sx, sy)


def test_oracle_warp_matches_reference_composition_cpu(oracle):
    """orc_ada_warp (float64 C) against pad -> upsample2d -> affine_grid + grid_sample -> downsample2d (reference ada_augment.py:286-301) in
    float64, margins from the same rule (:275-284); the identity map with zero margins must also reproduce the low-pass round trip."""
    pipe = AugmentPipe(**TRAIN_SRES_KW)
    torch.manual_seed(0)
    for (n, k, h, w), seed in (((2, 3, 14, 22), 1), ((3, 2, 9, 7), 2), ((1, 4, 20, 12), 3)):
        x = torch.randn(n, k, h, w, dtype=torch.float64)
        g_inv = _random_maps(n, w, h, seed)
        margins = [int(v) for v in pipe._warp_margins(g_inv, w, h).tolist()]
        want = pipe._warp_composed(x, g_inv.double(), margins).numpy()
        got = oracle.ada_warp(x.numpy(), g_inv.double().numpy(), pipe.Hz_geom.double().numpy(), margins)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
    x = torch.randn(1, 2, 12, 10, dtype=torch.float64)
    eye = torch.eye(3).unsqueeze(0)
    want = pipe._warp_composed(x, eye.double(), [0, 0, 0, 0]).numpy()
    np.testing.assert_allclose(oracle.ada_warp(x.numpy(), eye.numpy(), pipe.Hz_geom.double().numpy(), [0, 0, 0, 0]), want, rtol=0, atol=1e-10)


def test_oracle_colour_matches_reference_expressions_cpu(oracle):
    """orc_ada_colour against C[:3, :3] @ x + C[:3, 3:], + noise * sigma, cutout mask (reference ada_augment.py:376-381, :407-427)."""
    torch.manual_seed(1)
    n, t, h, w = 3, 2, 6, 8
    x = torch.randn(n, 3, t, h, w, dtype=torch.float64)
    cmat = torch.eye(4, dtype=torch.float64).repeat(n, 1, 1) + 0.3 * torch.randn(n, 4, 4, dtype=torch.float64)
    noise, sigma = torch.randn_like(x), torch.rand(n, dtype=torch.float64)
    cut = torch.tensor([[0.5, 0.5, 0.5, 0.5], [0.1, 0.9, 0.5, 0.5], [0.3, 0.3, 0.0, 0.0]], dtype=torch.float64)
    flat = cmat[:, :3, :3] @ x.reshape(n, 3, -1) + cmat[:, :3, 3:]
    want = flat.reshape(x.shape) + noise * sigma.reshape(n, 1, 1, 1, 1)
    xs = (torch.arange(w).reshape(1, 1, 1, -1) + 0.5) / w
    ys = (torch.arange(h).reshape(1, 1, -1, 1) + 0.5) / h
    keep = torch.logical_or((xs - cut[:, 0].reshape(n, 1, 1, 1)).abs() >= cut[:, 2].reshape(n, 1, 1, 1) / 2,
                            (ys - cut[:, 1].reshape(n, 1, 1, 1)).abs() >= cut[:, 3].reshape(n, 1, 1, 1) / 2)
    want = want * keep.reshape(n, 1, 1, h, w).to(torch.float64)
    got = oracle.ada_colour(x.numpy(), cmat.numpy(), noise.numpy(), sigma.numpy(), cut.numpy())
    np.testing.assert_allclose(got, want.numpy(), rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [((2, 3, 14, 22), 1, 1.0), ((3, 2, 9, 7), 2, 1.0), ((1, 4, 20, 12), 3, 1.0), ((2, 5, 72, 128), 4, 1.0), ((1, 2, 40, 56), 5, 6.0)],
                         ids=['small', 'tiny', 'tall', 'sres_quarter', 'zoomed_out'])
def test_hip_warp_matches_oracle_gpu(case, oracle):
    """lvg_ada_warp (one launch, float32, coordinates in double) against orc_ada_warp (float64). 'zoomed_out': a footprint larger than the
    LDS source patch (the direct-read path) and mostly outside the image (zeros)."""
    from torch_utils.ops import ada_ops
    from lvg import ada_augment as aa
    (n, k, h, w), seed, extra_zoom = case
    pipe = AugmentPipe(**TRAIN_SRES_KW).cuda()
    torch.manual_seed(seed)
    x = torch.randn(n, k, h, w, device='cuda')
    g_inv = (_random_maps(n, w, h, seed) @ aa.zoom2(extra_zoom, extra_zoom)).cuda()
    margins = pipe._warp_margins(g_inv, w, h)
    assert ada_ops.warp_supported(x, pipe.Hz_geom)
    got = ada_ops.ada_warp(x, g_inv, margins, pipe.Hz_geom).cpu().numpy()
    want = oracle.ada_warp(x.cpu().numpy(), g_inv.cpu().double().numpy(), pipe.Hz_geom.cpu().double().numpy(), [int(v) for v in margins.tolist()])
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-5 * max(1.0, np.abs(want).max()))
    # ... and the whole stage as the pipeline calls it (fused) against its composition of library ops (float32 both)
    comp = pipe._warp_composed(x, g_inv, [int(v) for v in margins.tolist()]).cpu().numpy()
    np.testing.assert_allclose(pipe._warp(x, g_inv).cpu().numpy(), comp, rtol=0, atol=1e-4 * max(1.0, np.abs(want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize('case', [((2, 3, 14, 22), 1, 1.0), ((1, 4, 20, 12), 3, 1.0), ((2, 5, 72, 128), 4, 1.0), ((1, 2, 40, 56), 5, 6.0)],
                         ids=['small', 'tall', 'sres_quarter', 'zoomed_out'])
def test_hip_warp_backward_is_the_adjoint_gpu(case):
    """lvg_ada_warp_adjoint: <A x, r> = <x, A^T r> against the forward kernel, the input gradient against autograd of the reference's
    composition (pad / upfirdn2d / grid_sample: their own backward passes), and the backward of the backward (R1) against the forward map."""
    from torch_utils.ops import ada_ops
    from lvg import ada_augment as aa
    (n, k, h, w), seed, extra_zoom = case
    pipe = AugmentPipe(**TRAIN_SRES_KW).cuda()
    torch.manual_seed(seed)
    x = torch.randn(n, k, h, w, device='cuda', requires_grad=True)
    r = torch.randn(n, k, h, w, device='cuda', requires_grad=True)
    g_inv = (_random_maps(n, w, h, seed) @ aa.zoom2(extra_zoom, extra_zoom)).cuda()
    margins = pipe._warp_margins(g_inv, w, h)
    y = ada_ops.ada_warp(x, g_inv, margins, pipe.Hz_geom)
    (gx,) = torch.autograd.grad(y, x, r, create_graph=True)
    lhs, rhs = float((y.double() * r.double()).sum()), float((x.double() * gx.double()).sum())
    scale = float((y.double() * r.double()).abs().sum())
    assert abs(lhs - rhs) <= 2e-5 * scale, (lhs, rhs, scale)
    xc = x.detach().clone().requires_grad_(True)
    yc = pipe._warp_composed(xc, g_inv, [int(v) for v in margins.tolist()])
    (gc,) = torch.autograd.grad(yc, xc, r.detach())
    torch.testing.assert_close(gx.detach(), gc, rtol=0, atol=1e-4 * max(1.0, float(gc.abs().max())))
    # second order: d/d r <gx, s> = A s
    s_ = torch.randn_like(gx)
    (gr,) = torch.autograd.grad(gx, r, s_)
    want = ada_ops.ada_warp(s_, g_inv, margins, pipe.Hz_geom)
    torch.testing.assert_close(gr, want, rtol=0, atol=1e-5 * max(1.0, float(want.abs().max())))


@pytest.mark.gpu
def test_hip_colour_pass_matches_oracle_and_autograd_gpu(oracle):
    from torch_utils.ops import ada_ops
    torch.manual_seed(2)
    n, t, h, w = 3, 4, 18, 26
    x = torch.randn(n, 3, t, h, w, device='cuda', requires_grad=True)
    cmat = (torch.eye(4).repeat(n, 1, 1) + 0.3 * torch.randn(n, 4, 4)).cuda()
    noise, sigma = torch.randn(n, 3, t, h, w, device='cuda'), torch.rand(n, device='cuda')
    cut = torch.tensor([[0.5, 0.5, 0.5, 0.5], [0.1, 0.9, 0.5, 0.5], [0.3, 0.3, 0.0, 0.0]], device='cuda')
    for args in ((cmat, noise, sigma, cut), (cmat, None, None, None), (None, noise, sigma, None), (None, None, None, cut)):
        y = ada_ops.ada_colour(x, *args)
        want = oracle.ada_colour(x.detach().cpu().numpy(), *[None if a is None else a.cpu().numpy() for a in args])
        np.testing.assert_allclose(y.detach().cpu().numpy(), want, rtol=0, atol=2e-6)
    # first and second order: y is affine in x, so d y / d x applied to r is the linear part, whatever the order
    y = ada_ops.ada_colour(x, cmat, noise, sigma, cut)
    r = torch.randn_like(y, requires_grad=True)
    (gx,) = torch.autograd.grad(y, x, r, create_graph=True)
    lin = oracle.ada_colour(np.zeros((n, 3, t, h, w)), None, None, None, None)          # (shape helper)
    keep = torch.tensor(oracle.ada_colour(np.ones((n, 3, t, h, w)), None, None, None, cut.cpu().numpy()), device='cuda', dtype=torch.float32)
    want_gx = torch.einsum('nij,nithw->njthw', cmat[:, :3, :3], r * keep)
    torch.testing.assert_close(gx, want_gx, rtol=1e-5, atol=1e-5)
    s = torch.randn_like(gx)
    (gr,) = torch.autograd.grad(gx, r, s)
    want_gr = torch.einsum('nij,njthw->nithw', cmat[:, :3, :3], s) * keep
    torch.testing.assert_close(gr, want_gr, rtol=1e-5, atol=1e-5)
    assert lin.shape == (n, 3, t, h, w)


@pytest.mark.gpu
def test_fused_stages_are_what_the_pipeline_runs_gpu():
    """The GPU pipeline goes through both fused launches (no silent fall-back to the composition) and reads nothing back to the host in
    its forward pass."""
    from torch_utils.ops import ada_ops
    calls = []
    orig_w, orig_c = ada_ops.ada_warp, ada_ops.ada_colour
    ada_ops.ada_warp = lambda *a, **k: (calls.append('warp'), orig_w(*a, **k))[1]
    ada_ops.ada_colour = lambda *a, **k: (calls.append('colour'), orig_c(*a, **k))[1]
    try:
        pipe = AugmentPipe(**{**TRAIN_SRES_KW, 'noise': 1, 'cutout': 1}).cuda()
        v = sample_video().cuda()
        torch.manual_seed(0)
        out = pipe(v)
    finally:
        ada_ops.ada_warp, ada_ops.ada_colour = orig_w, orig_c
    assert calls == ['warp', 'colour'] and out.shape == v.shape and torch.isfinite(out).all()


def test_second_call_builds_no_new_constants_cpu():
    """The pipeline must be replayable from a hipGraph after one eager pass (SuperResTrainer(use_graphs=True)): every constant vector / matrix
    it needs is built from Python numbers ONCE per (values, device) -- a tensor built from a list is a host-to-device copy, which a stream
    capture does not allow (and which blocked the host in the middle of the eager pipeline: train_sres 608 -> 496 ms) -- and later
    calls only read the caches. The cached tensors are shared: the pipeline's results must not depend on call order (nobody writes to them)."""
    from lvg import ada_augment as aa
    pipe = AugmentPipe(**TRAIN_SRES_KW).train()
    pipe.p.fill_(0.8)
    clip = sample_video()
    torch.manual_seed(3)
    first = pipe(clip)
    built = (len(aa._CONST), len(aa._MAT_BASE))
    snapshot = {k: v.clone() for k, v in aa._CONST.items()}
    torch.manual_seed(3)
    second = pipe(clip)
    assert (len(aa._CONST), len(aa._MAT_BASE)) == built and built[0] > 0
    assert all(torch.equal(v, aa._CONST[k]) for k, v in snapshot.items())          # read-only by contract
    assert torch.equal(first, second)
