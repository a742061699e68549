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
