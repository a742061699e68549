"""1 x 1 convolutions with a 3-channel side (csrc/pointwise_thin.hip, torch_utils/ops/pointwise_thin.py; reference model/generator_lres.py:600-640 ToRGB,
model/discriminator_lres.py:169 with kernel size 1). CPU: the oracle's definition against F.conv2d. GPU: all three passes against the oracle (one output
rounding), reproducibility of the weight gradient, and that the networks take the kernels."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from torch_utils.ops import pointwise_thin as pt


def test_oracle_is_the_1x1_convolution_cpu(oracle):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 16, 5, 7, generator=g, dtype=torch.float64)
    w = torch.randn(3, 16, generator=g, dtype=torch.float64)
    want = F.conv2d(x, w[:, :, None, None]).permute(0, 2, 3, 1).reshape(-1, 3)
    got = oracle.pointwise(x.permute(0, 2, 3, 1).reshape(-1, 16).numpy(), w.numpy())
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-12, atol=1e-12)


# (frames, H, W, Ci, Co): ToRGB and the discriminator's first layer at small sizes, every wide width, ragged pixel counts
CASES = [(3, 5, 7, 64, 3), (3, 5, 7, 3, 32), (2, 9, 16, 128, 3), (1, 3, 3, 8, 1), (5, 4, 6, 4, 16), (2, 36, 64, 64, 3), (2, 64, 64, 3, 32), (1, 1, 1, 2, 8)]


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', CASES)
def test_three_passes_match_oracle_gpu(oracle, case, dtype):
    f, h, w, ci, co = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(f, ci, h, w, generator=g).to(dtype)
    wt = (torch.randn(co, ci, generator=g) / ci ** 0.5).to(dtype)
    gy = torch.randn(f, co, h, w, generator=g).to(dtype)
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = wt.cuda().requires_grad_(True)
    if ci == 1 or co == 1 or xd.stride() != (h * w * ci, 1, w * ci, ci):
        xd = torch.empty_strided((f, ci, h, w), (h * w * ci, 1, w * ci, ci), dtype=dtype, device='cuda').copy_(x).requires_grad_(True)
    assert pt.supported(xd, wd)
    y = pt.pointwise_thin(xd, wd)
    assert y.dtype == dtype and y.stride() == (h * w * co, 1, w * co, co)
    gx, gw = torch.autograd.grad(y, (xd, wd), gy.cuda())
    xm = x.double().permute(0, 2, 3, 1).reshape(-1, ci).numpy()
    gm = gy.double().permute(0, 2, 3, 1).reshape(-1, co).numpy()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11          # one rounding of the output type
    want_y = oracle.pointwise(xm, wt.double().numpy())
    got_y = y.detach().double().cpu().permute(0, 2, 3, 1).reshape(-1, co).numpy()
    assert np.abs(got_y - want_y).max() <= eps * np.abs(want_y).max() + 1e-6
    want_gx = oracle.pointwise(gm, wt.double().t().contiguous().numpy())
    got_gx = gx.double().cpu().permute(0, 2, 3, 1).reshape(-1, ci).numpy()
    assert np.abs(got_gx - want_gx).max() <= eps * np.abs(want_gx).max() + 1e-6
    want_gw = gm.T @ xm
    assert np.abs(gw.double().cpu().numpy() - want_gw).max() <= eps * np.abs(want_gw).max() + 1e-5 * np.sqrt(f * h * w)
    # reproducible: fixed pixel ranges, fixed summation order
    gw2 = torch.autograd.grad(pt.pointwise_thin(xd, wd), wd, gy.cuda())[0]
    assert torch.equal(gw, gw2)


@pytest.mark.gpu
def test_networks_take_the_kernels_gpu(monkeypatch):
    from lvg.models import lres
    calls = []
    real = pt.pointwise_thin
    monkeypatch.setattr(lres.pointwise_thin, 'pointwise_thin', lambda x, w: (calls.append(tuple(w.shape)), real(x, w))[1])
    torch.manual_seed(0)
    G = lres.VideoGenerator().cuda()
    D = lres.VideoDiscriminator(seq_length=16, max_edge=64).cuda()
    video = G(1, 16, dtype=torch.bfloat16)
    D(video, dtype=torch.bfloat16).sum().backward()
    assert (3, 64) in calls and (32, 3) in calls, calls


@pytest.mark.gpu
def test_model_gradients_match_library_route_gpu():
    """Generator + discriminator, bfloat16: the same update with the thin kernels on and off. Two bf16 evaluations of these networks differ from run
    to run where library kernels with atomics remain (reproducibility), so the yardstick is the distance between two library-route runs."""
    from lvg.models import lres

    def run(on):
        lres.THIN_POINTWISE = on
        try:
            torch.manual_seed(0)
            G = lres.VideoGenerator().cuda()
            D = lres.VideoDiscriminator(seq_length=16, max_edge=64).cuda()
            video = G(1, 16, dtype=torch.bfloat16)
            F.softplus(-D(video, dtype=torch.bfloat16)).mean().backward()
            return (video.detach().float(), G.to_rgb.weight.grad.detach().float().clone(), D.blocks[0].conv_vid.weight.grad.detach().float().clone())
        finally:
            lres.THIN_POINTWISE = True

    # Round 6: the gate used to be max |difference| / max against ONE library-vs-library sample; that statistic is carried by single elements and its yardstick
    # ranged over 0.02 .. 0.115 from run to run (two of eight runs failed on an unchanged tree). Now: relative L2 distances, three library runs as the yardstick.
    thin, libs = run(True), [run(False) for _ in range(3)]
    for k, name in enumerate(('video', 'ToRGB weight gradient', 'first-layer weight gradient')):
        def dist(a, b):
            return float((a - b).norm() / b.norm())
        noise = max(dist(libs[i][k], libs[j][k]) for i in range(3) for j in range(3) if i != j)
        err = min(dist(thin[k], lib[k]) for lib in libs)
        print(f'[measured] {name}: thin vs library {err:.3g} (relative L2), library vs library {noise:.3g}')
        assert err <= 2 * noise + 2e-2, (name, err, noise)


# ---- weight gradient of the wide 1 x 1 convolutions (csrc/pointwise_wgrad.hip) ----------------------------------------------------------------
# (frames, H, W, Ci, Co): skip convolutions of both networks at small sizes; several channel tiles, a last K-step that is not full, one step only
WGRAD_CASES = [(3, 4, 8, 64, 64), (2, 9, 16, 128, 64), (5, 3, 8, 64, 192), (1, 2, 4, 64, 64), (7, 5, 8, 256, 128), (16, 36, 64, 64, 64)]


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', WGRAD_CASES)
def test_pointwise_wgrad_matches_oracle_gpu(oracle, case, dtype):
    from torch_utils.ops import conv3d_frames as cf
    f, h, w, ci, co = case
    g = torch.Generator().manual_seed(2)
    x = torch.randn(f, ci, h, w, generator=g).to(dtype)
    dy = torch.randn(f, co, h, w, generator=g).to(dtype)
    xd, dyd = x.cuda().contiguous(memory_format=torch.channels_last), dy.cuda().contiguous(memory_format=torch.channels_last)
    assert cf.pointwise_wgrad_supported(xd, dyd)
    gw = cf.pointwise_wgrad(xd, dyd)
    assert gw.shape == (co, ci) and gw.dtype == torch.float32
    xm = x.double().permute(0, 2, 3, 1).reshape(-1, ci).numpy()
    gm = dy.double().permute(0, 2, 3, 1).reshape(-1, co).numpy()
    want = oracle.pointwise(gm.T.copy(), xm.T.copy())                 # [co, pixels] x [ci, pixels]^T: the same contraction through the oracle's definition
    err = np.abs(gw.double().cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 2e-5, err                                            # exact products of 16-bit operands, float32 accumulation
    assert torch.equal(gw, cf.pointwise_wgrad(xd, dyd))               # fixed pixel ranges, fixed summation order
    # channel slices of wider tensors (what the tap-stacked backward hands over) and pixel-pair views
    wide = torch.randn(f, 2 * co, h, w, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    sl = wide[:, co:2 * co]
    assert cf.pointwise_wgrad_supported(xd, sl)
    want2 = sl.double().permute(0, 2, 3, 1).reshape(-1, co).t() @ xd.double().permute(0, 2, 3, 1).reshape(-1, ci)
    err2 = float((cf.pointwise_wgrad(xd, sl).double() - want2).abs().max() / want2.abs().max())
    assert err2 < 2e-5, err2


@pytest.mark.gpu
def test_discriminator_skip_convolutions_take_the_hand_kernels_gpu(monkeypatch):
    from lvg.models import lres
    from torch_utils.ops import conv3d_frames as cf
    shapes = []
    real = cf.pointwise_wgrad
    monkeypatch.setattr(cf, 'pointwise_wgrad', lambda x, dy: (shapes.append((x.shape[1], dy.shape[1])), real(x, dy))[1])
    torch.manual_seed(0)
    D = lres.VideoDiscriminator(seq_length=16, max_edge=64).cuda()
    video = torch.randn(1, 3, 16, 36, 64, device='cuda', requires_grad=True)
    D(video, dtype=torch.bfloat16).sum().backward()
    # the 32 -> 64 skip runs as pixel pairs (64 -> 128) next to the 64 -> 128 one; the deeper skips of this one-clip batch have fewer tiles than
    # HAND_CONV_MIN_TILES and stay on the library
    assert shapes.count((64, 128)) >= 2, shapes


# ---- weight gradient of frames of a few pixels: taps unrolled into channels + pointwise_wgrad (lres.small_frame_wgrad) --------------------------
def test_unrolled_taps_give_the_weight_gradient_cpu(oracle):
    from lvg.models import lres
    g = torch.Generator().manual_seed(3)
    t, n, ci, co, h, w, kt = 5, 2, 4, 3, 3, 4, 3
    x = torch.randn(t * n, ci, h, w, generator=g, dtype=torch.float64)
    dy = torch.randn(t * n, co, h, w, generator=g, dtype=torch.float64)
    col = lres.unroll_taps(x.contiguous(memory_format=torch.channels_last), kt, 3, 3, n)
    assert col.shape == (t * n, kt * 9 * ci, h, w) and col.is_contiguous(memory_format=torch.channels_last)
    gw = torch.einsum('fohw,fkhw->ok', dy, col).reshape(co, kt, 3, 3, ci).permute(0, 4, 1, 2, 3)
    np.testing.assert_allclose(gw.numpy(), oracle.conv3d_frames_wgrad(x.numpy(), dy.numpy(), kt, 3, 3, shift=n), rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', [(6, 4, 64, 64, 3, 4, 3), (8, 2, 128, 64, 3, 4, 1), (4, 2, 64, 128, 2, 4, 5), (24, 8, 512, 512, 3, 4, 3)])
def test_small_frame_wgrad_matches_oracle_gpu(oracle, case, dtype, monkeypatch):
    from lvg.models import lres
    monkeypatch.setattr(lres, 'SMALL_FRAME_WGRAD', True)              # (off by default: measured slower than the library on the two layers it applies to)
    t, n, ci, co, h, w, kt = case
    g = torch.Generator().manual_seed(4)
    x = torch.randn(t * n, ci, h, w, generator=g).to(dtype)
    dy = torch.randn(t * n, co, h, w, generator=g).to(dtype)
    xd, dyd = x.cuda().contiguous(memory_format=torch.channels_last), dy.cuda().contiguous(memory_format=torch.channels_last)
    wt = torch.empty(co, ci, kt, 3, 3)
    assert lres._small_frame_wgrad_takes(xd, co, wt, (1, 1))
    gw = lres.small_frame_wgrad(xd, dyd, kt, 3, 3, n)
    assert gw.shape == (co, ci, kt, 3, 3)
    if ci * co <= 128 * 128:                                          # (the full-size case would take the scalar oracle minutes: float64 matmul instead)
        want = torch.from_numpy(oracle.conv3d_frames_wgrad(x.double().numpy(), dy.double().numpy(), kt, 3, 3, shift=n))
    else:
        col = lres.unroll_taps(x.double().contiguous(memory_format=torch.channels_last), kt, 3, 3, n)
        want = torch.einsum('fohw,fkhw->ok', dy.double(), col).reshape(co, kt, 3, 3, ci).permute(0, 4, 1, 2, 3)
    err = float((gw.double().cpu() - want).abs().max() / want.abs().max())
    assert err < 2e-5, err
