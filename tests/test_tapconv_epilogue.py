"""Tap-stacked temporal convolution + fused gather/epilogue (csrc/tapconv_epilogue.hip,
lvg.models.lres._TapConvEpilogue) against the existing formulation (kt accumulated 2-D convolutions
followed by the separate epilogue), forward and all gradients; launch-level functions against their
explicit PyTorch formulas; whole generator against the reference golden with the flag on."""

import numpy as np
import pytest
import torch

from conftest import load_golden
from torch_utils.ops import modconv_epilogue as me

from lvg.models import lres


def _gather_case(seed, t, n, c, h, w, taps, dtype, device, with_res=False):
    g = torch.Generator().manual_seed(seed)
    f = t * n
    z = torch.randn(f, taps * c, h, w, generator=g).to(device=device, dtype=dtype).contiguous(memory_format=torch.channels_last)
    pre = (0.5 + torch.rand(f, c, generator=g)).to(device)
    post = None if with_res else torch.randn(f, c, generator=g).to(device)
    b = None if with_res else (0.3 * torch.randn(c, generator=g)).to(device=device, dtype=dtype)
    res = torch.randn(f, c, h, w, generator=g).to(device=device, dtype=dtype).contiguous(memory_format=torch.channels_last) if with_res else None
    return z, pre, b, res, post


def _compose(z, pre, b, res, post, taps, shift, act, clamp):
    """Differentiable float64 composition: shifted tap sum, then the epilogue definition."""
    f, kc, h, w = z.shape
    c = kc // taps
    y = torch.zeros(f, c, h, w, dtype=torch.float64, device=z.device)
    for k in range(taps):
        r = me._tap_ranges(k, taps, shift, f)
        if r is not None:
            pad = torch.zeros_like(y)
            pad[r[1]] = z[r[0], k * c:(k + 1) * c].double()
            y = y + pad
    spec, alpha, gain, cl = me._resolve(act, None, None, clamp)
    u = y * pre.double()[:, :, None, None]
    if b is not None:
        u = u + b.double()[None, :, None, None]
    if res is not None:
        u = u + res.double()
    if act == 'lrelu':
        u = torch.nn.functional.leaky_relu(u, alpha)
    u = u * gain
    if cl >= 0:
        u = u.clamp(-cl, cl)
    msq = u.detach().square().mean()
    return (u if post is None else u * post.double()[:, :, None, None]), y, msq


@pytest.mark.parametrize('taps,with_res', [(3, False), (5, False), (3, True)])
def test_launch_level_reference_formulas_cpu(oracle, taps, with_res):
    z, pre, b, res, post = _gather_case(0, 6, 2, 8, 3, 5, taps, torch.float32, 'cpu', with_res)
    act, clamp = ('linear', None) if with_res else ('lrelu', 1.5)
    out, ysum, msq = me.tap_gather_forward(z, pre, b, res, post, taps, 2, act=act, clamp=clamp, want_msq=True)
    np_ = lambda t: None if t is None else t.double().numpy()
    o_out, o_sum, o_msq = oracle.modconv_epilogue(np_(z), np_(pre), np_(b), np_(res), np_(post), taps=taps, shift=2, act=act, clamp=clamp)
    np.testing.assert_allclose(out.numpy(), o_out, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ysum.numpy(), o_sum, rtol=1e-5, atol=1e-5)
    assert abs(float(msq) - o_msq) < 1e-5 * o_msq
    zz = z.clone().requires_grad_(True)
    prer = pre.clone().requires_grad_(True)
    postr = post.clone().requires_grad_(True) if post is not None else None
    want, y, want_msq = _compose(zz, prer, b, res, postr, taps, 2, act, clamp)
    torch.testing.assert_close(out.double(), want.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ysum.double(), y.detach(), rtol=1e-5, atol=1e-5)
    assert abs(float(msq) - float(want_msq)) < 1e-5 * float(want_msq)
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    dz, d_pre, d_post, d_sum = me.tap_gather_backward(go, ysum, pre, b, res, post, taps, 2, act=act, clamp=clamp)
    inputs = [zz, prer] + ([postr] if postr is not None else [])
    grads = torch.autograd.grad(want, inputs, go.double())
    torch.testing.assert_close(dz.double(), grads[0].double(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(d_pre.double(), grads[1].double(), rtol=1e-4, atol=1e-4)
    if postr is not None:
        torch.testing.assert_close(d_post.double(), grads[2].double(), rtol=1e-4, atol=1e-4)


def _block_pair():
    torch.manual_seed(0)
    blk = lres.Synthesis3dResBlock(latent_dim=16, in_channels=8, out_channels=16, temporal_ksize=3, spatial_ksize=3)
    with torch.no_grad():
        blk.bias_0.normal_(0, 0.2)
        blk.bias_1.normal_(0, 0.2)
    return blk


def _run_block(blk, x, latent, flag, beta, monkeypatch):
    import copy
    monkeypatch.setattr(lres, 'TAP_STACK', flag)
    blk = copy.deepcopy(blk)                     # beta < 1 updates the magnitude EMAs: every run starts from the same state
    xr = x.clone().requires_grad_(True)
    out = blk.forward_frames(lres.frames_from_video(xr), latent, beta)
    (out * torch.linspace(-1, 1, out.numel(), device=out.device).reshape(out.shape)).sum().backward()
    return out.detach(), xr.grad, {k: v.grad.clone() for k, v in blk.named_parameters()}


def test_block_with_stacked_taps_equals_accumulated_taps_cpu(monkeypatch):
    blk = _block_pair()
    x = torch.randn(2, 8, 6, 5, 7)
    latent = torch.randn(2, 16, 6)
    a = _run_block(blk, x, latent, False, 1.0, monkeypatch)
    b = _run_block(blk, x, latent, True, 1.0, monkeypatch)
    torch.testing.assert_close(b[0], a[0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(b[1], a[1], rtol=1e-4, atol=1e-5)
    for k in a[2]:
        torch.testing.assert_close(b[2][k], a[2][k], rtol=1e-3, atol=1e-5, msg=k)


def test_generator_matches_reference_golden_with_stacked_taps_cpu(monkeypatch):
    monkeypatch.setattr(lres, 'TAP_STACK', True)
    torch.set_num_threads(8)
    from helpers.named_fill import fill_named
    g = load_golden('lres_models')
    G = lres.VideoGenerator()
    fill_named(G)
    with torch.no_grad():
        ws = G.compute_latent_ws(G.temporal_emb.blur(torch.tensor(g['noise'])), 16)
        video = G.synthesize_video(G._temporal_input(ws), ws, 16)
    np.testing.assert_allclose(video.numpy(), g['video'], rtol=0, atol=1e-3)


def _disc_run(D, video, flag, monkeypatch):
    monkeypatch.setattr(lres, 'TAP_STACK', flag)
    v = video.clone().requires_grad_(True)
    logits = D(v)
    (g,) = torch.autograd.grad(torch.nn.functional.softplus(-logits).mean(), v)
    return logits.detach(), g


def test_discriminator_with_stacked_taps_equals_accumulated_taps_cpu(monkeypatch):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    D = lres.VideoDiscriminator(seq_length=16, max_edge=64).requires_grad_(False)
    video = torch.rand(1, 3, 16, 36, 64) * 2 - 1
    a = _disc_run(D, video, False, monkeypatch)
    b = _disc_run(D, video, True, monkeypatch)
    torch.testing.assert_close(b[0], a[0], rtol=1e-4, atol=1e-5)
    scale = float(a[1].abs().max())
    assert float((b[1] - a[1]).abs().max()) <= 1e-3 * scale


def test_second_order_scope_falls_back_to_differentiable_form_cpu(monkeypatch):
    """R1: inside lres.second_order() the stacked form is not used, so grad-of-grad exists."""
    monkeypatch.setattr(lres, 'TAP_STACK', True)
    torch.manual_seed(0)
    layer = lres.Conv3dLayer(4, 8, spatial_ksize=3, temporal_ksize=3, activation='lrelu', conv_clamp=256)
    x = torch.randn(6, 4, 5, 7, requires_grad=True)
    with lres.second_order():
        y = layer.forward_frames(x, 2)
    (g,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    g.square().sum().backward()
    assert layer.weight.grad is not None and torch.isfinite(layer.weight.grad).all()
    y = layer.forward_frames(x, 2)                                  # outside the scope: first-order only, and it says so
    (g,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    with pytest.raises(RuntimeError):
        g.square().sum().backward()


CASES = [(8, 2, 64, 9, 16, 3), (6, 1, 8, 5, 7, 3), (12, 2, 32, 4, 6, 5), (4, 3, 256, 3, 4, 3)]


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('t,n,c,h,w,taps', CASES)
@pytest.mark.parametrize('with_res', [False, True])
def test_kernels_match_formulas_gpu(oracle, dtype, t, n, c, h, w, taps, with_res):
    z, pre, b, res, post = _gather_case(3, t, n, c, h, w, taps, dtype, 'cuda', with_res)
    act, clamp = ('linear', None) if with_res else ('lrelu', 2.0)
    out, ysum, msq = me.tap_gather_forward(z, pre, b, res, post, taps, n, act=act, clamp=clamp, want_msq=True)
    np_ = lambda v: None if v is None else v.double().cpu().numpy()
    o_out, o_sum, _ = oracle.modconv_epilogue(np_(z), np_(pre), np_(b), np_(res), np_(post), taps=taps, shift=n, act=act, clamp=clamp)
    o_eps = 2e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(out.double().cpu().numpy(), o_out, rtol=o_eps, atol=o_eps)
    np.testing.assert_allclose(ysum.double().cpu().numpy(), o_sum, rtol=o_eps, atol=o_eps)
    cpu = lambda v: None if v is None else v.cpu()
    out_r, ysum_r, msq_r = me.tap_gather_forward(cpu(z).float(), cpu(pre), cpu(b.float()) if b is not None else None,
                                                 cpu(res.float()) if res is not None else None, cpu(post), taps, n, act=act, clamp=clamp, want_msq=True)
    eps = 2e-5 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(out.float().cpu(), out_r, rtol=eps, atol=eps)
    torch.testing.assert_close(ysum.float().cpu(), ysum_r, rtol=eps, atol=eps)
    assert abs(float(msq) - float(msq_r)) < 1e-3 * float(msq_r)
    go = torch.randn(out.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)).to(dtype).contiguous(memory_format=torch.channels_last)
    dz, d_pre, d_post, d_sum = me.tap_gather_backward(go, ysum, pre, b, res, post, taps, n, act=act, clamp=clamp)
    ref = me.tap_gather_backward(cpu(go).float(), cpu(ysum).float(), cpu(pre), cpu(b.float()) if b is not None else None,
                                 cpu(res.float()) if res is not None else None, cpu(post), taps, n, act=act, clamp=clamp)
    for name, a, r in zip(('dz', 'd_pre', 'd_post', 'd_sum'), (dz, d_pre, d_post, d_sum), ref):
        if r is None:
            assert a is None
            continue
        scale = float(r.abs().max()) + 1e-12
        tol = 3e-5 if dtype == torch.float32 else 2e-2
        assert float((a.float().cpu() - r).abs().max()) <= tol * scale, (name, float((a.float().cpu() - r).abs().max()), scale)


@pytest.mark.gpu
def test_block_with_stacked_taps_equals_accumulated_taps_gpu(monkeypatch):
    blk = _block_pair().cuda()
    x = torch.randn(2, 8, 6, 5, 7, device='cuda')
    latent = torch.randn(2, 16, 6, device='cuda')
    a = _run_block(blk, x, latent, False, 0.9, monkeypatch)
    b = _run_block(blk, x, latent, True, 0.9, monkeypatch)
    torch.testing.assert_close(b[0], a[0], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(b[1], a[1], rtol=1e-3, atol=1e-4)
    for k in a[2]:
        torch.testing.assert_close(b[2][k], a[2][k], rtol=5e-3, atol=1e-4, msg=k)
