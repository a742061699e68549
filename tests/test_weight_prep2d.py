"""Weight side of the super-resolution generator's modulated convolution in one launch each way (csrc/weight_prep.hip,
lvg_weight_prep2d / lvg_weight_prep2d_backward): the packed 16-bit weights, the energy and both gradient paths against the
tensor expressions of model/generator_sres.py:50-58 evaluated in float64; the generator layer with and without it."""

import math

import pytest
import torch

from torch_utils.ops import conv2d_frames, weight_prep


def test_reference_expression_cpu():
    """_ref2d is the reference's normalisation (generator_sres.py:50-52) with the 1 / sqrt(fan_in) of the demodulated weight folded in."""
    torch.manual_seed(0)
    w = torch.randn(5, 7, 3, 3, dtype=torch.float64)
    wn, energy = weight_prep._ref2d(w, 0.25)
    want = w * w.square().mean([1, 2, 3], keepdim=True).rsqrt() * 0.25
    torch.testing.assert_close(wn, want)
    torch.testing.assert_close(energy, want.square().sum(dim=(2, 3)))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(64, 64), (512, 515), (61, 130), (3, 64)], ids=['aligned', 'wide_odd', 'ragged', 'few_rows'])
def test_forward_and_backward_vs_float64(shape, dtype):
    co, ci = shape
    dev = torch.device('cuda')
    torch.manual_seed(co * 1000 + ci)
    w = (torch.randn(co, ci, 3, 3, device=dev) * torch.rand(co, 1, 1, 1, device=dev).add(0.1)).requires_grad_(True)
    scale = 1.0 / math.sqrt(ci * 9)
    assert weight_prep.supported2d(w, dtype)
    prep = weight_prep.prepare2d(w, scale, dtype)
    co_pad, ci_pad = prep.wp.shape[2], prep.wp.shape[3]
    assert co_pad % 64 == 0 and ci_pad % 64 == 0 and co_pad >= co and ci_pad >= ci
    w64 = w.detach().double().requires_grad_(True)
    wn64, e64 = weight_prep._ref2d(w64, scale)
    # forward: the packed weights are the float32 expression rounded once to the 16-bit type; padding is zero
    want_wp = conv2d_frames.pack_weight(wn64.detach().float(), dtype, ci_pad, co_pad)
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    err = (prep.wp.double() - want_wp.double()).abs().max().item()
    assert err <= ulp * wn64.abs().max().item(), err                       # at most one unit in the last place of the largest element
    assert prep.wp[:, :, co:].abs().max().item() == 0 if co_pad > co else True
    assert prep.wp[:, :, :, ci:].abs().max().item() == 0 if ci_pad > ci else True
    want_wt = conv2d_frames.pack_weight_dgrad(prep.wp[:, :, :co, :ci].permute(2, 3, 0, 1), dtype, ci_pad, co_pad)
    assert torch.equal(prep.wt, want_wt)
    torch.testing.assert_close(prep.energy.double(), e64.detach(), rtol=1e-5, atol=1e-7)
    # backward: d L / d w for L = <g, w'> + <g_e, energy>, through the conv path (grad_from_conv) and the energy path (autograd)
    g = torch.randn(3, 3, co_pad, ci_pad, device=dev)
    g_e = torch.randn(co, ci, device=dev)
    (wn64.permute(2, 3, 0, 1) * g[:, :, :co, :ci].double()).sum().backward(retain_graph=True)
    want_conv = w64.grad.clone(); w64.grad = None
    (e64 * g_e.double()).sum().backward()
    want_energy = w64.grad.clone()
    got_conv = prep.grad_from_conv(g)
    (prep.energy * g_e).sum().backward()
    got_energy = w.grad
    for got, want in ((got_conv, want_conv), (got_energy, want_energy)):
        assert (got.double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_layer_with_and_without(dtype, monkeypatch):
    """One 16-bit 3 x 3 generator layer, forward and gradients: the one-launch weight side against the tensor expressions."""
    from lvg.models import sres
    dev = torch.device('cuda')
    torch.manual_seed(3)
    layer = sres.SynthesisLayer(w_dim=32, is_torgb=False, is_critically_sampled=False, use_fp16=True, in_channels=67, out_channels=64,
                                in_size=(36, 28), out_size=(36, 28), in_sampling_rate=16, out_sampling_rate=16, in_cutoff=4, out_cutoff=4,
                                in_half_width=3, out_half_width=3).to(dev)
    layer.compute_dtype = dtype
    x = torch.randn(2, 64, 28, 36, device=dev)
    cond = torch.randn(2, 3, 28, 36, device=dev)
    w = torch.randn(2, 32, device=dev)
    outs = {}
    for on in (False, True):
        monkeypatch.setattr(sres, 'WEIGHT_PREP', on)
        layer.zero_grad()
        xx = x.clone().requires_grad_(True)
        y = layer(xx.to(dtype), w, cond=cond.to(dtype))
        (y.float() * torch.linspace(-1, 1, y.numel(), device=dev).reshape(y.shape)).sum().backward()
        outs[on] = (y.float(), xx.grad, layer.weight.grad.clone(), layer.affine.weight.grad.clone())
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    for a, b in zip(outs[False], outs[True]):
        assert (a - b).abs().max().item() <= tol * b.abs().max().item()
