"""lvg.models.sres vs golden vectors produced by the REFERENCE super-resolution generator and
discriminator (tests/golden/make_golden_sres.py): same name-keyed weights and inputs; the
conditioning pyramid, the generated frames, the logits and a set of parameter gradients, plus the
layer schedule (sizes / factors / paddings / filter taps) of the full 256x144 configuration.
CPU run exercises the plain-PyTorch op definitions, GPU run the HIP kernels (fused
filtered_lrelu for every synthesis layer). Tolerance: north star's 1e-3 in float32."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from helpers.named_fill import fill_named, analytic_buffers
from helpers.sres_cfg import SMALL_G, SMALL_D, FULL_G, small_inputs, video_ramp

from lvg.models import sres


def _schedule(net):
    rows = []
    for l in net.synthesis.layers():
        rows.append([l.in_channels, l.out_channels, *map(int, l.in_size), *map(int, l.out_size), l.in_sampling_rate,
                     l.out_sampling_rate, l.up_factor, l.down_factor, l.up_taps, l.down_taps, *l.padding, int(l.use_fp16)])
    return np.array(rows, dtype=np.int64)


def _check_buffers(g, prefix, net):
    for name, buf in analytic_buffers(net).items():
        want = g[f'buf_{prefix}_{name}']
        got = np.array([float(buf.double().sum()), float(buf.double().abs().sum()), float(buf.numel())])
        np.testing.assert_allclose(got, want, rtol=1e-6, err_msg=name)


def _build(device):
    G = sres.Generator(**SMALL_G)
    D = sres.VideoDiscriminator(**SMALL_D)
    fill_named(G)
    fill_named(D)
    return G.to(device).requires_grad_(True), D.to(device).requires_grad_(True)


def _run(device, rtol_grad, force_fp32):
    g = load_golden('sres_models')
    G, D = _build(device)
    _check_buffers(g, 'G', G)
    _check_buffers(g, 'D', D)
    assert sorted(G.state_dict().keys()) == list(g['small_keys'])
    assert sorted(D.state_dict().keys()) == list(g['small_d_keys'])
    np.testing.assert_array_equal(_schedule(G), g['small_schedule'])

    z, lr_video = (t.to(device) for t in small_inputs())
    ctx = SMALL_G['cond_context']
    conds = G.prep_cond(lr_video)
    np.testing.assert_array_equal(np.array([list(c.shape) for c in conds]), g['cond_shapes'])
    np.testing.assert_allclose(np.array([float(c.square().mean().sqrt()) for c in conds]), g['cond_rms'], rtol=1e-4)
    np.testing.assert_allclose(conds[3].cpu().numpy(), g['cond_3'], atol=1e-5)
    np.testing.assert_allclose(conds[12].cpu().numpy(), g['cond_12'], atol=1e-5)

    kw = dict(force_fp32=True) if force_fp32 else {}
    video = G(z, lr_video, **kw)
    np.testing.assert_allclose(video.detach().cpu().numpy(), g['video'], rtol=0, atol=1e-3)
    logits = D(lr_video[:, :, ctx:-ctx], video)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g['logits'], rtol=1e-3, atol=1e-3)
    loss = F.softplus(-logits).mean() + (video * video_ramp(video)).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(g['loss'])) < 1e-3 * max(1.0, abs(float(g['loss'])))
    layers = G.synthesis.layers()
    first, mid, last = layers[0], layers[7], layers[-1]
    pairs = dict(g_first_weight=first.weight, g_mid_weight=mid.weight, g_mid_bias=mid.bias, g_mid_affine_weight=mid.affine.weight,
                 g_last_weight=last.weight, g_map_fc0_weight=G.mapping.fc0.weight,
                 d_b64_fromrgb_weight=D.b64.fromrgb.weight, d_b16_conv1_weight=D.b16.conv1.weight,
                 d_b16_skip_weight=D.b16.skip.weight, d_b4_fc_bias=D.b4.fc.bias)
    for key, param in pairs.items():
        want = g[key]
        got = param.grad.detach().cpu().numpy()
        scale = np.abs(want).max() + 1e-12
        assert np.abs(got - want).max() <= rtol_grad * scale, (key, float(np.abs(got - want).max()), float(scale))


def test_full_config_schedule_and_state_dict_match_reference():
    """BASELINE.json configs[3] (8 x 144x256 from 36x64): every layer's geometry, every analytic
    filter and every state_dict key / size must be the reference's."""
    g = load_golden('sres_models')
    net = sres.VideoGenerator(hr_height=144, hr_width=256, lr_height=36, lr_width=64)
    G = net.SG3
    np.testing.assert_array_equal(_schedule(G), g['full_schedule'])
    assert G.synthesis.layer_names == list(g['full_names'])
    assert sorted(G.state_dict().keys()) == list(g['full_keys'])
    scales = [getattr(r, 'scale', 1) * (-1 if isinstance(r, sres.KaiserDownsample) else 1) for r in G.resamples]
    assert scales == list(g['full_resample_scales'])
    _check_buffers(g, 'F', G)
    D = sres.VideoDiscriminator(seq_length=8, lr_height=36, lr_width=64, hr_height=144, hr_width=256)
    sd = D.state_dict()
    assert sorted(sd.keys()) == list(g['full_d_keys'])
    assert [sd[k].numel() for k in sorted(sd.keys())] == list(g['full_d_shapes'])
    assert all(FULL_G[k] == getattr(G, k) for k in ('z_dim', 'w_dim', 'img_width', 'img_height', 'cond_context'))


def test_modulated_conv2d_matches_per_sample_weights():
    """Activation-side modulation == the textbook per-sample weight modulation (incl. gradients)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 5, 9, 11, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(7, 5, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    s = (1 + 0.3 * torch.randn(3, 5, generator=g, dtype=torch.float64)).requires_grad_(True)
    gain = torch.tensor(0.7, dtype=torch.float64)
    y = sres.modulated_conv2d(x, w, s, demodulate=True, padding=2, input_gain=gain)
    wn = w * w.square().mean(dim=(1, 2, 3), keepdim=True).rsqrt()
    sn = s * s.square().mean().rsqrt()
    wm = wn[None] * sn[:, None, :, None, None]
    wm = wm * (wm.square().sum(dim=(2, 3, 4), keepdim=True) + 1e-8).rsqrt() * gain
    want = torch.cat([F.conv2d(x[i:i + 1], wm[i], padding=2) for i in range(3)])
    torch.testing.assert_close(y, want, rtol=1e-10, atol=1e-10)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    got = torch.autograd.grad((y * gy).sum(), (x, w, s))
    ref = torch.autograd.grad((want * gy).sum(), (x, w, s))
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=1e-9, atol=1e-9)


def test_generator_discriminator_match_reference_cpu():
    torch.set_num_threads(8)
    _run('cpu', rtol_grad=2e-3, force_fp32=True)


def test_sample_video_segments_shares_latent_cpu():
    torch.set_num_threads(8)
    net = sres.VideoGenerator(hr_height=36, hr_width=64, lr_height=9, lr_width=16, temporal_context=1, latent_z_dim=32,
                              latent_w_dim=48, channel_base=1024, channel_max=24, num_fp16_res=2)
    lr = torch.randn(1, 3, 6, 9, 16)
    with torch.no_grad():
        segs = list(net.sample_video_segments(lr, segment_length=2, generator_z=torch.Generator().manual_seed(5)))
        z = net.sample_latent_z(1, torch.Generator().manual_seed(5))
        whole = net.SG3(z, lr)
    assert len(segs) == 2 and segs[0].shape == (1, 3, 2, 36, 64)
    torch.testing.assert_close(torch.cat(segs, dim=2), whole, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_generator_discriminator_match_reference_gpu_fp32():
    _run('cuda', rtol_grad=5e-3, force_fp32=True)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_reduced_precision_forward_close_to_fp32_gpu(dtype):
    g = load_golden('sres_models')
    G, _ = _build('cuda')
    for layer in G.synthesis.layers():
        layer.compute_dtype = dtype
    z, lr_video = (t.cuda() for t in small_inputs())
    with torch.no_grad():
        video = G(z, lr_video)
    err = np.abs(video.cpu().numpy() - g['video'])
    tol = (2e-2, 2e-3) if dtype == torch.float16 else (1e-1, 1e-2)
    assert err.max() < tol[0] and err.mean() < tol[1], (float(err.max()), float(err.mean()))


@pytest.mark.gpu
def test_full_size_segment_forward_backward_gpu():
    """configs[3] at full size: 8 frames 144x256 from 36x64 (+-4 context), float16 layers on the
    fused kernels; checks shapes, finiteness and that the fused path equals the generic
    upfirdn2d -> act -> upfirdn2d path on the same weights."""
    from torch_utils.ops import filtered_lrelu as fl
    torch.manual_seed(0)
    net = sres.VideoGenerator(hr_height=144, hr_width=256, lr_height=36, lr_width=64).cuda()
    D = sres.VideoDiscriminator(seq_length=8, lr_height=36, lr_width=64, hr_height=144, hr_width=256).cuda()
    lr = torch.randn(1, 3, 16, 36, 64, device='cuda').clamp(-1, 1)
    z = net.sample_latent_z(1, torch.Generator(device='cuda').manual_seed(1))
    video = net(lr, latent_z=z)
    assert video.shape == (1, 3, 8, 144, 256) and torch.isfinite(video).all()
    logits = D(lr[:, :, 4:-4], video)
    assert logits.shape == (1, 1)
    F.softplus(-logits).mean().backward()
    grads = [p.grad for p in net.parameters()]
    assert all(gr is not None and torch.isfinite(gr).all() for gr in grads)
    for layer in net.SG3.synthesis.layers()[:-1]:
        assert fl._fused_supported(layer.up_filter, layer.down_filter, layer.up_factor, layer.down_factor, torch.float16), layer.padding
    with torch.no_grad():
        fused = net(lr, latent_z=z)
        fl.FORCE_GENERIC = True
        try:
            generic = net(lr, latent_z=z)
        finally:
            fl.FORCE_GENERIC = False
    err = (fused - generic).abs()
    assert float(err.max()) < 2e-2 and float(err.mean()) < 1e-3, (float(err.max()), float(err.mean()))
