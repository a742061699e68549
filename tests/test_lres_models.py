"""lvg.models.lres vs golden vectors produced by the REFERENCE generator/discriminator
(tests/golden/make_golden_models.py): same name-keyed weights, same injected noise; outputs,
logits and a set of parameter gradients. CPU run exercises the plain-PyTorch op definitions,
GPU run the HIP kernels. Tolerance: north star's 1e-3 (float32, outputs in [-1, 1])."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, record_measured
from helpers.named_fill import fill_named, analytic_buffers

from lvg.models.lres import VideoGenerator, VideoDiscriminator

T = 16


def _build(device):
    G = VideoGenerator()
    D = VideoDiscriminator(seq_length=T, max_edge=64)
    fill_named(G)
    fill_named(D)
    return G.to(device).requires_grad_(True), D.to(device).requires_grad_(True)


def _pre_activation_spy(records, pin_to=None, pinned=None, where=None):
    """Patch of lres._TapConvEpilogue._backward that also records z = ysum * pre + b (+ res) of every activated call (the argument of the
    leaky ReLU whose derivative the backward pass takes): returns the original to restore.

    `pin_to` (records of another run, same call order): where the sign of z differs from that run's, the saved sum handed to the backward
    pass is moved so that z takes the OTHER run's sign (magnitude 1e-4 of the layer's maximum) -- the leaky-ReLU derivative of those
    elements is then the one the other run used; nothing else changes. `pinned` collects |z_other| / max |z_other| of the moved elements."""
    from lvg.models import lres
    orig = lres._TapConvEpilogue._backward
    other = {}
    for key, z in (pin_to or []):
        other.setdefault(key, []).append(z)
    seen = {}

    def spy(ctx, x, weight, ysum, pre, b, res, post, dout, wt_packed, need):
        if ctx.cfg[2] != 'linear' and ysum is not None:
            scale = None if pre is None else pre.detach().double().reshape(ysum.shape[0], -1, 1, 1)
            shift = torch.zeros((), dtype=torch.float64, device=ysum.device)
            if b is not None:
                shift = shift + b.detach().double().reshape(1, -1, 1, 1)
            if res is not None:
                shift = shift + res.detach().double()
            z = ysum.detach().double() * (1.0 if scale is None else scale) + shift
            key = (tuple(x.shape), tuple(weight.shape))
            i = seen.get(key, 0)
            seen[key] = i + 1
            records.append((key, z.cpu()))
            if pin_to is not None and key in other and i < len(other[key]):
                zo = other[key][i].to(z.device)
                differs = zo.sign() != z.sign()
                if bool(differs.any()):
                    pinned.extend(float(v) for v in (zo[differs].abs() / zo.abs().max()))
                    if where is not None:       # which layer (activation shape [frames, channels, h, w]), how many elements, terms per bias-gradient sum
                        where.append(dict(x=list(key[0]), w=list(key[1]), flipped=int(differs.sum()), of=int(z.numel()),
                                          terms_per_channel_sum=int(z.numel() // z.shape[1])))
                    target = zo.sign() * 1e-4 * zo.abs().max()
                    want_sum = (target - shift) / (1.0 if scale is None else scale)
                    ysum = torch.where(differs, want_sum.to(ysum.dtype), ysum)
        return orig(ctx, x, weight, ysum, pre, b, res, post, dout, wt_packed, need)
    lres._TapConvEpilogue._backward = staticmethod(spy)
    return orig


def _forward_backward(device, g, records=None, pin_to=None, pinned=None, where=None):
    """One generator + discriminator pass on the golden inputs -> (features' rms, video, logits, loss, the seven parameter gradients)."""
    from lvg.models import lres
    G, D = _build(device)
    orig = _pre_activation_spy(records, pin_to, pinned, where) if records is not None else None
    try:
        noise = torch.tensor(g['noise'], device=device)
        emb = G.temporal_emb.blur(noise)
        ws = G.compute_latent_ws(emb, T)
        feats = G.synthesize_video(G._temporal_input(ws), ws, T, return_features=True)
        video = feats[-1]
        rms = np.array([float(f.detach().float().square().mean().sqrt()) for f in feats])
        logits = D(video)
        loss = F.softplus(-logits).mean()
        loss.backward()
    finally:
        if orig is not None:
            lres._TapConvEpilogue._backward = staticmethod(orig)
    pairs = dict(g_spatial_input=G.spatial_input, g_to_rgb_weight=G.to_rgb.weight, g_t0_bias_0=G.temporal_layers[0].bias_0,
                 g_s3_weight_1=G.spatial_layers[3].weight_1, g_map_l1_bias=G.latent_mapping.layer_1.bias,
                 d_b0_conv_vid_weight=D.blocks[0].conv_vid.weight, d_ep_linear_1_weight=D.epilogue.linear_1.weight)
    grads = {k: p.grad.detach().double().cpu().numpy() for k, p in pairs.items()}
    return (G, D), rms, video.detach().cpu().numpy(), logits.detach().cpu().numpy(), float(loss.detach()), grads


def _gradient_errors(grads, golden):
    return {k: float(np.abs(v - golden[k]).max() / (np.abs(golden[k]).max() + 1e-12)) for k, v in grads.items()}


def _run(device, rtol_grad):
    """Forward values and seven parameter gradients against the reference's float32 run (gate `rtol_grad` of each gradient's scale) and
    against the reference run in float64 (tests/golden/make_golden_models_f64.py; gate 1e-3), where the reference's own float32 run sits at
    5e-7 .. 1e-5. This repo's networks order the float32 arithmetic differently (modulation on the activations, demodulation on the output,
    fused epilogues) and sit at 1e-6 .. 5e-4 on the CPU.

    Leaky-ReLU kinks. The 16-frame networks evaluate ~2e7 leaky-ReLU arguments; the nearest to zero lie 4e-8 .. 5e-7 of their layer's maximum
    away from it, and two correct float32 evaluations agree in those arguments to 4e-6 .. 7e-6 only (13 824-term sums). An element that lands
    on the other side of zero changes that layer's derivative from 1 to 0.2: 9e-2 in the layer's gradients, 9e-3 in the gradient of the
    network input (profiles/r03_f32_kink_flips.log) -- a discontinuity of the function, not an error of either evaluation. No choice of seed
    removes it (the expected number of arguments within 1e-5 of zero is several hundred for ANY input of this size), so the fixture is kept
    and the EVALUATION is made independent of the kink instead: on the GPU the hand-written float32 route runs a second time
    with the sign of every leaky-ReLU argument pinned to the sign the library route of the same device computed (`_pre_activation_spy`,
    `pin_to`), and THAT run is held to the gates as they stand -- `rtol_grad` against the float32 golden, 1e-3 against the float64 one. The
    pinned elements must lie within 1e-5 of zero (relative to the layer's maximum); the library route must meet the same gates unpinned. The
    unpinned errors of the hand route are recorded for the record (profiles/r04_parity_measured.json)."""
    g = load_golden('lres_models')
    g64 = load_golden('lres_models_f64')
    records = [] if device != 'cpu' else None
    (G, D), rms, video, logits, loss, grads = _forward_backward(device, g, records)
    # analytic buffers (firwin designs, bilinear ramps) must be the same numbers as the reference's
    for prefix, net in (('G', G), ('D', D)):
        for name, buf in analytic_buffers(net).items():
            want = g[f'buf_{prefix}_{name}']
            got = np.array([float(buf.double().sum()), float(buf.double().abs().sum()), float(buf.numel())])
            np.testing.assert_allclose(got, want, rtol=1e-6, err_msg=name)
    np.testing.assert_allclose(rms, g['feat_rms'], rtol=1e-3)
    np.testing.assert_allclose(video, g['video'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(logits, g['logits'], rtol=1e-3, atol=1e-3)
    assert abs(loss - float(g['loss'])) < 1e-3
    vs32, vs64 = _gradient_errors(grads, g), _gradient_errors(grads, g64)
    record_measured(f'lres_T16_f32_grads_vs_reference_f64_{device}', **vs64)
    if device == 'cpu':
        assert max(vs32.values()) <= rtol_grad and max(vs64.values()) < 1e-3, (vs32, vs64)
        return
    from lvg.models import lres
    assert lres.SPLIT_F32, 'the hand-written float32 route is the one under test'
    lres.SPLIT_F32 = False
    try:
        lib_records = []
        _, _, _, _, _, lib_grads = _forward_backward(device, g, lib_records)
    finally:
        lres.SPLIT_F32 = True
    lib32, lib64 = _gradient_errors(lib_grads, g), _gradient_errors(lib_grads, g64)
    assert max(lib32.values()) <= rtol_grad and max(lib64.values()) < 1e-3, ('library route', lib32, lib64)
    pinned, pin_records, where = [], [], []
    _, _, pin_video, pin_logits, _, pin_grads = _forward_backward(device, g, pin_records, pin_to=lib_records, pinned=pinned, where=where)
    record_measured(f'lres_T16_f32_kink_layers_{device}', layers=where)
    pin32, pin64 = _gradient_errors(pin_grads, g), _gradient_errors(pin_grads, g64)
    record_measured(f'lres_T16_f32_kink_pinned_{device}', pinned_elements=len(pinned), largest_relative_distance_from_zero=max(pinned, default=0.0),
                    hand_route_unpinned_vs_f64=max(vs64.values()), hand_route_pinned_vs_f64=max(pin64.values()), hand_route_pinned_vs_f32=max(pin32.values()),
                    library_route_vs_f64=max(lib64.values()))
    assert max(pinned, default=0.0) < 1e-5, ('a pinned leaky-ReLU argument is not at the kink', sorted(pinned)[-4:])
    np.testing.assert_allclose(pin_video, g['video'], rtol=0, atol=1e-3)
    assert max(pin32.values()) <= rtol_grad and max(pin64.values()) < 1e-3, ('hand-written float32 route, kinks pinned', pin32, pin64, len(pinned))


def test_state_dict_keys_match_reference_layout():
    G = VideoGenerator()
    keys = set(G.state_dict().keys())
    for must in ('spatial_input', 'temporal_emb.blur_filters', 'temporal_emb.output_scale', 'latent_mapping.layer_0.weight',
                 'latent_mapping.layer_1.bias', 'temporal_downsample_latent.filter', 'w_to_temp_input.weight',
                 'temporal_layers.0.affine_0.weight', 'temporal_layers.0.weight_skip', 'temporal_layers.0.input_magnitude_ema_1.magnitude_ema',
                 'temporal_layers.1.spatial_upsample.filter', 'temporal_layers.4.temporal_upsample.filter',
                 'spatial_layers.3.bias_1', 'to_rgb.affine.bias', 'to_rgb.input_magnitude_ema.magnitude_ema'):
        assert must in keys, must
    assert sum(p.numel() for p in G.parameters()) == 83215939       # SURVEY.md 2.3 (measured on the reference)
    D = VideoDiscriminator(seq_length=128, max_edge=64)
    assert sum(p.numel() for p in D.parameters()) == 46424609
    dkeys = set(D.state_dict().keys())
    for must in ('blocks.0.conv_vid.weight', 'blocks.0.conv_vid._bias', 'blocks.1.conv_1.downsample._downsample_filter',
                 'blocks.3.conv_skip.weight', 'epilogue.conv1d_0.weight', 'epilogue.conv1d_3._bias', 'epilogue.linear_1.weight'):
        assert must in dkeys, must


def test_generator_discriminator_match_reference_cpu():
    torch.set_num_threads(8)
    _run('cpu', rtol_grad=2e-3)


@pytest.mark.gpu
def test_generator_discriminator_match_reference_gpu():
    _run('cuda', rtol_grad=5e-3)


@pytest.mark.gpu
def test_bf16_forward_close_to_fp32_gpu():
    """bfloat16 activations / contraction vs the float32 golden of the reference. SURVEY.md 7 measured the deviation of the
    reference's OWN bf16 run from its float32 run at 1.1e-2 max / 1.5e-3 mean (outputs in [-0.43, 0.26]). Gate: 1.5 x this repository's
    measured deviation (1.39e-2 max / 2.20e-3 mean, profiles/r05_parity_measured.json; the kernels are bit-reproducible, so the margin only
    has to cover other boxes' library builds): 2.1e-2 / 3.3e-3. Measured values are printed and recorded."""
    g = load_golden('lres_models')
    G, _ = _build('cuda')
    with torch.no_grad():
        ws = G.compute_latent_ws(G.temporal_emb.blur(torch.tensor(g['noise'], device='cuda')), T)
        video = G.synthesize_video(G._temporal_input(ws), ws, T, dtype=torch.bfloat16)
    err = np.abs(video.cpu().numpy() - g['video'])
    record_measured('lres_T16_bf16_video_vs_reference_f32', max_abs=err.max(), mean_abs=err.mean(), ref_range=np.abs(g['video']).max())
    assert err.max() < 2.1e-2 and err.mean() < 3.3e-3, (float(err.max()), float(err.mean()))


def _run_t128(device, dtype=None):
    """BASELINE.json configs[1] shape: 128-frame generator forward + discriminator logits vs the reference
    (tests/golden/make_golden_models_full.py; the stored video is float16: +-2.5e-4 of storage error)."""
    g = load_golden('lres_models_full')
    T128 = 128
    G = VideoGenerator()
    D = VideoDiscriminator(seq_length=T128, max_edge=64)
    fill_named(G)
    fill_named(D)
    G, D = G.to(device), D.to(device)
    noise = torch.randn(*[int(v) for v in g['t128_noise_shape']], generator=torch.Generator().manual_seed(int(g['t128_noise_seed'])))
    assert abs(float(noise.double().sum()) - float(g['t128_noise_sum'])) < 1e-6      # same CPU random stream as the reference run
    with torch.no_grad():
        ws = G.compute_latent_ws(G.temporal_emb.blur(noise.to(device)), T128)
        kw = {} if dtype is None else dict(dtype=dtype)
        video = G.synthesize_video(G._temporal_input(ws), ws, T128, **kw)
        logits = D(video.float(), **kw)
    want = g['t128_video'].astype(np.float32)
    assert tuple(video.shape) == want.shape == (1, 3, 128, 36, 64)
    return np.abs(video.float().cpu().numpy() - want), logits.float().cpu().numpy(), g


def test_t128_generator_matches_reference_cpu():
    torch.set_num_threads(8)
    err, logits, g = _run_t128('cpu')
    record_measured('lres_T128_f32_video_vs_reference_cpu', max_abs=err.max(), mean_abs=err.mean())
    assert err.max() <= 1e-3, float(err.max())      # north star: 1e-3, the golden's float16 storage error (2.5e-4) included; measured 2.5e-4
    np.testing.assert_allclose(logits, g['t128_logits'], rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_t128_generator_matches_reference_gpu():
    err, logits, g = _run_t128('cuda')
    record_measured('lres_T128_f32_video_vs_reference_cuda', max_abs=err.max(), mean_abs=err.mean())
    assert err.max() <= 1e-3, float(err.max())      # north star: 1e-3, the golden's float16 storage error (2.5e-4) included
    np.testing.assert_allclose(logits, g['t128_logits'], rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_t128_bf16_generator_gate_gpu():
    """configs[1] runs bf16 activations: the 128-frame bf16 video against the float32 reference video (range +-0.84). Gates: 1.5 x the
    measured deviation (1.56e-2 max / 2.08e-3 mean, logit 3.2e-3; profiles/r05_parity_measured.json): 2.4e-2 / 3.2e-3 / 5e-3
    (before: 4.3e-2 / 5.9e-3 / 0.1). Measured values are printed and recorded."""
    err, logits, g = _run_t128('cuda', dtype=torch.bfloat16)
    record_measured('lres_T128_bf16_video_vs_reference_f32', max_abs=err.max(), mean_abs=err.mean(), ref_range=np.abs(g['t128_video'].astype(np.float32)).max(),
                    logit_abs=abs(float(logits.reshape(-1)[0]) - float(g['t128_logits'].reshape(-1)[0])))
    assert err.max() < 2.4e-2 and err.mean() < 3.2e-3, (float(err.max()), float(err.mean()))
    assert abs(float(logits.reshape(-1)[0]) - float(g['t128_logits'].reshape(-1)[0])) < 5e-3


def _run_r1(device):
    """R1 penalty (video_gan_lres.py:184-194): logits, d logits / d video, penalty and the parameter gradients
    of the penalty (double backward through every op of the discriminator) vs the reference."""
    import lvg.models.lres as lres
    g = load_golden('lres_models_full')
    D = VideoDiscriminator(seq_length=T, max_edge=64)
    fill_named(D)
    D = D.to(device).requires_grad_(True)
    real = (torch.rand(2, 3, T, 36, 64, generator=torch.Generator().manual_seed(int(g['r1_real_seed']))) * 2 - 1).to(device).requires_grad_(True)
    with lres.second_order():
        logits = D(real)
    (r1_grad,) = torch.autograd.grad(outputs=[logits.sum()], inputs=[real], create_graph=True)
    penalty = r1_grad.square().sum(dim=(1, 2, 3, 4))
    (penalty * 0.5).mean().backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g['r1_logits'], rtol=1e-3, atol=1e-4)
    want = g['r1_input_grad']
    np.testing.assert_allclose(r1_grad.detach().cpu().numpy(), want, rtol=0, atol=2e-3 * float(np.abs(want).max()))
    np.testing.assert_allclose(penalty.detach().cpu().numpy(), g['r1_penalty'], rtol=5e-3)
    named = dict(D.named_parameters())
    without = sorted(k for k, p in named.items() if p.grad is None)
    assert ','.join(without) == str(g['r1_params_without_grad'])
    for key in [k for k in g if k.startswith('r1_g_') and k.endswith('_sample')]:
        stem = key[len('r1_g_'):-len('_sample')]
        name = [n for n in named if n.replace('.', '_') == stem]
        assert len(name) == 1, stem
        flat = named[name[0]].grad.detach().cpu().numpy().reshape(-1)
        norm = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
        want_norm = float(g['r1_g_' + stem + '_norm'])
        assert abs(norm - want_norm) <= 5e-3 * want_norm, (stem, norm, want_norm)
        sample = flat[:: max(1, flat.size // 4096)][:4096]
        want_s = g[key]
        assert np.abs(sample - want_s).max() <= 5e-3 * (np.abs(want_s).max() + 1e-30), (stem, float(np.abs(sample - want_s).max()), float(np.abs(want_s).max()))


def test_r1_penalty_gradients_match_reference_cpu():
    torch.set_num_threads(8)
    _run_r1('cpu')


@pytest.mark.gpu
def test_r1_penalty_gradients_match_reference_gpu():
    _run_r1('cuda')


def test_temporal_conv_frames_matches_conv3d_to_second_order():
    """The time-major decomposition (kt 2-D convs + hand-written backward) equals conv3d with zero
    padding, including the double backward that the R1 penalty needs."""
    import lvg.models.lres as lres
    torch.manual_seed(3)
    for kt, n, t in ((3, 2, 5), (5, 1, 4), (1, 2, 3), (5, 2, 2)):
        x5 = torch.randn(n, 4, t, 6, 7, dtype=torch.float64, requires_grad=True)
        w = torch.randn(5, 4, kt, 3, 3, dtype=torch.float64, requires_grad=True)
        ref = F.conv3d(x5, w, padding=(kt // 2, 1, 1))
        xf = lres.frames_from_video(x5)
        got = lres.video_from_frames(lres.temporal_conv_frames(lres._cl(xf), w, n, (1, 1)), n)
        torch.testing.assert_close(got, ref, rtol=1e-10, atol=1e-10)
        probe = torch.randn_like(ref)
        for out in (ref, got):
            out.backward(probe, retain_graph=True)
        gx_ref, gw_ref = torch.autograd.grad(ref, [x5, w], probe, create_graph=True)
        gx_got, gw_got = torch.autograd.grad(got, [x5, w], probe, create_graph=True)
        torch.testing.assert_close(gx_got, gx_ref, rtol=1e-9, atol=1e-9)
        torch.testing.assert_close(gw_got, gw_ref, rtol=1e-9, atol=1e-9)
        # R1-style: d/dw of |d out / d x|^2
        (pen_ref,) = torch.autograd.grad(gx_ref.square().sum(), w)
        (pen_got,) = torch.autograd.grad(gx_got.square().sum(), w)
        torch.testing.assert_close(pen_got, pen_ref, rtol=1e-8, atol=1e-8)


def test_crop_frames_backward_keeps_memory_format():
    import lvg.models.lres as lres
    x = torch.randn(6, 8, 5, 7, requires_grad=True)
    xc = x.contiguous(memory_format=torch.channels_last)
    y = lres.crop_frames(xc, n=2, seq_length=1, height=3, width=4)
    assert y.shape == (2, 8, 3, 4) and y.is_contiguous(memory_format=torch.channels_last)
    (g,) = torch.autograd.grad(y.sum(), x)
    ref = torch.zeros(6, 8, 5, 7)
    ref[2:4, :, 1:4, 1:5] = 1
    assert torch.equal(g, ref)


def test_frames_block_tracks_input_magnitude_like_the_layer_definition():
    """forward_frames (fused epilogues, statistic measured inside them) vs the literal NCTHW layer:
    same output and the same EMA buffers when magnitude_ema_beta < 1."""
    import copy
    from lvg.models.lres import Synthesis3dResBlock, ToRGB, frames_from_video, video_from_frames
    torch.manual_seed(0)
    blk = Synthesis3dResBlock(latent_dim=16, in_channels=8, out_channels=12, temporal_ksize=3, spatial_ksize=3, spatial_up=True)
    with torch.no_grad():
        blk.bias_0.normal_(0, 0.2); blk.bias_1.normal_(0, 0.2)
        blk.input_magnitude_ema_0.magnitude_ema.fill_(0.7); blk.input_magnitude_ema_1.magnitude_ema.fill_(1.6)
    blk2 = copy.deepcopy(blk)
    x = torch.randn(2, 8, 6, 5, 7)
    latent = torch.randn(2, 16, 6)
    want = blk(x, latent, 0.9)
    got = video_from_frames(blk2.forward_frames(frames_from_video(x), latent, 0.9), 2)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    for name in ('input_magnitude_ema_0', 'input_magnitude_ema_1'):
        torch.testing.assert_close(getattr(blk2, name).magnitude_ema, getattr(blk, name).magnitude_ema, rtol=1e-5, atol=1e-6)
    rgb = ToRGB(latent_dim=16, in_channels=8)
    rgb2 = copy.deepcopy(rgb)
    want = rgb(x, latent, 0.9)
    got = video_from_frames(rgb2.forward_frames(frames_from_video(x), latent, 0.9), 2)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rgb2.input_magnitude_ema.magnitude_ema, rgb.input_magnitude_ema.magnitude_ema, rtol=1e-5, atol=1e-6)


def test_block_boundary_plumbing_cpu():
    """Two generator blocks + ToRGB chained through `Modulated` (the block-final bias_act fused with the next layer's modulation,
    lres.FUSE_BOUNDARY) against the separate passes, float32 on the CPU (plain-PyTorch definitions of the ops): values and gradients."""
    from lvg.models import lres
    torch.manual_seed(1)
    n, t, L = 2, 4, 32
    a = lres.Synthesis3dResBlock(L, 16, 16, out_width=8, out_height=6, temporal_ksize=3, spatial_ksize=3)
    b = lres.Synthesis3dResBlock(L, 16, 8, out_width=8, out_height=6, temporal_ksize=1, spatial_ksize=3)
    rgb = lres.ToRGB(L, 8)
    x0 = torch.randn(t * n, 16, 6, 8)
    lat = [torch.randn(n, L, t) for _ in range(3)]
    params = [p for m in (a, b, rgb) for p in m.parameters()]

    def run(fused):
        x = x0.clone().requires_grad_(True)
        ta, tb, tr = a.frame_terms(lat[0], torch.float32), b.frame_terms(lat[1], torch.float32), rgb.frame_terms(lat[2], torch.float32)
        ka = dict(boundary=(tb[1], torch.float32, True, True)) if fused else {}
        kb = dict(boundary=(tr[1], torch.float32, True, False)) if fused else {}
        h = a.forward_frames(x, lat[0], 0.9, out_seq_length=t, dtype=torch.float32, terms=ta, **ka)
        assert isinstance(h, lres.Modulated) == fused
        h = b.forward_frames(h, lat[1], 0.9, out_seq_length=t, dtype=torch.float32, terms=tb, **kb)
        assert isinstance(h, lres.Modulated) == fused and (not fused or h.plain is None)
        y = rgb.forward_frames(h, lat[2], 0.9, dtype=torch.float32, terms=tr)
        grads = torch.autograd.grad((y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum(), [x] + params, allow_unused=True)
        emas = [float(m.magnitude_ema) for mod in (a, b, rgb) for m in mod.modules() if isinstance(m, lres.MagnitudeEMA)]
        return y.detach(), grads, emas

    def reset():
        for mod in (a, b, rgb):
            for m in mod.modules():
                if isinstance(m, lres.MagnitudeEMA):
                    m.magnitude_ema.fill_(1.0)
    reset(); y0, g0, e0 = run(False)
    reset(); y1, g1, e1 = run(True)
    assert float((y1 - y0).abs().max()) <= 1e-5 * float(y0.abs().max())
    assert max(abs(p - q) for p, q in zip(e0, e1)) < 1e-6
    for p, q in zip(g0, g1):
        assert (p is None) == (q is None)
        if p is not None:
            assert float((p - q).abs().max()) <= 2e-4 * (float(p.abs().max()) + 1e-12)
