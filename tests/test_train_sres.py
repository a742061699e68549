"""Super-resolution training step (lvg.train_sres.SuperResTrainer): generator update through the
ADA-augmented discriminator, discriminator update, R1 (double backward through ADA's upfirdn2d /
grid_sample and the discriminator's resampling convs), ADA probability control and the generator
EMA, on synthetic clips. CPU run = plain-PyTorch ops at reduced width; GPU run = HIP kernels."""

import pytest
import torch

from helpers.ada_cfg import TRAIN_SRES_KW

from lvg.train_sres import SuperResTrainer

SMALL = dict(seq_length=2, temporal_context=1, lr_height=9, lr_width=16, hr_height=36, hr_width=64,
             G_kwargs=dict(latent_z_dim=32, latent_w_dim=48, channel_base=1024, channel_max=24, num_fp16_res=2),
             D_kwargs=dict(channels_base=1024, channels_max=32, num_fp16_res=0),
             augment_kwargs=TRAIN_SRES_KW, augment_p_init=0.3, overlap_grad_sync=False)


def _step(device, dtype):
    torch.manual_seed(0)
    tr = SuperResTrainer(device=device, compute_dtype=dtype, **SMALL)
    g0 = [p.detach().clone() for p in tr.G.parameters()]
    d0 = [p.detach().clone() for p in tr.D.parameters()]
    lr = torch.rand(2, 3, 4, 9, 16, device=device) * 2 - 1
    hr = torch.rand(2, 3, 2, 36, 64, device=device) * 2 - 1
    p0 = float(tr.augment.p)
    tr.train_step(step=0, lr_video=lr, hr_video=hr, r1_interval=16, ada_interval=4)
    for p in list(tr.G.parameters()) + list(tr.D.parameters()):
        assert torch.isfinite(p).all()
    assert sum(int(not torch.equal(a, b)) for a, b in zip(g0, tr.G.parameters())) > len(g0) // 2
    assert sum(int(not torch.equal(a, b)) for a, b in zip(d0, tr.D.parameters())) > len(d0) // 2
    assert abs(abs(float(tr.augment.p) - p0) - 4 * tr.augment_p_update_rate) < 1e-7      # moved by rate * ada_interval
    mags = [float(l.magnitude_ema) for l in tr.G.SG3.synthesis.layers()]
    assert any(m != 1.0 for m in mags)                                                  # update_D tracked the magnitudes
    ema = dict(tr.G_ema.named_parameters())
    name, p = next(iter(tr.G.named_parameters()))
    assert not torch.equal(ema[name], p)                                                # EMA lags the updated weights
    return tr


def test_train_step_cpu():
    torch.set_num_threads(8)
    tr = _step('cpu', torch.float32)
    # ADA control law: real logits mostly positive -> p rises, mostly negative -> p falls, clamped to [0, p_max]
    tr.augment.p.fill_(0.4999)
    tr._real_sign_sum.copy_(torch.tensor([8.0, 8.0]))
    tr.update_ada(gain=4)
    assert float(tr.augment.p) == pytest.approx(0.5)
    tr._real_sign_sum.copy_(torch.tensor([-8.0, 8.0]))
    tr.update_ada(gain=4)
    assert float(tr.augment.p) == pytest.approx(0.5 - 4 * 0.000125)
    p = float(tr.augment.p)
    tr.update_ada(gain=4)                                                               # nothing collected: unchanged
    assert float(tr.augment.p) == p


def test_checkpoint_resumes_the_ada_state_cpu(tmp_path):
    """A resumed super-resolution run continues with the adapted ADA probability, the conditioning-side pipe and the real-sign
    statistics collected since the last probability update (what the reference's ckpt() keeps, model/video_gan_sres.py
    `augment` / `in_augment` / `real_sign_collector`; round-2 advisor finding), and then takes the same next step."""
    from lvg import checkpoint
    torch.set_num_threads(8)
    torch.manual_seed(0)
    a = SuperResTrainer(device='cpu', compute_dtype=torch.float32, **SMALL)
    lr = torch.rand(2, 3, 4, 9, 16) * 2 - 1
    hr = torch.rand(2, 3, 2, 36, 64) * 2 - 1
    a.augment.p.fill_(0.1375)
    a.update_D(lr, lr, hr)                                              # collects real-sign statistics, moves D
    assert float(a._real_sign_sum[1]) > 0
    path = tmp_path / 'sres.pt'
    checkpoint.save_checkpoint(path, a, step=3)
    torch.manual_seed(9)
    b = SuperResTrainer(device='cpu', compute_dtype=torch.float32, **SMALL)      # different init, p = 0.3
    assert checkpoint.load_checkpoint(path, b) == 3
    assert float(b.augment.p) == float(a.augment.p) == pytest.approx(0.1375)
    assert torch.equal(a._real_sign_sum, b._real_sign_sum)
    for (k, x), (_, y) in zip(a.in_augment.state_dict().items(), b.in_augment.state_dict().items()):
        assert torch.equal(x, y), k
    for tr in (a, b):
        torch.manual_seed(4)
        tr.update_D(lr, lr, hr)
        tr.update_ada(gain=4)
    assert float(a.augment.p) == float(b.augment.p)
    for (k, x), (_, y) in zip(a.D.state_dict().items(), b.D.state_dict().items()):
        assert torch.equal(x, y), k


def test_run_D_applies_one_transform_to_both_clips_cpu():
    torch.manual_seed(1)
    tr = SuperResTrainer(device='cpu', compute_dtype=torch.float32, **dict(SMALL, lr_cond_prob=1.0, in_augment_strength=0.0))
    tr.augment.p.fill_(1.0)
    seen = {}
    orig = tr.D.forward
    tr.D.forward = lambda lr_up, hr: seen.update(lr=lr_up, hr=hr) or orig(lr_up, hr)
    hr = torch.rand(1, 3, 2, 36, 64) * 2 - 1
    lr = torch.nn.functional.avg_pool3d(hr, (1, 4, 4))
    tr.run_D(lr, tr.D.upsample(lr) * 0 + hr)                       # hr clip == a clip at hr size; lr its 4x box average
    # both halves came out of one AugmentPipe call: same geometry => the augmented lr stays the blur of the augmented hr
    a, b = seen['lr'], seen['hr']
    assert a.shape == b.shape == (1, 3, 2, 36, 64)
    corr = torch.corrcoef(torch.stack((a.flatten(), torch.nn.functional.avg_pool3d(b, (1, 5, 5), stride=1, padding=(0, 2, 2)).flatten())))[0, 1]
    assert float(corr) > 0.5


@pytest.mark.gpu
def test_train_step_gpu_fp16():
    _step('cuda', torch.float16)


@pytest.mark.gpu
def test_graph_mode_trains_like_eager_mode_gpu():
    """use_graphs=True replays the compute of update_G / update_D (and the fake generation) from hipGraphs. With every random draw off
    (no ADA, no conditioning jitter, no conditioning dropout, one fixed latent per batch size) the two modes compute the same step, and in
    FLOAT32 that step is deterministic enough to be the yardstick (measured at full size, tools/diag_graph_determinism.py,
    profiles/r05_graph_determinism_sres.log: generator gradients of two eager runs agree to 7e-8 of the largest, the discriminator's -- bias
    gradients summed with atomics -- to 1e-4, the running statistics to 2e-10; float16 runs differ by 1e-2 / 5e-2 from run to run):
      generator gradients  eager vs graph <= 2e-5 of the largest,  running statistics <= 1e-6,
      discriminator weights <= 1e-3; its BIAS gradients <= 6e-2: at this small size they are sums of cancelling terms accumulated with
      atomics -- b32.conv1.bias takes one of a few values from run to run, up to 2.4e-2 of the largest gradient apart, in eager mode as
      well (round-5 diagnostic, removed from tools/ in round 6) --, so the tight gates sit on the weights and on the generator side, which sees the
      discriminator's whole backward pass,
    the sign statistics of the real logits counted once; then the float16 configuration: finite after three steps, one graph per phase."""
    kw = dict(SMALL, augment_p_init=0.0, augment_real_sign_target=None, in_augment_strength=0.0, lr_cond_prob=1.0,
              G_grad_accum=2, D_grad_accum=2)
    lr = hr = None
    grads, signs, stats, trainers = {}, {}, {}, {}
    for name, use_graphs, dtype in (('eager', False, torch.float32), ('graph', True, torch.float32), ('graph16', True, torch.float16)):
        torch.manual_seed(0)
        tr = SuperResTrainer(device='cuda', compute_dtype=dtype, use_graphs=use_graphs, **kw)
        assert tr.use_graphs == use_graphs and tr.augment is None and tr.in_augment is None
        if lr is None:
            lr = torch.rand(4, 3, 4, 9, 16, device='cuda') * 2 - 1
            hr = torch.rand(4, 3, 2, 36, 64, device='cuda') * 2 - 1
        draw, fixed = tr.G.sample_latent_z, {}

        def same_z(batch_size, generator_z=None, draw=draw, fixed=fixed):      # (captured and eager execution number the device generator differently)
            if batch_size not in fixed:
                fixed[batch_size] = draw(batch_size, torch.Generator(device='cuda').manual_seed(7 + batch_size))
            return fixed[batch_size]
        tr.G.sample_latent_z = same_z
        torch.manual_seed(5)
        tr.train_step(step=1, lr_video=lr, hr_video=hr, r1_interval=0, ada_interval=0)
        grads[name] = (tr.G_sync.flat.clone(), tr.D_sync.flat.clone())
        signs[name] = tr._real_sign_sum.clone()
        stats[name] = torch.cat([b.float().flatten() for n, b in tr.G.named_buffers() if n.endswith('magnitude_ema') or n.endswith('w_avg')])
        for step in (2, 3):
            tr.train_step(step=step, lr_video=lr, hr_video=hr, r1_interval=0, ada_interval=0)
        trainers[name] = tr
    torch.cuda.synchronize()
    (eG, eD), (gG, gD) = grads['eager'], grads['graph']
    assert torch.isfinite(gG).all() and torch.isfinite(gD).all() and float(eG.abs().max()) > 0 and float(eD.abs().max()) > 0
    assert float((eG - gG).abs().max()) <= 2e-5 * float(eG.abs().max()), ('generator', float((eG - gG).abs().max()), float(eG.abs().max()))
    # discriminator: weights (well-conditioned sums) tight, biases (cancelling sums with atomics, see above) loose
    sync_e, sync_g = trainers['eager'].D_sync, trainers['graph'].D_sync
    is_bias = torch.zeros_like(eD, dtype=torch.bool)
    for (n, _), v in zip(trainers['eager'].D.named_parameters(), sync_e.views):
        if n.endswith('bias'):
            o = (v.data_ptr() - sync_e.flat.data_ptr()) // 4
            is_bias[o:o + v.numel()] = True
    dD, mD = (eD - gD).abs(), float(eD.abs().max())
    assert float(dD[~is_bias].max()) <= 1e-3 * mD, ('discriminator weights', float(dD[~is_bias].max()), mD)
    assert float(dD[is_bias].max()) <= 6e-2 * mD, ('discriminator biases', float(dD[is_bias].max()), mD)
    assert float((stats['eager'] - stats['graph']).abs().max()) <= 1e-6 and not torch.equal(stats['eager'], torch.ones_like(stats['eager']))
    assert float(signs['graph'][1]) == float(signs['graph16'][1]) == float(signs['eager'][1]) == 4.0          # four real logits counted once (the capture's warm-up is rolled back)
    for name in ('graph', 'graph16'):
        graph = trainers[name]
        for p in list(graph.G.parameters()) + list(graph.D.parameters()):
            assert torch.isfinite(p).all()
        assert {k[0] for k in graph._phase_graphs.graphs} == {'G', 'Dgen', 'D'}


def test_graph_mode_is_ignored_without_a_gpu_cpu():
    tr = SuperResTrainer(device='cpu', compute_dtype=torch.float32, use_graphs=True, **SMALL)
    assert tr.use_graphs is False


def test_step_body_matches_the_reference_trainer_on_stand_in_networks_cpu():
    """SuperResTrainer.update_G / update_D / update_r1 / update_ada / update_G_ema against the REFERENCE's SuperResVideoGAN methods
    (video_gan_sres.py:150-276) driving the stand-in networks of tests/helpers/stub_nets.py through three iterations of the loop of
    train_sres.py:241-264 (R1 and the ADA update on the first and third) from the same seed -- the reference side with the reference's
    AugmentPipe and its statistics collector, this side with lvg.ada_augment and the on-device sign sums: conditioning jitter on whole
    batches, micro-batch accumulation and gains, one ADA transform for the (upsampled lr, hr) pair, conditioning dropout, the three losses,
    gradient exchange semantics, Adam, the ADA probability control, the generator EMA -- and the order and number of every random draw.
    Fixture: tests/golden/make_golden_trainer_glue.py."""
    import numpy as np
    from conftest import load_golden
    from helpers.stub_nets import StubSresG, StubSresD
    from lvg import ddp
    from lvg.ada_augment import AugmentPipe
    from lvg.optim import FlatAdam
    g = load_golden('trainer_glue')
    spec = g['sres_spec']
    seq, ctx, batch = spec['seq_length'], spec['temporal_context'], spec['batch']
    tr = object.__new__(SuperResTrainer)
    tr.seq_length, tr.temporal_context, tr.context_seq_length, tr.channels = seq, ctx, seq + 2 * ctx, 3
    tr.lr_size, tr.hr_size, tr.device = (9, 16), (36, 64), torch.device('cpu')
    tr.G_grad_accum, tr.D_grad_accum, tr.G_magnitude_ema_beta = 2, 2, 0.999
    tr.G_ema_beta, tr.G_ema_warmup_steps, tr.r1_gamma, tr.lr_cond_prob = 0.99985, 25000, 1.0, spec['lr_cond_prob']
    tr.augment_p_max, tr.augment_p_update_rate, tr.augment_real_sign_target = 0.5, spec['augment_p_update_rate'], 0.6
    tr.G_lrate, tr.D_lrate, tr.G_warmup_steps, tr.D_warmup_steps = 0.003, 0.002, 0, 0
    tr.use_graphs, tr._static = False, {}
    tr.G, tr.G_ema, tr.D = StubSresG(ctx), StubSresG(ctx), StubSresD(seq)
    for net in (tr.G, tr.G_ema, tr.D):
        net.requires_grad_(False)
    tr.G_opt = FlatAdam(tr.G.parameters(), lr=0.003, betas=(0.0, 0.99), ema_params=tr.G_ema.parameters())
    tr.D_opt = FlatAdam(tr.D.parameters(), lr=0.002, betas=(0.0, 0.99))
    tr.G_sync = ddp.FlatGradSync(tr.G.parameters(), overlap=False)
    tr.D_sync = ddp.FlatGradSync(tr.D.parameters(), overlap=False)
    tr.augment = AugmentPipe(**TRAIN_SRES_KW).requires_grad_(False).train()
    tr.augment.p.fill_(spec['augment_p_init'])
    tr._real_sign_sum = torch.zeros(2)
    k = 8.0
    tr.in_augment = AugmentPipe(scale=1, scale_std=0.01 * k, rotate=1, rotate_max=0.002 * k, aniso=1, aniso_std=0.01 * k,
                                xfrac=1, xfrac_std=0.002 * k, noise=1, noise_std=0.01 * k).requires_grad_(False).train()
    tr.in_augment.p.fill_(0.5)
    gen = torch.Generator().manual_seed(spec['data_seed'])
    lr = torch.rand(batch, 3, seq + 2 * ctx, 9, 16, generator=gen) * 2 - 1
    hr = torch.rand(batch, 3, seq, 36, 64, generator=gen) * 2 - 1
    from torch_utils.ops import conv2d_gradfix, grid_sample_gradfix
    conv2d_gradfix.enabled = grid_sample_gradfix.enabled = True
    torch.manual_seed(spec['seed'])
    for step in range(spec['steps']):
        tr.train_step(step, lr, hr, r1_interval=spec['r1_interval'], ada_interval=spec['ada_interval'])
    after = torch.rand(3).numpy()
    assert np.array_equal(after, g['sres_next_rand'])                    # the same number of draws from the shared generator
    assert abs(float(tr.augment.p) - float(g['sres_augment_p'])) < 1e-7
    for net_name in ('G', 'D', 'G_ema'):
        net = getattr(tr, net_name)
        for n, t in list(net.named_parameters()) + list(net.named_buffers()):
            want = g[f'sres_{net_name}_{n}']
            got = t.detach().double().numpy()
            assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (net_name, n, float(np.abs(got - want).max()))


@pytest.mark.gpu
def test_graph_mode_with_ada_pipelines_gpu():
    """The default super-resolution configuration inside the captured phases: the discriminator-side ADA pipeline (p > 0,
    adapted from the real logits' signs) and the conditioning augmentation both draw on the device and update no buffer the capture's
    roll-back does not cover: after the first step (eager warm-up, roll-back, capture, replay) the sign statistic counts every real logit
    ONCE, the networks' and the pipelines' buffers equal those of an eager trainer wherever no random draw enters, everything stays
    finite over three steps, and the ADA update moves p."""
    kw = dict(SMALL, augment_p_init=0.3, G_grad_accum=2, D_grad_accum=2)
    lr = hr = None
    out = {}
    for name, use_graphs in (('eager', False), ('graph', True)):
        torch.manual_seed(0)
        tr = SuperResTrainer(device='cuda', compute_dtype=torch.float16, use_graphs=use_graphs, **kw)
        assert tr.use_graphs == use_graphs and tr.augment is not None and tr.in_augment is not None
        if lr is None:
            lr = torch.rand(4, 3, 4, 9, 16, device='cuda') * 2 - 1
            hr = torch.rand(4, 3, 2, 36, 64, device='cuda') * 2 - 1
        torch.manual_seed(5)
        tr.update_lrates(1)
        tr.update_G(lr, ema_step=1)
        tr.update_D(lr, lr, hr)
        signs = tr._real_sign_sum.clone()
        aug_buffers = {n: b.clone() for n, b in list(tr.augment.named_buffers()) + [('in.' + n, b) for n, b in tr.in_augment.named_buffers()]}
        p0 = float(tr.augment.p)
        tr.update_ada(gain=4)
        for step in (2, 3):
            tr.train_step(step=step, lr_video=lr, hr_video=hr, r1_interval=0, ada_interval=4)
        torch.cuda.synchronize()
        for p in list(tr.G.parameters()) + list(tr.D.parameters()) + list(tr.G.buffers()) + list(tr.D.buffers()):
            assert torch.isfinite(p).all()
        out[name] = (signs, aug_buffers, p0, float(tr.augment.p))
    for name, (signs, aug_buffers, p0, p1) in out.items():
        assert float(signs[1]) == 4.0 and abs(float(signs[0])) <= 4.0, (name, signs)          # four real logits, each counted once
        assert p0 == pytest.approx(0.3) and p1 != p0, (name, p0, p1)
    for n, b in out['eager'][1].items():                                                        # the pipelines' constants are untouched by the capture
        assert torch.equal(b, out['graph'][1][n]), n


@pytest.mark.gpu
def test_r1_phase_from_a_graph_equals_eager_gpu():
    """Round 6: in graph mode update_r1 is replayed from a hipGraph as well, with the discriminator's dense convolutions as nodes closed under
    differentiation (conv2d_gradfix.closed_nodes). float32, every random draw off: the gradients left in the exchange buffer, eager vs
    graph (first call = warm-up + capture + replay, second call = replay with other inputs), and eager with the library's own
    second-order graph (closed nodes off) as the third party. Weights tight, biases loose (cancelling sums, see the test above)."""
    from lvg import train_sres
    kw = dict(SMALL, augment_p_init=0.0, augment_real_sign_target=None, in_augment_strength=0.0, lr_cond_prob=1.0, D_grad_accum=2)
    g = torch.Generator(device='cuda').manual_seed(11)
    data = [(torch.rand(4, 3, 2, 9, 16, device='cuda', generator=g) * 2 - 1, torch.rand(4, 3, 2, 36, 64, device='cuda', generator=g) * 2 - 1) for _ in range(2)]
    flats, trainers = {}, {}
    for name, use_graphs, closed in (('eager', False, True), ('graph', True, True), ('library', False, False)):
        torch.manual_seed(0)
        tr = SuperResTrainer(device='cuda', compute_dtype=torch.float32, use_graphs=use_graphs, **kw)
        train_sres.R1_CLOSED_NODES = closed
        try:
            flats[name] = []
            for lr, hr in data:
                tr.D_opt.lr = 0.0
                tr.update_r1(lr, hr, gain=16.0)
                flats[name].append(tr.D_sync.flat.clone())
        finally:
            train_sres.R1_CLOSED_NODES = True
        trainers[name] = tr
        if use_graphs:
            assert any(k[0] == 'R1' for k in tr._phase_graphs.graphs) and not tr._phase_graphs.eager_keys
    sync = trainers['eager'].D_sync
    is_bias = torch.zeros_like(sync.flat, dtype=torch.bool)
    for (n, _), v in zip(trainers['eager'].D.named_parameters(), sync.views):
        if n.endswith('bias'):
            o = (v.data_ptr() - sync.flat.data_ptr()) // 4
            is_bias[o:o + v.numel()] = True
    for other in ('graph', 'library'):
        for e, o in zip(flats['eager'], flats[other]):
            m = float(e.abs().max())
            assert torch.isfinite(o).all() and m > 0
            d = (e - o).abs()
            assert float(d[~is_bias].max()) <= 1e-3 * m, (other, 'weights', float(d[~is_bias].max()), m)
            assert float(d[is_bias].max()) <= 6e-2 * m, (other, 'biases', float(d[is_bias].max()), m)
