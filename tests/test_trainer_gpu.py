"""One full iteration of the low-resolution training step on the GPU (update_G, update_D, R1 with
double backward through the HIP ops and the frames-layout convolutions, G-EMA), on synthetic video."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_step_runs_and_updates_parameters():
    from lvg.train_lres import LowResTrainer
    torch.manual_seed(0)
    tr = LowResTrainer(seq_length=8, height=36, width=64, device='cuda', compute_dtype=torch.float32,
                       G_grad_accum=1, D_grad_accum=1, overlap_grad_sync=False, with_ema=True, temp_scale_augment=1.0)
    g_before = [p.detach().clone() for p in list(tr.G.parameters())[:4]]
    d_before = [p.detach().clone() for p in list(tr.D.parameters())[:4]]
    real = torch.rand(1, 3, 8, 36, 64, device='cuda') * 2 - 1
    tr.train_step(step=0, real_video=real, r1_interval=16)       # step 0 includes the R1 update
    torch.cuda.synchronize()
    for p in list(tr.G.parameters()) + list(tr.D.parameters()):
        assert torch.isfinite(p).all()
    assert any(not torch.equal(a, b) for a, b in zip(g_before, list(tr.G.parameters())[:4]))
    assert any(not torch.equal(a, b) for a, b in zip(d_before, list(tr.D.parameters())[:4]))
    ema_mag = [b for n, b in tr.G.named_buffers() if n.endswith('magnitude_ema')]
    assert any(float(b) != 1.0 for b in ema_mag)                  # update_D ran G with beta = 0.999
    # gradients live in the persistent flat buffers
    assert tr.G_sync.flat.numel() >= sum(p.numel() for p in tr.G.parameters())      # (slices are padded to 16-byte boundaries)


def test_update_r1_gradients_match_reference_golden():
    """LowResTrainer.update_r1 with the augmentations off is deterministic: the gradients it leaves in the flat
    exchange buffer must be the reference's R1 gradients (tests/golden/make_golden_models_full.py, gamma = 1)."""
    import numpy as np
    from conftest import load_golden
    from helpers.named_fill import fill_named
    from lvg.train_lres import LowResTrainer
    g = load_golden('lres_models_full')
    tr = LowResTrainer(seq_length=16, height=36, width=64, device='cuda', compute_dtype=torch.float32, r1_gamma=1.0,
                       D_grad_accum=1, overlap_grad_sync=False, with_ema=False, temp_scale_augment=0.0, diffaug_policy='')
    fill_named(tr.D)
    real = (torch.rand(2, 3, 16, 36, 64, generator=torch.Generator().manual_seed(int(g['r1_real_seed']))) * 2 - 1).cuda()
    before = {n: p.detach().clone() for n, p in tr.D.named_parameters()}
    tr.update_r1(real, gain=1.0)
    torch.cuda.synchronize()
    named = dict(tr.D.named_parameters())
    views = {n: v for (n, _), v in zip(tr.D.named_parameters(), tr.D_sync.views)}
    for key in [k for k in g if k.startswith('r1_g_') and k.endswith('_sample')]:
        stem = key[len('r1_g_'):-len('_sample')]
        (name,) = [n for n in named if n.replace('.', '_') == stem]
        flat = views[name].detach().cpu().numpy().reshape(-1)
        want = g[key]
        got = flat[:: max(1, flat.size // 4096)][:4096]
        assert np.abs(got - want).max() <= 5e-3 * np.abs(want).max(), (stem, float(np.abs(got - want).max()), float(np.abs(want).max()))
    # the parameter autograd never reached keeps grad None -> Adam must not have touched it (reference: zero_grad(set_to_none=True))
    unused = str(g['r1_params_without_grad']).split(',')
    for n in unused:
        assert named[n].grad is None and torch.equal(named[n].detach(), before[n]), n
    assert any(not torch.equal(before[n], p.detach()) for n, p in named.items() if n not in unused)


def test_graph_mode_trains_like_eager_mode():
    """use_graphs=True replays the compute of update_G / update_D from hipGraphs, with the host-side draws (crop offsets, temporal stretch)
    handed over through static buffers. With the device-side randomness taken out (fixed temporal noise per shape, no DiffAugment -- captured
    and eager execution number the device generator differently) both modes compute the same step, and in FLOAT32 the step is
    deterministic enough to be the yardstick (measured, tools/diag_graph_determinism.py, profiles/r05_graph_determinism.log: two eager
    runs agree to 1e-6 of the largest generator gradient; the discriminator's bias gradients are sums with atomics and differ by 1e-4 ..
    2.5e-4 of the largest gradient between two EAGER runs; the running magnitudes agree exactly):
      generator gradients   eager vs graph  <= 2e-5 of the largest gradient   (measured 2e-7 .. 1e-6)
      discriminator         eager vs graph  <= 2e-3                           (measured 4e-6 .. 4e-5; eager vs eager up to 2.5e-4)
      running magnitudes    equal to 1e-6   (ONE update per update_D: the eager warm-up before the capture is rolled back)
    A stale draw buffer, a wrong stretch index or a phase that ran twice / not at all is off by 1e-2 .. 1 on these scales. Then the same
    step in bfloat16 (the bench's arithmetic): 16-bit gradients differ by ~10 % of their maximum between two eager runs, so that part only
    checks what it can: finite after three steps, one graph per phase and micro-batch shape."""
    from lvg.train_lres import LowResTrainer
    kw = dict(seq_length=8, height=36, width=64, device='cuda', G_grad_accum=2, D_grad_accum=2,
              overlap_grad_sync=False, with_ema=True, temp_scale_augment=1.0, diffaug_policy='')
    real = None
    grads, mags, trainers = {}, {}, {}
    for name, use_graphs, dtype in (('eager', False, torch.float32), ('graph', True, torch.float32), ('graph16', True, torch.bfloat16)):
        torch.manual_seed(0)
        tr = LowResTrainer(use_graphs=use_graphs, compute_dtype=dtype, **kw)
        assert tr.use_graphs == use_graphs
        if real is None:
            real = torch.rand(4, 3, 8, 36, 64, device='cuda') * 2 - 1
        draw, fixed = tr.G.sample_temporal_emb, {}

        def same_noise(batch, seq, generator=None, draw=draw, fixed=fixed):
            if (batch, seq) not in fixed:
                fixed[batch, seq] = draw(batch, seq, torch.Generator(device='cuda').manual_seed(100 * batch + seq))
            return fixed[batch, seq]
        tr.G.sample_temporal_emb = same_noise
        torch.manual_seed(5)
        tr.train_step(step=1, real_video=real, r1_interval=0)
        grads[name] = (tr.G_sync.flat.clone(), tr.D_sync.flat.clone())
        # (after ONE step: the statistics of the first update_D depend on the initial parameters only; later steps follow trajectories that
        # Adam with beta1 = 0 -- steps of +-lr by the gradient's sign -- drives apart from run to run)
        mags[name] = torch.stack([b.float().reshape(()) for n, b in tr.G.named_buffers() if n.endswith('magnitude_ema')])
        for step in (2, 3):
            tr.train_step(step=step, real_video=real, r1_interval=0)
        trainers[name] = tr
    torch.cuda.synchronize()
    for (e, g), gate, what in zip(zip(grads['eager'], grads['graph']), (2e-5, 2e-3), ('generator', 'discriminator')):
        assert torch.isfinite(g).all() and float(e.abs().max()) > 0
        assert float((e - g).abs().max()) <= gate * float(e.abs().max()), (what, float((e - g).abs().max()), float(e.abs().max()))
    moved = float((mags['eager'] - 1).abs().max())
    assert moved > 1e-4                                                 # one update at beta 0.999 ...
    assert float((mags['eager'] - mags['graph']).abs().max()) <= 1e-6, (mags['eager'], mags['graph'])   # ... not two, not none
    for name in ('graph', 'graph16'):
        graph = trainers[name]
        assert torch.isfinite(grads[name][0]).all() and torch.isfinite(grads[name][1]).all()
        for p in list(graph.G.parameters()) + list(graph.D.parameters()):
            assert torch.isfinite(p).all()
        assert {k[0] for k in graph._graphs if isinstance(k, tuple)} >= {'G', 'Dgen', 'D'}
