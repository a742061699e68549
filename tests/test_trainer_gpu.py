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
    deterministic enough to be the yardstick (measured, profiles/r05_graph_determinism.log: two eager
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


def _worker_graph_two_ranks(rank, world, port, out):
    """Two ranks on ONE device (gloo carries the device tensors through the host): LowResTrainer and SuperResTrainer with use_graphs=True.
    What is checked is the protocol at N > 1 with real captures: no collective ends up inside a captured phase (that aborts / hangs),
    the deferred statistics and the gradients are exchanged after the replays, and after two steps both ranks hold identical networks."""
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'long-video-gan_amd'))
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lvg.train_lres import LowResTrainer
        from lvg.train_sres import SuperResTrainer
        from helpers.ada_cfg import TRAIN_SRES_KW
        torch.cuda.set_device(0)

        def digest(nets):
            t = torch.cat([x.detach().flatten().double() for net in nets for x in list(net.parameters()) + list(net.buffers())])
            d = torch.stack([t.sum(), t.abs().sum(), t.square().sum()]).cpu()
            both = [torch.zeros_like(d) for _ in range(world)]
            dist.all_gather(both, d)
            assert torch.isfinite(d).all() and torch.equal(both[0], both[1]), f'ranks differ: {both}'

        torch.manual_seed(20 + rank)                                    # different init per rank: the broadcast must fix it
        tr = LowResTrainer(seq_length=8, device='cuda', compute_dtype=torch.bfloat16, use_graphs=True, with_ema=True)
        assert tr.use_graphs
        real = torch.rand(1, 3, 8, 36, 64, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3 + rank)) * 2 - 1
        for step in (1, 2):
            tr.train_step(step, real, r1_interval=2)                    # (step 2 runs R1: eager, exchange overlapped with its backward pass)
        emas = torch.stack([b.float().reshape(()) for n, b in tr.G.named_buffers() if n.endswith('magnitude_ema')])
        assert float((emas - 1).abs().max()) > 0                        # the deferred statistics were exchanged and applied
        digest((tr.G, tr.D, tr.G_ema))
        assert {k[0] for k in tr._graphs if isinstance(k, tuple)} >= {'G', 'Dgen', 'D'}
        del tr

        torch.manual_seed(40 + rank)
        ts = SuperResTrainer(device='cuda', compute_dtype=torch.float16, use_graphs=True, seq_length=2, temporal_context=1, lr_height=9, lr_width=16,
                             hr_height=36, hr_width=64, G_kwargs=dict(latent_z_dim=32, latent_w_dim=48, channel_base=1024, channel_max=24, num_fp16_res=2),
                             D_kwargs=dict(channels_base=1024, channels_max=32, num_fp16_res=0), augment_kwargs=TRAIN_SRES_KW, augment_p_init=0.3)
        g = torch.Generator(device='cuda').manual_seed(7 + rank)
        lr = torch.rand(2, 3, 4, 9, 16, device='cuda', generator=g) * 2 - 1
        hr = torch.rand(2, 3, 2, 36, 64, device='cuda', generator=g) * 2 - 1
        for step in (0, 1):
            ts.train_step(step=step, lr_video=lr, hr_video=hr, r1_interval=16, ada_interval=4)
        digest((ts.G, ts.D, ts.G_ema, ts.augment))
        out.put((rank, 'ok'))
    except Exception:
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_graph_trainers_two_ranks_on_one_device():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_graph_two_ranks, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == 'ok', f'rank {rank}: {msg}'


def test_flat_sync_assign_mode_inside_a_captured_step():
    """FlatGradSync.zero(assign=True) ... backward ... gather() captured into a hipGraph (bench.py's main step): every replay refills the flat
    buffer with that replay's gradients, equal to the eager accumulate-mode buffer bit for bit."""
    from lvg import ddp
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.Tanh(), torch.nn.Linear(96, 32)).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.Tanh(), torch.nn.Linear(96, 32)).cuda()
    ref.load_state_dict(net.state_dict())
    sync, sync_ref = ddp.FlatGradSync(net.parameters(), overlap=False), ddp.FlatGradSync(ref.parameters(), overlap=False)
    x = torch.zeros(16, 64, device='cuda')

    def compute():
        sync.zero(assign=True)
        net(x).square().sum().backward()
        sync.gather()

    x.normal_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        compute()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        compute()
    for _ in range(3):
        x.normal_()
        graph.replay()
        sync.finish()
        sync_ref.zero()
        ref(x).square().sum().backward()
        sync_ref.finish()
        torch.cuda.synchronize()
        assert torch.equal(sync.flat, sync_ref.flat)
        for p, v in zip(net.parameters(), sync.views):
            assert p.grad is v


def test_r1_phase_from_a_graph_equals_eager():
    """Round 6: in graph mode update_r1 (double backward through the discriminator, reference video_gan_lres.py:180-204) is replayed from a
    hipGraph like the other phases. float32, no DiffAugment (device-side draws are numbered differently under capture), temporal stretch
    on (host draws through the static buffers): the gradients the update leaves in the exchange buffer, eager vs graph -- first call (eager
    warm-up, rolled back, capture, replay) and a second call with other reals (replay only)."""
    from lvg.train_lres import LowResTrainer
    kw = dict(seq_length=8, height=36, width=64, device='cuda', D_grad_accum=2, overlap_grad_sync=False, with_ema=False,
              temp_scale_augment=1.0, diffaug_policy='', compute_dtype=torch.float32)
    reals = [torch.rand(4, 3, 8, 36, 64, device='cuda', generator=torch.Generator(device='cuda').manual_seed(s)) * 2 - 1 for s in (1, 2)]
    flats = {}
    for name, use_graphs in (('eager', False), ('graph', True)):
        torch.manual_seed(0)
        tr = LowResTrainer(use_graphs=use_graphs, **kw)
        torch.manual_seed(5)
        flats[name] = []
        for real in reals:
            tr.D_opt.lr = 0.0                                          # (the same parameters for the second call in both modes)
            tr.update_r1(real, gain=16.0)
            flats[name].append(tr.D_sync.flat.clone())
        if use_graphs:
            assert ('R1', 2) in tr._phase_graphs.graphs and not tr._phase_graphs.eager_keys
    for e, g in zip(flats['eager'], flats['graph']):
        assert torch.isfinite(g).all() and float(e.abs().max()) > 0
        assert float((e - g).abs().max()) <= 2e-3 * float(e.abs().max()), (float((e - g).abs().max()), float(e.abs().max()))
