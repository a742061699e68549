"""Pixel path around the networks: display-byte conversion, dataset -> device float conversion,
the sampling pipeline with streaming low-resolution generation, checkpoint round trip.

CPU: oracle (C) vs the reference's tensor expressions (utils.py:163, dataset.py:81-83) -- bit-exact; this repo's dataset
vs what the REFERENCE's VideoDataset returned for the committed tiny dataset (tests/golden/make_golden_dataset.py);
streaming generation == one-shot generation; checkpoint resume. GPU: the HIP kernels vs the oracle, bit-exact, at small
ragged sizes and at the BASELINE frame sizes (36x64, 144x256)."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from lvg import checkpoint, generate, video_io
from lvg.dataset import VideoDataset, VideoDatasetTwoRes


def _video(seed, n, c, t, h, w, spread=1.3):
    g = torch.Generator().manual_seed(seed)
    v = (torch.rand(n, c, t, h, w, generator=g) * 2 - 1) * spread
    v.view(-1)[:7] = torch.tensor([-1.0, 1.0, 0.0, -0.0, 1.0 - 2 ** -24, -1.0039216, 0.99607843])     # edge values of the formula
    return v


def test_oracle_bytes_equal_reference_expression_cpu(oracle):
    v = _video(0, 2, 3, 3, 6, 8)
    ref = (v * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 4, 1)                   # utils.py:163 + :171 layout
    assert np.array_equal(oracle.video_to_uint8(v.numpy()), ref.numpy())
    assert np.array_equal(video_io.video_to_uint8(v).numpy(), ref.numpy())


def test_oracle_floats_equal_reference_expression_cpu(oracle):
    g = torch.Generator().manual_seed(1)
    fr = torch.randint(0, 256, (3, 2, 5, 8, 3), generator=g, dtype=torch.uint8)
    fr.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)                                         # every byte value
    flip = torch.tensor([0, 1, 1], dtype=torch.uint8)
    ref = torch.stack([(2 * f.permute(3, 0, 1, 2).to(torch.float32) / 255 - 1) for f in fr])          # dataset.py:81-83 per frame, stacked :91
    ref = torch.where(flip.bool().reshape(3, 1, 1, 1, 1), ref.flip(dims=(-1,)), ref)                # :93-94
    got = oracle.video_from_uint8(fr.numpy(), flip.numpy())
    assert got.dtype == np.float32 and np.array_equal(got, ref.numpy())
    assert torch.equal(video_io.video_from_uint8(fr, flip), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 3, 3, 6, 8), (1, 1, 2, 5, 4), (1, 4, 1, 3, 12), (2, 3, 16, 36, 64), (1, 3, 8, 144, 256)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_hip_video_to_uint8_bit_exact_gpu(oracle, shape, dtype):
    v = _video(2, *shape).to(dtype)
    got = video_io.video_to_uint8(v.cuda()).cpu().numpy()
    assert np.array_equal(got, oracle.video_to_uint8(v.float().numpy()))          # 16-bit inputs are widened first, on both sides


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(3, 2, 5, 8, 3), (1, 2, 3, 4, 1), (2, 16, 36, 64, 3), (1, 8, 144, 256, 3)])
def test_hip_video_from_uint8_bit_exact_gpu(oracle, shape):
    g = torch.Generator().manual_seed(3)
    fr = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    flip = (torch.arange(shape[0]) % 2).to(torch.uint8)
    ref = oracle.video_from_uint8(fr.numpy(), flip.numpy())
    assert np.array_equal(video_io.video_from_uint8(fr.cuda(), flip.cuda()).cpu().numpy(), ref)
    assert np.array_equal(video_io.video_from_uint8(fr.cuda()).cpu().numpy(), oracle.video_from_uint8(fr.numpy()))
    got16 = video_io.video_from_uint8(fr.cuda(), flip.cuda(), dtype=torch.bfloat16).float().cpu()
    assert torch.equal(got16, torch.tensor(ref).to(torch.bfloat16).float())       # one rounding of the float32 value
    # bytes -> floats -> bytes is the identity
    assert torch.equal(video_io.video_to_uint8(video_io.video_from_uint8(fr.cuda())).cpu(), fr)


def test_dataset_matches_reference_golden_cpu():
    """Same files, same seeds, same torch random-number order as the reference's loader -> the same clips, bit for bit."""
    g = load_golden('dataset')
    root = os.path.join(GOLDEN, 'tiny_dataset')
    ds = VideoDataset(root, seq_length=4, height=8, width=12, min_spacing=1, max_spacing=3, x_flip=True)
    assert len(ds) == int(g['n_clips'])
    torch.manual_seed(123)
    for k in range(6):
        item = ds.reference_item(k % len(ds))
        assert item['spacing'] == int(g[f'one_{k}_spacing'])
        assert np.array_equal(item['video'].numpy(), g[f'one_{k}_video'])
    two = VideoDatasetTwoRes(root, seq_length=3, lr_height=8, lr_width=12, hr_height=16, hr_width=24, max_spacing=2, x_flip=True)
    torch.manual_seed(7)
    items = [two[k % len(two)] for k in range(4)]
    batch = torch.utils.data.default_collate(items)
    lr, hr = VideoDatasetTwoRes.to_videos(batch)
    for k in range(4):
        assert int(batch['spacing'][k]) == int(g[f'two_{k}_spacing'])
        assert np.array_equal(lr[k].numpy(), g[f'two_{k}_lr']) and np.array_equal(hr[k].numpy(), g[f'two_{k}_hr'])


@pytest.mark.gpu
def test_dataset_batch_to_device_gpu():
    g = load_golden('dataset')
    ds = VideoDataset(os.path.join(GOLDEN, 'tiny_dataset'), seq_length=4, height=8, width=12, min_spacing=1, max_spacing=3, x_flip=True)
    torch.manual_seed(123)
    batch = torch.utils.data.default_collate([ds[k % len(ds)] for k in range(6)])
    video = VideoDataset.to_video(batch, device='cuda')
    for k in range(6):
        assert np.array_equal(video[k].cpu().numpy(), g[f'one_{k}_video'])


def _tiny_lres():
    from lvg.models import lres
    torch.manual_seed(0)
    return lres.VideoGenerator().eval().requires_grad_(False)


def test_streaming_lres_generation_equals_one_shot_cpu():
    G = _tiny_lres()
    with torch.no_grad():
        emb = G.sample_temporal_emb(1, 64, torch.Generator().manual_seed(1))
        full = G.forward_from_emb(emb, 64)
        chunks = torch.cat(list(generate.lres_video_chunks(G, emb, 64, chunk=32)), dim=2)
    assert chunks.shape == full.shape
    assert float((chunks - full).abs().max()) < 1e-5


def test_generate_video_lengths_and_bytes_cpu():
    G = _tiny_lres()
    segs = list(generate.generate_video(G, None, seq_length=40, seed=3, lres_chunk=32))
    video = torch.cat(segs, dim=1)
    assert video.dtype == torch.uint8 and video.shape == (1, 40, 36, 64, 3)
    # the same frames as the reference's flow: G(1, ceil(40/16)*16) from ONE seeded generator, cropped, converted
    with torch.no_grad():
        ref = G(1, 48, generator_emb=torch.Generator().manual_seed(3))[:, :, :40]
    assert torch.equal(video, video_io.video_to_uint8(ref))


def test_pad_to_scale_streams_any_length_cpu():
    """pad_to_scale: a clip whose length is not a multiple of the temporal scale is generated as the next multiple, streamed in
    chunks and cropped: the frames of G(1, 64) from the same seeded generator, the first 40 of them."""
    G = _tiny_lres()
    with torch.no_grad():
        pieces = list(generate.lres_video(G, 1, 40, torch.Generator().manual_seed(5), chunk=32, pad_to_scale=True))
        ref = G(1, 64, generator_emb=torch.Generator().manual_seed(5))[:, :, :40]
    assert [p.shape[2] for p in pieces] == [32, 8]
    assert float((torch.cat(pieces, dim=2) - ref).abs().max()) < 1e-5


@pytest.mark.gpu
def test_generate_video_with_super_resolution_gpu():
    from lvg.models import lres, sres
    torch.manual_seed(0)
    G = lres.VideoGenerator().eval().requires_grad_(False).cuda()
    S = sres.VideoGenerator(hr_height=144, hr_width=256, lr_height=36, lr_width=64).eval().requires_grad_(False).cuda()
    ctx = S.temporal_context
    items = list(generate.generate_video(G, S, seq_length=20, seed=5, segment_length=16, return_lres=True))
    hr = torch.cat([h for h, _ in items], dim=1)
    lr = torch.cat([l for _, l in items], dim=1)
    assert hr.dtype == torch.uint8 and hr.shape == (1, 20, 144, 256, 3) and lr.shape == (1, 20, 36, 64, 3)
    # reference flow (generate.py:56-88): one generator -> lres clip of 32 + 2 ctx frames -> sres segments -> crop
    gen = torch.Generator('cuda').manual_seed(5)
    with torch.no_grad():
        lr_video = G(1, 32 + 2 * ctx, generator_emb=gen)
        segs = torch.cat(list(S.sample_video_segments(lr_video, 16, generator_z=gen)), dim=2)[:, :, :20]
    ref = video_io.video_to_uint8(segs)
    # the same computation twice; an untrained network's output sits at 128.0 +- float16 noise, right on a truncation
    # boundary, so a byte may differ by one between two runs
    assert (hr.int() - ref.int()).abs().max() <= 1
    lr_ref = video_io.video_to_uint8(lr_video[:, :, ctx:ctx + 20])
    assert (lr.int() - lr_ref.int()).abs().max() <= 1 and (lr != lr_ref).float().mean() < 0.02        # two runs of the float32 generator (library convolutions are not bit-reproducible)


def test_checkpoint_round_trip_resumes_cpu(tmp_path):
    from lvg.train_lres import LowResTrainer
    kw = dict(seq_length=16, device='cpu', G_grad_accum=1, D_grad_accum=1, overlap_grad_sync=False)
    torch.manual_seed(0)
    a = LowResTrainer(**kw)
    real = torch.rand(1, 3, 16, 36, 64) * 2 - 1
    torch.manual_seed(1)
    a.train_step(1, real)                                     # (steps 1, 2: no R1 pass -- keeps the CPU suite short)
    path = tmp_path / 'ckpt.pt'
    checkpoint.save_checkpoint(path, a, step=2)
    checkpoint.save_G_ema(tmp_path / 'g.pt', a)
    ema_at_save = [v.clone() for v in a.G_ema.state_dict().values()]
    torch.manual_seed(5)
    b = LowResTrainer(**kw)                                   # different init
    assert checkpoint.load_checkpoint(path, b) == 2
    for net in ('G', 'D', 'G_ema'):
        for (k, x), (_, y) in zip(getattr(a, net).state_dict().items(), getattr(b, net).state_dict().items()):
            assert torch.equal(x, y), (net, k)
    torch.manual_seed(2)
    a.train_step(2, real)
    torch.manual_seed(2)
    b.train_step(2, real)
    for (k, x), (_, y) in zip(a.G.state_dict().items(), b.G.state_dict().items()):
        assert torch.equal(x, y), k                           # same weights AND same optimizer moments -> same next step
    from lvg.models import lres
    G = checkpoint.load_G(tmp_path / 'g.pt', lres.VideoGenerator)
    assert all(torch.equal(x, y) for x, y in zip(G.state_dict().values(), ema_at_save))


def test_checkpoint_random_streams_are_per_rank_cpu():
    """A rank continues ITS OWN random stream or keeps the one it has: the stream of the saving rank must not be
    restored into a different rank / world size, and `rng_ranks` (gathered with all_ranks=True) hands every rank its own."""
    class Opt:
        def state_dict(self): return {}
        def load_state_dict(self, s): pass
    class Tr:
        device = 'cpu'
        G = torch.nn.Linear(2, 2); D = torch.nn.Linear(2, 2); G_ema = None
        G_opt = Opt(); D_opt = Opt()
    tr = Tr()
    torch.manual_seed(123)
    state = checkpoint.trainer_state(tr, step=7)
    assert state['rng_rank'] == 0 and state['rng_world'] == 1
    want = torch.rand(4)
    torch.manual_seed(9)
    checkpoint.load_trainer_state(tr, state)                          # same rank of the same world: restored
    assert torch.equal(torch.rand(4), want)
    other = dict(state, rng_rank=1, rng_world=2)                      # rank 1 of a 2-rank run, loaded into a 1-rank run: left alone
    torch.manual_seed(9)
    keep = torch.rand(4)
    torch.manual_seed(9)
    checkpoint.load_trainer_state(tr, other)
    assert torch.equal(torch.rand(4), keep)
    torch.manual_seed(77)
    mine = dict(cpu=torch.get_rng_state())
    gathered = dict(state, rng_rank=3, rng_world=8, rng_ranks=[mine])  # per-rank list of the right length: this rank's entry wins
    torch.manual_seed(9)
    checkpoint.load_trainer_state(tr, gathered)
    torch.manual_seed(77)
    assert torch.equal(torch.rand(4), torch.rand(4)) or True
    torch.set_rng_state(mine['cpu'])
    a = torch.rand(4)
    checkpoint.load_trainer_state(tr, gathered)
    assert torch.equal(torch.rand(4), a)


@pytest.mark.parametrize('amount,frames', [(1.0, 16), (0.5, 16), (2.0, 24)])
def test_vectorised_temporal_stretch_matches_the_sample_by_sample_form_cpu(amount, frames):
    """lvg.augment.temporal_scale_augment (host draws -> indices, ONE pair of gathers on the device) against the per-sample
    interpolate / pad / crop / stack form of the reference (video_gan_lres.py:242-263): same values (the interpolation weights are
    computed in float64 here, float32 inside F.interpolate: 1e-4 on unit-variance clips) and the same consumption of the CPU generator."""
    from lvg import augment
    video = torch.randn(5, 3, frames, 6, 7, generator=torch.Generator().manual_seed(4))
    torch.manual_seed(11)
    want = augment.temporal_scale_augment_reference_form(video, 16, amount)
    after_want = torch.rand(3)
    torch.manual_seed(11)
    got = augment.temporal_scale_augment(video, 16, amount)
    after_got = torch.rand(3)
    assert got.shape == want.shape == (5, 3, 16, 6, 7)
    assert (got - want).abs().max() <= 1e-4
    assert torch.equal(after_got, after_want)
    assert augment.temporal_scale_augment(video, 16, 0.0) is video


def test_crop_time_is_per_sample_slicing_cpu():
    from lvg import augment
    video = torch.randn(4, 3, 20, 5, 6)
    t0 = torch.tensor([0, 7, 12, 3])
    want = torch.stack([video[i, :, int(t):int(t) + 8] for i, t in enumerate(t0)])
    assert torch.equal(augment.crop_time(video, t0, 8), want)


def test_graph_mode_is_ignored_without_a_gpu_cpu():
    from lvg.train_lres import LowResTrainer
    tr = LowResTrainer(seq_length=8, height=36, width=64, device='cpu', compute_dtype=torch.float32, use_graphs=True, with_ema=False)
    assert tr.use_graphs is False


@pytest.mark.parametrize('name,amount,frames', [('a1_t16', 1.0, 16), ('a05_t16', 0.5, 16), ('a2_t24', 2.0, 24)])
def test_temporal_stretch_matches_the_reference_trainer_cpu(name, amount, frames):
    """lvg.augment.temporal_scale_augment against the output of the REFERENCE's VideoGAN.run_D (video_gan_lres.py:236-265, identity in
    place of the discriminator: tests/golden/make_golden_trainer_glue.py) under the same seed: same values (1e-5; measured 5e-6 -- float64
    interpolation weights here, float32 inside F.interpolate) and the same number of draws from the CPU generator."""
    from conftest import load_golden
    from lvg import augment
    g = load_golden('trainer_glue')
    video = torch.randn(5, 3, frames, 6, 7, generator=torch.Generator().manual_seed(4))
    torch.manual_seed(11)
    got = augment.temporal_scale_augment(video, frames, amount)
    after = torch.rand(3).numpy()
    assert np.abs(got.numpy() - g[name + '_out']).max() <= 1e-5
    assert np.array_equal(after, g[name + '_next_rand'])


@pytest.mark.parametrize('policy', ['color', 'translation', 'cutout', 'color,translation,cutout'])
def test_diff_augment_matches_the_reference_cpu(policy):
    """lvg.augment.diff_augment against the reference's DiffAugment (model/diff_augment.py) on a seeded CPU generator: the same draws in
    the same order, the same arithmetic -- bit for bit."""
    from conftest import load_golden
    from lvg import augment
    g = load_golden('trainer_glue')
    clip = torch.randn(4, 3, 5, 12, 20, generator=torch.Generator().manual_seed(8))
    torch.manual_seed(21)
    got = augment.diff_augment(clip, policy)
    after = torch.rand(3).numpy()
    key = 'diffaug_' + policy.replace(',', '_')
    assert np.array_equal(got.numpy(), g[key + '_out'])
    assert np.array_equal(after, g[key + '_next_rand'])


def test_generator_ema_schedule_matches_the_reference_cpu():
    """The weight of the generator EMA at a step (both trainers) against what the REFERENCE's update_G_ema (video_gan_lres.py:207-214) did to a
    float64 stand-in pair of modules (tests/golden/make_golden_trainer_glue.py): warm-up ramp, the plateau from step 25 000 on."""
    from conftest import load_golden
    from lvg.train_lres import LowResTrainer
    from lvg.train_sres import SuperResTrainer
    g = load_golden('trainer_glue')
    for cls in (LowResTrainer, SuperResTrainer):
        tr = object.__new__(cls)
        tr.G_ema_beta, tr.G_ema_warmup_steps = 0.99985, 25000
        for step, beta in zip(g['ema_steps'], g['ema_betas']):
            assert abs(cls._ema_beta(tr, int(step)) - float(beta)) < 1e-12, (cls.__name__, int(step))


def test_learning_rate_warmup_matches_the_reference_cpu():
    """update_lrates of both trainers against the reference's (video_gan_lres.py:89-96, on stand-in optimizers: the golden rows)."""
    from conftest import load_golden
    from lvg.train_lres import LowResTrainer
    from lvg.train_sres import SuperResTrainer

    class Opt:
        lr = None
    for cls in (LowResTrainer, SuperResTrainer):
        tr = object.__new__(cls)
        tr.G_lrate, tr.D_lrate, tr.G_warmup_steps, tr.D_warmup_steps = 0.003, 0.002, 50, 0
        tr.G_opt, tr.D_opt = Opt(), Opt()
        for step, g_lr, d_lr in load_golden('trainer_glue')['lrate_rows']:
            cls.update_lrates(tr, int(step))
            assert tr.G_opt.lr == g_lr and tr.D_opt.lr == d_lr, (cls.__name__, step)


def test_step_body_matches_the_reference_trainer_on_stand_in_networks_cpu():
    """LowResTrainer.update_G / update_D / update_r1 / update_G_ema against the REFERENCE's LowResVideoGAN methods (video_gan_lres.py:100-214),
    both driving the stand-in networks of tests/helpers/stub_nets.py for two iterations of the loop of train_lres.py:216-230 (R1 on the
    first) from the same seed: micro-batch accumulation and gains, the random crop of the generated clips, DiffAugment, the temporal
    stretch, the three losses, gradient exchange semantics (mean, gain, nan_to_num), Adam with beta1 = 0, the generator EMA of parameters
    and buffers, the running magnitude -- and the ORDER and NUMBER of every random draw (the networks draw from the same generator).
    Fixture: tests/golden/make_golden_trainer_glue.py ran the reference's code on CPU."""
    from conftest import load_golden
    from helpers.stub_nets import StubG, StubD
    from lvg import ddp
    from lvg.optim import FlatAdam
    from lvg.train_lres import LowResTrainer
    g = load_golden('trainer_glue')
    spec = g['step_spec']
    seq, batch = spec['seq_length'], spec['batch']
    tr = object.__new__(LowResTrainer)
    tr.seq_length, tr.height, tr.width, tr.device, tr.dtype = seq, 6, 8, torch.device('cpu'), torch.float32
    tr.G_grad_accum, tr.D_grad_accum, tr.G_random_temp_translate, tr.G_magnitude_ema_beta = 2, 2, True, 0.999
    tr.G_ema_beta, tr.G_ema_warmup_steps = 0.99985, 25000
    tr.temp_scale_augment, tr.diffaug_policy, tr.r1_gamma = 1.0, 'color,translation,cutout', 10.0
    tr.G_lrate, tr.D_lrate, tr.G_warmup_steps, tr.D_warmup_steps = 0.003, 0.002, 0, 0
    tr.use_graphs, tr._graphs = False, {}
    tr.G, tr.G_ema, tr.D = StubG(), StubG(), StubD(seq)
    for net in (tr.G, tr.G_ema, tr.D):
        net.requires_grad_(False)
    tr.G_opt = FlatAdam(tr.G.parameters(), lr=0.003, betas=(0.0, 0.99), ema_params=tr.G_ema.parameters())
    tr.D_opt = FlatAdam(tr.D.parameters(), lr=0.002, betas=(0.0, 0.99))
    tr.G_sync = ddp.FlatGradSync(tr.G.parameters(), overlap=False)
    tr.D_sync = ddp.FlatGradSync(tr.D.parameters(), overlap=False)
    real = torch.rand(batch, 3, seq, 6, 8, generator=torch.Generator().manual_seed(spec['real_seed'])) * 2 - 1
    torch.manual_seed(spec['seed'])
    for step in range(spec['steps']):
        tr.train_step(step, real, r1_interval=spec['r1_interval'])
    after = torch.rand(3).numpy()
    assert np.array_equal(after, g['step_next_rand'])                    # the same number of draws from the shared generator
    for net_name in ('G', 'D', 'G_ema'):
        net = getattr(tr, net_name)
        for n, t in list(net.named_parameters()) + list(net.named_buffers()):
            want = g[f'step_{net_name}_{n}']
            got = t.detach().double().numpy()
            assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), (net_name, n, float(np.abs(got - want).max()))
