"""Temporal noise filter bank (csrc/noise_bank.hip, torch_utils/ops/noise_bank.py; reference model/generator_lres.py:323-388, BlurredNoise).
CPU: the oracle (orc_noise_filter_bank) and this repo's BlurredNoise against outputs of the REFERENCE's BlurredNoise.blur
(tests/golden/make_golden_noise_bank.py), the packed operand form against its definition. GPU: the HIP kernel against the oracle on ragged
cases, at the generator's full size against the dense product, reproducibility, and that the generator really takes the kernel."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from torch_utils.ops import noise_bank as nb

CASES = ['norm1', 'norm0', 'linear_rates']


def _golden(name):
    g = np.load(os.path.join(GOLDEN, 'noise_bank.npz'))
    kw = eval(str(g[f'{name}/kwargs']))  # pylint: disable=eval-used  (a dict literal written by the golden script)
    return g, kw


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_cpu(oracle, name):
    g, kw = _golden(name)
    noise, bank, feat = g[f'{name}/noise'], g[f'{name}/bank'], g[f'{name}/features']
    n, c, length = noise.shape
    npf = float(g[f'{name}/normalize_per_filter'])
    scale = 1 + npf * (g[f'{name}/scale'].astype(np.float64) - 1) if npf > 0 else None
    out = oracle.noise_filter_bank(noise.reshape(n * c, length), bank, scale)          # [R, F, T]
    out = out.reshape(n, c * bank.shape[0], -1)
    np.testing.assert_allclose(out, feat, rtol=0, atol=2e-6 * np.abs(feat).max())


@pytest.mark.parametrize('name', CASES)
def test_module_matches_reference_cpu(name):
    from lvg.models.lres import BlurredNoise
    g, kw = _golden(name)
    mod = BlurredNoise(**kw)
    np.testing.assert_allclose(mod.blur_filters[:, 0, :].numpy(), g[f'{name}/bank'], rtol=0, atol=1e-7)      # the same analytic bank
    feat = mod.blur(torch.from_numpy(g[f'{name}/noise']))
    ref = g[f'{name}/features']
    np.testing.assert_allclose(feat.numpy(), ref, rtol=0, atol=5e-6 * np.abs(ref).max())


def test_packed_bank_layout_cpu():
    """bankP[(pairOff[g] + p), lane] = bank[32 g + lane % 32][K - 2 pairs[g] + 2 p + lane // 32]; every non-zero tap of the bank appears exactly once."""
    g, _ = _golden('norm1')
    bank = torch.from_numpy(g['norm1/bank'])                          # 40 filters: two groups, the second one partial
    f, k = bank.shape
    packed, off, max_pairs = nb.pack_bank(bank)
    off = off.numpy()
    pairs = np.diff(off)
    assert len(pairs) == 2 and all(p % 64 == 0 for p in pairs) and max_pairs == pairs.max() and packed.shape == (off[-1] + 8, 64)
    taps = k - (g['norm1/bank'] != 0).argmax(axis=1)
    for grp in range(2):
        assert 2 * pairs[grp] >= taps[grp * 32:(grp + 1) * 32].max() > 2 * (pairs[grp] - 64)
    total = 0.0
    for grp in range(2):
        for p in (0, 1, int(pairs[grp]) - 1):
            for lane in (0, 5, 31, 32, 40, 63):
                filt, tap = 32 * grp + lane % 32, k - 2 * int(pairs[grp]) + 2 * p + lane // 32
                want = float(bank[filt, tap]) if filt < f and 0 <= tap < k else 0.0
                assert float(packed[off[grp] + p, lane]) == want
        total += float(packed[off[grp]:off[grp + 1]].double().abs().sum())
    assert abs(total - float(bank.double().abs().sum())) < 1e-9


def _random_bank(f, k, seed, staircase=True):
    g = torch.Generator().manual_seed(seed)
    bank = torch.randn(f, k, generator=g)
    if staircase:
        for i in range(f):
            taps = max(1, int(k * (0.05 + 0.95 * (i + 1) / f)))
            bank[i, :k - taps] = 0
    return bank


# (rows, frames, filters, taps, staircase): ragged in every dimension -- odd row counts, frames not a multiple of 32 or 4, partial filter group,
# odd tap counts, a bank without zeros, groups shorter than one wave share
GPU_CASES = [(5, 37, 40, 301, True), (2, 64, 32, 64, True), (1, 3, 7, 33, True), (8, 100, 96, 517, False), (3, 640, 128, 1000, True)]


@pytest.mark.gpu
@pytest.mark.parametrize('case', GPU_CASES)
def test_hip_matches_oracle_gpu(oracle, case):
    rows, frames, filters, taps, stair = case
    bank = _random_bank(filters, taps, 1, stair)
    g = torch.Generator().manual_seed(2)
    noise = torch.randn(rows, frames + taps - 1, generator=g)
    scale = torch.rand(filters, generator=g) + 0.5
    want = oracle.noise_filter_bank(noise.numpy(), bank.numpy(), scale.numpy())
    dn, db = noise.cuda(), bank.cuda()
    assert nb.supported(dn, db)
    got = nb.noise_filter_bank(dn, db, nb.pack_bank(db), scale.cuda())
    err = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 2e-6, err                                            # float32 products, float32 accumulation over <= 1000 taps
    got2 = nb.noise_filter_bank(dn, db, nb.pack_bank(db), None)
    np.testing.assert_allclose(got2.cpu().numpy(), want / scale.numpy()[None, :, None], rtol=0, atol=2e-6 * np.abs(want).max() * 2)


@pytest.mark.gpu
def test_full_size_matches_dense_product_and_is_reproducible_gpu():
    """The generator's bank (128 filters x 5000 taps) at 8 clips x 640 frames: kernel == dense window product (the previous route), twice the same bits."""
    from lvg.models import lres
    mod = lres.BlurredNoise().cuda()
    g = torch.Generator(device='cuda').manual_seed(3)
    noise = torch.randn(8, mod.noise_channels, 640 + mod.kernel_size - 1, device='cuda', generator=g)
    assert lres.NOISE_BANK_HIP
    y = mod.blur(noise)
    y2 = mod.blur(noise)
    assert torch.equal(y, y2)
    lres.NOISE_BANK_HIP = False
    try:
        dense = mod.blur(noise)
    finally:
        lres.NOISE_BANK_HIP = True
    assert y.shape == dense.shape == (8, 1024, 640)
    err = float((y - dense).abs().max() / dense.abs().max())
    assert err < 2e-5, err                                            # two float32 summation orders over 5000 taps
    # float64 check of a few rows through the oracle's definition (full size on the CPU would take minutes)
    bank = mod.blur_filters[:, 0, :].double().cpu()
    row = noise[3, 5].double().cpu()
    scale = (1 + mod.normalize_per_filter * (mod.output_scale.reshape(-1).double().cpu() - 1))
    for f, t in ((0, 0), (17, 333), (127, 639), (64, 100)):
        want = float((row[t:t + mod.kernel_size] * bank[f]).sum() * scale[f])
        assert abs(float(y[3, 5 * 128 + f, t]) - want) < 2e-5 * float(dense.abs().max())


@pytest.mark.gpu
def test_generator_takes_the_kernel_gpu(monkeypatch):
    from lvg.models import lres
    calls = []
    real = nb.noise_filter_bank
    monkeypatch.setattr(lres.noise_bank, 'noise_filter_bank', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    mod = lres.BlurredNoise().cuda()
    out = mod(2, 16)
    assert out.shape == (2, 1024, 16) and len(calls) == 1


def test_packed_bank_is_repacked_in_place_cpu():
    """A version bump of the bank (what a trainer's buffer roll-back does) must not replace the packed tensors: captured graphs hold their addresses."""
    bank = _random_bank(40, 1024, 4)                                  # two groups; the first one needs 448 of the 512 tap pairs
    cache = nb.PackedBank()
    first = cache.get(bank)
    ptrs = (first[0].data_ptr(), first[1].data_ptr())
    assert cache.get(bank) is first
    bank.copy_(bank.clone())                                          # same values, new version
    again = cache.get(bank)
    assert again is first and (again[0].data_ptr(), again[1].data_ptr()) == ptrs
    bank[0, -1] += 1.0                                                # new values, same layout: visible through the same tensors
    changed = cache.get(bank)
    assert changed is first and float(changed[0].abs().sum()) != 0 and torch.equal(changed[0], nb.pack_bank(bank)[0])
    bank[0, 0] = 1.0                                                  # filter 0 now has K taps: another layout -> new tensors, old ones kept alive
    other = cache.get(bank)
    assert other is not first and cache._retired and cache._retired[0] is first


@pytest.mark.gpu
def test_graph_survives_a_repack_gpu():
    """Regression: capture a pass, bump the bank's version and run an eager pass (which repacks), replay -- the graph must still read valid memory."""
    from lvg.models import lres
    mod = lres.BlurredNoise().cuda()
    noise = torch.randn(1, mod.noise_channels, 64 + mod.kernel_size - 1, device='cuda')
    want = mod.blur(noise).clone()                                    # eager pass: packs
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out = mod.blur(noise)
    torch.cuda.current_stream().wait_stream(side)
    mod.blur_filters.copy_(mod.blur_filters.clone())                  # what a roll-back of the buffers does
    for _ in range(3):
        eager = mod.blur(noise)                                       # repacks (in place)
        junk = [torch.empty(1 << 20, device='cuda') for _ in range(8)]   # churn the allocator
        del junk
    torch.cuda.empty_cache()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(eager, want)
