"""HIP filtered_lrelu (fused kernel where available, generic path otherwise) vs the oracle and
the golden fixtures, including the sign-mask round trip used by the backward pass."""

import warnings

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

from torch_utils.ops import filtered_lrelu

DEV = 'cuda'
# float64: gain/slope/clamp cross the C ABI as float32 (as in the reference plugin), hence 1e-6.
TOL = {torch.float32: dict(rtol=5e-5, atol=5e-6), torch.float64: dict(rtol=2e-6, atol=2e-7),
       # 16-bit gates at about twice the measured worst case of the MFMA kernels (tools/flrelu_check: float16 forward 9e-4, the
       # up-2 / down-4 backward 1.6e-3 of |want| + 1; bfloat16 3.2e-3): VERDICT r03 weak 2
       torch.float16: dict(rtol=2e-3, atol=2e-3), torch.bfloat16: dict(rtol=1e-2, atol=1e-2)}


def dev(a, dtype, grad=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV, requires_grad=grad)


def host(t):
    return t.detach().to(torch.float64).cpu().numpy()


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_golden_forward_backward(dtype):
    g = load_golden('filtered_lrelu')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        for i in range(int(g['num_cases'])):
            p = f'c{i}_'
            sp = g[p + 'spec']
            fu = torch.tensor(g[p + 'fu'], device=DEV) if p + 'fu' in g else None
            fd = torch.tensor(g[p + 'fd'], device=DEV) if p + 'fd' in g else None
            x, b = dev(g[p + 'x'], dtype, True), dev(g[p + 'b'], dtype, True)
            y = filtered_lrelu.filtered_lrelu(x, fu, fd, b, **sp['kw'])
            tol = TOL[dtype]
            np.testing.assert_allclose(host(y), g[p + 'y'], err_msg=str(sp), **tol)
            dx, db = torch.autograd.grad(y, [x, b], dev(g[p + 'dy'], dtype))
            # float32: a pre-activation within rounding of 0 or of the clamp may flip its mask bit;
            # such elements are measure-zero, allow a handful of outliers.
            err = np.abs(host(dx) - g[p + 'dx'])
            lim = tol['atol'] * 10 + tol['rtol'] * 10 * np.abs(g[p + 'dx'])
            assert (err > lim).mean() <= (0.0 if dtype == torch.float64 else 2e-3), (str(sp), float(err.max()))
            np.testing.assert_allclose(host(db), g[p + 'db'], rtol=1e-3 if dtype == torch.float32 else 1e-5, atol=1e-3 if dtype == torch.float32 else 1e-5)


def test_sign_mask_bit_exact_with_exact_arithmetic(oracle):
    """Integer inputs and dyadic taps make float32 and float64 arithmetic both exact, so the
    2-bit mask written by the GPU must equal the oracle's byte for byte (active region)."""
    rs = np.random.RandomState(11)
    x = rs.randint(-8, 9, size=(2, 3, 10, 13)).astype(np.float32)
    b = rs.randint(-2, 3, size=(3,)).astype(np.float32)
    f = np.array([0.125, 0.375, 0.375, 0.125], dtype=np.float32)
    kw = dict(up=2, down=2, padding=[3, 2, 3, 2], gain=2.0, slope=0.25, clamp=4.0)
    xt = dev(x, torch.float32, True)
    ft = torch.tensor(f, device=DEV)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        y = filtered_lrelu.filtered_lrelu(xt, ft, ft, dev(b, torch.float32), **kw)
    yo, so = oracle.filtered_lrelu(x, f, f, b, write_signs=True, **kw)
    np.testing.assert_array_equal(host(y), yo)
    s_gpu = y.grad_fn.saved_tensors[0].cpu().numpy()
    assert s_gpu.shape == so.shape
    sh, swb, sw_active = oracle.sign_shape(yo.shape[2], yo.shape[3], 2, 4, 4)
    for xx in range(sw_active):
        got = (s_gpu[..., xx >> 2] >> ((xx & 3) * 2)) & 3
        want = (so[..., xx >> 2] >> ((xx & 3) * 2)) & 3
        np.testing.assert_array_equal(got, want)
    dy = rs.randint(-4, 5, size=yo.shape).astype(np.float32)
    (dx,) = torch.autograd.grad(y, xt, dev(dy, torch.float32))
    pp = [3 + 3 - 3, x.shape[3] * 2 - yo.shape[3] * 2 + 3 - 1, 3 + 3 - 3, x.shape[2] * 2 - yo.shape[2] * 2 + 3 - 1]
    dxo = oracle.filtered_lrelu(dy, f, f, None, up=2, down=2, padding=pp, gain=2.0, slope=0.25, clamp=None,
                                flip_filter=True, signs=so, sign_ofs=(-3 + 3, -3 + 3))
    np.testing.assert_array_equal(host(dx), dxo)


SRES = [
    # (name, x shape, up, down, fu taps, fd taps, padding)
    ('L0_like', [2, 16, 31, 38], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('up4', [2, 8, 31, 38], 4, 2, 24, 12, [-6, -9, -6, -9]),
    ('bwd_shape_up2_down4', [2, 8, 40, 44], 2, 4, 12, 24, [11, 10, 11, 10]),
    ('final_crop', [1, 8, 60, 70], 2, 2, 12, 12, [-11, -12, -11, -12]),
    ('odd', [1, 3, 17, 23], 2, 2, 12, 12, [9, 8, 9, 8]),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('case', SRES, ids=[s[0] for s in SRES])
def test_sres_layer_shapes_vs_oracle(case, dtype, oracle):
    import scipy.signal
    name, shape, up, down, nu, nd, pad = case
    fu = scipy.signal.firwin(numtaps=nu, cutoff=0.9 / up, width=0.6 / up, fs=2.0).astype(np.float32)
    fd = scipy.signal.firwin(numtaps=nd, cutoff=0.9 / down, width=0.6 / down, fs=2.0).astype(np.float32)
    rs = np.random.RandomState(6)
    x = dev(rs.randn(*shape), dtype, True)
    b = dev(rs.randn(shape[1]) * 0.3, dtype, True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        y = filtered_lrelu.filtered_lrelu(x, torch.tensor(fu, device=DEV), torch.tensor(fd, device=DEV), b,
                                          up=up, down=down, padding=pad, gain=np.sqrt(2), slope=0.2, clamp=256)
        ref = oracle.filtered_lrelu(host(x), fu, fd, host(b), up=up, down=down, padding=pad, gain=np.sqrt(2), slope=0.2, clamp=256)
        assert tuple(y.shape) == ref.shape and y.dtype == dtype
        np.testing.assert_allclose(host(y), ref, err_msg=name, **TOL[dtype])
        _check_backward(oracle, x, b, y, fu, fd, up, down, pad, np.sqrt(2), 0.2, dtype, name)


def _mask_pixels(s, sw_active):
    """[n, c, sh, sw_active] array of 2-bit codes from a packed mask."""
    cols = np.arange(sw_active)
    return (s[..., cols >> 2] >> ((cols & 3) * 2)) & 3


def _check_backward(oracle, x, b, y, fu, fd, up, down, pad, gain, slope, dtype, name, mask_ref=None):
    """dx and db of the HIP op (the fused kernel in sign-READ mode with the filter roles swapped,
    reference filtered_lrelu.py:239-268) vs the oracle run in READ mode on the mask the GPU wrote, and
    the GPU's mask vs the oracle's own (computed in float64 from the same inputs)."""
    px0, px1, py0, py1 = pad
    nu, nd = len(fu), len(fd)
    rs = np.random.RandomState(3)
    dy = dev(rs.randn(*y.shape), dtype)
    s_gpu = y.grad_fn.saved_tensors[0].cpu().numpy()
    dx, db = torch.autograd.grad(y, [x, b], dy)
    xh, xw = x.shape[2:]
    yh, yw = y.shape[2:]
    pp = [(nu - 1) + (nd - 1) - px0, xw * up - yw * down + px0 - (up - 1),
          (nu - 1) + (nd - 1) - py0, xh * up - yh * down + py0 - (up - 1)]
    dxo = oracle.filtered_lrelu(host(dy), fd, fu, None, up=down, down=up, padding=pp, gain=gain * up ** 2 / down ** 2,
                                slope=slope, clamp=None, flip_filter=True, signs=s_gpu,
                                sign_ofs=(-(nu - 1) + px0, -(nu - 1) + py0))
    assert dxo.shape == tuple(x.shape)
    tol = TOL[dtype]
    np.testing.assert_allclose(host(dx), dxo, err_msg=name + ' dx', **tol)
    dbo = dxo.sum(axis=(0, 2, 3))
    # db is a torch reduction of the (rounded) dx
    np.testing.assert_allclose(host(db), dbo, rtol=max(tol['rtol'], 1e-4) * 4, atol=tol['atol'] * np.sqrt(dxo[0, 0].size) * 4, err_msg=name + ' db')
    if mask_ref is not None:
        sh, swb, sw_active = oracle.sign_shape(yh, yw, down, nd, nd)
        assert s_gpu.shape == mask_ref.shape == (x.shape[0], x.shape[1], sh, swb)
        diff = _mask_pixels(s_gpu, sw_active) != _mask_pixels(mask_ref, sw_active)
        # a pre-activation within rounding of 0 (or of the clamp) may land on the other side in float64. float32 kernels round at
        # 2^-24: 2e-4 of the pixels at most. The 16-bit MFMA kernels carry T' and U in float16 (2^-11): measured 0.5e-4 .. 2.2e-4 of the
        # pixels (clamp 2.5 doubles the tie zone; tools/flrelu_check prints the fraction per case), gate at twice the worst
        assert diff.mean() <= (2e-4 if dtype == torch.float32 else 4.5e-4), (name, float(diff.mean()))
        assert not s_gpu[..., (sw_active + 3) >> 2:].any(), 'padding bytes of the mask must be 0'


# True layer geometry of the 144x256 super-resolution generator: several tiles of the
# fused kernels in x AND y (tile seams, mask-byte ownership, the READ-mode column re-alignment), few planes.
FULL = [
    ('L8_u2d2', [1, 4, 94, 150], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('L10_u4d2', [1, 3, 94, 150], 4, 2, 24, 12, [-6, -9, -6, -9]),
    ('L13_final_crop', [1, 3, 166, 278], 2, 2, 12, 12, [-11, -12, -11, -12]),
    ('L4_u2d2_small', [2, 3, 40, 54], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('L5_u4d2_small', [2, 3, 40, 54], 4, 2, 24, 12, [-6, -9, -6, -9]),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('case', FULL, ids=[s[0] for s in FULL])
def test_full_size_layers_forward_backward_vs_oracle(case, dtype, oracle):
    import scipy.signal
    name, shape, up, down, nu, nd, pad = case
    fu = scipy.signal.firwin(numtaps=nu, cutoff=0.9 / up, width=0.6 / up, fs=2.0).astype(np.float32)
    fd = scipy.signal.firwin(numtaps=nd, cutoff=0.9 / down, width=0.6 / down, fs=2.0).astype(np.float32)
    rs = np.random.RandomState(16)
    x = dev(rs.randn(*shape), dtype, True)
    b = dev(rs.randn(shape[1]) * 0.3, dtype, True)
    clamp = 2.5      # low enough that the "clamped" bit of the mask is exercised
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        y = filtered_lrelu.filtered_lrelu(x, torch.tensor(fu, device=DEV), torch.tensor(fd, device=DEV), b,
                                          up=up, down=down, padding=pad, gain=np.sqrt(2), slope=0.2, clamp=clamp)
        ref, so = oracle.filtered_lrelu(host(x), fu, fd, host(b), up=up, down=down, padding=pad, gain=np.sqrt(2), slope=0.2,
                                        clamp=clamp, write_signs=True)
        assert tuple(y.shape) == ref.shape and y.dtype == dtype
        np.testing.assert_allclose(host(y), ref, err_msg=name, **TOL[dtype])
        _check_backward(oracle, x, b, y, fu, fd, up, down, pad, np.sqrt(2), 0.2, dtype, name, mask_ref=so)


def test_no_grad_writes_no_mask_and_torgb_1x1(oracle):
    rs = np.random.RandomState(8)
    x = dev(rs.randn(2, 5, 9, 11), torch.float32)
    b = dev(rs.randn(5), torch.float32)
    y = filtered_lrelu.filtered_lrelu(x, None, None, b, up=1, down=1, gain=1, slope=1, clamp=256)
    assert y.grad_fn is None
    np.testing.assert_allclose(host(y), oracle.filtered_lrelu(host(x), None, None, host(b), gain=1, slope=1, clamp=256), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(3, 5, 92, 148), (2, 7, 13, 9), (1, 1, 1, 1), (4, 64, 36, 64)])
def test_bias_gradient_plane_sums(shape, dtype):
    """filtered_lrelu's bias gradient dx.sum([0, 2, 3]) (reference filtered_lrelu.py:254) through lvg_plane_sum: float32 accumulation of
    every plane, then the samples -- against the float64 sum of the same values."""
    from torch_utils.ops import filtered_lrelu as fl
    torch.manual_seed(5)
    dx = torch.randn(shape, device='cuda').to(dtype)
    got = fl._bias_grad(dx)
    assert got.dtype == dtype and got.shape == (shape[1],)
    want = dx.double().sum([0, 2, 3])
    eps = {torch.float32: 2.0 ** -23, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[dtype]
    scale = dx.double().abs().sum([0, 2, 3])
    assert ((got.double() - want).abs() <= eps * want.abs() + 2.0 ** -22 * scale + 1e-30).all()


@pytest.fixture
def band_kernel_everywhere():
    """Route every float16 call the row-band kernel can take to it (lvg_filtered_lrelu_set_impl(4); what it cannot take falls back to the
    wave kernel), instead of only the plane widths it is the default for."""
    from torch_utils.ops import _hip
    prev = _hip.lib().lvg_filtered_lrelu_set_impl(4)
    assert prev >= 0
    yield
    _hip.lib().lvg_filtered_lrelu_set_impl(prev)


# Geometry the row-band kernel (csrc/filtered_lrelu_band.hip) has code paths for: one to six column strips, planes shorter than a
# block of 32 output rows / exactly one block / two blocks finishing in the same step, more planes than workgroups are resident
# (several planes per workgroup: LDS ring, bias rows and mask prefetch cross plane boundaries), the up-4 and down-4 chunk patterns,
# negative padding (crop), wide margins of the backward passes.
BAND = [
    ('one_strip', [3, 7, 40, 54], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('two_strips_u4', [2, 3, 40, 54], 4, 2, 24, 12, [-6, -9, -6, -9]),
    ('three_strips', [1, 4, 94, 150], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('five_strips_crop', [1, 3, 166, 278], 2, 2, 12, 12, [-11, -12, -11, -12]),
    ('five_strips_u4', [1, 3, 94, 150], 4, 2, 24, 12, [-6, -9, -6, -9]),
    ('exactly_32_rows', [1, 3, 34, 62], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('37_rows', [1, 3, 39, 118], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('tall', [1, 2, 300, 20], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('many_planes', [16, 512, 18, 22], 2, 2, 12, 12, [9, 8, 9, 8]),
]


@pytest.fixture
def strip_kernel_everywhere():
    """Route every float16 call the strip kernel (round 6, csrc/filtered_lrelu_strip.hip) can take to it (lvg_filtered_lrelu_set_impl(5)),
    instead of only the plane widths it is the default for. LVG_FLRELU_STRIP_MAXGRID is read once per process by the launcher, so the
    several-items-per-wave case is covered by shapes with more (plane, strip) items than resident waves ('many_planes')."""
    from torch_utils.ops import _hip
    prev = _hip.lib().lvg_filtered_lrelu_set_impl(5)
    assert prev >= 0
    yield
    _hip.lib().lvg_filtered_lrelu_set_impl(prev)


# Extra geometry for the strip kernel: a plane's first strip starting left of the image at every residue of the padding (DMA origin a
# multiple of 8, transpose-read base 0 / 4, fragment shift 0..3), strips whose ring row reaches past the end of the image row (the
# zeroed straddling piece) while not being the last strip, widths of 24 k + {2, 4} columns (narrow last strip), up 2 / down 4 forward.
STRIP = BAND + [
    ('pad_residues_a', [2, 3, 40, 54], 2, 2, 12, 12, [10, 7, 8, 9]),
    ('pad_residues_b', [2, 3, 40, 54], 2, 2, 12, 12, [7, 10, 11, 6]),
    ('pad_residues_c', [1, 3, 40, 54], 2, 2, 12, 12, [6, 11, 5, 12]),
    ('narrow_last_strip', [1, 3, 30, 45], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('width_50', [1, 2, 20, 52], 2, 2, 12, 12, [9, 8, 9, 8]),
    ('d4_forward', [1, 3, 60, 100], 2, 4, 12, 24, [9, 8, 9, 8]),
]


@pytest.mark.parametrize('clamp', [2.5, 256.0], ids=['clamp2.5', 'clamp256'])
@pytest.mark.parametrize('case', STRIP, ids=[s[0] for s in STRIP])
def test_strip_kernel_forward_backward_vs_oracle(case, clamp, oracle, strip_kernel_everywhere):
    """The strip kernel on everything it can take, float16: the same checks as for the row-band kernel below."""
    _kernel_forward_backward_vs_oracle(case, clamp, oracle)


@pytest.mark.parametrize('clamp', [2.5, 256.0], ids=['clamp2.5', 'clamp256'])
@pytest.mark.parametrize('case', BAND, ids=[s[0] for s in BAND])
def test_band_kernel_forward_backward_vs_oracle(case, clamp, oracle, band_kernel_everywhere):
    _kernel_forward_backward_vs_oracle(case, clamp, oracle)


def _kernel_forward_backward_vs_oracle(case, clamp, oracle):
    """One fused kernel on everything it can take, float16: forward with the mask against the float64 oracle, the mask against the
    oracle's, the backward pass (sign-READ mode, filter roles swapped: for up 4 / down 2 layers that is the up 2 / down 4 instance)
    against the oracle run on the mask the GPU wrote. clamp 2.5 exercises the clamp and its mask bit, clamp 256 the proof that skips them."""
    import scipy.signal
    name, shape, up, down, nu, nd, pad = case
    dtype = torch.float16
    fu = scipy.signal.firwin(numtaps=nu, cutoff=0.9 / up, width=0.6 / up, fs=2.0).astype(np.float32)
    fd = scipy.signal.firwin(numtaps=nd, cutoff=0.9 / down, width=0.6 / down, fs=2.0).astype(np.float32)
    rs = np.random.RandomState(21)
    x = dev(rs.randn(*shape), dtype, True)
    b = dev(rs.randn(shape[1]) * 0.3, dtype, True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        y = filtered_lrelu.filtered_lrelu(x, torch.tensor(fu, device=DEV), torch.tensor(fd, device=DEV), b,
                                          up=up, down=down, padding=pad, gain=np.sqrt(2), slope=0.2, clamp=clamp)
        ref, so = oracle.filtered_lrelu(host(x), fu, fd, host(b), up=up, down=down, padding=pad, gain=np.sqrt(2), slope=0.2,
                                        clamp=clamp, write_signs=True)
        assert tuple(y.shape) == ref.shape and y.dtype == dtype
        np.testing.assert_allclose(host(y), ref, err_msg=name, **TOL[dtype])
        _check_backward(oracle, x, b, y, fu, fd, up, down, pad, np.sqrt(2), 0.2, dtype, name, mask_ref=so)
        # no-mask forward (inference): same values as the mask-writing forward
        with torch.no_grad():
            y2 = filtered_lrelu.filtered_lrelu(x, torch.tensor(fu, device=DEV), torch.tensor(fd, device=DEV), b,
                                               up=up, down=down, padding=pad, gain=np.sqrt(2), slope=0.2, clamp=clamp)
        assert torch.equal(y2, y.detach())


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape,cl', [((2, 155, 148, 276), False), ((16, 362, 94, 150), True), ((3, 7, 33, 57), False), ((1, 5, 128, 36, 64), False)])
def test_mean_square_statistic(shape, cl, dtype):
    """torch_utils.ops.stats.mean_square (lvg_plane_sum_sq: one pass, float32 accumulation) == x.float().square().mean() of the reference's
    input-magnitude statistic (generator_sres.py:278-286), on contiguous and channels-last tensors, with a tail that is not a whole chunk."""
    from torch_utils.ops import stats
    torch.manual_seed(5)
    x = torch.randn(shape, device='cuda').to(dtype) * 3
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    got = stats.mean_square(x)
    want = x.double().square().mean()
    assert got.dtype == torch.float32 and got.shape == ()
    assert abs(float(got) - float(want)) <= 2e-6 * float(want)


def _random_case(seed):
    """A seeded random geometry inside what the sres layers span and a little beyond: plane sizes 6..130 x 6..180 (one to eight column
    strips, one to five row blocks), the three (up, down) pairs of the forward and backward passes, every residue of the four paddings
    (positive = margin, negative = crop) that leaves at least one output pixel, slope / gain / clamp drawn separately."""
    rs = np.random.RandomState(1000 + seed)
    up, down, nu, nd = [(2, 2, 12, 12), (4, 2, 24, 12), (2, 4, 12, 24)][seed % 3]
    n, c = int(rs.randint(1, 3)), int(rs.randint(1, 5))
    h, w = int(rs.randint(6, 131)), int(rs.randint(6, 181))
    while True:
        pad = [int(v) for v in rs.randint(-7, 15, size=4)]
        ow = (w * up + pad[0] + pad[1] - (nu - 1) - (nd - 1) + (down - 1)) // down
        oh = (h * up + pad[2] + pad[3] - (nu - 1) - (nd - 1) + (down - 1)) // down
        if ow >= 1 and oh >= 1:
            break
    slope = [0.2, 0.1, 0.5][int(rs.randint(3))]
    gain = [float(np.sqrt(2)), 1.0][int(rs.randint(2))]
    clamp = [2.5, 256.0, None][int(rs.randint(3))]
    return (f'seed{seed}_u{up}d{down}_{h}x{w}', [n, c, h, w], up, down, nu, nd, pad), slope, gain, clamp


@pytest.mark.parametrize('impl', [0, 5, 4, 3], ids=['routed', 'strip', 'band', 'wave'])
@pytest.mark.parametrize('seed', range(18))
def test_random_geometry_vs_oracle(seed, impl, oracle):
    """Seeded random plane sizes / paddings / activation constants, float16, on the default route and with each fused MFMA kernel forced
    (what a kernel cannot take falls back, so every case runs): forward and mask against the float64 oracle, backward against the oracle
    on the GPU's mask. The hand-picked lists above name the code paths; this sweep is for the combinations nobody picked."""
    import scipy.signal
    from torch_utils.ops import _hip
    case, slope, gain, clamp = _random_case(seed)
    name, shape, up, down, nu, nd, pad = case
    dtype = torch.float16
    fu = scipy.signal.firwin(numtaps=nu, cutoff=0.9 / up, width=0.6 / up, fs=2.0).astype(np.float32)
    fd = scipy.signal.firwin(numtaps=nd, cutoff=0.9 / down, width=0.6 / down, fs=2.0).astype(np.float32)
    rs = np.random.RandomState(seed)
    x = dev(rs.randn(*shape), dtype, True)
    b = dev(rs.randn(shape[1]) * 0.3, dtype, True)
    prev = _hip.lib().lvg_filtered_lrelu_set_impl(impl)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)
            y = filtered_lrelu.filtered_lrelu(x, torch.tensor(fu, device=DEV), torch.tensor(fd, device=DEV), b,
                                              up=up, down=down, padding=pad, gain=gain, slope=slope, clamp=clamp)
            ref, so = oracle.filtered_lrelu(host(x), fu, fd, host(b), up=up, down=down, padding=pad, gain=gain, slope=slope,
                                            clamp=clamp, write_signs=True)
            assert tuple(y.shape) == ref.shape and y.dtype == dtype, name
            np.testing.assert_allclose(host(y), ref, err_msg=name, **TOL[dtype])
            _check_backward(oracle, x, b, y, fu, fd, up, down, pad, gain, slope, dtype, name, mask_ref=so)
    finally:
        _hip.lib().lvg_filtered_lrelu_set_impl(prev)
