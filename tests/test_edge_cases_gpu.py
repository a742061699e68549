"""Edge cases the reference kernels define and a fast path can silently get wrong: non-finite inputs, clamp = 0,
slope > 1, zero-size tensors. HIP ops through the C ABI against the C oracle (which restates the CUDA kernels' expressions:
bias_act.cu:23-147, upfirdn2d.cu:29-187, filtered_lrelu.cu:484-579), on every implementation that can serve the call.

What the reference does, and what is pinned here:
  * bias_act: the clamp is written `(y > -c & y < c) ? y : (y >= 0 ? c : -c)`: NaN -> -clamp when a clamp is given (clamp >= 0),
    NaN stays NaN without one; +-inf -> +-clamp. Same expression here: bit-for-bit positions of the non-finite outputs.
  * upfirdn2d: a non-finite pixel reaches every output whose taps touch it (NaN * tap, also for taps that are 0: the kernels multiply
    every tap). Same here.
  * filtered_lrelu: `if (fabsf(v) > clamp) v = copysign(clamp, v)` leaves NaN alone; leaky ReLU of NaN is NaN. The float32 kernel
    follows that pixel for pixel. The 16-bit MFMA kernels (round-2 MFMA, wave, band) run the FIR stages as banded matrix products:
    a non-finite value multiplies the ZERO entries of the band too (NaN), and their packed min / max are compiled without NaN
    handling, so inside the up-sampled blocks that contain the bad pixel the outputs are unspecified (NaN, or +-clamp); what they
    guarantee -- and what is tested -- is that every output FARTHER than one block (128 up-sampled pixels) from it is unaffected,
    that nothing faults, and that the mask bytes there are those of the clean input.
  * clamp = 0: every output is +-0 and (filtered_lrelu) every mask code says "clamped"; slope > 1 (leaky ReLU is then NOT max(x, s x)):
    the wave and band kernels hand the call to the round-2 kernel -- result vs oracle.
  * zero-size x: bias_act returns an empty tensor (the reference plugin launches nothing); upfirdn2d / filtered_lrelu refuse
    ("x has zero size" / "x is empty": TORCH_CHECK there, AssertionError here)."""

import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from torch_utils.ops import bias_act, filtered_lrelu, upfirdn2d

DEV = 'cuda'
BAD = [float('nan'), float('inf'), -float('inf')]


def host(t):
    return t.detach().to(torch.float64).cpu().numpy()


def same_nonfinite(got, want):
    """NaNs at the same places, infinities equal with sign, finite values left to the caller."""
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    inf = np.isinf(want)
    np.testing.assert_array_equal(np.isinf(got), inf)
    np.testing.assert_array_equal(got[inf], want[inf])


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('act', ['linear', 'lrelu', 'relu', 'sigmoid', 'swish'])
@pytest.mark.parametrize('clamp', [None, 0.0, 4.0])
def test_bias_act_nonfinite_inputs_and_zero_clamp(act, clamp, dtype, oracle):
    rs = np.random.RandomState(3)
    x = rs.randn(2, 5, 7, 3)
    x[0, 1, 2, 0], x[1, 0, 0, 1], x[1, 4, 6, 2] = BAD
    x[0, 0, 0, 0] = 1e4          # finite, beyond the clamp
    b = rs.randn(5) * 0.3
    xt = torch.tensor(x, dtype=dtype, device=DEV)
    bt = torch.tensor(b, dtype=dtype, device=DEV)
    y = bias_act.bias_act(xt, bt, dim=1, act=act, clamp=clamp)
    want = oracle.bias_act(host(xt), host(bt), dim=1, act=act, clamp=clamp)
    got = host(y)
    same_nonfinite(got, want)
    fin = np.isfinite(want)
    tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    np.testing.assert_allclose(got[fin], want[fin], rtol=tol, atol=tol)
    if clamp == 0.0:
        assert not got.any()                                           # +-0 everywhere, NaN included (-clamp = -0)
    # first derivative with a non-finite incoming gradient and a non-finite forward value
    if act in ('linear', 'lrelu', 'relu'):
        xg = xt.clone().requires_grad_(True)
        yg = bias_act.bias_act(xg, bt, dim=1, act=act, clamp=clamp)
        dy = rs.randn(*x.shape)
        dy[0, 2, 3, 1] = float('nan')
        dyt = torch.tensor(dy, dtype=dtype, device=DEV)
        (dx,) = torch.autograd.grad(yg, xg, dyt)
        want_dx = oracle.bias_act(host(dyt), None, dim=1, act=act, clamp=clamp, grad=1, xref=host(xt) + host(bt).reshape(1, -1, 1, 1), yref=host(yg), dy=host(dyt))
        same_nonfinite(host(dx), want_dx)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_bias_act_zero_size(dtype):
    x = torch.zeros(0, 3, 4, device=DEV, dtype=dtype, requires_grad=True)
    b = torch.zeros(3, device=DEV, dtype=dtype, requires_grad=True)
    y = bias_act.bias_act(x, b, dim=1, act='lrelu', clamp=1.0)
    assert y.shape == x.shape and y.dtype == dtype
    dx, db = torch.autograd.grad(y.sum(), [x, b])
    assert dx.shape == x.shape and db.shape == b.shape and not db.any()


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('up,down', [(2, 1), (1, 2), (1, 1)])
def test_upfirdn2d_nonfinite_inputs(up, down, dtype, oracle):
    rs = np.random.RandomState(5)
    x = rs.randn(2, 3, 19, 23)
    x[0, 1, 4, 5], x[1, 2, 10, 11], x[1, 0, 17, 2] = BAD
    f = np.array([1.0, 3.0, 3.0, 1.0], dtype=np.float32)
    f = (np.outer(f, f) / 64.0).astype(np.float32)
    xt = torch.tensor(x, dtype=dtype, device=DEV)
    y = upfirdn2d.upfirdn2d(xt, torch.tensor(f, device=DEV), up=up, down=down, padding=[2, 1, 2, 1], gain=up * up)
    want = oracle.upfirdn2d(host(xt), f, up=up, down=down, padding=[2, 1, 2, 1], gain=up * up)
    got = host(y)
    # a NaN reaches exactly the outputs whose window holds the bad pixel; +inf and -inf alone stay infinities of their sign
    np.testing.assert_array_equal(np.isfinite(got), np.isfinite(want))
    same_nonfinite(got, want)
    fin = np.isfinite(want)
    tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    np.testing.assert_allclose(got[fin], want[fin], rtol=tol, atol=tol)


def test_zero_size_is_refused():
    f = torch.ones(4, device=DEV)
    with pytest.raises((AssertionError, RuntimeError), match='zero size'):
        upfirdn2d.upfirdn2d(torch.zeros(2, 0, 8, 8, device=DEV), f)
    with pytest.raises((AssertionError, RuntimeError), match='empty'):
        filtered_lrelu.filtered_lrelu(torch.zeros(0, 3, 8, 8, device=DEV), f, f, torch.zeros(3, device=DEV), up=2, down=2, padding=1)


def _taps(n, rate):
    import scipy.signal
    return scipy.signal.firwin(numtaps=n, cutoff=0.9 / rate, width=0.6 / rate, fs=2.0).astype(np.float32)


@pytest.fixture
def flrelu_impl():
    """lvg_filtered_lrelu_set_impl for the duration of a test: 1 fp32-VALU, 2 round-2 MFMA, 3 wave, 4 band, 5 strip."""
    from torch_utils.ops import _hip
    prev = []

    def use(impl):
        prev.append(_hip.lib().lvg_filtered_lrelu_set_impl(impl))
    yield use
    if prev:
        _hip.lib().lvg_filtered_lrelu_set_impl(prev[0])


def test_filtered_lrelu_float32_nonfinite_follows_the_reference(oracle):
    rs = np.random.RandomState(7)
    x = rs.randn(1, 2, 30, 40)
    x[0, 0, 7, 9], x[0, 1, 20, 30], x[0, 1, 3, 3] = BAD
    b = rs.randn(2) * 0.3
    fu, fd = _taps(12, 2), _taps(12, 2)
    kw = dict(up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=0.2, clamp=2.5)
    xt, bt = torch.tensor(x, dtype=torch.float32, device=DEV), torch.tensor(b, dtype=torch.float32, device=DEV)
    with torch.no_grad():
        y = filtered_lrelu.filtered_lrelu(xt, torch.tensor(fu, device=DEV), torch.tensor(fd, device=DEV), bt, **kw)
    want = oracle.filtered_lrelu(x, fu, fd, b, **kw)
    got = host(y)
    np.testing.assert_array_equal(np.isfinite(got), np.isfinite(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=5e-5, atol=5e-6)
    assert (~fin).sum() > 0 and fin.sum() > 0


@pytest.mark.parametrize('impl', [2, 3, 4, 5], ids=['mfma', 'wave', 'band', 'strip'])
@pytest.mark.parametrize('bad', BAD, ids=['nan', 'inf', '-inf'])
def test_filtered_lrelu_16bit_nonfinite_stays_local(impl, bad, oracle, flrelu_impl):
    """One bad pixel in a 94 x 150 plane: outputs more than a block (128 up-sampled = 64 output pixels, + the filters' reach) away from it
    equal those of the clean plane bit for bit, and so do their mask bytes; the other planes are untouched."""
    flrelu_impl(impl)
    rs = np.random.RandomState(9)
    x = rs.randn(1, 3, 94, 150)
    fu, fd = _taps(12, 2), _taps(12, 2)
    kw = dict(up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=0.2, clamp=256)
    bt = torch.tensor(rs.randn(3) * 0.3, dtype=torch.float16, device=DEV)
    fut, fdt = torch.tensor(fu, device=DEV), torch.tensor(fd, device=DEV)
    clean = torch.tensor(x, dtype=torch.float16, device=DEV, requires_grad=True)
    py, px = 40, 70
    xb = x.copy()
    xb[0, 1, py, px] = bad
    dirty = torch.tensor(xb, dtype=torch.float16, device=DEV, requires_grad=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        yc = filtered_lrelu.filtered_lrelu(clean, fut, fdt, bt, **kw)
        yd = filtered_lrelu.filtered_lrelu(dirty, fut, fdt, bt, **kw)
    torch.cuda.synchronize()
    yc_h, yd_h = host(yc), host(yd)
    far = np.ones(yc_h.shape, dtype=bool)
    reach = 64 + 12
    far[0, 1, max(0, py - reach):py + reach, max(0, px - reach):px + reach] = False
    assert far[0, 1].sum() > 0
    np.testing.assert_array_equal(yd_h[far], yc_h[far])
    assert np.isfinite(yd_h[far]).all()
    sc, sd = yc.grad_fn.saved_tensors[0].cpu().numpy(), yd.grad_fn.saved_tensors[0].cpu().numpy()
    np.testing.assert_array_equal(sd[0, 0], sc[0, 0])
    np.testing.assert_array_equal(sd[0, 2], sc[0, 2])
    # mask rows / bytes of the dirty plane far from the pixel (2 up-sampled rows per output row, 4 pixels per byte)
    rows = np.ones(sc.shape[2], dtype=bool); rows[max(0, 2 * (py - reach)):2 * (py + reach)] = False
    np.testing.assert_array_equal(sd[0, 1][rows], sc[0, 1][rows])


@pytest.mark.parametrize('impl', [0, 2, 3, 4, 5], ids=['default', 'mfma', 'wave', 'band', 'strip'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_filtered_lrelu_zero_clamp_and_large_slope(impl, dtype, oracle, flrelu_impl):
    flrelu_impl(impl)
    rs = np.random.RandomState(11)
    x = rs.randn(2, 3, 40, 54)
    b = rs.randn(3) * 0.3
    fu, fd = _taps(12, 2), _taps(12, 2)
    xt, bt = torch.tensor(x, dtype=dtype, device=DEV, requires_grad=True), torch.tensor(b, dtype=dtype, device=DEV)
    fut, fdt = torch.tensor(fu, device=DEV), torch.tensor(fd, device=DEV)
    tol = dict(rtol=5e-5, atol=5e-6) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-3)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        # clamp = 0: every pre-activation is "clamped" (|v| > 0) or exactly 0; the output is zero, every mask code with |v| > 0 is 2
        y0 = filtered_lrelu.filtered_lrelu(xt, fut, fdt, bt, up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=0.2, clamp=0)
        assert not host(y0).any()
        want, so = oracle.filtered_lrelu(host(xt), fu, fd, host(bt), up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=0.2, clamp=0, write_signs=True)
        s0 = y0.grad_fn.saved_tensors[0].cpu().numpy()
        sh, swb, sw_active = oracle.sign_shape(want.shape[2], want.shape[3], 2, 12, 12)
        cols = np.arange(sw_active)
        codes = (s0[..., cols >> 2] >> ((cols & 3) * 2)) & 3
        want_codes = (so[..., cols >> 2] >> ((cols & 3) * 2)) & 3
        assert (codes != want_codes).mean() <= 4.5e-4 and (codes == 2).mean() > 0.99
        (dx0,) = torch.autograd.grad(y0, xt, torch.ones_like(y0))
        assert not host(dx0).any()                                     # every pixel clamped: no gradient
        # slope > 1: max(x, slope x) would be wrong; vs the oracle, forward and backward
        kw = dict(up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=1.7, clamp=256)
        y1 = filtered_lrelu.filtered_lrelu(xt, fut, fdt, bt, **kw)
        np.testing.assert_allclose(host(y1), oracle.filtered_lrelu(host(xt), fu, fd, host(bt), **kw), **tol)
        dy = torch.tensor(rs.randn(*y1.shape), dtype=dtype, device=DEV)
        s1 = y1.grad_fn.saved_tensors[0].cpu().numpy()
        (dx1,) = torch.autograd.grad(y1, xt, dy)
        pp = [11 + 11 - 9, x.shape[3] * 2 - y1.shape[3] * 2 + 9 - 1, 11 + 11 - 9, x.shape[2] * 2 - y1.shape[2] * 2 + 9 - 1]
        dxo = oracle.filtered_lrelu(host(dy), fd, fu, None, up=2, down=2, padding=pp, gain=np.sqrt(2), slope=1.7, clamp=None,
                                    flip_filter=True, signs=s1, sign_ofs=(-11 + 9, -11 + 9))
        np.testing.assert_allclose(host(dx1), dxo, **tol)
