"""Pins oracle/ (the CPU restatement) to the golden fixtures produced by the reference's own
impl='ref' path (tests/golden/make_golden.py). CPU only."""

import numpy as np
import pytest

from conftest import load_golden

TOL = dict(rtol=1e-9, atol=1e-10)  # both sides are float64 (filters are float32 values on both sides)


def _tol_for_up(up):
    """The reference's Python ref path folds gain^(ndim/2) into the float32 filter BEFORE casting it
    (upfirdn2d.py:196-197) while its CUDA kernel -- which the oracle follows -- scales the
    accumulator (upfirdn2d.cu:194). Identical for power-of-two gains (up in 1,2,4: gain=up^2);
    float32-rounding apart (~1e-7 relative) otherwise."""
    return TOL if up in (1, 2, 4) else dict(rtol=2e-6, atol=2e-7)


def test_bias_act_forward_and_grads(oracle):
    g = load_golden('bias_act')
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        x, b = g[p + 'x'], g.get(p + 'b')
        kw = dict(dim=sp['dim'], act=sp['act'], alpha=sp['alpha'], gain=sp['gain'], clamp=sp['clamp'])
        y = oracle.bias_act(x, b, **kw)
        np.testing.assert_allclose(y, g[p + 'y'], err_msg=str(sp), **TOL)
        # first-order: kernel form grad=1 takes (dy, yref=y) (and xref=x for swish)
        dx = oracle.bias_act(g[p + 'dy'], b, grad=1, xref=x, yref=y, **kw)
        np.testing.assert_allclose(dx, g[p + 'dx'], err_msg='dx ' + str(sp), **TOL)
        if b is not None:
            axes = tuple(a for a in range(x.ndim) if a != sp['dim'])
            np.testing.assert_allclose(dx.sum(axis=axes), g[p + 'db'], err_msg='db ' + str(sp), rtol=1e-9, atol=1e-9)
        # second-order: grad=2 takes (d_dx, dy, yref, xref)
        if p + 'd_x' in g:
            d_x = oracle.bias_act(g[p + 'ddx'], b, grad=2, xref=x, yref=y, dy=g[p + 'dy'], **kw)
            np.testing.assert_allclose(d_x, g[p + 'd_x'], err_msg='d_x ' + str(sp), rtol=1e-8, atol=1e-9)


def test_upfirdn2d_forward_and_backward(oracle):
    g = load_golden('upfirdn2d')
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        fn = dict(upfirdn2d=oracle.upfirdn2d, upsample2d=oracle.upsample2d, downsample2d=oracle.downsample2d,
                  filter2d=None)[sp['entry']]
        x, f = g[p + 'x'], g.get(p + 'f')
        kw = dict(sp['kw'])
        if sp['entry'] == 'filter2d':
            fw = fh = f.shape[0]
            kw['padding'] = [fw // 2, (fw - 1) // 2, fh // 2, (fh - 1) // 2]  # upfirdn2d.py:301-309
            fn = oracle.upfirdn2d
        y = fn(x, f, **kw)
        assert y.shape == g[p + 'y'].shape, sp
        np.testing.assert_allclose(y, g[p + 'y'], err_msg=str(sp), **TOL)


def test_upfirdn2d_backward_is_transposed_op(oracle):
    """dx from the reference's autograd == the op with up<->down, flipped filter and the padding
    of upfirdn2d.py:256-266, evaluated by the oracle."""
    g = load_golden('upfirdn2d')
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        if sp['entry'] != 'upfirdn2d':
            continue
        kw = sp['kw']
        x, f, dy = g[p + 'x'], g.get(p + 'f'), g[p + 'dy']
        up = kw.get('up', 1); down = kw.get('down', 1)
        upx, upy = (up, up) if isinstance(up, int) else up
        downx, downy = (down, down) if isinstance(down, int) else down
        pad = kw.get('padding', 0)
        pad = [pad] * 4 if isinstance(pad, int) else (list(pad) if len(pad) == 4 else [pad[0], pad[0], pad[1], pad[1]])
        f2 = np.ones([1, 1]) if f is None else (np.outer(f, f) if f.ndim == 1 else f)
        fh, fw = f2.shape
        ih, iw = x.shape[2:]
        oh, ow = dy.shape[2:]
        pp = [fw - pad[0] - 1, iw * upx - ow * downx + pad[0] - upx + 1, fh - pad[2] - 1, ih * upy - oh * downy + pad[2] - upy + 1]
        dx = oracle.upfirdn2d(dy, f, up=(downx, downy), down=(upx, upy), padding=pp,
                              flip_filter=not kw.get('flip_filter', False), gain=kw.get('gain', 1))
        np.testing.assert_allclose(dx, g[p + 'dx'], err_msg=str(sp), **TOL)


def test_filtered_lrelu_forward(oracle):
    g = load_golden('filtered_lrelu')
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        y = oracle.filtered_lrelu(g[p + 'x'], g.get(p + 'fu'), g.get(p + 'fd'), g[p + 'b'], **sp['kw'])
        np.testing.assert_allclose(y, g[p + 'y'], err_msg=str(sp), **_tol_for_up(sp['kw']['up']))


def test_filtered_lrelu_backward_via_signs(oracle):
    """The reference's backward (filtered_lrelu.py:239-268) = same op with up<->down, fu<->fd,
    gain*up^2/down^2, no clamp, flipped filters, reading the forward's sign tensor at offset.
    Checked against autograd of the reference's ref path."""
    g = load_golden('filtered_lrelu')
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        kw = dict(sp['kw'])
        x, fu, fd, b, dy = g[p + 'x'], g.get(p + 'fu'), g.get(p + 'fd'), g[p + 'b'], g[p + 'dy']
        y, s = oracle.filtered_lrelu(x, fu, fd, b, write_signs=True, **kw)
        up, down = kw['up'], kw['down']
        pad = kw['padding']
        pad = [pad] * 4 if isinstance(pad, int) else list(pad)
        fu_w = 1 if fu is None else fu.shape[-1]; fu_h = 1 if fu is None else fu.shape[0]
        fd_w = 1 if fd is None else fd.shape[-1]; fd_h = 1 if fd is None else fd.shape[0]
        xh, xw = x.shape[2:]; yh, yw = y.shape[2:]
        pp = [(fu_w - 1) + (fd_w - 1) - pad[0], xw * up - yw * down + pad[0] - (up - 1),
              (fu_h - 1) + (fd_h - 1) - pad[2], xh * up - yh * down + pad[2] - (up - 1)]
        gg = kw['gain'] * up ** 2 / down ** 2
        sx = -(fu_w - 1) + pad[0]
        sy = -(fu_h - 1) + pad[2]
        dx = oracle.filtered_lrelu(dy, fd, fu, None, up=down, down=up, padding=pp, gain=gg, slope=kw['slope'], clamp=None,
                                   flip_filter=not kw.get('flip_filter', False), signs=s, sign_ofs=(sx, sy))
        tol = _tol_for_up(up)
        np.testing.assert_allclose(dx, g[p + 'dx'], err_msg='dx ' + str(sp), **tol)
        np.testing.assert_allclose(dx.sum(axis=(0, 2, 3)), g[p + 'db'], err_msg='db ' + str(sp), rtol=tol['rtol'] * 10, atol=tol['atol'] * 100)


def test_filtered_lrelu_act_matches_fused_signs(oracle):
    """generic path middle step: signs written by act_ on the up-sampled tensor equal the fused
    op's signs (same bit semantics) when no value is exactly -0.0."""
    rs = np.random.RandomState(7)
    x = rs.randn(2, 3, 9, 13)
    out, s = oracle.filtered_lrelu_act(x, gain=1.4, slope=0.2, clamp=0.9, write_signs=True)
    ref = x * 1.4
    neg = ref < 0
    ref = np.where(neg, ref * 0.2, ref)
    clamped = np.abs(ref) > 0.9
    ref = np.clip(ref, -0.9, 0.9)
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=0)
    bits = np.where(clamped, 2, neg.astype(int))
    for xx in range(13):
        got = (s[:, :, :, xx >> 2] >> ((xx & 3) * 2)) & 3
        assert (got == bits[:, :, :, xx]).all()
    back = oracle.filtered_lrelu_act(np.ones_like(x), gain=1.0, slope=0.2, clamp=None, signs=s)
    np.testing.assert_allclose(back, np.where(clamped, 0.0, np.where(neg, 0.2, 1.0)))


def test_fma(oracle):
    g = load_golden('misc_ops')
    np.testing.assert_allclose(oracle.fma(g['fma_a'], g['fma_b'], g['fma_c']), g['fma_y'], **TOL)
