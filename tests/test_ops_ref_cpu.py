"""The plain-PyTorch definitions shipped in torch_utils.ops (impl='ref' / CPU tensors) agree with
the golden fixtures made from the reference's ref path, and with the oracle. CPU only."""

import numpy as np
import pytest
import torch

from conftest import load_golden

from torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu, conv2d_resample, fma


def t64(a, grad=False):
    return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=grad)


def test_bias_act_ref_matches_golden():
    g = load_golden('bias_act')
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        x = t64(g[p + 'x'], True)
        b = t64(g[p + 'b'], True) if p + 'b' in g else None
        y = bias_act.bias_act(x, b, dim=sp['dim'], act=sp['act'], alpha=sp['alpha'], gain=sp['gain'], clamp=sp['clamp'])
        np.testing.assert_allclose(y.detach().numpy(), g[p + 'y'], rtol=1e-12, atol=1e-12)
        dx = torch.autograd.grad(y, x, t64(g[p + 'dy']))[0]
        np.testing.assert_allclose(dx.numpy(), g[p + 'dx'], rtol=1e-12, atol=1e-12)


def test_upfirdn2d_ref_matches_golden():
    g = load_golden('upfirdn2d')
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        f = torch.tensor(g[p + 'f']) if p + 'f' in g else None
        x = t64(g[p + 'x'], True)
        y = getattr(upfirdn2d, sp['entry'])(x, f, **sp['kw'])
        np.testing.assert_allclose(y.detach().numpy(), g[p + 'y'], rtol=1e-12, atol=1e-12, err_msg=str(sp))
        dx = torch.autograd.grad(y, x, t64(g[p + 'dy']))[0]
        np.testing.assert_allclose(dx.numpy(), g[p + 'dx'], rtol=1e-12, atol=1e-12, err_msg=str(sp))


def test_filtered_lrelu_ref_matches_golden():
    g = load_golden('filtered_lrelu')
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        fu = torch.tensor(g[p + 'fu']) if p + 'fu' in g else None
        fd = torch.tensor(g[p + 'fd']) if p + 'fd' in g else None
        x, b = t64(g[p + 'x'], True), t64(g[p + 'b'], True)
        y = filtered_lrelu.filtered_lrelu(x, fu, fd, b, **sp['kw'])
        np.testing.assert_allclose(y.detach().numpy(), g[p + 'y'], rtol=1e-12, atol=1e-12, err_msg=str(sp))
        dx, db = torch.autograd.grad(y, [x, b], t64(g[p + 'dy']))
        np.testing.assert_allclose(dx.numpy(), g[p + 'dx'], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(db.numpy(), g[p + 'db'], rtol=1e-11, atol=1e-11)


def test_conv2d_resample_and_fma_and_setup_filter_match_golden():
    g = load_golden('misc_ops')
    bil = upfirdn2d.setup_filter([1, 3, 3, 1])
    for i in range(int(g['num_resample'])):
        p = f'r{i}_'
        kw = dict(g[p + 'spec']['kw'])
        f = bil if kw.pop('f', False) else None
        y = conv2d_resample.conv2d_resample(torch.tensor(g[p + 'x']), torch.tensor(g[p + 'w']), f=f, **kw)
        np.testing.assert_allclose(y.numpy(), g[p + 'y'], rtol=1e-5, atol=1e-5, err_msg=str(g[p + 'spec']))
    a, b, c = t64(g['fma_a'], True), t64(g['fma_b'], True), t64(g['fma_c'], True)
    y = fma.fma(a, b, c)
    np.testing.assert_allclose(y.detach().numpy(), g['fma_y'], rtol=1e-13)
    da, db, dc = torch.autograd.grad(y, [a, b, c], t64(g['fma_dy']))
    for got, key in ((da, 'fma_da'), (db, 'fma_db'), (dc, 'fma_dc')):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-12, atol=1e-12)
    for j in range(int(g['num_sf'])):
        sp = g[f'sf{j}_spec']
        np.testing.assert_array_equal(upfirdn2d.setup_filter(sp['f'], **sp['kw']).numpy(), g[f'sf{j}'])


def test_argument_errors_are_assertions():
    x = torch.zeros(1, 2, 4, 4)
    with pytest.raises(AssertionError):
        bias_act.bias_act(x, torch.zeros(3))            # bias length mismatch
    with pytest.raises(AssertionError):
        bias_act.bias_act(x, clamp=-1.0)
    with pytest.raises(AssertionError):
        upfirdn2d.upfirdn2d(x, None, up=0)
    with pytest.raises(AssertionError):
        upfirdn2d.upfirdn2d(x, torch.ones(9), padding=0)  # filter larger than padded image
    with pytest.raises(AssertionError):
        filtered_lrelu.filtered_lrelu(x, gain=-1)
    with pytest.raises(KeyError):
        bias_act.bias_act(x, act='nope')


@pytest.mark.parametrize('stride,padding', [(1, 1), (2, 1), (1, 0)])
def test_conv2d_gradfix_closed_nodes_match_the_library_graph_to_second_order(stride, padding):
    """conv2d_gradfix.closed_nodes(): value, input gradient and the gradients of an R1-style penalty on that gradient (with respect to the
    weight AND the input) against torch.nn.functional.conv2d's own graph; outside the scope, with a bias or with groups the call is
    F.conv2d itself."""
    import torch.nn.functional as F
    from torch_utils.ops import conv2d_gradfix
    torch.manual_seed(0)
    x = torch.randn(2, 3, 9, 8, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(4, 3, 3, 3, dtype=torch.float64) * 0.3).requires_grad_(True)

    def r1(fn):
        y = fn(x, w)
        (g,) = torch.autograd.grad(y.tanh().sum(), [x], create_graph=True)
        gw, gx = torch.autograd.grad(g.square().sum(), [w, x])
        return y.detach(), g.detach(), gw, gx

    with conv2d_gradfix.closed_nodes():
        assert conv2d_gradfix.enabled
        y = conv2d_gradfix.conv2d(x, w, stride=stride, padding=padding)
        assert type(y.grad_fn).__name__ == '_ConvBackward'
        got = r1(lambda x, w: conv2d_gradfix.conv2d(x, w, stride=stride, padding=padding))
        assert type(conv2d_gradfix.conv2d(x, w, bias=torch.zeros(4, dtype=torch.float64)).grad_fn).__name__ != '_ConvBackward'
    assert not conv2d_gradfix.enabled
    assert type(conv2d_gradfix.conv2d(x, w).grad_fn).__name__ != '_ConvBackward'
    want = r1(lambda x, w: F.conv2d(x, w, stride=stride, padding=padding))
    for a, b, name in zip(got, want, ['y', 'dx', 'd penalty / dw', 'd penalty / dx']):
        assert float((a - b).abs().max()) <= 1e-10 * float(b.abs().max()) + 1e-12, name
