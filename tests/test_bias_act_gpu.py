"""HIP bias_act (through the Python API -> ctypes -> C ABI) vs the CPU oracle and the golden
fixtures: forward, first- and second-order gradients, dtypes, layouts. Tolerances: float32
1e-5 relative (north star: 1e-3), float16 2e-3, bfloat16 1.6e-2 (one rounding of the output)."""

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

from torch_utils.ops import bias_act

DEV = 'cuda'
# float64: alpha/gain/clamp cross the C ABI as float32 (as in the reference plugin), hence 1e-6.
TOL = {torch.float32: dict(rtol=1e-5, atol=1e-6), torch.float64: dict(rtol=1e-6, atol=1e-7),
       torch.float16: dict(rtol=2e-3, atol=2e-3), torch.bfloat16: dict(rtol=1.6e-2, atol=1.6e-2)}


def dev(a, dtype, grad=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV, requires_grad=grad)


def host(t):
    return t.detach().to(torch.float64).cpu().numpy()


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_golden_all_activations_fwd_bwd_bwd2(dtype, oracle):
    g = load_golden('bias_act')
    tol = TOL[dtype]
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        x = dev(g[p + 'x'], dtype, True)
        b = dev(g[p + 'b'], dtype, True) if p + 'b' in g else None
        y = bias_act.bias_act(x, b, dim=sp['dim'], act=sp['act'], alpha=sp['alpha'], gain=sp['gain'], clamp=sp['clamp'])
        np.testing.assert_allclose(host(y), g[p + 'y'], err_msg=str(sp), **tol)
        grads = torch.autograd.grad(y, [x] + ([b] if b is not None else []), dev(g[p + 'dy'], dtype), create_graph=True)
        mask = np.ones_like(g[p + 'dx'], dtype=bool)
        if sp['clamp'] is not None:
            # clamp boundary decided on the float32 forward output: skip elements within rounding of it
            mask = np.abs(np.abs(g[p + 'y']) - sp['clamp']) > 1e-5
        np.testing.assert_allclose(host(grads[0])[mask], g[p + 'dx'][mask], err_msg='dx ' + str(sp), **tol)
        if b is not None:
            np.testing.assert_allclose(host(grads[1]), g[p + 'db'], rtol=tol['rtol'] * 20, atol=tol['atol'] * 200, err_msg='db ' + str(sp))
        if p + 'd_x' in g and grads[0].requires_grad:
            d_x = torch.autograd.grad(grads[0], x, dev(g[p + 'ddx'], dtype), allow_unused=True)[0]
            if d_x is not None:
                np.testing.assert_allclose(host(d_x)[mask], g[p + 'd_x'][mask], rtol=tol['rtol'] * 10, atol=tol['atol'] * 10, err_msg='d_x ' + str(sp))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape,dim', [([640, 1024], 1), ([1, 128, 32, 48], 1), ([2, 64, 16, 36, 64], 1),
                                       ([1, 512, 20, 3, 4], 1), ([3, 5, 7, 9, 11], 1), ([3, 5, 7, 9, 11], 3), ([4099], 0)])
def test_model_shapes_vs_oracle(dtype, shape, dim, oracle):
    rs = np.random.RandomState(1)
    x = dev(rs.randn(*shape), dtype)
    b = dev(rs.randn(shape[dim]), dtype)
    for act, clamp, gain in (('lrelu', 256, None), ('lrelu', 0.5, 1.0), ('linear', 256, None), ('linear', None, np.sqrt(2))):
        y = bias_act.bias_act(x, b, dim=dim, act=act, clamp=clamp, gain=gain)
        ref = oracle.bias_act(host(x), host(b), dim=dim, act=act, clamp=clamp, gain=gain)
        assert y.dtype == dtype and y.shape == x.shape
        np.testing.assert_allclose(host(y), ref, err_msg=f'{act} {clamp} {gain}', **TOL[dtype])
    y = bias_act.bias_act(x, None, act='lrelu')
    np.testing.assert_allclose(host(y), oracle.bias_act(host(x), None, act='lrelu'), **TOL[dtype])


def test_channels_last_and_noncontiguous_inputs(oracle):
    rs = np.random.RandomState(2)
    x = dev(rs.randn(2, 8, 6, 10), torch.float32)
    b = dev(rs.randn(8), torch.float32)
    ref = oracle.bias_act(host(x), host(b), act='lrelu', clamp=1.0)
    xcl = x.contiguous(memory_format=torch.channels_last)
    y = bias_act.bias_act(xcl, b, act='lrelu', clamp=1.0)
    assert y.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(host(y), ref, rtol=1e-5, atol=1e-6)
    big = dev(rs.randn(2, 8, 12, 10), torch.float32)
    view = big[:, :, 3:9, :]          # center_crop-style view (generator_lres.py:131-155)
    y = bias_act.bias_act(view, b, act='lrelu', clamp=1.0)
    np.testing.assert_allclose(host(y), oracle.bias_act(host(view), host(b), act='lrelu', clamp=1.0), rtol=1e-5, atol=1e-6)
    odd = dev(rs.randn(1, 3, 5, 7), torch.float32)[:, :, :, 1:]   # misaligned base pointer -> scalar kernel
    y = bias_act.bias_act(odd, None, act='relu')
    np.testing.assert_allclose(host(y), oracle.bias_act(host(odd), None, act='relu'), rtol=1e-5, atol=1e-6)


def test_identity_and_empty():
    x = torch.randn(2, 3, 4, device=DEV)
    assert bias_act.bias_act(x, act='linear').data_ptr() == x.data_ptr()   # no launch for a no-op
    e = torch.empty(0, 3, device=DEV)
    assert bias_act.bias_act(e, torch.zeros(3, device=DEV), act='lrelu').shape == e.shape


def test_large_stream_linearity_and_idempotence():
    """Full-size property checks (lres-G largest activation, [1,64,128,36,64] = 18.9M elements):
    relu is idempotent; linear+bias is affine; clamp bounds hold exactly."""
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(1, 64, 128, 36, 64, device=DEV, generator=g)
    b = torch.randn(64, device=DEV, generator=g)
    r1 = bias_act.bias_act(x, None, act='relu', gain=1)
    assert torch.equal(bias_act.bias_act(r1, None, act='relu', gain=1), r1)
    lin = bias_act.bias_act(x, b, act='linear', gain=2.0)
    assert torch.allclose(lin, (x + b.view(1, -1, 1, 1, 1)) * 2.0, rtol=1e-6, atol=1e-6)
    c = bias_act.bias_act(x, b, act='lrelu', clamp=0.75)
    assert float(c.abs().max()) == 0.75
    ref = torch.nn.functional.leaky_relu(x + b.view(1, -1, 1, 1, 1), 0.2) * np.sqrt(2)
    assert torch.allclose(c, ref.clamp(-0.75, 0.75), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape,act,clamp', [([64, 64, 36, 64], 'lrelu', 256.0), ([16, 128, 18, 32], 'lrelu', None), ([8, 8, 64, 64], 'relu', 0.5),
                                             ([4, 512, 9, 16], 'tanh', None), ([32, 64, 16, 16], 'linear', 1.0)])
def test_fused_bias_gradient_channels_last(dtype, shape, act, clamp, oracle, monkeypatch):
    """dx + bias gradient from one pass (lvg_bias_act_grad_bias): dx is bit-identical to the two-pass form, db is the sum of the STORED
    dx over the pixels (float64 reduction as the reference value; oracle for dx)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
    b = (0.3 * torch.randn(shape[1], generator=g)).to(dtype).to(DEV)
    dy = torch.randn(shape, generator=g).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(bias_act, 'FUSED_BIAS_GRAD', fused)
        xq, bq = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = bias_act.bias_act(xq, bq, act=act, clamp=clamp)
        out[fused] = torch.autograd.grad(y, [xq, bq], dy)
    (dx_f, db_f), (dx_p, db_p) = out[True], out[False]
    assert dx_f.is_contiguous(memory_format=torch.channels_last) and torch.equal(dx_f, dx_p)
    ref_db = host(dx_f).sum(axis=(0, 2, 3))
    scale = np.abs(ref_db).max()
    tol = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]        # one rounding of the result to dtype
    assert np.abs(host(db_f) - ref_db).max() <= tol * scale + 1e-6
    assert np.abs(host(db_p) - ref_db).max() <= 4 * tol * scale + 1e-6                    # the tensor reduction accumulates in dtype order
    ydet = bias_act.bias_act(x, b, act=act, clamp=clamp)
    ref_dx = oracle.bias_act(host(dy), None, dim=1, act=act, clamp=clamp, grad=1, xref=host(x), yref=host(ydet))
    np.testing.assert_allclose(host(dx_f), ref_dx, **TOL[dtype])
    # under create_graph the gradient stays a differentiable function of dy (two-launch form), with the same values
    monkeypatch.setattr(bias_act, 'FUSED_BIAS_GRAD', True)
    xq, bq = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    dyq = dy.clone().requires_grad_(True)
    y = bias_act.bias_act(xq, bq, act=act, clamp=clamp)
    dx_g, db_g = torch.autograd.grad(y, [xq, bq], dyq, create_graph=True)
    assert db_g.requires_grad and torch.equal(dx_g, dx_f)
    assert np.abs(host(db_g) - ref_db).max() <= 4 * tol * scale + 1e-6
