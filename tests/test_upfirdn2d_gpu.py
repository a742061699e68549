"""HIP upfirdn2d vs the CPU oracle / golden fixtures: every case of SURVEY.md Appendix G row
"upfirdn2d", forward + backward (+ double backward), dtypes, strided inputs."""

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

from torch_utils.ops import upfirdn2d

DEV = 'cuda'
TOL = {torch.float32: dict(rtol=2e-5, atol=2e-6), torch.float64: dict(rtol=1e-11, atol=1e-12),
       torch.float16: dict(rtol=4e-3, atol=4e-3), torch.bfloat16: dict(rtol=2e-2, atol=2e-2)}


def dev(a, dtype, grad=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV, requires_grad=grad)


def host(t):
    return t.detach().to(torch.float64).cpu().numpy()


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_golden_forward_backward(dtype):
    g = load_golden('upfirdn2d')
    for i in range(int(g['num_cases'])):
        p = f'c{i}_'
        sp = g[p + 'spec']
        f = torch.tensor(g[p + 'f'], device=DEV) if p + 'f' in g else None
        x = dev(g[p + 'x'], dtype, True)
        y = getattr(upfirdn2d, sp['entry'])(x, f, **sp['kw'])
        tol = TOL[dtype] if dtype == torch.float64 or sp['kw'].get('gain', 1) in (1, 2, 4, 16) else dict(rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(host(y), g[p + 'y'], err_msg=str(sp), **tol)
        dy = dev(g[p + 'dy'], dtype, True)
        dx = torch.autograd.grad(y, x, dy, create_graph=True)[0]
        np.testing.assert_allclose(host(dx), g[p + 'dx'], err_msg='dx ' + str(sp), **tol)
        # double backward: dx is linear in dy, so d<dx, v>/d(dy) is the forward op applied to v
        v = torch.randn_like(x)
        (d_dy,) = torch.autograd.grad(dx, dy, v)
        fwd_v = getattr(upfirdn2d, sp['entry'])(v, f, **sp['kw'])
        np.testing.assert_allclose(host(d_dy), host(fwd_v), err_msg='double backward ' + str(sp), **tol)


SHAPES = [
    # name, x shape, filter (1-D unless given 2-D), kwargs, entry
    ('lres_spatial_up2', [1, 512, 18, 32], [0.125, 0.375, 0.375, 0.125], dict(up=2), 'upsample2d'),
    ('lres_spatial_up2_tiny', [1, 1024, 3, 4], [0.125, 0.375, 0.375, 0.125], dict(up=2), 'upsample2d'),
    ('lres_D_down2', [1, 256, 64, 64], [0.125, 0.375, 0.375, 0.125], dict(down=2), 'downsample2d'),
    ('lres_temporal_up2', [1, 64, 80, 144], np.array([0.125, 0.375, 0.375, 0.125])[:, None], dict(up=(1, 2), padding=[0, 0, 2, 1], gain=2), 'upfirdn2d'),
    ('lres_temporal_down2', [1, 32, 128, 256], np.array([0.125, 0.375, 0.375, 0.125])[:, None], dict(down=(1, 2), padding=[0, 0, 1, 1]), 'upfirdn2d'),
    ('wide_plane', [1, 3, 70, 300], [0.125, 0.375, 0.375, 0.125], dict(down=2), 'downsample2d'),
    ('sresD_4x4_down2', [1, 16, 256, 256], np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0, dict(down=2, padding=1), 'upfirdn2d'),
    ('sresD_4x4_blur', [1, 16, 128, 128], np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0, dict(padding=2), 'upfirdn2d'),
    ('generic_up3', [2, 3, 17, 19], [0.1, 0.2, 0.4, 0.2, 0.1], dict(up=3, down=2, padding=[2, 1, 3, 0]), 'upfirdn2d'),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('case', SHAPES, ids=[s[0] for s in SHAPES])
def test_model_shapes_vs_oracle(case, dtype, oracle):
    name, shape, f, kw, entry = case
    rs = np.random.RandomState(3)
    x = dev(rs.randn(*shape), dtype)
    ft = torch.tensor(np.asarray(f, dtype=np.float32), device=DEV)
    y = getattr(upfirdn2d, entry)(x, ft, **kw)
    ref = getattr(oracle, entry)(host(x), np.asarray(f, dtype=np.float32), **kw)
    assert y.dtype == dtype and tuple(y.shape) == ref.shape
    np.testing.assert_allclose(host(y), ref, err_msg=name, **TOL[dtype])


def test_kaiser_filters_sres_prep_cond(oracle):
    import scipy.signal
    rs = np.random.RandomState(4)
    k12 = scipy.signal.firwin(numtaps=12, cutoff=0.45, width=0.3, fs=2.0).astype(np.float32)
    k24 = scipy.signal.firwin(numtaps=24, cutoff=0.22, width=0.15, fs=2.0).astype(np.float32)
    x = dev(rs.randn(8, 27, 44, 46), torch.float32)
    for f, kw in ((k12, dict(up=2, padding=[4, 3, 4, 3], gain=4)), (k24, dict(up=4, padding=[9, 6, 9, 6], gain=16)),
                  (k12, dict(down=2, padding=[3, 3, 3, 3])), (k24, dict(down=4, padding=[6, 6, 6, 6])),
                  (k12, dict(down=2, padding=[-2, -3, -1, -4]))):
        y = upfirdn2d.upfirdn2d(x, torch.tensor(f, device=DEV), **kw)
        np.testing.assert_allclose(host(y), oracle.upfirdn2d(host(x), f, **kw), err_msg=str(kw), rtol=2e-5, atol=2e-6)


def test_strided_latent_view_and_channels_last(oracle):
    """Permuted latents `(n t) c -> n c t 1` (time stride = C) go through without a copy
    (generator_lres.py:277,478); channels_last keeps its format (upfirdn2d.cpp:38)."""
    import scipy.signal
    rs = np.random.RandomState(5)
    k12 = scipy.signal.firwin(numtaps=12, cutoff=0.45, width=0.3, fs=2.0).astype(np.float32)
    lat = dev(rs.randn(1, 650, 96), torch.float32)            # [n, t, c]
    view = lat.permute(0, 2, 1).unsqueeze(3)                   # [n, c, t, 1], stride(c)=1
    assert not view.is_contiguous()
    f = torch.tensor(k12[:, None], device=DEV)
    y = upfirdn2d.upfirdn2d(view, f, down=(1, 2), padding=[0, 0, 5, 5])
    ref = oracle.upfirdn2d(host(view), k12[:, None], down=(1, 2), padding=[0, 0, 5, 5])
    np.testing.assert_allclose(host(y), ref, rtol=2e-5, atol=2e-6)
    x = dev(rs.randn(2, 6, 9, 11), torch.float32).contiguous(memory_format=torch.channels_last)
    bil = torch.tensor([0.125, 0.375, 0.375, 0.125], device=DEV)
    y = upfirdn2d.upsample2d(x, bil, up=2)
    assert y.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(host(y), oracle.upsample2d(host(x), [0.125, 0.375, 0.375, 0.125], up=2), rtol=2e-5, atol=2e-6)


def test_round_trip_and_linearity_at_full_size():
    """BASELINE-size properties: [1,8192,18,32] -> 36x64. Linearity; DC gain 1 for a
    normalised filter with upsample gain; adjointness <up(x), y> == <x, up^T(y)> via autograd."""
    bil = torch.tensor([0.125, 0.375, 0.375, 0.125], device=DEV)
    g = torch.Generator(device=DEV).manual_seed(1)
    x1 = torch.randn(1, 8192, 18, 32, device=DEV, generator=g)
    x2 = torch.randn(1, 8192, 18, 32, device=DEV, generator=g)
    u1, u2 = upfirdn2d.upsample2d(x1, bil), upfirdn2d.upsample2d(x2, bil)
    u12 = upfirdn2d.upsample2d(x1 * 0.5 + x2 * 2.0, bil)
    assert u1.shape == (1, 8192, 36, 64)
    assert torch.allclose(u12, u1 * 0.5 + u2 * 2.0, rtol=1e-5, atol=1e-5)
    ones = torch.ones(1, 4, 18, 32, device=DEV)
    interior = upfirdn2d.upsample2d(ones, bil)[:, :, 2:-2, 2:-2]
    assert torch.allclose(interior, torch.ones_like(interior), atol=1e-6)
    xr = x1[:, :64].clone().requires_grad_(True)
    y = upfirdn2d.upsample2d(xr, bil)
    w = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, xr, w)
    lhs = (y.detach() * w).sum().double()
    rhs = (xr.detach() * gx).sum().double()
    assert abs(float(lhs - rhs)) <= 1e-4 * max(1.0, abs(float(lhs)))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_channels_last_kernel_vs_oracle(dtype, oracle):
    """NHWC kernel (frames-layout networks keep activations channels-last)."""
    rs = np.random.RandomState(9)
    bil = [0.125, 0.375, 0.375, 0.125]
    ft = torch.tensor(bil, device=DEV)
    for shape, kind in (([6, 64, 9, 16], 'up'), ([6, 64, 18, 32], 'down'), ([3, 16, 5, 8], 'up'), ([3, 24, 8, 8], 'down'), ([2, 8, 7, 5], 'blur')):
        x = dev(rs.randn(*shape), dtype).contiguous(memory_format=torch.channels_last)
        if kind == 'up':
            y, ref = upfirdn2d.upsample2d(x, ft, up=2), oracle.upsample2d(host(x), bil, up=2)
        elif kind == 'down':
            y, ref = upfirdn2d.downsample2d(x, ft, down=2), oracle.downsample2d(host(x), bil, down=2)
        else:
            y, ref = upfirdn2d.filter2d(x, ft), oracle.upfirdn2d(host(x), bil, padding=[2, 1, 2, 1])
        assert y.is_contiguous(memory_format=torch.channels_last) and tuple(y.shape) == ref.shape
        np.testing.assert_allclose(host(y), ref, err_msg=f'{shape} {kind}', **TOL[dtype])
        # cropped view (the networks crop H/W before bias_act) keeps the NHWC path
        xc = x[:, :, 1:, :-1]
        yc = upfirdn2d.upsample2d(xc, ft, up=2)
        np.testing.assert_allclose(host(yc), oracle.upsample2d(host(xc), bil, up=2), **TOL[dtype])
