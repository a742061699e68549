"""Generate tests/golden/*.npz from the REFERENCE's own Python `impl='ref'` path on CPU.

Run in the build container only (needs the read-only reference checkout):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [/root/reference]

Nothing is copied out of the reference: it is imported, called on small seeded inputs, and the
inputs + outputs are stored. The fixtures pin oracle/ (tests/test_oracle_golden.py) and are a
second checker for the HIP path (tests/test_*_gpu.py). The reference ships no golden vectors
of its own, so these are "outputs of the reference itself run here"."""

import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402
from torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu, conv2d_resample, fma  # noqa: E402

assert os.path.realpath(bias_act.__file__).startswith(os.path.realpath(REF)), 'must import the reference, not this repo'
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def t(a, grad=False, dtype=torch.float64):
    return torch.tensor(np.asarray(a), dtype=dtype, requires_grad=grad)


def gen_bias_act():
    rs = np.random.RandomState(100)
    cases = {}
    idx = 0
    specs = []
    for act in bias_act.activation_funcs:
        specs.append(dict(act=act, shape=[2, 3, 4, 5], dim=1, bias=True, alpha=None, gain=None, clamp=None))
        specs.append(dict(act=act, shape=[3, 7], dim=1, bias=True, alpha=None, gain=1.5, clamp=0.7))
    specs += [
        dict(act='lrelu', shape=[2, 4, 3, 3, 5], dim=1, bias=True, alpha=0.2, gain=np.sqrt(2), clamp=256),
        dict(act='lrelu', shape=[2, 4, 3, 3, 5], dim=1, bias=True, alpha=0.2, gain=np.sqrt(2), clamp=0.4),
        dict(act='linear', shape=[1, 3, 4, 3, 6], dim=1, bias=True, alpha=None, gain=None, clamp=0.9),
        dict(act='linear', shape=[2, 5, 4], dim=2, bias=True, alpha=None, gain=2.0, clamp=None),
        dict(act='lrelu', shape=[6, 5], dim=0, bias=True, alpha=0.1, gain=None, clamp=None),
        dict(act='relu', shape=[11], dim=0, bias=False, alpha=None, gain=None, clamp=None),
        dict(act='lrelu', shape=[2, 3, 4, 5], dim=1, bias=False, alpha=None, gain=1.0, clamp=None),
    ]
    for sp in specs:
        x = rs.randn(*sp['shape']) * 1.5
        b = rs.randn(sp['shape'][sp['dim']]) if sp['bias'] else None
        dy = rs.randn(*sp['shape'])
        ddx = rs.randn(*sp['shape'])
        xt, bt = t(x, True), (t(b, True) if b is not None else None)
        y = bias_act.bias_act(xt, bt, dim=sp['dim'], act=sp['act'], alpha=sp['alpha'], gain=sp['gain'], clamp=sp['clamp'], impl='ref')
        grads = torch.autograd.grad(y, [xt] + ([bt] if bt is not None else []), t(dy), create_graph=True)
        dx = grads[0]
        db = grads[1] if bt is not None else None
        # second order: derivative of <dx, ddx> w.r.t. x (what BiasActCudaGrad.backward's grad=2 call returns)
        if dx.requires_grad:
            d_x = torch.autograd.grad(dx, xt, t(ddx), allow_unused=True)[0]
        else:
            d_x = None
        pre = f'c{idx}_'
        cases[pre + 'x'], cases[pre + 'dy'], cases[pre + 'ddx'] = x, dy, ddx
        if b is not None:
            cases[pre + 'b'] = b
            cases[pre + 'db'] = db.detach().numpy()
        cases[pre + 'y'] = y.detach().numpy()
        cases[pre + 'dx'] = dx.detach().numpy()
        if d_x is not None:
            cases[pre + 'd_x'] = d_x.numpy()
        cases[pre + 'spec'] = np.array(repr(dict(sp, gain=None if sp['gain'] is None else float(sp['gain']))))
        idx += 1
    cases['num_cases'] = np.array(idx)
    np.savez_compressed(os.path.join(OUT, 'bias_act.npz'), **cases)
    print('bias_act:', idx, 'cases')


def kaiser(taps, cutoff, width, fs):
    import scipy.signal
    return scipy.signal.firwin(numtaps=taps, cutoff=cutoff, width=width, fs=fs)


def gen_upfirdn2d():
    rs = np.random.RandomState(200)
    bil = np.array([0.125, 0.375, 0.375, 0.125])
    k12 = kaiser(12, 0.45, 0.3, 2.0)
    k24 = kaiser(24, 0.22, 0.15, 2.0)
    asym6 = np.array([0.015, -0.08, 0.33, 0.81, 0.45, -0.12])
    specs = [
        # (name, x shape, f, kwargs, entry) -- entry in {'upfirdn2d','upsample2d','downsample2d','filter2d'}
        ('up2_bilinear_sep', [2, 3, 5, 6], bil, dict(up=2), 'upsample2d'),
        ('down2_bilinear_sep', [2, 3, 8, 10], bil, dict(down=2), 'downsample2d'),
        ('temporal_up2', [1, 4, 6, 7], bil[:, None], dict(up=(1, 2), padding=[0, 0, 2, 1], gain=2), 'upfirdn2d'),
        ('temporal_down2', [1, 4, 12, 5], bil[:, None], dict(down=(1, 2), padding=[0, 0, 1, 1]), 'upfirdn2d'),
        ('temporal_down2_k12_latent', [1, 6, 30, 1], k12[:, None], dict(down=(1, 2), padding=[0, 0, 5, 5]), 'upfirdn2d'),
        ('temporal_negpad', [1, 3, 14, 4], bil[:, None], dict(down=(1, 2), padding=[0, 0, -1, -2]), 'upfirdn2d'),
        ('kaiser12_up2', [2, 2, 9, 11], k12 * 1.0, dict(up=2, padding=[4, 3, 4, 3], gain=4), 'upfirdn2d'),
        ('kaiser24_up4', [1, 2, 8, 9], k24 * 1.0, dict(up=4, padding=[9, 6, 9, 6], gain=16), 'upfirdn2d'),
        ('kaiser12_down2', [1, 2, 20, 22], k12 * 1.0, dict(down=2, padding=[3, 3, 3, 3]), 'upfirdn2d'),
        ('kaiser24_down4', [1, 2, 40, 36], k24 * 1.0, dict(down=4, padding=[6, 6, 6, 6]), 'upfirdn2d'),
        ('f2d_4x4_down2', [1, 3, 10, 12], np.outer(bil, bil) * 8, dict(down=2, padding=1), 'upfirdn2d'),
        ('f2d_4x4_blur_pad2', [1, 3, 7, 9], np.outer(bil, bil), dict(padding=2), 'upfirdn2d'),
        ('flip_asym_sep', [1, 2, 9, 8], asym6, dict(up=2, padding=3, flip_filter=True), 'upfirdn2d'),
        ('noflip_asym_sep', [1, 2, 9, 8], asym6, dict(down=2, padding=3, flip_filter=False), 'upfirdn2d'),
        ('flip_asym_2d', [1, 2, 6, 7], np.outer(asym6[:3], asym6[1:5]), dict(up=(2, 1), down=(1, 2), padding=[2, 1, 3, 0], flip_filter=True), 'upfirdn2d'),
        ('identity_f_none', [2, 2, 4, 5], None, dict(up=2, padding=[0, 1, 1, 0], gain=3), 'upfirdn2d'),
        ('generic_up3_down2_5x3', [1, 2, 7, 6], rs.randn(5, 3), dict(up=3, down=2, padding=[2, 3, 1, 4]), 'upfirdn2d'),
        ('tiny', [1, 1, 1, 1], bil, dict(up=4, padding=[3, 0, 3, 0]), 'upfirdn2d'),
        ('crop_both', [1, 2, 12, 12], bil, dict(padding=[-2, -1, -3, 0]), 'upfirdn2d'),
        ('filter2d_sep', [1, 2, 6, 6], bil, dict(), 'filter2d'),
        ('sres_bilinear_up4', [1, 2, 5, 6], np.array([1, 3, 5, 7, 7, 5, 3, 1]) / 32.0, dict(up=4), 'upsample2d'),
    ]
    cases = {}
    for i, (name, shape, f, kw, entry) in enumerate(specs):
        x = rs.randn(*shape)
        dy_seed = rs.randint(1 << 30)
        xt = t(x, True)
        ft = None if f is None else torch.tensor(f, dtype=torch.float32)
        y = getattr(upfirdn2d, entry)(xt, ft, impl='ref', **kw)
        dy = np.random.RandomState(dy_seed).randn(*y.shape)
        dx = torch.autograd.grad(y, xt, t(dy))[0]
        pre = f'c{i}_'
        cases[pre + 'x'] = x
        if f is not None:
            cases[pre + 'f'] = np.asarray(f, dtype=np.float32)
        cases[pre + 'y'], cases[pre + 'dy'], cases[pre + 'dx'] = y.detach().numpy(), dy, dx.numpy()
        cases[pre + 'spec'] = np.array(repr(dict(name=name, entry=entry, kw=kw)))
    cases['num_cases'] = np.array(len(specs))
    np.savez_compressed(os.path.join(OUT, 'upfirdn2d.npz'), **cases)
    print('upfirdn2d:', len(specs), 'cases')


def gen_filtered_lrelu():
    rs = np.random.RandomState(300)
    k12 = kaiser(12, 0.45, 0.3, 2.0).astype(np.float32)
    k24 = kaiser(24, 0.22, 0.15, 2.0).astype(np.float32)
    bil = np.array([0.125, 0.375, 0.375, 0.125], dtype=np.float32)
    specs = [
        ('up2_down2_k12', [2, 3, 14, 17], k12, k12, dict(up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=0.2, clamp=256)),
        ('up2_down2_k12_smallclamp', [1, 2, 14, 17], k12, k12, dict(up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=0.2, clamp=0.3)),
        ('up4_down2_k24_k12_negpad', [1, 2, 16, 20], k24, k12, dict(up=4, down=2, padding=[-6, -9, -6, -9], gain=np.sqrt(2), slope=0.2, clamp=256)),
        ('up2_down4_k12_k24', [1, 2, 24, 26], k12, k24, dict(up=2, down=4, padding=[11, 10, 11, 10], gain=1.3, slope=0.2, clamp=None)),
        ('crop_final', [1, 2, 24, 30], k12, k12, dict(up=2, down=2, padding=[-3, -4, -3, -4], gain=np.sqrt(2), slope=0.2, clamp=256)),
        ('torgb_1x1', [2, 3, 6, 7], None, None, dict(up=1, down=1, padding=0, gain=1, slope=1, clamp=256)),
        ('up1_down1_lrelu', [2, 3, 6, 7], None, None, dict(up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=0.5)),
        ('bilinear_up2_down2_flip', [1, 2, 7, 8], bil, bil, dict(up=2, down=2, padding=[3, 2, 3, 2], gain=1.0, slope=0.1, clamp=None, flip_filter=True)),
        ('generic_up3_down1', [1, 2, 6, 5], np.array([0.2, 0.5, 0.3, 0.4, 0.1, 0.25], dtype=np.float32), None, dict(up=3, down=1, padding=[2, 3, 2, 3], gain=1.1, slope=0.3, clamp=0.8)),
        ('f2d_up2_down2', [1, 2, 6, 6], np.outer(bil, bil).astype(np.float32), np.outer(bil, bil).astype(np.float32), dict(up=2, down=2, padding=3, gain=1.2, slope=0.2, clamp=None)),
    ]
    cases = {}
    for i, (name, shape, fu, fd, kw) in enumerate(specs):
        x = rs.randn(*shape)
        b = rs.randn(shape[1]) * 0.5
        xt, bt = t(x, True), t(b, True)
        fut = None if fu is None else torch.tensor(fu)
        fdt = None if fd is None else torch.tensor(fd)
        y = filtered_lrelu.filtered_lrelu(xt, fut, fdt, bt, impl='ref', **kw)
        dy = rs.randn(*y.shape)
        dx, db = torch.autograd.grad(y, [xt, bt], t(dy))
        pre = f'c{i}_'
        cases[pre + 'x'], cases[pre + 'b'], cases[pre + 'dy'] = x, b, dy
        if fu is not None:
            cases[pre + 'fu'] = fu
        if fd is not None:
            cases[pre + 'fd'] = fd
        cases[pre + 'y'], cases[pre + 'dx'], cases[pre + 'db'] = y.detach().numpy(), dx.numpy(), db.numpy()
        kw2 = dict(kw, gain=float(kw['gain']))
        cases[pre + 'spec'] = np.array(repr(dict(name=name, kw=kw2)))
    cases['num_cases'] = np.array(len(specs))
    np.savez_compressed(os.path.join(OUT, 'filtered_lrelu.npz'), **cases)
    print('filtered_lrelu:', len(specs), 'cases')


def gen_misc():
    rs = np.random.RandomState(400)
    cases = {}
    # conv2d_resample branches used by discriminator_sres.Conv2dLayer (up=1, down in {1,2}), plus up=2.
    bil2d = upfirdn2d.setup_filter([1, 3, 3, 1])
    specs = [
        ('k3_plain', [2, 4, 9, 9], [5, 4, 3, 3], dict(padding=1)),
        ('k3_down2', [2, 4, 10, 10], [5, 4, 3, 3], dict(down=2, padding=1, f=True)),
        ('k1_down2', [2, 4, 10, 10], [5, 4, 1, 1], dict(down=2, f=True)),
        ('k1_plain', [2, 4, 7, 7], [5, 4, 1, 1], dict()),
        ('k3_up2', [2, 4, 6, 6], [5, 4, 3, 3], dict(up=2, padding=1, f=True)),
        ('k1_up2', [2, 4, 6, 6], [5, 4, 1, 1], dict(up=2, f=True)),
        ('k3_noflip', [1, 2, 8, 8], [3, 2, 3, 3], dict(padding=1, flip_weight=False)),
    ]
    for i, (name, xs, ws, kw) in enumerate(specs):
        x, w = rs.randn(*xs), rs.randn(*ws) * 0.3
        kw2 = dict(kw)
        f = bil2d.double() if kw2.pop('f', False) else None
        # the reference ref-path requires float32 filters; run this one in float32
        y = conv2d_resample.conv2d_resample(t(x, dtype=torch.float32), t(w, dtype=torch.float32), f=None if f is None else f.float(), **kw2)
        pre = f'r{i}_'
        cases[pre + 'x'], cases[pre + 'w'], cases[pre + 'y'] = x.astype(np.float32), w.astype(np.float32), y.numpy()
        cases[pre + 'spec'] = np.array(repr(dict(name=name, kw=kw)))
    cases['num_resample'] = np.array(len(specs))
    a, b, c = rs.randn(3, 1, 5), rs.randn(1, 4, 5), rs.randn(3, 4, 1)
    at, bt_, ct = t(a, True), t(b, True), t(c, True)
    y = fma.fma(at, bt_, ct)
    dy = rs.randn(*y.shape)
    da, db, dc = torch.autograd.grad(y, [at, bt_, ct], t(dy))
    cases.update(fma_a=a, fma_b=b, fma_c=c, fma_y=y.detach().numpy(), fma_dy=dy, fma_da=da.numpy(), fma_db=db.numpy(), fma_dc=dc.numpy())
    # setup_filter behaviours
    sf = {}
    for j, (f, kw) in enumerate([([1, 3, 3, 1], {}), ([1, 3, 3, 1], dict(separable=True)), (list(range(1, 9)), {}),
                                 ([1, 2, 3], dict(flip_filter=True, gain=2)), (None, {}), (3.0, dict(normalize=False)),
                                 ([[1, 2], [3, 4]], dict(flip_filter=True, gain=4))]):
        sf[f'sf{j}'] = upfirdn2d.setup_filter(f, **kw).numpy()
        sf[f'sf{j}_spec'] = np.array(repr(dict(f=f, kw=kw)))
    sf['num_sf'] = np.array(7)
    cases.update(sf)
    np.savez_compressed(os.path.join(OUT, 'misc_ops.npz'), **cases)
    print('misc:', len(specs), 'resample cases + fma + setup_filter')


if __name__ == '__main__':
    gen_bias_act()
    gen_upfirdn2d()
    gen_filtered_lrelu()
    gen_misc()
