"""numpy front end of the CPU oracle (oracle/lvg_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of lvg_oracle.c. Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from long-video-gan_amd/.

Parity status: PINNED against tests/golden/ (fixtures generated from the reference's own
impl='ref' Python path by tests/golden/make_golden.py; checked by tests/test_oracle_golden.py).

All functions take/return numpy arrays. Arithmetic is float64 inside; results are returned as
float64 (callers cast/compare at the tolerance of the dtype under test)."""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liblvg_oracle.so')
_lib = None

ACT_IDS = dict(linear=1, relu=2, lrelu=3, tanh=4, sigmoid=5, elu=6, selu=7, softplus=8, swish=9)
ACT_DEFAULTS = dict(  # (def_alpha, def_gain) -- reference bias_act.py:21-31
    linear=(0, 1), relu=(0, np.sqrt(2)), lrelu=(0.2, np.sqrt(2)), tanh=(0, 1), sigmoid=(0, 1),
    elu=(0, 1), selu=(0, 1), softplus=(0, 1), swish=(0, np.sqrt(2)))


def build(force=False):
    """Compile lvg_oracle.c with gcc (seconds). Building the checker is not using it."""
    src = os.path.join(_HERE, 'lvg_oracle.c')
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(['make', '-C', _HERE, '-s'], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _f64(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a), dtype=np.float64)


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, grad=0, xref=None, yref=None, dy=None):
    """grad=0: forward. grad=1: x is dy. grad=2: x is d_dx and `dy` the first-order cotangent."""
    x = _f64(x)
    def_alpha, def_gain = ACT_DEFAULTS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    y = np.empty_like(x)
    b64 = _f64(b)
    step = int(np.prod(x.shape[dim + 1:])) if b is not None else 1
    size_b = b64.shape[0] if b is not None else 1
    xr, yr, dyv = _f64(xref), _f64(yref), _f64(dy)
    rc = lib().orc_bias_act(_dp(x), _dp(b64), _dp(xr), _dp(yr), _dp(dyv), _dp(y),
                            ctypes.c_int64(x.size), ctypes.c_int64(size_b), ctypes.c_int64(step),
                            ctypes.c_int(grad), ctypes.c_int(ACT_IDS[act]),
                            ctypes.c_double(alpha), ctypes.c_double(gain), ctypes.c_double(clamp))
    assert rc == 0, rc
    return y


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _pad4(p):
    if isinstance(p, int):
        p = [p, p]
    p = list(p)
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    return p


def _filter2d(f):
    """None -> 1x1 identity; 1-D -> outer product (separable, upfirdn2d.py:205-207)."""
    if f is None:
        return np.ones([1, 1], dtype=np.float64)
    f = _f64(f)
    if f.ndim == 1:
        f = np.outer(f, f)
    return np.ascontiguousarray(f)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    x = _f64(x)
    n, c, ih, iw = x.shape
    upx, upy = _pair(up)
    downx, downy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    f2 = _filter2d(f)
    fh, fw = f2.shape
    ow = (iw * upx + px0 + px1 - fw + downx) // downx
    oh = (ih * upy + py0 + py1 - fh + downy) // downy
    assert ow >= 1 and oh >= 1
    y = np.empty([n, c, oh, ow], dtype=np.float64)
    rc = lib().orc_upfirdn2d(_dp(x), _dp(f2), _dp(y), ctypes.c_int64(n * c), ih, iw, fh, fw,
                             upx, upy, downx, downy, px0, px1, py0, py1, int(bool(flip_filter)),
                             ctypes.c_double(gain), oh, ow)
    assert rc == 0, rc
    return y


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    """Padding rule of upfirdn2d.py:339-348."""
    upx, upy = _pair(up)
    px0, px1, py0, py1 = _pad4(padding)
    f2 = _filter2d(f)
    fh, fw = f2.shape
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1):
    """Padding rule of upfirdn2d.py:378-387."""
    downx, downy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    f2 = _filter2d(f)
    fh, fw = f2.shape
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


def sign_shape(oh, ow, down, fdh, fdw):
    """Rows and bytes-per-row of the sign plane (filtered_lrelu.cpp:87-94)."""
    sw_active = ow * down - (down - 1) + (fdw - 1)
    sh = oh * down - (down - 1) + (fdh - 1)
    return sh, ((sw_active + 15) & ~15) >> 2, sw_active


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, signs=None, sign_ofs=(0, 0), write_signs=False):
    """Returns y, or (y, s) with write_signs. `signs` (uint8 [n,c,sh,swb]) switches to READ mode."""
    x = _f64(x)
    n, c, ih, iw = x.shape
    fu2, fd2 = _filter2d(fu), _filter2d(fd)
    px0, px1, py0, py1 = _pad4(padding)
    cw = iw * up + px0 + px1 - (fu2.shape[1] - 1)
    ch = ih * up + py0 + py1 - (fu2.shape[0] - 1)
    ow = (cw - (fd2.shape[1] - 1) + (down - 1)) // down
    oh = (ch - (fd2.shape[0] - 1) + (down - 1)) // down
    y = np.empty([n, c, oh, ow], dtype=np.float64)
    mode, s, sh, swb = 0, None, 0, 0
    if signs is not None:
        mode, s = 2, np.ascontiguousarray(signs, dtype=np.uint8)
        sh, swb = s.shape[2], s.shape[3]
    elif write_signs:
        sh, swb, _ = sign_shape(oh, ow, down, fd2.shape[0], fd2.shape[1])
        mode, s = 1, np.zeros([n, c, sh, swb], dtype=np.uint8)
    clamp_v = float('inf') if clamp is None else float(clamp)
    b64 = _f64(b)
    rc = lib().orc_filtered_lrelu(
        _dp(x), _dp(fu2), _dp(fd2), _dp(b64), _dp(y),
        None if s is None else s.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
        ctypes.c_int64(n), ctypes.c_int64(c), ih, iw, fu2.shape[0], fu2.shape[1], fd2.shape[0], fd2.shape[1],
        up, down, px0, px1, py0, py1, int(sign_ofs[0]), int(sign_ofs[1]), sh, swb,
        ctypes.c_double(gain), ctypes.c_double(slope), ctypes.c_double(clamp_v), int(bool(flip_filter)), mode, oh, ow)
    assert rc == 0, rc
    return (y, s) if write_signs else y


def filtered_lrelu_act(x, gain, slope, clamp, signs=None, sign_ofs=(0, 0), write_signs=False):
    x = _f64(x).copy()
    n, c, h, w = x.shape
    mode, s, sh, swb = 0, None, 0, 0
    if signs is not None:
        mode, s = 2, np.ascontiguousarray(signs, dtype=np.uint8)
        sh, swb = s.shape[2], s.shape[3]
    elif write_signs:
        sh, swb = h, ((w + 15) & ~15) >> 2
        mode, s = 1, np.zeros([n, c, sh, swb], dtype=np.uint8)
    clamp_v = float('inf') if clamp is None else float(clamp)
    rc = lib().orc_filtered_lrelu_act(
        _dp(x), None if s is None else s.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
        ctypes.c_int64(n * c), h, w, int(sign_ofs[0]), int(sign_ofs[1]), sh, swb,
        ctypes.c_double(gain), ctypes.c_double(slope), ctypes.c_double(clamp_v), mode)
    assert rc == 0, rc
    return (x, s) if write_signs else x


def fma(a, b, c):
    a, b, c = np.broadcast_arrays(_f64(a), _f64(b), _f64(c))
    a, b, c = (np.ascontiguousarray(v) for v in (a, b, c))
    y = np.empty_like(a)
    rc = lib().orc_fma(_dp(a), _dp(b), _dp(c), _dp(y), ctypes.c_int64(a.size))
    assert rc == 0
    return y


def modconv_epilogue(z, pre=None, b=None, res=None, post=None, taps=1, shift=1, act='linear', alpha=None, gain=None, clamp=None):
    """Modulated-conv epilogue (orc_modconv_epilogue). z [frames, taps*C, H, W] (tap-major channels);
    pre / post [frames, C]; b [C]; res [frames, C, H, W]. Returns (out, ysum, mean_square)."""
    z = _f64(z)
    f, kc, h, w = z.shape
    c = kc // taps
    assert c * taps == kc
    def_alpha, def_gain = ACT_DEFAULTS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    pre64, b64, res64, post64 = _f64(pre), _f64(b), _f64(res), _f64(post)
    out = np.empty((f, c, h, w), dtype=np.float64)
    ysum = np.empty_like(out)
    msq = ctypes.c_double(0.0)
    rc = lib().orc_modconv_epilogue(_dp(z), _dp(pre64), _dp(b64), _dp(res64), _dp(post64), _dp(out), _dp(ysum), ctypes.byref(msq),
                                    ctypes.c_int64(f), ctypes.c_int(c), ctypes.c_int(h * w), ctypes.c_int(taps), ctypes.c_int64(shift),
                                    ctypes.c_int(ACT_IDS[act]), ctypes.c_double(alpha), ctypes.c_double(gain), ctypes.c_double(clamp))
    assert rc == 0, rc
    return out, ysum, float(msq.value)


def conv3d_frames(x, w, shift=1):
    """conv3d with 'same' zero padding over time-major frames (orc_conv3d_frames).
    x [frames, Ci, H, W] (frame f = t * shift + n), w [Co, Ci, kt, kh, kw] -> y [frames, Co, H, W]."""
    x, w = _f64(x), _f64(w)
    f, ci, h, wd = x.shape
    co, ci2, kt, kh, kw = w.shape
    assert ci == ci2
    y = np.empty((f, co, h, wd), dtype=np.float64)
    rc = lib().orc_conv3d_frames(_dp(x), _dp(w), _dp(y), ctypes.c_int64(f), ctypes.c_int(ci), ctypes.c_int(co), ctypes.c_int(h),
                                 ctypes.c_int(wd), ctypes.c_int(kt), ctypes.c_int(kh), ctypes.c_int(kw), ctypes.c_int64(shift))
    assert rc == 0, rc
    return y


def conv3d_frames_wgrad(x, dy, kt, kh, kw, shift=1):
    """Weight gradient of `conv3d_frames` (orc_conv3d_frames_wgrad): x [frames, Ci, H, W], dy [frames, Co, H, W]
    -> gw [Co, Ci, kt, kh, kw]."""
    x, dy = _f64(x), _f64(dy)
    f, ci, h, wd = x.shape
    co = dy.shape[1]
    gw = np.empty((co, ci, kt, kh, kw), dtype=np.float64)
    rc = lib().orc_conv3d_frames_wgrad(_dp(x), _dp(dy), _dp(gw), ctypes.c_int64(f), ctypes.c_int(ci), ctypes.c_int(co), ctypes.c_int(h),
                                       ctypes.c_int(wd), ctypes.c_int(kt), ctypes.c_int(kh), ctypes.c_int(kw), ctypes.c_int64(shift))
    assert rc == 0, rc
    return gw


def conv2d(x, w, padding=0):
    """Plain cross-correlation with symmetric zero padding (orc_conv2d): x [N, Ci, H, W], w [Co, Ci, kh, kw] -> [N, Co, H', W']."""
    x, w = _f64(x), _f64(w)
    n, ci, h, wd = x.shape
    co, ci2, kh, kw = w.shape
    assert ci == ci2
    y = np.empty((n, co, h + 2 * padding - kh + 1, wd + 2 * padding - kw + 1), dtype=np.float64)
    rc = lib().orc_conv2d(_dp(x), _dp(w), _dp(y), ctypes.c_int64(n), ctypes.c_int(ci), ctypes.c_int(co), ctypes.c_int(h), ctypes.c_int(wd),
                          ctypes.c_int(kh), ctypes.c_int(kw), ctypes.c_int(padding))
    assert rc == 0, rc
    return y


def conv2d_wgrad(x, dy, kh, kw, padding=0):
    """Weight gradient of `conv2d` (orc_conv2d_wgrad): x [N, Ci, H, W], dy [N, Co, H', W'] -> [Co, Ci, kh, kw]."""
    x, dy = _f64(x), _f64(dy)
    n, ci, h, wd = x.shape
    co = dy.shape[1]
    assert dy.shape[2:] == (h + 2 * padding - kh + 1, wd + 2 * padding - kw + 1)
    gw = np.empty((co, ci, kh, kw), dtype=np.float64)
    rc = lib().orc_conv2d_wgrad(_dp(x), _dp(dy), _dp(gw), ctypes.c_int64(n), ctypes.c_int(ci), ctypes.c_int(co), ctypes.c_int(h), ctypes.c_int(wd),
                                ctypes.c_int(kh), ctypes.c_int(kw), ctypes.c_int(padding))
    assert rc == 0, rc
    return gw


def conv2d_dgrad(dy, w, h, wd, padding=0):
    """Data gradient of `conv2d` (orc_conv2d_dgrad): dy [N, Co, H', W'], w [Co, Ci, kh, kw] -> [N, Ci, h, wd]."""
    dy, w = _f64(dy), _f64(w)
    n, co = dy.shape[:2]
    co2, ci, kh, kw = w.shape
    assert co == co2 and dy.shape[2:] == (h + 2 * padding - kh + 1, wd + 2 * padding - kw + 1)
    dx = np.empty((n, ci, h, wd), dtype=np.float64)
    rc = lib().orc_conv2d_dgrad(_dp(dy), _dp(w), _dp(dx), ctypes.c_int64(n), ctypes.c_int(ci), ctypes.c_int(co), ctypes.c_int(h), ctypes.c_int(wd),
                                ctypes.c_int(kh), ctypes.c_int(kw), ctypes.c_int(padding))
    assert rc == 0, rc
    return dx


def modconv2d_prologue(x, cond, mod, c_pad):
    """cat(x, cond) * mod with zero channels up to c_pad. x may be None. NCHW in, NCHW out."""
    cond = _f64(cond)
    n, cb, h, w = cond.shape
    x64 = _f64(x)
    ca = 0 if x is None else x64.shape[1]
    out = np.empty([n, c_pad, h, w], dtype=np.float64)
    m = _f64(mod)
    rc = lib().orc_modconv2d_prologue(_dp(x64), _dp(cond), _dp(m), _dp(out), ctypes.c_int64(n), ca, cb, c_pad, ctypes.c_int64(h * w))
    assert rc == 0, rc
    return out


def modconv2d_epilogue(y, demod, c_out):
    """y[:, :c_out] * demod. NCHW in, NCHW out."""
    y = _f64(y)
    n, c_pad, h, w = y.shape
    out = np.empty([n, c_out, h, w], dtype=np.float64)
    d = _f64(demod)
    rc = lib().orc_modconv2d_epilogue(_dp(y), _dp(d), _dp(out), ctypes.c_int64(n), c_pad, c_out, ctypes.c_int64(h * w))
    assert rc == 0, rc
    return out


def style_prep(style, w2):
    """Style side of the modulated convolution (orc_style_prep): style [T, N, Ci] (frames order), w2 [Co, Ci]
    -> (modulation [(T N), Ci], demodulation [(T N), Co])."""
    s, w2 = _f64(style), _f64(w2)
    t, n, ci = s.shape
    co = w2.shape[0]
    mod = np.empty((t * n, ci), dtype=np.float64)
    demod = np.empty((t * n, co), dtype=np.float64)
    rc = lib().orc_style_prep(_dp(s), _dp(w2), _dp(mod), _dp(demod), ctypes.c_int(t), ctypes.c_int(n), ctypes.c_int(ci), ctypes.c_int(co))
    assert rc == 0, rc
    return mod, demod


def video_to_uint8(video):
    """[N, C, T, H, W] float32 -> [N, T, H, W, C] uint8, (x * 127.5 + 128).clamp(0, 255) truncated (orc_video_to_uint8)."""
    v = np.ascontiguousarray(video, dtype=np.float32)
    n, c, t, h, w = v.shape
    out = np.empty((n, t, h, w, c), dtype=np.uint8)
    rc = lib().orc_video_to_uint8(v.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n), c, t, h, w)
    assert rc == 0
    return out


def video_from_uint8(frames, flip=None):
    """[N, T, H, W, C] uint8 -> [N, C, T, H, W] float32, 2 * x / 255 - 1; samples with flip[i] != 0 mirrored in x."""
    b = np.ascontiguousarray(frames, dtype=np.uint8)
    n, t, h, w, c = b.shape
    out = np.empty((n, c, t, h, w), dtype=np.float32)
    fl = None if flip is None else np.ascontiguousarray(flip, dtype=np.uint8)
    rc = lib().orc_video_from_uint8(b.ctypes.data_as(ctypes.c_void_p), None if fl is None else fl.ctypes.data_as(ctypes.c_void_p),
                                    out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n), c, t, h, w)
    assert rc == 0
    return out


def noise_filter_bank(noise, bank, scale=None):
    """BlurredNoise.blur (orc_noise_filter_bank; reference generator_lres.py:378-388): noise [R, L], bank [F, K], scale [F] or None ->
    [R, F, L - K + 1] float64."""
    noise, bank = _f64(noise), _f64(bank)
    r, length = noise.shape
    f, k = bank.shape
    out = np.empty((r, f, length - k + 1), dtype=np.float64)
    sc = None if scale is None else _f64(np.asarray(scale).reshape(-1))
    rc = lib().orc_noise_filter_bank(noise.ctypes.data_as(ctypes.c_void_p), bank.ctypes.data_as(ctypes.c_void_p),
                                     None if sc is None else sc.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), r, length, f, k)
    assert rc == 0
    return out


def pointwise(x, w):
    """1 x 1 convolution on channels-last pixels (orc_pointwise): x [M, Ci], w [Co, Ci] -> [M, Co] float64."""
    x, w = _f64(x), _f64(w)
    m, ci = x.shape
    co = w.shape[0]
    out = np.empty((m, co), dtype=np.float64)
    rc = lib().orc_pointwise(x.ctypes.data_as(ctypes.c_void_p), w.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(m), ci, co)
    assert rc == 0
    return out


def ada_warp(x, g_inv, f, margins):
    """ADA geometric stage (orc_ada_warp; reference ada_augment.py:271-304): x [N, K, H, W], g_inv [N, 3, 3] (pixel units, centred),
    f [taps] the normalised 1-D low-pass, margins (mx0, my0, mx1, my1) the reflect padding of :283 -> [N, K, H, W]."""
    x, g_inv, f = _f64(x), _f64(g_inv), _f64(f)
    n, k, h, w = x.shape
    y = np.empty_like(x)
    rc = lib().orc_ada_warp(_dp(x), _dp(g_inv), _dp(f), ctypes.c_int(f.shape[0]), _dp(y), ctypes.c_int(n), ctypes.c_int(k), ctypes.c_int(h),
                            ctypes.c_int(w), *[ctypes.c_int(int(m)) for m in margins])
    assert rc == 0, rc
    return y


def ada_colour(x, cmat=None, noise=None, sigma=None, cut=None):
    """ADA colour matrix / additive noise / cutout in one pass (orc_ada_colour; reference ada_augment.py:376-381, :407-427):
    x [N, C, T, H, W]; cmat [N, 4, 4]; noise like x with sigma [N]; cut [N, 4] = (cx, cy, sx, sy) -> like x."""
    x = _f64(x)
    n, c, t, h, w = x.shape
    y = np.empty_like(x)
    cm = None if cmat is None else _f64(cmat)
    nz = None if noise is None else _f64(noise)
    sg = None if sigma is None else _f64(sigma)
    ct = None if cut is None else _f64(cut)
    null = ctypes.POINTER(ctypes.c_double)()
    rc = lib().orc_ada_colour(_dp(x), null if cm is None else _dp(cm), null if nz is None else _dp(nz), null if sg is None else _dp(sg),
                              null if ct is None else _dp(ct), _dp(y), ctypes.c_int(n), ctypes.c_int(c), ctypes.c_int(t), ctypes.c_int(h), ctypes.c_int(w))
    assert rc == 0, rc
    return y
