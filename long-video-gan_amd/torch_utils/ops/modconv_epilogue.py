"""Epilogue of a style-modulated convolution fused with the prologue of the next one:

    out[f, c] = clamp(act(y[f, c] * pre[f, c] + b[c]) * gain) * post[f, c]     (+ mean(value^2) before `post`)

`y` is the raw convolution output over frames f (sample x time), `pre` its demodulation
coefficients, `post` the style modulation of the convolution that consumes `out`, and the mean
square is the input-magnitude statistic the next layer tracks. The reference spells this as three
elementwise passes plus a reduction (model/generator_lres.py:122 `output * demodulation`, :570
bias_act, :101 `input * style`, :574 magnitude EMA); here GPU tensors take ONE pass,
`lvg_modconv_epilogue` (csrc/modconv_epilogue.hip), whose backward recomputes the activation and
returns the per-(frame, channel) reductions d_pre / d_post / d_bias from the same pass. CPU tensors
run the plain-PyTorch composition (also the definition the GPU tests compare against).

Activations: linear, relu, lrelu (what the modulated layers use). First-order gradients only --
the generator losses of this repo never differentiate twice through a modulated layer (R1 acts on
the discriminator)."""

import torch

from .. import custom_ops
from . import _hip
from .bias_act import activation_funcs

_FUSED_ACTS = ('linear', 'relu', 'lrelu')
_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        custom_ops.get_plugin(module_name='modconv_epilogue_plugin')
        _plugin = _hip.lib()
    return True


def _resolve(act, alpha, gain, clamp):
    assert act in _FUSED_ACTS, f'modconv_epilogue supports {_FUSED_ACTS}, got {act!r}'
    spec = activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    return spec, alpha, gain, clamp


def _ref(y, pre, b, post, act, alpha, gain, clamp, want_msq, res=None, want_mid=False):
    """Plain-PyTorch definition (float32 arithmetic, one rounding at the end)."""
    u = y.float()
    if pre is not None:
        u = u * pre[:, :, None, None]
    if b is not None:
        u = u + b.float()[None, :, None, None]
    if res is not None:
        u = u + res.float()
    if act == 'relu':
        u = torch.relu(u)
    elif act == 'lrelu':
        u = torch.nn.functional.leaky_relu(u, alpha)
    if gain != 1:
        u = u * gain
    if clamp >= 0:
        u = u.clamp(-clamp, clamp)
    msq = u.detach().square().mean() if want_msq else None
    if want_mid:
        mid = u.to(y.dtype)
    if post is not None:
        u = u * post[:, :, None, None]
    if want_mid:
        return u.to(y.dtype), mid, msq
    return u.to(y.dtype), msq


def _layout(t):
    """(dense tensor, channels_last flag) as handed to the kernel."""
    if t.shape[1] > 1 and t.stride(1) == 1 and t.is_contiguous(memory_format=torch.channels_last):
        return t, 1
    if t.is_contiguous():
        return t, 0
    if t.shape[1] > 1 and t.stride(1) == 1:
        return t.contiguous(memory_format=torch.channels_last), 1
    return t.contiguous(), 0


def _sum_slots(red, used):
    """[3, slots, frames, channels] partial sums -> [3, frames, channels] (fixed order; unused rows are never read)."""
    if red.shape[1] == 1:
        return red[:, 0]
    out = torch.empty((3,) + tuple(red.shape[2:]), dtype=red.dtype, device=red.device)
    first = next(i for i, u in enumerate(used) if u)
    if all(used[first:]):                                   # the used rows are one contiguous block: ONE reduction launch instead of one per row
        torch.sum(red[first:], dim=1, out=out[first:])
        return out
    for i, u in enumerate(used):
        if u:
            torch.sum(red[i], dim=0, out=out[i])
    return out


def _launch_fwd(y, pre, b, post, cl, act_id, alpha, gain, clamp, want_msq, want_mid=False):
    """One lvg_modconv_epilogue launch on y's device and the current stream -> (out, msq per frame | None) -- with `want_mid`
    (channels-last only) the dual form: (out, mid, msq per frame | None), mid = the value before `post`."""
    f, c, h, w = y.shape
    out = torch.empty_like(y)
    assert out.stride() == y.stride()
    # partial sums per slot (chunk of a frame / channel plane): summed below in a fixed order -- no atomics, reproducible
    slots = _hip.lib().lvg_modconv_epilogue_slots(f, c, h * w, cl, _hip.dtype_code(y.dtype), 0) if want_msq else 0
    msq = torch.empty((slots, f), dtype=torch.float32, device=y.device) if want_msq else None
    if want_mid:
        assert cl, 'modconv_epilogue: the dual form needs channels-last tensors'
        mid = torch.empty_like(y)
        with torch.cuda.device(y.device):
            rc = _hip.lib().lvg_modconv_epilogue_dual(
                y.data_ptr(), _hip.ptr(pre), _hip.ptr(b), _hip.ptr(post), out.data_ptr(), mid.data_ptr(), _hip.ptr(msq),
                f, c, h * w, _hip.dtype_code(y.dtype), act_id, alpha, gain, clamp, _hip.stream(y.device))
        _hip.check(rc, 'modconv_epilogue_dual')
        return out, mid, msq
    with torch.cuda.device(y.device):
        rc = _hip.lib().lvg_modconv_epilogue(
            y.data_ptr(), _hip.ptr(pre), _hip.ptr(b), _hip.ptr(post), out.data_ptr(), _hip.ptr(msq),
            f, c, h * w, cl, _hip.dtype_code(y.dtype), act_id, alpha, gain, clamp, _hip.stream(y.device))
    _hip.check(rc, 'modconv_epilogue')
    return out, msq


def _launch_bwd(dout, y, pre, b, post, cl, act_id, alpha, gain, clamp, dmid=None):
    """One lvg_modconv_epilogue_backward launch -> (dy, [d_pre, d_post, d_sum] float32 [3, frames, channels]); `dmid`: the gradient
    of the dual form's second output (channels-last only)."""
    f, c, h, w = y.shape
    dy = torch.empty_like(y)
    slots = _hip.lib().lvg_modconv_epilogue_slots(f, c, h * w, cl, _hip.dtype_code(y.dtype), 1)
    red = torch.empty(3, slots, f, c, dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        if dmid is not None:
            assert cl and dmid.stride() == y.stride() and dmid.dtype == y.dtype
            rc = _hip.lib().lvg_modconv_epilogue_dual_backward(
                dout.data_ptr(), dmid.data_ptr(), y.data_ptr(), _hip.ptr(pre), _hip.ptr(b), _hip.ptr(post), dy.data_ptr(),
                red[0].data_ptr() if pre is not None else None, red[1].data_ptr() if post is not None else None, red[2].data_ptr(),
                f, c, h * w, _hip.dtype_code(y.dtype), act_id, alpha, gain, clamp, _hip.stream(y.device))
        else:
            rc = _hip.lib().lvg_modconv_epilogue_backward(
                dout.data_ptr(), y.data_ptr(), _hip.ptr(pre), _hip.ptr(b), _hip.ptr(post), dy.data_ptr(),
                red[0].data_ptr() if pre is not None else None, red[1].data_ptr() if post is not None else None, red[2].data_ptr(),
                f, c, h * w, cl, _hip.dtype_code(y.dtype), act_id, alpha, gain, clamp, _hip.stream(y.device))
    _hip.check(rc, 'modconv_epilogue_backward')
    return dy, _sum_slots(red, (pre is not None, post is not None, True))


class _Epilogue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, pre, b, post, act_id, alpha, gain, clamp, want_msq):
        y, cl = _layout(y)
        pre = pre.contiguous() if pre is not None else None
        post = post.contiguous() if post is not None else None
        b = b.contiguous() if b is not None else None
        out, msq = _launch_fwd(y, pre, b, post, cl, act_id, alpha, gain, clamp, want_msq)
        ctx.save_for_backward(y, pre, b, post)
        ctx.cfg = (cl, act_id, alpha, gain, clamp)
        mean_sq = msq.sum() / float(y.numel()) if want_msq else None
        if want_msq:
            ctx.mark_non_differentiable(mean_sq)
        return out, mean_sq

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout, _dmsq):
        y, pre, b, post = ctx.saved_tensors
        cl, act_id, alpha, gain, clamp = ctx.cfg
        dout = dout.contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
        assert dout.stride() == y.stride() and dout.dtype == y.dtype
        dy, red = _launch_bwd(dout, y, pre, b, post, cl, act_id, alpha, gain, clamp)
        d_pre = red[0] if (pre is not None and ctx.needs_input_grad[1]) else None
        d_b = red[2].sum(dim=0).to(b.dtype) if (b is not None and ctx.needs_input_grad[2]) else None
        d_post = red[1] if (post is not None and ctx.needs_input_grad[3]) else None
        return dy, d_pre, d_b, d_post, None, None, None, None, None


class _EpilogueDual(torch.autograd.Function):
    """(out, mid, mean_square): `out` = mid * post. Channels-last GPU tensors."""

    @staticmethod
    def forward(ctx, y, pre, b, post, act_id, alpha, gain, clamp, want_msq):
        pre = pre.contiguous() if pre is not None else None
        post = post.contiguous() if post is not None else None
        b = b.contiguous() if b is not None else None
        out, mid, msq = _launch_fwd(y, pre, b, post, 1, act_id, alpha, gain, clamp, want_msq, want_mid=True)
        ctx.save_for_backward(y, pre, b, post)
        ctx.cfg = (act_id, alpha, gain, clamp)
        mean_sq = msq.sum() / float(y.numel()) if want_msq else None
        if want_msq:
            ctx.mark_non_differentiable(mean_sq)
        return out, mid, mean_sq

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout, dmid, _dmsq):
        y, pre, b, post = ctx.saved_tensors
        act_id, alpha, gain, clamp = ctx.cfg
        if dout is None:                                    # only `mid` was used downstream
            dout = torch.zeros_like(y)
        dout = dout.contiguous(memory_format=torch.channels_last)
        if dmid is not None:
            dmid = dmid.contiguous(memory_format=torch.channels_last)
        assert dout.stride() == y.stride() and dout.dtype == y.dtype
        dy, red = _launch_bwd(dout, y, pre, b, post, 1, act_id, alpha, gain, clamp, dmid=dmid)
        d_pre = red[0] if (pre is not None and ctx.needs_input_grad[1]) else None
        d_b = red[2].sum(dim=0).to(b.dtype) if (b is not None and ctx.needs_input_grad[2]) else None
        d_post = red[1] if (post is not None and ctx.needs_input_grad[3]) else None
        return dy, d_pre, d_b, d_post, None, None, None, None, None


def dual_supported(y):
    """True when `modconv_epilogue_dual` runs as ONE kernel on y (channels-last GPU tensor, 16-bit or float32, a power-of-two number
    of 16-byte channel vectors)."""
    if not (y.is_cuda and y.ndim == 4 and y.dtype in (torch.float16, torch.bfloat16, torch.float32)):
        return False
    v = 4 if y.dtype == torch.float32 else 8
    c = y.shape[1]
    cv = c // v
    return c % v == 0 and 0 < cv <= 256 and (cv & (cv - 1)) == 0 and y.shape[0] <= 65535 and y.is_contiguous(memory_format=torch.channels_last) \
        and y.stride(1) == 1


def modconv_epilogue_dual(y, pre=None, b=None, post=None, act='linear', alpha=None, gain=None, clamp=None, want_msq=False):
    r"""(out, mid[, mean_square]): mid = clamp(act(y * pre + b) * gain), out = mid * post -- both from ONE pass over y (3 streams where
    bias_act followed by the next layer's modulation moves 4), and one backward pass for both gradients (4 streams instead of 9).
    Same arguments as `modconv_epilogue`; other tensors take that function twice."""
    assert y.ndim == 4
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if dual_supported(y) and _init():
        out, mid, msq = _EpilogueDual.apply(y, pre, b, post, spec.cuda_idx, alpha, gain, clamp, bool(want_msq))
    elif y.device.type == 'cuda':
        mid = modconv_epilogue(y, pre=pre, b=b, act=act, alpha=alpha, gain=gain, clamp=clamp)
        res = modconv_epilogue(mid, post=post, want_msq=want_msq)
        out, msq = res if want_msq else (res, None)
    else:
        out, mid, msq = _ref(y, pre, b, post, act, alpha, gain, clamp, want_msq, want_mid=True)
    return (out, mid, msq) if want_msq else (out, mid)


def modconv_epilogue(y, pre=None, b=None, post=None, act='linear', alpha=None, gain=None, clamp=None, want_msq=False, impl='cuda'):
    r"""out = clamp(act(y * pre + b) * gain) * post, optionally with mean(value before post ** 2).

    Args:
        y:     [frames, channels, H, W], contiguous or channels-last; float32 / float16 / bfloat16.
        pre:   float32 [frames, channels] or None.
        b:     [channels] in y's dtype, or None.
        post:  float32 [frames, channels] or None.
        act, alpha, gain, clamp: as `bias_act` ('linear', 'relu', 'lrelu').
        want_msq: also return the (detached, float32 scalar) mean square of the value before `post`.

    Returns `out`, or `(out, mean_square)` with `want_msq`."""
    assert y.ndim == 4
    f, c = y.shape[:2]
    for name, t in (('pre', pre), ('post', post)):
        assert t is None or (t.shape == (f, c) and t.dtype == torch.float32 and t.device == y.device), f'{name} must be float32 [frames, channels]'
    assert b is None or (b.shape == (c,) and b.dtype == y.dtype and b.device == y.device), 'b must be [channels] in y\'s dtype'
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if impl == 'cuda' and y.device.type == 'cuda' and _init():
        out, msq = _Epilogue.apply(y, pre, b, post, spec.cuda_idx, alpha, gain, clamp, bool(want_msq))
    else:
        out, msq = _ref(y, pre, b, post, act, alpha, gain, clamp, want_msq)
    return (out, msq) if want_msq else out


# ----------------------------------------------------------------------------------------------------
# Temporal-tap gather + epilogue (csrc/tapconv_epilogue.hip). z [(T N), taps*C, H, W] channels-last holds the
# output of ONE 2-D convolution whose output channels stack the temporal taps (tap-major); frame f of the
# result sums tap k from frame f + (k - taps//2) * shift. These two functions are the launch-level interface
# (used by lvg.models.lres._TapConvEpilogue); CPU tensors take the explicit PyTorch formulas below, which
# are also what the GPU tests compare against.

def _tap_ranges(k, taps, shift, total):
    """(source frames of z, destination frames of the sum) for tap k, or None if they never overlap."""
    d = (k - taps // 2) * shift
    if abs(d) >= total:
        return None
    if d == 0:
        return slice(None), slice(None)
    return (slice(d, None), slice(None, -d)) if d > 0 else (slice(None, d), slice(-d, None))


def tap_gather_forward(z, pre, b, res, post, taps, shift, act='linear', alpha=None, gain=None, clamp=None, want_msq=False, keep_sum=True):
    """-> (out, ysum | None, mean_square | None). out/ysum: [(T N), C, H, W] in z's dtype and memory format."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    f, kc, h, w = z.shape
    c = kc // taps
    assert c * taps == kc
    if z.device.type == 'cuda' and _init():
        assert z.is_contiguous(memory_format=torch.channels_last), 'tap gather needs channels-last z'
        out = torch.empty((f, c, h, w), dtype=z.dtype, device=z.device, memory_format=torch.channels_last)
        ysum = torch.empty_like(out) if keep_sum else None
        slots = _hip.lib().lvg_tapconv_epilogue_slots(f, c, h * w, _hip.dtype_code(z.dtype))
        msq = torch.empty((slots, f), dtype=torch.float32, device=z.device) if want_msq else None
        if res is not None:
            res = res.contiguous(memory_format=torch.channels_last)
        with torch.cuda.device(z.device):
            rc = _hip.lib().lvg_tapconv_epilogue(
                z.data_ptr(), _hip.ptr(pre), _hip.ptr(b), _hip.ptr(res), _hip.ptr(post), out.data_ptr(), _hip.ptr(ysum), _hip.ptr(msq),
                f, c, h * w, taps, shift, _hip.dtype_code(z.dtype), spec.cuda_idx, alpha, gain, clamp, _hip.stream(z.device))
        _hip.check(rc, 'tapconv_epilogue')
        return out, ysum, (msq.sum() / float(out.numel()) if want_msq else None)
    ysum = torch.zeros((f, c, h, w), dtype=torch.float32, device=z.device)
    for k in range(taps):
        r = _tap_ranges(k, taps, shift, f)
        if r is not None:
            ysum[r[1]] += z[r[0], k * c:(k + 1) * c].float()
    ysum = ysum.to(z.dtype)                                   # what the kernel saves (rounded); `out` uses the exact sum
    out, msq = _ref(ysum, pre, b, post, act, alpha, gain, clamp, want_msq, res=res)
    return out, (ysum if keep_sum else None), msq


def tap_gather_backward(dout, ysum, pre, b, res, post, taps, shift, act='linear', alpha=None, gain=None, clamp=None):
    """-> (dz [(T N), taps*C, H, W], d_pre | None, d_post | None, d_sum [(T N), C]); float32 reductions."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    f, c, h, w = ysum.shape
    if ysum.device.type == 'cuda' and _init():
        dout = dout.contiguous(memory_format=torch.channels_last)
        assert ysum.is_contiguous(memory_format=torch.channels_last) and dout.dtype == ysum.dtype
        dz = torch.empty((f, taps * c, h, w), dtype=ysum.dtype, device=ysum.device, memory_format=torch.channels_last)
        slots = _hip.lib().lvg_tapconv_epilogue_slots(f, c, h * w, _hip.dtype_code(ysum.dtype))
        red = torch.empty(3, slots, f, c, dtype=torch.float32, device=ysum.device)
        if res is not None:
            res = res.contiguous(memory_format=torch.channels_last)
        with torch.cuda.device(ysum.device):
            rc = _hip.lib().lvg_tapconv_epilogue_backward(
                dout.data_ptr(), ysum.data_ptr(), _hip.ptr(pre), _hip.ptr(b), _hip.ptr(res), _hip.ptr(post), dz.data_ptr(),
                red[0].data_ptr() if pre is not None else None, red[1].data_ptr() if post is not None else None, red[2].data_ptr(),
                f, c, h * w, taps, shift, _hip.dtype_code(ysum.dtype), spec.cuda_idx, alpha, gain, clamp, _hip.stream(ysum.device))
        _hip.check(rc, 'tapconv_epilogue_backward')
        red = _sum_slots(red, (pre is not None, post is not None, True))
        return dz, (red[0] if pre is not None else None), (red[1] if post is not None else None), red[2]
    y = ysum.float()
    u = y if pre is None else y * pre[:, :, None, None]
    if b is not None:
        u = u + b.float()[None, :, None, None]
    if res is not None:
        u = u + res.float()
    if act == 'relu':
        a, slope = torch.relu(u), (u > 0).float()
    elif act == 'lrelu':
        a, slope = torch.nn.functional.leaky_relu(u, alpha), torch.where(u > 0, torch.ones_like(u), torch.full_like(u, alpha))
    else:
        a, slope = u, torch.ones_like(u)
    g = a * gain
    inside = torch.ones_like(g) if clamp < 0 else ((g > -clamp) & (g < clamp)).float()
    gc = g if clamp < 0 else g.clamp(-clamp, clamp)
    go = dout.float()
    du = go * inside * gain * slope
    if post is not None:
        du = du * post[:, :, None, None]
    dy = du if pre is None else du * pre[:, :, None, None]
    dz = torch.zeros((f, taps * c, h, w), dtype=torch.float32, device=ysum.device)
    for k in range(taps):
        r = _tap_ranges(k, taps, shift, f)
        if r is not None:
            dz[r[0], k * c:(k + 1) * c] = dy[r[1]]
    d_pre = (du * y).sum(dim=(2, 3)) if pre is not None else None
    d_post = (go * gc).sum(dim=(2, 3)) if post is not None else None
    return dz.to(ysum.dtype), d_pre, d_post, du.sum(dim=(2, 3))
