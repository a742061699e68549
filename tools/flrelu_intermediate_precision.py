"""How much accuracy does filtered_lrelu lose if the three intermediates between its four separable FIR
stages are kept in f16 / bf16 (what an MFMA formulation with 16-bit operands needs) instead of float32?
Pure numpy emulation against the float64 oracle; no GPU. Prints max / mean absolute error next to the error
of the current scheme (float32 intermediates, ONE rounding at the output)."""
import os
import sys

import numpy as np
import scipy.signal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

try:
    import torch
except ImportError:  # pragma: no cover
    torch = None


def rnd(a, kind):
    if kind == 'f32':
        return a.astype(np.float32).astype(np.float64)
    if kind == 'f16':
        return a.astype(np.float16).astype(np.float64)
    return torch.tensor(a, dtype=torch.float64).to(torch.bfloat16).double().numpy()      # bf16


def fir_up(x, f, up, axis):
    """zero-insert by `up` along `axis` and correlate with f * up (valid part), float64."""
    shape = list(x.shape)
    shape[axis] *= up
    z = np.zeros(shape)
    idx = [slice(None)] * x.ndim
    idx[axis] = slice(0, None, up)
    z[tuple(idx)] = x
    return np.apply_along_axis(lambda v: np.convolve(v, f * up, mode='valid'), axis, z)


def fir_down(x, f, down, axis):
    y = np.apply_along_axis(lambda v: np.convolve(v, f, mode='valid'), axis, x)
    idx = [slice(None)] * x.ndim
    idx[axis] = slice(0, None, down)
    return y[tuple(idx)]


def run(up, down, fu_taps, fd_taps, io, mid, seed=0):
    rs = np.random.RandomState(seed)
    fu = scipy.signal.firwin(fu_taps, cutoff=0.9 / up, width=0.6 / up, fs=2.0)
    fd = scipy.signal.firwin(fd_taps, cutoff=0.9 / down, width=0.6 / down, fs=2.0)
    x = rnd(rs.randn(2, 3, 40, 44) * 1.5, io)                       # activations after a demodulated conv: O(1)
    b = rnd(rs.randn(3) * 0.1, io)
    ref = oracle.filtered_lrelu(x, fu, fd, b, up=up, down=down, padding=0, gain=np.sqrt(2), slope=0.2, clamp=256)
    t = x + b[None, :, None, None]
    t = rnd(fir_up(t, fu, up, 3), mid)
    t = rnd(fir_up(t, fu, up, 2), mid)
    t = np.where(t < 0, t * 0.2, t) * np.sqrt(2)
    t = np.clip(t, -256, 256)
    t = rnd(t, mid)
    t = rnd(fir_down(t, fd, down, 3), mid)
    t = fir_down(t, fd, down, 2)
    out = rnd(t, io)
    h, w = min(out.shape[2], ref.shape[2]), min(out.shape[3], ref.shape[3])
    err = np.abs(out[:, :, :h, :w] - ref[:, :, :h, :w])
    return err.max(), err.mean(), np.abs(ref).mean()


if __name__ == '__main__':
    oracle.build()
    print('config            io    intermediates   max abs err   mean abs err   mean |y|')
    for up, down, nu, nd in ((2, 2, 12, 12), (4, 2, 24, 12), (2, 4, 12, 24)):
        for io in ('f16', 'bf16'):
            for mid in ('f32', io):
                mx, mean, mag = run(up, down, nu, nd, io, mid)
                print(f'up{up} down{down} {nu}/{nd}   {io:5s} {mid:5s}          {mx:10.3e}    {mean:10.3e}    {mag:8.3f}')
