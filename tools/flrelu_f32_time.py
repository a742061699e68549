"""MEASUREMENT TOOL: forward (mask write) and backward time of the float32 filtered_lrelu layers of the sres generator
(L0-L2: [16, 512, 31, 38] -> [16, 512, 29, 36], up 2 / down 2, 12 / 12 taps) and parity of the outputs against the float64 reference path."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'long-video-gan_amd'))
import numpy as np, scipy.signal, torch
from torch_utils.ops import filtered_lrelu

dev = torch.device('cuda')
torch.manual_seed(0)
f = torch.tensor(scipy.signal.firwin(numtaps=12, cutoff=0.45, width=0.3, fs=2.0).astype(np.float32), device=dev)
for dtype in (torch.float32,):
    x = torch.randn(16, 512, 31, 38, device=dev, dtype=dtype, requires_grad=True)
    b = torch.randn(512, device=dev, dtype=dtype) * 0.3
    kw = dict(fu=f, fd=f, b=b, up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=0.2, clamp=256)
    y = filtered_lrelu.filtered_lrelu(x, **kw)
    dy = torch.randn_like(y)
    y.backward(dy)
    xs = x.detach()[:1, :8].double().cpu().requires_grad_(True)
    yr = filtered_lrelu.filtered_lrelu(xs, fu=f.cpu(), fd=f.cpu(), b=b[:8].double().cpu(), up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=0.2, clamp=256, impl='ref')
    yr.backward(dy[:1, :8].double().cpu())
    print('parity fwd %.2e bwd %.2e' % ((y[:1, :8].double().cpu() - yr).abs().max().item(), (x.grad[:1, :8].double().cpu() - xs.grad).abs().max().item()))
    def timeit(fn, reps=20):
        fn(); torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        return a.elapsed_time(e) / reps * 1e3
    t_fwd = timeit(lambda: filtered_lrelu.filtered_lrelu(x, **kw))
    def fb():
        yy = filtered_lrelu.filtered_lrelu(x, **kw); yy.backward(dy)
    t_fb = timeit(fb)
    nbytes = (x.numel() + y.numel()) * x.element_size()
    print(f'{dtype}: fwd+mask {t_fwd:.1f} us, fwd+bwd {t_fb:.1f} us; algorithmic {nbytes/1e6:.1f} MB per pass -> fwd {nbytes / t_fwd / 1e3:.0f} GB/s')
