"""Per-op achieved-bandwidth probe on one MI355X (algorithmic bytes / event-timed launch)."""
import os, sys, json, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import numpy as np
import torch
from torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu

dev = torch.device('cuda')
warnings.simplefilter('ignore')


def timeit(fn, iters=20, warm=3):
    """GPU time per call: `iters` launches queued back to back between two events (a single
    launch between events would include the host's launch latency when the GPU is idle)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) * 1e-3 / iters
        best = t if best is None else min(best, t)
    return best


def report(name, secs, nbytes):
    print(json.dumps(dict(op=name, us=round(secs * 1e6, 1), GBps=round(nbytes / secs / 1e9, 1), frac_8TBps=round(nbytes / secs / 8e12, 3))), flush=True)


def main():
    print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, 'CUs')
    # copy ceiling
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev); b = torch.empty_like(a)
    report('torch_copy_1GiB_f32', timeit(lambda: b.copy_(a)), 2 * a.numel() * 4)
    del a, b
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        s = torch.finfo(dtype).bits // 8
        x = torch.randn(1, 64, 128, 36, 64, device=dev).to(dtype); bb = torch.randn(64, device=dev).to(dtype)
        report(f'bias_act_lrelu_fwd[1,64,128,36,64]{dtype}', timeit(lambda: bias_act.bias_act(x, bb, act='lrelu', clamp=256)), 2 * x.numel() * s)
        x2 = torch.randn(4, 64, 128, 36, 64, device=dev).to(dtype)
        report(f'bias_act_lrelu_fwd[4,64,128,36,64]{dtype}', timeit(lambda: bias_act.bias_act(x2, bb, act='lrelu', clamp=256)), 2 * x2.numel() * s)
        y2 = bias_act.bias_act(x2, bb, act='lrelu', clamp=256)
        from torch_utils.ops.bias_act import _launch
        report(f'bias_act_lrelu_bwd[4,64,128,36,64]{dtype}', timeit(lambda: _launch(x2, bb, None, y2, None, 1, 1, 3, 0.2, 1.414, 256.0)), 3 * x2.numel() * s)
        del x2, y2
        for shp in ([512, 64, 36, 64], [640, 512, 9, 16], [96, 512, 3, 4]):
            xf = torch.randn(*shp, device=dev).to(dtype); bf_ = torch.randn(shp[1], device=dev).to(dtype)
            report(f'bias_act_frames{shp}{dtype}', timeit(lambda: bias_act.bias_act(xf, bf_, act='lrelu', clamp=256)), 2 * xf.numel() * s)
        f = torch.tensor([0.125, 0.375, 0.375, 0.125], device=dev)
        x = torch.randn(1, 8192, 18, 32, device=dev).to(dtype)
        y = upfirdn2d.upsample2d(x, f)
        report(f'upsample2d[1,8192,18,32]{dtype}', timeit(lambda: upfirdn2d.upsample2d(x, f)), (x.numel() + y.numel()) * s)
        x = torch.randn(4, 8192, 18, 32, device=dev).to(dtype)
        y = upfirdn2d.upsample2d(x, f)
        report(f'upsample2d[4,8192,18,32]{dtype}', timeit(lambda: upfirdn2d.upsample2d(x, f)), (x.numel() + y.numel()) * s)
        x = torch.randn(4, 8192, 64, 64, device=dev).to(dtype)
        y = upfirdn2d.downsample2d(x, f)
        report(f'downsample2d[4,8192,64,64]{dtype}', timeit(lambda: upfirdn2d.downsample2d(x, f)), (x.numel() + y.numel()) * s)
        ft = f[:, None]
        x = torch.randn(4, 256, 80, 144, device=dev).to(dtype)
        kw = dict(up=(1, 2), padding=[0, 0, 2, 1], gain=2)
        y = upfirdn2d.upfirdn2d(x, ft, **kw)
        report(f'temporal_up2[4,256,80,144]{dtype}', timeit(lambda: upfirdn2d.upfirdn2d(x, ft, **kw)), (x.numel() + y.numel()) * s)
        x = torch.randn(4, 4096, 32, 32, device=dev).to(dtype)
        y = upfirdn2d.upsample2d(x, f)
        report(f'upsample2d_Dbwd[4,4096,32,32]{dtype}', timeit(lambda: upfirdn2d.upsample2d(x, f)), (x.numel() + y.numel()) * s)
        x = torch.randn(4, 16384, 3, 4, device=dev).to(dtype)
        y = upfirdn2d.upsample2d(x, f)
        report(f'upsample2d_tiny[4,16384,3,4]{dtype}', timeit(lambda: upfirdn2d.upsample2d(x, f)), (x.numel() + y.numel()) * s)
        x = torch.randn(4, 128, 128, 256, device=dev).to(dtype)
        kw = dict(down=(1, 2), padding=[0, 0, 1, 1])
        y = upfirdn2d.upfirdn2d(x, ft, **kw)
        report(f'temporal_down2[4,128,128,256]{dtype}', timeit(lambda: upfirdn2d.upfirdn2d(x, ft, **kw)), (x.numel() + y.numel()) * s)
    if '--quick' in sys.argv:
        return
    import scipy.signal
    k12 = torch.tensor(scipy.signal.firwin(numtaps=12, cutoff=0.45, width=0.3, fs=2.0).astype(np.float32), device=dev)
    k24 = torch.tensor(scipy.signal.firwin(numtaps=24, cutoff=0.22, width=0.15, fs=2.0).astype(np.float32), device=dev)
    for dtype in (torch.float16, torch.float32):
        s = torch.finfo(dtype).bits // 8
        for name, shape, up, down, fu, fd, pad in (('L8_up2down2', [8, 512, 94, 150], 2, 2, k12, k12, [9, 8, 9, 8]),
                                                   ('L10_up4down2', [8, 256, 94, 150], 4, 2, k24, k12, [-6, -9, -6, -9]),
                                                   ('L12_up2down2', [8, 128, 166, 278], 2, 2, k12, k12, [9, 8, 9, 8])):
            x = torch.randn(*shape, device=dev).to(dtype); bb = torch.randn(shape[1], device=dev).to(dtype)
            fn = lambda: filtered_lrelu.filtered_lrelu(x, fu, fd, bb, up=up, down=down, padding=pad, clamp=256)
            y = fn()
            report(f'filtered_lrelu_{name}{shape}{dtype}', timeit(fn, iters=5, warm=1), (x.numel() + y.numel()) * s)


if __name__ == '__main__':
    main()
