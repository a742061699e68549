"""Times the two transposing kernels of the 2-D modulated convolution (csrc/modconv2d_layout.hip) on the super-resolution generator's
shapes and checks them against the PyTorch composition. usage: python tools/layout_bench.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import torch

from torch_utils.ops import modconv2d_layout as ml

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps): fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best * 1e3

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda')
    torch.manual_seed(0)
    shapes = [(8, 512, 0, 92, 148), (8, 509, 3, 92, 148), (8, 256, 0, 92, 148), (8, 128, 0, 164, 276), (8, 61, 3, 164, 276)]
    big = torch.randn(8, 512, 92, 148, device=dev, dtype=torch.float16); big2 = torch.empty_like(big)
    t0 = timeit(lambda: big2.copy_(big), reps)
    print(f'reference: plain copy of {big.numel() * 2 / 1e6:.0f} MB {t0:7.1f} us {big.numel() * 4 / t0 / 1e6:6.2f} TB/s', flush=True)
    for (n, ca, cb, h, w) in shapes:
        c = ca + cb
        cp = (c + 63) // 64 * 64
        xa = torch.randn(n, ca, h, w, device=dev, dtype=torch.float16)
        xb = torch.randn(n, cb, h, w, device=dev, dtype=torch.float16) if cb else None
        scale = torch.rand(n, c, device=dev) + 0.5
        hd, wd = (h + 2 + 3) // 4 * 4 + 2, (w + 2 + 15) // 16 * 16 + 2
        dst = torch.full((n, hd, wd, cp), 7.0, device=dev, dtype=torch.float16)
        oth = torch.randn(n, h, w, cp, device=dev, dtype=torch.float16)
        # forward prologue
        ml._nchw_to_nhwc_padded(xa, xb, scale, dst, (2, 2), zero_border=True)
        src = xa if xb is None else torch.cat((xa, xb), 1)
        ref = (src.float() * scale[:, :, None, None]).permute(0, 2, 3, 1).half()
        err = (dst[:, 2:2 + h, 2:2 + w, :c].float() - ref.float()).abs().max().item()
        border = dst[:, :2].abs().max().item() + dst[:, :, :2].abs().max().item() + dst[:, 2 + h:].abs().max().item() + dst[:, :, 2 + w:].abs().max().item() + (dst[..., c:].abs().max().item() if cp > c else 0.0)
        t1 = timeit(lambda: ml._nchw_to_nhwc_padded(xa, xb, scale, dst, (2, 2), zero_border=True), reps)
        byt = n * c * h * w * 2 + n * h * w * cp * 2
        # with the reduction (backward of the epilogue)
        part = ml._nchw_to_nhwc_padded(xa, xb, scale, dst, (2, 2), oth=oth, zero_border=True)
        pref = (src.float().permute(0, 2, 3, 1) * oth[..., :c].float()).sum(dim=(1, 2))
        perr = ((part.sum(1) - pref).abs().max() / pref.abs().max()).item()
        t2 = timeit(lambda: ml._nchw_to_nhwc_padded(xa, xb, scale, dst, (2, 2), oth=oth, zero_border=True), reps)
        # epilogue
        y = torch.randn(n, h, w, cp, device=dev, dtype=torch.float16)
        sc2 = torch.rand(n, c, device=dev) + 0.5
        out, _ = ml._frames_to_nchw(y, sc2, c)
        ref2 = (y[..., :c].float().permute(0, 3, 1, 2) * sc2[:, :, None, None]).half()
        err2 = (out.float() - ref2.float()).abs().max().item()
        t3 = timeit(lambda: ml._frames_to_nchw(y, sc2, c), reps)
        out2, part2 = ml._frames_to_nchw(y, sc2, c, oth_a=xa, oth_b=xb)
        pref2 = (y[..., :c].float().permute(0, 3, 1, 2) * src.float()).sum(dim=(2, 3))
        perr2 = ((part2.sum(1) - pref2).abs().max() / pref2.abs().max()).item()
        t4 = timeit(lambda: ml._frames_to_nchw(y, sc2, c, oth_a=xa, oth_b=xb), reps)
        byt3 = n * h * w * cp * 2 + n * c * h * w * 2
        print(f'[{n},{ca}+{cb},{h},{w}] to_nhwc {t1:7.1f} us {byt / t1 / 1e6:6.2f} TB/s (err {err:.1e} border {border:.1e}) | +reduce {t2:7.1f} us '
              f'{(byt + n * h * w * cp * 2) / t2 / 1e6:6.2f} TB/s (err {perr:.1e}) | to_nchw {t3:7.1f} us {byt3 / t3 / 1e6:6.2f} TB/s (err {err2:.1e}) | +reduce {t4:7.1f} us '
              f'{(byt3 + n * c * h * w * 2) / t4 / 1e6:6.2f} TB/s (err {perr2:.1e})', flush=True)

main()
