"""Time BlurredNoise.blur at the generator's size (8 clips x 640 frames): float32-MFMA kernel against the dense window product. MEASUREMENT TOOL (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import torch
from lvg.models import lres

mod = lres.BlurredNoise().cuda()
for clips, frames in ((8, 640), (8, 128), (2, 640), (32, 672)):
    noise = torch.randn(clips, mod.noise_channels, frames + mod.kernel_size - 1, device='cuda')
    res = {}
    for hip in (True, False):
        lres.NOISE_BANK_HIP = hip
        for _ in range(3):
            y = mod.blur(noise)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            y = mod.blur(noise)
        b.record(); torch.cuda.synchronize()
        res[hip] = (a.elapsed_time(b) / 20 * 1e3, y)
    err = float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max())
    flops = 2.0 * clips * mod.noise_channels * frames * float((mod.blur_filters != 0).sum())
    print(f'{clips} clips x {frames} frames: kernel {res[True][0]:8.1f} us ({flops / res[True][0] / 1e6:6.1f} TFLOP/s on the non-zero taps), dense product {res[False][0]:8.1f} us, max difference {err:.2e}')
