"""float32 upfirdn2d on the GPU (both memory layouts) against the float64 reference path, forward and input gradient, at the shapes of the
lres discriminator's down-sampling blocks. MEASUREMENT TOOL (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import torch
from torch_utils.ops import upfirdn2d as U
torch.manual_seed(0)
f = torch.tensor([1.0, 3.0, 3.0, 1.0]) / 8
cases = []
for (n, c, h, w) in [(16, 128, 32, 32), (8, 256, 16, 16), (4, 512, 8, 8), (16, 64, 36, 64)]:
    for fmt in ('nchw', 'nhwc'):
        x = torch.randn(n, c, h, w)
        for name, fn in (('down2', lambda t, ff: U.downsample2d(t, ff, down=2)), ('up2', lambda t, ff: U.upsample2d(t, ff, up=2)),
                         ('tdown2', lambda t, ff: U.downsample2d(t.reshape(n // 2, c, 2 * h, w) if False else t, ff.unsqueeze(1), down=(1, 2)))):
            xr = x.double().requires_grad_(True)
            yr = fn(xr, f)
            gy = torch.randn_like(yr)
            (yr * gy).sum().backward()
            xg = x.cuda()
            if fmt == 'nhwc':
                xg = xg.contiguous(memory_format=torch.channels_last)
            xg.requires_grad_(True)
            yg = fn(xg, f.cuda())
            gyg = gy.float().cuda()
            if fmt == 'nhwc':
                gyg = gyg.contiguous(memory_format=torch.channels_last)
            (yg * gyg).sum().backward()
            e1 = float((yg.double().cpu() - yr).abs().max() / yr.abs().max())
            e2 = float((xg.grad.double().cpu() - xr.grad).abs().max() / xr.grad.abs().max())
            print(f'[{n},{c},{h},{w}] {fmt} {name:6s} fwd {e1:.1e} grad {e2:.1e}{"   <<<" if max(e1, e2) > 1e-5 else ""}', flush=True)
