import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import torch
from lvg.models import lres
from torch_utils.ops import bias_act
torch.manual_seed(0)
taps = torch.tensor([1.0, 3.0, 3.0, 1.0]) / 8
for (tn, c, h, w) in [(16, 128, 16, 16), (8, 256, 8, 8), (16, 64, 32, 32), (4, 512, 4, 4)]:
    for fmt in ('nchw', 'nhwc'):
        for name in ('tdown', 'tup', 'bias_act'):
            x = torch.randn(tn, c, h, w)
            b = torch.randn(c)
            def fn(t, tp, bb):
                if name == 'tdown': return lres.resample_time_frames(t, tp, 1, down=2)
                if name == 'tup': return lres.resample_time_frames(t, tp, 1, up=2)
                return bias_act.bias_act(t, bb, act='lrelu', clamp=256.0)
            xr = x.double().requires_grad_(True); br = b.double().requires_grad_(True)
            yr = fn(xr, taps, br)
            gy = torch.randn_like(yr)
            (yr * gy).sum().backward()
            xg = x.cuda()
            if fmt == 'nhwc': xg = xg.contiguous(memory_format=torch.channels_last)
            xg.requires_grad_(True); bg = b.cuda().requires_grad_(True)
            yg = fn(xg, taps.cuda(), bg)
            gyg = gy.float().cuda()
            if fmt == 'nhwc': gyg = gyg.contiguous(memory_format=torch.channels_last)
            (yg * gyg).sum().backward()
            e1 = float((yg.detach().double().cpu() - yr.detach()).abs().max() / yr.detach().abs().max())
            e2 = float((xg.grad.double().cpu() - xr.grad).abs().max() / xr.grad.abs().max())
            e3 = float((bg.grad.double().cpu() - br.grad).abs().max() / br.grad.abs().max()) if name == 'bias_act' else 0.0
            print(f'[{tn},{c},{h},{w}] {fmt} {name:8s} fwd {e1:.1e} grad {e2:.1e} db {e3:.1e}{"   <<<" if max(e1, e2, e3) > 1e-5 else ""}', flush=True)
