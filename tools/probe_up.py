import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import torch
from torch_utils.ops import upfirdn2d
dev = torch.device('cuda')
f = torch.tensor([0.125, 0.375, 0.375, 0.125], device=dev)
def gtime(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for dtype in (torch.bfloat16, torch.float32):
    for shp in ([64, 512, 3, 4], [160, 512, 5, 8], [256, 128, 9, 16], [256, 64, 18, 32], [256, 64, 32, 32]):
        x = torch.randn(*shp, device=dev).to(dtype)
        tu = gtime(lambda: upfirdn2d.upsample2d(x, f))
        xd = torch.randn(shp[0], shp[1], shp[2] * 2, shp[3] * 2, device=dev).to(dtype)
        td = gtime(lambda: upfirdn2d.downsample2d(xd, f))
        nb = (x.numel() * 5) * x.element_size()
        print(f'{dtype} {shp}: up2 {tu:7.1f} us {nb/tu/1e3:7.1f} GB/s | down2 {td:7.1f} us {nb/td/1e3:7.1f} GB/s')
print('--- channels_last ---')
for dtype in (torch.bfloat16, torch.float32):
    for shp in ([1024, 64, 18, 32], [1024, 64, 32, 32], [512, 128, 9, 16]):
        x = torch.randn(*shp, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        tu = gtime(lambda: upfirdn2d.upsample2d(x, f))
        xd = torch.randn(shp[0], shp[1], shp[2] * 2, shp[3] * 2, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        td = gtime(lambda: upfirdn2d.downsample2d(xd, f))
        nb = (x.numel() * 5) * x.element_size()
        print(f'NHWC {dtype} {shp}: up2 {tu:7.1f} us {nb/tu/1e3:7.1f} GB/s | down2 {td:7.1f} us {nb/td/1e3:7.1f} GB/s')
