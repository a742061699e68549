"""Channels-last x2 down- / up-sampling (upfirdn2d_nhwc_stream_kernel) at the step's shapes; LVG_UPFIRDN_CHUNKS = row chunks per frame.
MEASUREMENT TOOL (GPU).   python tools/upfirdn_chunk_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import torch
from torch_utils.ops import upfirdn2d

f = torch.tensor([1.0, 3.0, 3.0, 1.0], device='cuda') / 8            # 1-D taps: the separable form the models pass
SHAPES = [(1024, 64, 36, 64), (1024, 128, 18, 32), (1024, 64, 64, 64), (1024, 128, 32, 32), (512, 256, 16, 16)]      # output of up2 / input of down2
for n, c, h, w in SHAPES:
    big = torch.randn(n, c, h, w, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    small = torch.randn(n, c, h // 2, w // 2, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for name, fn, nbytes in (('down2', lambda: upfirdn2d.downsample2d(big, f, down=2), big.numel() * 2 * 1.25), ('up2', lambda: upfirdn2d.upsample2d(small, f, up=2), big.numel() * 2 * 1.25)):
        with torch.no_grad():
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f'{name:6s} [{n},{c},{h},{w}]  {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s')
