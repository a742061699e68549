"""MEASUREMENT TOOL (GPU): lvg_conv2d_frames (channels-last result) + lvg_modconv2d_nhwc_to_nchw against lvg_conv2d_frames_planes (the convolution stores the NCHW
planes itself) on the super-resolution generator's forward layer shapes, 16 frames, float16."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'long-video-gan_amd'))
import torch
from torch_utils.ops import conv2d_frames as c2, modconv2d_layout as ml

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

n = 16
for ci, co, h, w in [(539, 512, 36, 36), (539, 512, 52, 52), (539, 512, 84, 84), (539, 512, 148, 148), (539, 362, 148, 148), (389, 256, 148, 148), (283, 181, 276, 276), (208, 128, 276, 276), (155, 128, 276, 276)]:
    cip, cop = c2.round_up(ci, c2.CH), c2.round_up(co, c2.CH)
    ho, wo = h + 2, w + 2
    x = torch.randn(n, ho + 6, wo + 18, cip, device='cuda', dtype=torch.float16)
    wp = torch.randn(3, 3, cop, cip, device='cuda', dtype=torch.float16) * 0.02
    demod = torch.rand(n, co, device='cuda') + 0.5
    t_cl = timeit(lambda: c2.conv2d_valid(x, wp, ho, wo, offset=(2, 2)))
    y = c2.conv2d_valid(x, wp, ho, wo, offset=(2, 2))
    t_tr = timeit(lambda: ml._frames_to_nchw(y, demod, co))
    t_pl = timeit(lambda: c2.conv2d_valid_planes(x, wp, ho, wo, co, offset=(2, 2), pre=demod))
    print(f'{ci:4d} -> {co:3d} @ {ho} x {wo}: channels-last {t_cl:7.1f} us + transposition {t_tr:6.1f} us = {t_cl + t_tr:7.1f} | planes {t_pl:7.1f} us ({t_pl / t_cl:.2f} x the convolution)')
