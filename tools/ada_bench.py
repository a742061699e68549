"""ADA augmentation pipeline on the GPU at the shape of the sres discriminator input (16 clips x 8 frames 144x256): the fused geometric
stage / colour pass (csrc/ada_augment.hip) against the composition of library ops (pad, upfirdn2d up, grid_sample, upfirdn2d down with the
margins read back to the host; bmm + elementwise). 'fused' = the shipped default (fused forward, gather-form adjoint backward), 'hybrid' =
LVG_ADA_WARP_GRAD=composed (fused launch only where no gradient is needed). MEASUREMENT TOOL (GPU).  python tools/ada_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from helpers.ada_cfg import TRAIN_SRES_KW
from lvg.ada_augment import AugmentPipe
from torch_utils.ops import ada_ops

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best

torch.manual_seed(0)
v = torch.randn(16, 3, 8, 144, 256, device='cuda')
pipe = AugmentPipe(**{**TRAIN_SRES_KW, 'noise': 1, 'cutout': 1}).cuda()
pipe.p.fill_(1.0)
sup_w, sup_c = ada_ops.warp_supported, ada_ops.colour_supported
from lvg import ada_augment as aa
for name, fused in (('fused', True), ('composed', False), ('hybrid', True), ('fused', True)):
    aa.WARP_GRAD = 'composed' if name == 'hybrid' else 'adjoint'
    ada_ops.warp_supported = sup_w if fused else (lambda *a: False)
    ada_ops.colour_supported = sup_c if fused else (lambda *a: False)
    with torch.no_grad():
        t_fwd = timeit(lambda: pipe(v))
    vv = v.clone().requires_grad_(True)
    def fb():
        vv.grad = None
        pipe(vv).square().mean().backward()
    t_fb = timeit(fb, 5)
    print(f'{name:9s} forward {t_fwd:7.2f} ms   forward + backward {t_fb:7.2f} ms   ({v.numel() * 4 / 1e6:.0f} MB clip)')
