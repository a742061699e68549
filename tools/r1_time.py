"""MEASUREMENT TOOL: LowResTrainer.update_r1 (bf16, 128 frames, 8 clips per micro-batch) on the hand-written second-order nodes vs the
library's kt-convolution form. usage: python tools/r1_time.py [clips]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'long-video-gan_amd'))
import torch
from lvg.models import lres
from lvg.train_lres import LowResTrainer
from torch_utils.ops import conv3d_frames

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
tr = LowResTrainer(seq_length=128, device='cuda', compute_dtype=torch.bfloat16, G_grad_accum=1, D_grad_accum=1, overlap_grad_sync=False, with_ema=False)
real = torch.rand(clips, 3, 128, 36, 64, device='cuda') * 2 - 1
for hand in (True, False, True):
    lres.HAND_SECOND_ORDER = hand
    tr.update_r1(real, gain=16.0)
    torch.cuda.synchronize()
    n0 = conv3d_frames.stats['launches']
    t0 = time.perf_counter()
    for _ in range(3):
        tr.update_r1(real, gain=16.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'update_r1 {clips} clips x 128 frames bf16, hand second order = {hand}: {dt * 1e3:.1f} ms ({(conv3d_frames.stats["launches"] - n0) // 3} hand-kernel launches per update)')
