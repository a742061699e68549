"""Run one filtered_lrelu layer shape a few times (for rocprofv3)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import numpy as np, torch, scipy.signal
from torch_utils.ops import filtered_lrelu
warnings.simplefilter('ignore')
dev = torch.device('cuda')
case = sys.argv[1] if len(sys.argv) > 1 else 'L8'
dtype = dict(f16=torch.float16, f32=torch.float32, bf16=torch.bfloat16)[sys.argv[2] if len(sys.argv) > 2 else 'f16']
grad = len(sys.argv) > 3 and sys.argv[3] == 'grad'
k12 = torch.tensor(scipy.signal.firwin(numtaps=12, cutoff=0.45, width=0.3, fs=2.0).astype(np.float32), device=dev)
k24 = torch.tensor(scipy.signal.firwin(numtaps=24, cutoff=0.22, width=0.15, fs=2.0).astype(np.float32), device=dev)
cfg = dict(L8=([8, 512, 94, 150], 2, 2, k12, k12, [9, 8, 9, 8]), L10=([8, 256, 94, 150], 4, 2, k24, k12, [-6, -9, -6, -9]),
           L12=([8, 128, 166, 278], 2, 2, k12, k12, [9, 8, 9, 8]))[case]
shape, up, down, fu, fd, pad = cfg
x = torch.randn(*shape, device=dev).to(dtype).requires_grad_(grad)
b = torch.randn(shape[1], device=dev).to(dtype)
for _ in range(5):
    y = filtered_lrelu.filtered_lrelu(x, fu, fd, b, up=up, down=down, padding=pad, clamp=256)
    if grad:
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
print('done', tuple(y.shape))
