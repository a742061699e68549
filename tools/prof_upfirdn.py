import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import torch
from torch_utils.ops import upfirdn2d
dev = torch.device('cuda')
case = sys.argv[1]; dtype = dict(f32=torch.float32, bf16=torch.bfloat16)[sys.argv[2]]
f = torch.tensor([0.125, 0.375, 0.375, 0.125], device=dev)
if case == 'down':
    x = torch.randn(4, 8192, 64, 64, device=dev).to(dtype); fn = lambda: upfirdn2d.downsample2d(x, f)
elif case == 'up':
    x = torch.randn(4, 8192, 18, 32, device=dev).to(dtype); fn = lambda: upfirdn2d.upsample2d(x, f)
elif case == 'tup':
    x = torch.randn(4, 256, 80, 144, device=dev).to(dtype); fn = lambda: upfirdn2d.upfirdn2d(x, f[:, None], up=(1, 2), padding=[0, 0, 2, 1], gain=2)
for _ in range(5): y = fn()
torch.cuda.synchronize(); print(y.shape)
