"""Every upfirdn2d launch of one training-shaped step of the lres pair (bench.py's main step: generator forward, discriminator forward,
backward; bfloat16, batch 8 x 128 frames), replayed alone: time and bytes per distinct call. MEASUREMENT TOOL (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
import torch, torch.nn.functional as F
from torch_utils.ops import upfirdn2d as U
from lvg.models.lres import VideoGenerator, VideoDiscriminator
calls = {}
orig = U._launch
def spy(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain):
    key = (tuple(x.shape), tuple(x.stride()), x.dtype, None if f is None else tuple(f.shape), upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter)
    if key not in calls:
        calls[key] = [0, f, gain]
    calls[key][0] += 1
    return orig(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)
U._launch = spy
torch.manual_seed(0)
B, T = int(os.environ.get('B', 8)), 128
G = VideoGenerator().cuda().requires_grad_(True)
D = VideoDiscriminator(seq_length=T, max_edge=64).cuda().requires_grad_(True)
video = G(B, T, dtype=torch.bfloat16) if 'dtype' in G.forward.__code__.co_varnames else G(B, T)
F.softplus(-D(video, dtype=torch.bfloat16)).mean().backward()       # (the discriminator in the compute dtype too: rounds 3-4 ran it in float32 here)
torch.cuda.synchronize()
U._launch = orig
rows = []
for key, (cnt, f, gain) in calls.items():
    shape, stride, dtype, fshape, upx, upy, downx, downy, px0, px1, py0, py1, flip = key
    n = max(s * st for s, st in zip(shape, stride))
    x = torch.randn(int(n) + 16, device='cuda').to(dtype).as_strided(shape, stride)
    fn = lambda: orig(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain)
    y = fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); b.synchronize()
    us = a.elapsed_time(b) * 100
    byts = (x.numel() + y.numel()) * x.element_size()
    rows.append((cnt * us, cnt, us, byts / us / 1e6, shape, stride[1] == 1, str(dtype)[6:], fshape, (upx, upy, downx, downy), tuple(y.shape)))
tot = sum(r[0] for r in rows); totb = sum(r[1] * r[3] * r[2] for r in rows)
print(f'total {tot:.0f} us in {sum(r[1] for r in rows)} launches; family average {totb / tot:.2f} TB/s')
for r in sorted(rows, reverse=True):
    print(f'{r[0]:8.0f} us = {r[1]:2d} x {r[2]:7.1f} us  {r[3]:5.2f} TB/s  x{r[4]} nhwc={r[5]} {r[6]} f{r[7]} up/down{r[8]} -> {r[9]}')
