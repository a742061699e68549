"""MEASUREMENT TOOL (GPU): what the library's own streaming kernels reach on this box by tensor size (copy_, add, sum, fill; bfloat16) -- the practical ceiling
next to which tools/stream_bw.py puts the hand-written streams (round 6: add 6.0-6.2 TB/s on 0.3-2.4 GB tensors, copy_ 4.6-5.4, fill 6.9, sum 3.7)."""
import torch
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it*1e-3
for mb in (64, 128, 302, 604, 1208, 2416):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device='cuda', dtype=torch.bfloat16); y = torch.empty_like(x)
    dt = t(lambda: y.copy_(x)); print(f'copy {mb} MB: {2*n*2/dt/1e12:.2f} TB/s ({dt*1e6:.0f} us)')
    dt = t(lambda: torch.add(x, 1.0, out=y)); print(f'  add  {mb} MB: {2*n*2/dt/1e12:.2f} TB/s')
    dt = t(lambda: x.sum()); print(f'  sum  {mb} MB (read only): {n*2/dt/1e12:.2f} TB/s')
    dt = t(lambda: y.fill_(1.0)); print(f'  fill {mb} MB (write only): {n*2/dt/1e12:.2f} TB/s')
