"""MEASUREMENT TOOL (GPU): the streaming kernels of the lres step on step-sized channels-last tensors next to a library elementwise kernel of the same traffic."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'long-video-gan_amd'))
import torch
from torch_utils.ops import bias_act, modconv_epilogue
from torch_utils.ops.modconv_epilogue import tap_gather_backward

def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e-3

for shape in ([1024, 64, 36, 64], [1024, 128, 18, 32], [512, 512, 9, 16]):
    x = torch.randn(*shape, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(shape[1], device='cuda').to(torch.bfloat16)
    nbytes = x.numel() * 2
    y = torch.empty_like(x)
    dt = t(lambda: torch.add(x, 1.0, out=y)); print(f'{shape}: aten add (1 read + 1 write) {2 * nbytes / dt / 1e12:.2f} TB/s ({dt * 1e6:.0f} us)')
    dt = t(lambda: bias_act.bias_act(x, b, act='lrelu', clamp=256)); print(f'   bias_act lrelu fwd {2 * nbytes / dt / 1e12:.2f} TB/s ({dt * 1e6:.0f} us)')
    pre = torch.rand(shape[0], shape[1], device='cuda') + 0.5
    post = torch.rand(shape[0], shape[1], device='cuda') + 0.5
    fn = lambda: modconv_epilogue._launch_fwd(x, pre, b, post, True, 3, 0.2, 2.0 ** 0.5, 256.0, True)
    dt = t(fn); print(f'   modconv_epilogue fwd (+msq) {2 * nbytes / dt / 1e12:.2f} TB/s ({dt * 1e6:.0f} us)')
    fn2 = lambda: modconv_epilogue._launch_fwd(x, pre, b, post, True, 3, 0.2, 2.0 ** 0.5, 256.0, True, True)
    dt = t(fn2); print(f'   modconv_epilogue dual fwd (1 read + 2 writes) {3 * nbytes / dt / 1e12:.2f} TB/s ({dt * 1e6:.0f} us)')
    dout = torch.randn_like(x)
    fn3 = lambda: tap_gather_backward(dout, x, pre, b, None, post, 1, 8, act='lrelu', clamp=256.0)
    dt = t(fn3); print(f'   tap_gather_backward, 1 tap (2 reads + 1 write + reductions) {3 * nbytes / dt / 1e12:.2f} TB/s ({dt * 1e6:.0f} us)')
