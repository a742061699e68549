import sys, os, torch, time
sys.path.insert(0, 'long-video-gan_amd')
from torch_utils.ops import filtered_lrelu as fl
for shape in [(16,512,92,148),(16,512,94,150),(16,128,166,278),(16,512,56,84),(16,181,164,276),(16,512,38,52),(3,5,7,9)]:
    dx = torch.randn(shape, device='cuda').half()
    for _ in range(3): fl._bias_grad(dx)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): got = fl._bias_grad(dx)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    want = dx.double().sum([0,2,3])
    err = float((got.double()-want).abs().max()/ (want.abs().max()+1e-9))
    print(shape, f'{dt*1e6:.1f} us', f'{dx.numel()*2/dt/1e9:.0f} GB/s', 'rel err', f'{err:.2e}')
