import torch, time, os, sys
import torch.nn.functional as F
dev='cuda'
def timeit(fn, iters=5):
    fn(); fn(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter()-t)/iters
shapes=[(2,512,80,9,16,512,(3,3,3)), (2,128,128,18,32,128,(1,3,3)), (2,64,128,36,64,64,(1,3,3)), (2,256,144,9,16,256,(3,3,3)), (2,64,128,32,32,128,(5,3,3))]
for dtype in (torch.bfloat16, torch.float32):
  for (n,ci,t,h,w,co,k) in shapes:
    pad=tuple(x//2 for x in k)
    flops=2*n*co*t*h*w*ci*k[0]*k[1]*k[2]
    for cl in (False, True):
        x=torch.randn(n,ci,t,h,w,device=dev,dtype=dtype); wt=torch.randn(co,ci,*k,device=dev,dtype=dtype)
        if cl:
            x=x.contiguous(memory_format=torch.channels_last_3d); wt=wt.contiguous(memory_format=torch.channels_last_3d)
        x.requires_grad_(True); wt.requires_grad_(True)
        try:
            tf=timeit(lambda: F.conv3d(x,wt,padding=pad))
            y=F.conv3d(x,wt,padding=pad); g=torch.randn_like(y)
            def fb():
                y=F.conv3d(x,wt,padding=pad); y.backward(g)
            tb=timeit(fb)
            print(f'{str(dtype):15s} cl={cl} {n,ci,t,h,w,co,k}: fwd {tf*1e3:8.2f} ms {flops/tf/1e12:7.1f} TF | fwd+bwd {tb*1e3:8.2f} ms {3*flops/tb/1e12:7.1f} TF  out_cl={y.is_contiguous(memory_format=torch.channels_last_3d)}', flush=True)
        except Exception as e:
            print('ERR', dtype, cl, (n,ci,t,h,w,co,k), str(e)[:200])
