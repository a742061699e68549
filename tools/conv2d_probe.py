import torch, time, sys
import torch.nn.functional as F
dev='cuda'
def timeit(fn, iters=5):
    fn(); fn(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter()-t)/iters
# (batch=frames, ci, h, w, co, k)
shapes=[(160,512,9,16,512,3), (288,256,9,16,256,3), (256,128,18,32,128,3), (256,64,36,64,64,3), (256,64,32,32,128,3), (96,512,5,8,512,3)]
dtype=torch.bfloat16
for (n,ci,h,w,co,k) in shapes:
    flops=2*n*co*h*w*ci*k*k
    for cl in (False, True):
        x=torch.randn(n,ci,h,w,device=dev,dtype=dtype); wt=torch.randn(co,ci,k,k,device=dev,dtype=dtype)
        if cl:
            x=x.contiguous(memory_format=torch.channels_last); wt=wt.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True); wt.requires_grad_(True)
        try:
            tf=timeit(lambda: F.conv2d(x,wt,padding=k//2))
            y=F.conv2d(x,wt,padding=k//2); g=torch.randn_like(y)
            def fb():
                y=F.conv2d(x,wt,padding=k//2); y.backward(g)
            tb=timeit(fb)
            print(f'cl={cl} {n,ci,h,w,co,k}: fwd {tf*1e3:7.3f} ms {flops/tf/1e12:7.1f} TF | fwd+bwd {tb*1e3:7.3f} ms {3*flops/tb/1e12:7.1f} TF', flush=True)
        except Exception as e:
            print('ERR', cl, (n,ci,h,w,co,k), str(e)[:200])
