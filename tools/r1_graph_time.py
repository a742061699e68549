import os, sys, time
sys.path.insert(0, "long-video-gan_amd")
import torch
from lvg.train_lres import LowResTrainer
tr = LowResTrainer(seq_length=128, device="cuda", compute_dtype=torch.bfloat16, G_grad_accum=1, D_grad_accum=1, overlap_grad_sync=True, with_ema=False, use_graphs=True)
real = torch.rand(8, 3, 128, 36, 64, device="cuda") * 2 - 1
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.update_r1(real, gain=16.0)
    torch.cuda.synchronize()
    print(f"update_r1 graph mode call {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms; eager keys {tr._phase_graphs.eager_keys}; graphs {[k for k in tr._phase_graphs.graphs]}")
