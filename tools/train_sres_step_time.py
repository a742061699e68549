"""train_sres step time (update_G + update_D + ADA / 4 + EMA, no R1; 16 segment pairs in micro-batches of 2, ADA p = 0.2), eager against graph mode.
MEASUREMENT TOOL (GPU).

    python tools/train_sres_step_time.py [steps] [segments per micro-batch = 2] [modes = eager,graph]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
if os.path.isdir(_DB) and os.access(_DB, os.W_OK):
    os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_DB, 'db'))
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_DB, 'cache'))

import torch
from lvg.train_sres import SuperResTrainer

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
MICRO = int(sys.argv[2]) if len(sys.argv) > 2 else 2
MODES = (sys.argv[3] if len(sys.argv) > 3 else 'eager,graph').split(',')
dev = torch.device('cuda', 0)
for graphs in [m == 'graph' for m in MODES]:
    torch.manual_seed(0)
    tr = SuperResTrainer(device=dev, compute_dtype=torch.float16, G_grad_accum=16 // MICRO, D_grad_accum=16 // MICRO, augment_p_init=0.2, overlap_grad_sync=not graphs,
                         with_ema=True, use_graphs=graphs)
    lr = torch.rand(16, 3, tr.context_seq_length, 36, 64, device=dev) * 2 - 1
    hr = torch.rand(16, 3, tr.seq_length, 144, 256, device=dev) * 2 - 1
    n = 1
    for _ in range(2):
        tr.train_step(n, lr, hr, r1_interval=0); n += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        tr.train_step(n, lr, hr, r1_interval=0); n += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    print(f'graphs={graphs}, {MICRO} segments per micro-batch: {dt * 1e3:.1f} ms/step, {16 * 8 / dt:.1f} frames/s, augment p {float(tr.augment.p):.6f}, memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
    del tr
    torch.cuda.empty_cache()
