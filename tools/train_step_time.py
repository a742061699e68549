"""train_lres step time (update_G + update_D + EMA, no R1), eager against graph mode. MEASUREMENT TOOL (GPU).

    python tools/train_step_time.py [clips] [micro-batches] [steps]"""
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
from lvg.train_lres import LowResTrainer

CLIPS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ACCUM = int(sys.argv[2]) if len(sys.argv) > 2 else 4
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device('cuda', 0)
for graphs in (False, True):
    torch.manual_seed(0)
    tr = LowResTrainer(seq_length=128, device=dev, compute_dtype=torch.bfloat16, G_grad_accum=ACCUM, D_grad_accum=ACCUM,
                       overlap_grad_sync=not graphs, with_ema=True, use_graphs=graphs)
    real = torch.rand(CLIPS, 3, 128, 36, 64, device=dev) * 2 - 1
    n = 1
    for _ in range(2):
        tr.train_step(n, real, r1_interval=0); n += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        tr.train_step(n, real, r1_interval=0); n += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    print(f'graphs={graphs}: {dt * 1e3:.1f} ms/step, {CLIPS * 128 / dt:.0f} frames/s, memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
    del tr
    torch.cuda.empty_cache()
