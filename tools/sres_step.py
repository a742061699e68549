"""The sres leg of bench.py alone (generator update of SuperResTrainer, 2 segments), for rocprofv3:
prints the duration of the timed window so tools/trace_window.py can cut the trace."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_DB, 'db'))
os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_DB, 'cache'))
import torch
from lvg.train_sres import SuperResTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
segments = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(0)
kw = {}
if os.environ.get('LVG_SRES_CL'):
    kw = dict(D_kwargs=dict(fp16_channels_last=True))
tr = SuperResTrainer(device='cuda', compute_dtype=torch.float16, augment_real_sign_target=None, augment_p_init=0.0,
                     in_augment_strength=0.0, lr_cond_prob=1.0, overlap_grad_sync=False, with_ema=False, **kw)
lr = torch.rand(segments, 3, tr.context_seq_length, 36, 64, device='cuda') * 2 - 1
import torch.nn.functional as F
for _ in range(2):
    tr.update_G(lr)
torch.cuda.synchronize()
graph = None
if os.environ.get('LVG_SRES_STEP_GRAPH', '1') == '1':          # the compute part replayed from a hipGraph, as bench.py's sres leg does
    def compute():
        tr.G.requires_grad_(True)
        tr.G_sync.zero()
        logits = tr.run_D(tr.crop_to_seq_length(lr), tr.G(lr))
        F.softplus(-logits).mean().backward()
        tr.G.requires_grad_(False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        compute()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        compute()
    graph.replay(); tr.G_sync.finish(); tr.G_opt.step()
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    if graph is not None:
        graph.replay(); tr.G_sync.finish(); tr.G_opt.step()
    else:
        tr.update_G(lr)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(window_ms=dt * 1e3, steps=steps, ms_per_step=dt * 1e3 / steps, frames_per_s=segments * 8 * steps / dt, mode='hipgraph' if graph is not None else 'eager')))
