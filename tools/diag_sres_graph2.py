"""MEASUREMENT TOOL (round 5): forward / backward of the sres networks replayed from a hipGraph vs eager, piece by piece."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'long-video-gan_amd'))
import torch
import torch.nn.functional as F
from lvg.train_sres import SuperResTrainer

dtype = torch.float32 if len(sys.argv) < 2 else getattr(torch, sys.argv[1])
kw = dict(augment_real_sign_target=None, augment_p_init=0.0, in_augment_p=0.0, lr_cond_prob=1.0, G_grad_accum=1, D_grad_accum=1, overlap_grad_sync=False)
torch.manual_seed(0)
tr = SuperResTrainer(device='cuda', compute_dtype=dtype, use_graphs=False, **kw)
g = torch.Generator(device='cuda').manual_seed(1)
lr = torch.rand(2, 3, tr.context_seq_length, 36, 64, device='cuda', generator=g) * 2 - 1
hr = torch.rand(2, 3, tr.seq_length, 144, 256, device='cuda', generator=g) * 2 - 1

def rel(a, b): return float((a - b).abs().max()) / max(float(a.abs().max()), 1e-30)

def capture(fn):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): out = fn()
    return gr, out

# 1. generator forward
with torch.no_grad():
    ye = tr.G(lr).clone()
    gr, yg = capture(lambda: tr.G(lr))
    gr.replay(); torch.cuda.synchronize()
    print('G forward eager vs eager', rel(ye, tr.G(lr)), ' eager vs graph', rel(ye, yg), flush=True)
# 2. discriminator forward
lrc = tr.crop_to_seq_length(lr)
with torch.no_grad():
    de = tr.run_D(lrc, hr).clone()
    gr, dg = capture(lambda: tr.run_D(lrc, hr))
    gr.replay(); torch.cuda.synchronize()
    print('D forward eager vs graph', rel(de, dg), flush=True)
# 3. generator forward + backward through a plain loss
tr.G.requires_grad_(True)
def gb():
    tr.G_sync.zero()
    tr.G(lr).square().mean().backward()
gb(); ge = tr.G_sync.flat.clone(); gb(); ge2 = tr.G_sync.flat.clone()
gr, _ = capture(gb); gr.replay(); torch.cuda.synchronize()
print('G fwd+bwd eager vs eager', rel(ge, ge2), ' eager vs graph', rel(ge, tr.G_sync.flat), flush=True)
# 4. update_G's phase: through the discriminator
def gd():
    tr.G_sync.zero()
    F.softplus(-tr.run_D(lrc, tr.G(lr))).mean().backward()
gd(); ge = tr.G_sync.flat.clone()
gr, _ = capture(gd); gr.replay(); torch.cuda.synchronize()
print('G through D eager vs graph', rel(ge, tr.G_sync.flat), flush=True)
tr.G.requires_grad_(False)
# 5. discriminator fwd + bwd
tr.D.requires_grad_(True)
def db():
    tr.D_sync.zero()
    F.softplus(tr.run_D(lrc, hr)).mean().backward()
db(); dge = tr.D_sync.flat.clone()
gr, _ = capture(db); gr.replay(); torch.cuda.synchronize()
print('D fwd+bwd eager vs graph', rel(dge, tr.D_sync.flat), flush=True)
