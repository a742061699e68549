"""The two float32 GPU routes of the lres networks (library vs split operands), call by call through _TapConvEpilogue._backward: relative
difference of the incoming gradient and of every result. The first call whose inputs agree and whose outputs do not is the faulty one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F
from conftest import load_golden
from helpers.named_fill import fill_named
from lvg.models import lres
from lvg.models.lres import VideoGenerator, VideoDiscriminator
T = 16
g = load_golden('lres_models')
rec = []
orig = lres._TapConvEpilogue._backward
def spy(ctx, x, weight, ysum, pre, b, res, post, dout, wt_packed, need):
    out = orig(ctx, x, weight, ysum, pre, b, res, post, dout, wt_packed, need)
    rec[-1].append(dict(shape=(tuple(x.shape), tuple(weight.shape)), cfg=ctx.cfg, plain=ctx.plain, dout=dout.detach().double().cpu(),
                        ysum=None if ysum is None else ysum.detach().double().cpu(), pre=None if pre is None else pre.detach().double().cpu(), b=None if b is None else b.detach().double().cpu(), res=None if res is None else res.detach().double().cpu(),
                        outs=[None if t is None else t.detach().double().cpu() for t in out]))
    return out
lres._TapConvEpilogue._backward = staticmethod(spy)


def run(split):
    lres.SPLIT_F32 = split
    rec.append([])
    G, D = VideoGenerator(), VideoDiscriminator(seq_length=T, max_edge=64)
    fill_named(G); fill_named(D)
    G, D = G.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
    noise = torch.tensor(g['noise'], device='cuda')
    ws = G.compute_latent_ws(G.temporal_emb.blur(noise), T)
    video = G.synthesize_video(G._temporal_input(ws), ws, T)
    F.softplus(-D(video)).mean().backward()

run(False); run(True)
rel = lambda a, b: float('nan') if a is None or b is None else float((a - b).abs().max() / (a.abs().max() + 1e-300))
names = ('gx', 'gw', 'd_pre', 'd_b', 'd_res', 'd_post')
print(len(rec[0]), len(rec[1]))
def keyed(lst):
    seen, out = {}, {}
    for r in lst:
        k = (r['shape'], r['cfg'][2], r['plain'])
        seen[k] = seen.get(k, 0) + 1
        out[k + (seen[k],)] = r
    return out
A, B = keyed(rec[0]), keyed(rec[1])
for i, r in enumerate(rec[1]):
    k = [kk for kk, v in B.items() if v is r][0]
    if k not in A:
        print(f'{i:3d} x{r["shape"][0]} w{r["shape"][1]} act={r["cfg"][2]} plain={r["plain"]}   (split route only)')
        continue
    a, b = A[k], r
    line = f'{i:3d} x{a["shape"][0]} w{a["shape"][1]} act={a["cfg"][2]} plain={a["plain"]} dout {rel(a["dout"], b["dout"]):.1e} ysum {rel(a["ysum"], b["ysum"]):.1e} |'
    for nme, p, q in zip(names, a['outs'], b['outs']):
        if p is not None and q is not None:
            line += f' {nme} {rel(p, q):.1e}'
    print(line)

# sign pattern of the pre-activations of every activated call: z = ysum * pre + b (+ res)
def z_of(r):
    z = r['ysum']
    f = z.shape[0]
    if r['pre'] is not None: z = z * r['pre'].reshape(f, -1, 1, 1)
    if r['b'] is not None: z = z + r['b'].reshape(1, -1, 1, 1)
    if r['res'] is not None: z = z + r['res']
    return z
for i, r in enumerate(rec[1]):
    k = [kk for kk, v in B.items() if v is r][0]
    if k not in A or r['cfg'][2] == 'linear' or r['ysum'] is None: continue
    za, zb = z_of(A[k]), z_of(r)
    flips = (za.sign() != zb.sign())
    print(f'{i:3d} x{r["shape"][0]} sign flips {int(flips.sum())} of {za.numel()}; |z| of the flipped / max|z|: {[float(v) for v in (za[flips].abs() / za.abs().max())][:6]}  smallest |z|/max overall {float(za.abs().min() / za.abs().max()):.1e}')
