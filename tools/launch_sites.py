"""Which source lines launch the step's kernels: one eager generator update (G forward, D forward, backward) under torch.profiler with Python
stacks; every device kernel launch is attributed to the innermost frame inside this repository. Prints launches per (file:line, kernel family),
small kernels (< 12 us) first -- the map behind the per-step fixed cost.
usage: python tools/launch_sites.py [--batch 1] [--frames 128] [--top 80]"""
import argparse
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_DB, 'db'))
os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_DB, 'cache'))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

from lvg import ddp  # noqa: E402
from lvg.models import lres  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--frames', type=int, default=128)
ap.add_argument('--top', type=int, default=80)
ap.add_argument('--net', default='lres', choices=['lres', 'sres', 'train_sres', 'train_lres', 'r1_lres', 'r1_sres', 'd_sres'])
ap.add_argument('--first-iter', type=int, default=1, help='iteration number of the first profiled-run warm-up (train_lres: R1 runs when it divides by the R1 interval)')
ap.add_argument('--big', type=float, default=15.0, help='also list aten kernels of at least this many us (passes over activations written as tensor expressions)')
args = ap.parse_args()

torch.manual_seed(0)
dev = torch.device('cuda')
if args.net == 'lres':
    G = lres.VideoGenerator().to(dev).requires_grad_(True).train()
    D = lres.VideoDiscriminator(seq_length=args.frames, max_edge=64).to(dev).requires_grad_(False).train()
    sync = ddp.FlatGradSync(G.parameters(), overlap=False)
    dtype = torch.bfloat16

    def step():
        sync.zero()
        with lres.deferred_magnitude_sync():
            video = G(args.batch, args.frames, magnitude_ema_beta=0.999, dtype=dtype)
        F.softplus(-D(video, dtype=dtype)).mean().backward()
elif args.net == 'train_sres':
    # one full iteration of SuperResTrainer (update_G + update_D with fake generation, ADA on), eager, micro-batches of 2 segments
    from lvg.train_sres import SuperResTrainer
    tr = SuperResTrainer(device='cuda', compute_dtype=torch.float16, G_grad_accum=2, D_grad_accum=2, augment_p_init=0.2, overlap_grad_sync=False, with_ema=True)
    lr_clip = torch.rand(4, 3, tr.context_seq_length, 36, 64, device='cuda') * 2 - 1
    hr_clip = torch.rand(4, 3, tr.seq_length, 144, 256, device='cuda') * 2 - 1
    state = dict(n=1)

    def step():
        tr.train_step(state['n'], lr_clip, hr_clip)
        state['n'] += 1
elif args.net in ('r1_sres', 'd_sres'):
    # the R1 update / the discriminator update of SuperResTrainer alone (eager), `--batch` segments
    from lvg.train_sres import SuperResTrainer
    tr = SuperResTrainer(device='cuda', compute_dtype=torch.float16, G_grad_accum=1, D_grad_accum=1, augment_p_init=0.2, overlap_grad_sync=False, with_ema=False, use_graphs=False)
    nb = max(2, args.batch)
    lr_clip = torch.rand(nb, 3, tr.context_seq_length, 36, 64, device='cuda') * 2 - 1
    hr_clip = torch.rand(nb, 3, tr.seq_length, 144, 256, device='cuda') * 2 - 1

    def step():
        if args.net == 'r1_sres':
            tr.update_r1(tr.crop_to_seq_length(lr_clip), hr_clip, gain=16)
        else:
            tr.update_D(lr_clip, lr_clip, hr_clip)
elif args.net == 'r1_lres':
    # the R1 update alone (second-order pass through the discriminator, reference video_gan_lres.py:180-204)
    from lvg.train_lres import LowResTrainer
    tr = LowResTrainer(seq_length=args.frames, device=dev, compute_dtype=torch.bfloat16, G_grad_accum=1, D_grad_accum=1, overlap_grad_sync=False, with_ema=True)
    real = torch.rand(max(2, args.batch), 3, args.frames, 36, 64, device=dev) * 2 - 1

    def step():
        tr.update_r1(real, gain=16)
elif args.net == 'train_lres':
    from lvg.train_lres import LowResTrainer
    tr = LowResTrainer(seq_length=args.frames, device=dev, compute_dtype=torch.bfloat16, G_grad_accum=1, D_grad_accum=1, overlap_grad_sync=False, with_ema=True)
    real = torch.rand(max(2, args.batch), 3, args.frames, 36, 64, device=dev) * 2 - 1
    state = dict(n=args.first_iter)

    def step():
        tr.train_step(state['n'], real)
        state['n'] += 1
else:
    # the sres leg of bench.py: generator update of SuperResTrainer on two segments
    from lvg.train_sres import SuperResTrainer
    tr = SuperResTrainer(device='cuda', compute_dtype=torch.float16, augment_real_sign_target=None, augment_p_init=0.0,
                         in_augment_strength=0.0, lr_cond_prob=1.0, overlap_grad_sync=False, with_ema=False)
    lr_clip = torch.rand(2, 3, tr.context_seq_length, 36, 64, device='cuda') * 2 - 1

    def step():
        tr.G.requires_grad_(True)
        tr.G_sync.zero()
        logits = tr.run_D(tr.crop_to_seq_length(lr_clip), tr.G(lr_clip))
        F.softplus(-logits).mean().backward()
        tr.G.requires_grad_(False)


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()

events = prof.events()
# device kernels carry no stack; their launching CPU op does. Walk: kernel -> correlated launch -> enclosing cpu op with a stack.
by_site = collections.Counter()
dur_site = collections.Counter()
small_site = collections.Counter()
pat = re.compile(r'(long-video-gan_amd/[^\s(]+\.py)\((\d+)\): (\S+)')
# every CPU op event lists the device kernels it launched; count them at the innermost op and attribute them to the first frame inside this repository
for ev in events:
    if ev.device_type == torch.autograd.DeviceType.CUDA or not ev.kernels:
        continue
    if ev.cpu_children:            # count kernels at the innermost op only
        inner = [c for c in ev.cpu_children if c.kernels]
        if inner:
            continue
    site = 'autograd / unknown'
    cur = ev
    while cur is not None and site == 'autograd / unknown':
        for frame in (cur.stack or []):
            m = pat.search(frame)
            if m:
                site = f'{m.group(1).replace("long-video-gan_amd/", "")}:{m.group(2)} {m.group(3)}'
                break
        if site == 'autograd / unknown' and cur.name and ('Backward' in cur.name or 'AccumulateGrad' in cur.name):
            site = 'autograd node ' + cur.name[:60]
        cur = cur.cpu_parent
    for k in ev.kernels:
        name = re.sub(r'\(anonymous namespace\)::|void |at::native::', '', k.name)[:60]
        key = (site, ev.name[:40] + (' ' + str(ev.input_shapes)[:90] if getattr(ev, 'input_shapes', None) and k.duration >= args.big else ''), name)
        by_site[key] += 1
        dur_site[key] += k.duration
        if k.duration < 30:
            small_site[key] += 1
total = sum(by_site.values())
print(f'kernels in one eager step: {total}; under 30 us: {sum(small_site.values())}')
agg = collections.Counter()
agg_small = collections.Counter()
agg_us = collections.Counter()
for (site, op, name), c in by_site.items():
    agg[site] += c
    agg_small[site] += small_site[(site, op, name)]
    agg_us[site] += dur_site[(site, op, name)]
print('--- by source line (launches, of them small, total us)')
for site, c in agg.most_common(args.top):
    print(f'{c:5d} {agg_small[site]:5d} {agg_us[site]:9.0f}  {site}')
print(f'--- aten / library kernels of at least {args.big} us: launches, avg us, total us')
big = collections.Counter(); big_us = collections.Counter()
for (site, op, name), c in by_site.items():
    avg = dur_site[(site, op, name)] / c
    if avg >= args.big and (op.startswith('aten::') or 'Backward' in site) and not any(k in name for k in ('igemm', 'wgrad', 'filtered_lrelu', 'upfirdn', 'bias_act', 'epilogue', 'nhwc', 'nchw', 'Cijk', 'ada_', 'style_', 'weight_prep')):
        big[(site, op, name)] += c; big_us[(site, op, name)] += dur_site[(site, op, name)]
for key, us in big_us.most_common(args.top):
    print(f'{big[key]:5d} {us / big[key]:8.1f} {us:9.0f}  {key[0]} | {key[1]} | {key[2]}')
print('--- by (source line, op, kernel), small kernels')
for key, c in small_site.most_common(args.top):
    print(f'{c:5d} {dur_site[key] / max(by_site[key], 1):7.1f} us  {key[0]} | {key[1]} | {key[2]}')
