"""MEASUREMENT TOOL (round 5): how far apart are two eager runs / an eager and a graph-replayed run of one trainer step with every random draw
pinned? Decides the yardstick of tests/test_trainer_gpu.py::test_graph_mode_trains_like_eager_mode and of the sres twin.
usage: python tools/diag_graph_determinism.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'long-video-gan_amd'))
import torch


def lres_runs(dtype, temp_scale):
    from lvg.train_lres import LowResTrainer
    kw = dict(seq_length=8, height=36, width=64, device='cuda', compute_dtype=dtype, G_grad_accum=2, D_grad_accum=2,
              overlap_grad_sync=False, with_ema=True, temp_scale_augment=temp_scale, diffaug_policy='')
    real = None
    out = {}
    for name, use_graphs in (('eager', False), ('eager2', False), ('graph', True), ('segmented', 'segmented')):
        torch.manual_seed(0)
        tr = LowResTrainer(use_graphs=use_graphs, **kw)
        if real is None:
            real = torch.rand(4, 3, 8, 36, 64, device='cuda') * 2 - 1
        draw, fixed = tr.G.sample_temporal_emb, {}

        def same_noise(batch, seq, generator=None, draw=draw, fixed=fixed):
            if (batch, seq) not in fixed:
                fixed[batch, seq] = draw(batch, seq, torch.Generator(device='cuda').manual_seed(100 * batch + seq))
            return fixed[batch, seq]
        tr.G.sample_temporal_emb = same_noise
        torch.manual_seed(5)
        tr.train_step(step=1, real_video=real, r1_interval=0)
        out[name] = (tr.G_sync.flat.clone(), tr.D_sync.flat.clone(),
                     torch.stack([b.float().reshape(()) for n, b in tr.G.named_buffers() if n.endswith('magnitude_ema')]))
        del tr
    return out


def sres_runs(dtype):
    from lvg.train_sres import SuperResTrainer
    kw = dict(augment_real_sign_target=None, augment_p_init=0.0, in_augment_p=0.0, lr_cond_prob=1.0, G_grad_accum=2, D_grad_accum=2, overlap_grad_sync=False)
    out = {}
    lr = hr = None
    for name, use_graphs in (('eager', False), ('eager2', False), ('graph', True), ('segmented', 'segmented')):
        torch.manual_seed(0)
        tr = SuperResTrainer(device='cuda', compute_dtype=dtype, use_graphs=use_graphs, **kw)
        if lr is None:
            lr = torch.rand(4, 3, tr.context_seq_length, 36, 64, device='cuda') * 2 - 1
            hr = torch.rand(4, 3, tr.seq_length, 144, 256, device='cuda') * 2 - 1
        draw, fixed = tr.G.sample_latent_z, {}

        def same_z(batch_size, generator_z=None, draw=draw, fixed=fixed):      # (captured and eager execution number the device generator differently)
            if batch_size not in fixed:
                fixed[batch_size] = draw(batch_size, torch.Generator(device='cuda').manual_seed(7 + batch_size))
            return fixed[batch_size]
        tr.G.sample_latent_z = same_z
        torch.manual_seed(5)
        tr.train_step(step=1, lr_video=lr, hr_video=hr, r1_interval=0, ada_interval=0)
        out[name] = (tr.G_sync.flat.clone(), tr.D_sync.flat.clone(),
                     torch.cat([b.float().flatten() for n, b in tr.G.named_buffers() if n.endswith('magnitude_ema') or n.endswith('w_avg')]))
        del tr
    return out


def report(tag, out):
    for i, what in enumerate(('G grad', 'D grad', 'stats')):
        e, e2, g = out['eager'][i], out['eager2'][i], out['graph'][i]
        m = float(e.abs().max())
        seg = f"  eager-segmented {float((e - out['segmented'][i]).abs().max()) / m:.3e}" if 'segmented' in out else ''
        print(f'{tag:28s} {what:7s} max |x| {m:.3e}  eager-eager2 {float((e - e2).abs().max()) / m:.3e}  eager-graph {float((e - g).abs().max()) / m:.3e}{seg}  (of max |x|)', flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', 'lres'):
        for dt, ts in ((torch.float32, 0.0), (torch.float32, 1.0), (torch.bfloat16, 1.0)):
            report(f'lres {str(dt)[6:]} stretch {ts}', lres_runs(dt, ts))
    if which in ('all', 'sres'):
        for dt in (torch.float32, torch.float16):
            report(f'sres {str(dt)[6:]}', sres_runs(dt))
