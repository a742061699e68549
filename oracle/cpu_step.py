"""CPU leg of bench.py (`cpu_baseline`): the same step -- low-resolution generator forward (+
discriminator forward and backward) -- restated on the host cores in float32.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see lvg_oracle.c header). The dense contractions run on
PyTorch's CPU kernels; every custom op takes the plain-PyTorch definition (the same arithmetic as
the reference's CPU fallback, `impl='ref'`: bias_act.py:91, upfirdn2d.py:167, pinned by
tests/test_ops_ref_cpu.py against fixtures made from the reference itself). kind = "port".

Bounded sample: 16-frame clips (BASELINE.json configs[0] shape) instead of 128, batch 1, a few
repetitions -- about 10-30 s of CPU work. The rate is reported in the metric's unit (frames/s)."""

import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'long-video-gan_amd')
if PKG not in sys.path:
    sys.path.insert(0, PKG)


MAX_THREADS = 16   # more intra-op threads than this only adds contention on these small tensors
                   # (measured: 256 threads on the GPU box's host made one step take 353 s instead of ~3 s)


def run(forward_only=False, frames=16, budget_s=20.0, max_reps=5):
    os.environ['LVG_CHANNELS_LAST'] = os.environ.get('LVG_CHANNELS_LAST', '1')
    from lvg.models import lres
    from lvg.models.lres import VideoDiscriminator, VideoGenerator
    cores = min(os.cpu_count() or 1, MAX_THREADS)
    prev = torch.get_num_threads()
    prev_cl = lres.CHANNELS_LAST
    torch.set_num_threads(cores)
    lres.CHANNELS_LAST = False          # NCHW is the fast layout for PyTorch's CPU convolutions
    try:
        torch.manual_seed(0)
        G = VideoGenerator().requires_grad_(not forward_only)
        D = VideoDiscriminator(seq_length=frames, max_edge=64).requires_grad_(False)

        def step():
            if forward_only:
                with torch.no_grad():
                    return G(1, frames)
            for p in G.parameters():
                p.grad = None
            F.softplus(-D(G(1, frames))).mean().backward()

        times = []
        t_begin = time.perf_counter()
        t0 = time.perf_counter()
        step()                                            # first step doubles as warm-up ...
        first = time.perf_counter() - t0
        while len(times) < max_reps and (time.perf_counter() - t_begin) + (times[-1] if times else first) < budget_s:
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        if not times:                                     # ... and is the sample if the budget is already spent
            times = [first]
        times.sort()
        med = times[len(times) // 2]
        return dict(value=round(frames / med, 3), unit='frames/s', cores=cores, kind='port',
                    sample=f'{len(times)} x [G(1,{frames}) float32 ' + ('forward' if forward_only else 'forward + D forward + backward') +
                           f'] on {cores} of {os.cpu_count()} host threads, median {med:.2f} s/step; plain-PyTorch op definitions (reference CPU-fallback arithmetic)')
    finally:
        torch.set_num_threads(prev)
        lres.CHANNELS_LAST = prev_cl


if __name__ == '__main__':
    import json
    print(json.dumps(run(forward_only='--forward-only' in sys.argv)))
