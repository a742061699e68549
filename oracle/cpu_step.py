"""CPU leg of bench.py (`cpu_baseline`): the low-resolution generator step on the host cores in float32.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see lvg_oracle.c header).

Two kinds, reported in `kind`:
  * "reference": when the reference checkout is importable (REFERENCE_ROOT, default /root/reference -- the build
    container), its OWN networks (model/generator_lres.py, model/discriminator_lres.py) are timed; on CPU tensors
    their ops dispatch to the reference's `impl='ref'` path (bias_act.py:84-86, upfirdn2d.py:160-162).
  * "port": anywhere else (the GPU box has no reference checkout): this repo's networks with every custom op
    taking the plain-PyTorch definition -- the same arithmetic as the reference's CPU fallback (pinned by
    tests/test_ops_ref_cpu.py and the model goldens made from the reference itself).
The dense contractions are PyTorch's CPU kernels in both.

Bounded samples (~10-30 s of CPU work in total), each the median of a few repetitions after a warm-up:
  * the step of the GPU leg (G forward, D forward, backward) at 16 frames, batch 1 (BASELINE.json configs[0] size)
    on `cores` threads -- the headline `value` -- and on ONE thread;
  * the forward pass at the GPU leg's 128 frames on `cores` threads (BASELINE.md section 2 shape).
`cores` = min(host threads, 16): more intra-op threads only add contention on these small tensors (measured on the
GPU box's 256-thread host: one step took 353 s instead of ~3 s)."""

import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'long-video-gan_amd')
REFERENCE_ROOT = os.environ.get('REFERENCE_ROOT', '/root/reference')

MAX_THREADS = 16


def _networks(frames, forward_only):
    """(G, D, kind). The reference's own classes when its checkout is present, else this repo's."""
    ref_ok = os.path.isfile(os.path.join(REFERENCE_ROOT, 'model', 'generator_lres.py')) and not os.environ.get('LVG_CPU_BASELINE_PORT')
    if ref_ok:
        sys.dont_write_bytecode = True
        sys.path.insert(0, REFERENCE_ROOT)
        try:
            from model import generator_lres, discriminator_lres
            kind = 'reference'
            G = generator_lres.VideoGenerator()
            D = discriminator_lres.VideoDiscriminator(seq_length=frames, max_edge=64)
            return G.requires_grad_(not forward_only), D.requires_grad_(False), kind
        except Exception:  # pylint: disable=broad-except
            sys.path.remove(REFERENCE_ROOT)
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    from lvg.models import lres
    lres.CHANNELS_LAST = False          # NCHW is the fast layout for PyTorch's CPU convolutions
    G = lres.VideoGenerator().requires_grad_(not forward_only)
    D = lres.VideoDiscriminator(seq_length=frames, max_edge=64).requires_grad_(False)
    return G, D, 'port'


def _median_time(fn, budget_s, max_reps):
    t_begin = time.perf_counter()
    t0 = time.perf_counter()
    fn()                                                  # first call doubles as warm-up ...
    first = time.perf_counter() - t0
    times = []
    while len(times) < max_reps and (time.perf_counter() - t_begin) + (times[-1] if times else first) < budget_s:
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    if not times:                                         # ... and is the sample if the budget is already spent
        times = [first]
    times.sort()
    return times[len(times) // 2], len(times)


def run(forward_only=False, frames=16, budget_s=14.0, max_reps=5):
    cores = min(os.cpu_count() or 1, MAX_THREADS)
    prev = torch.get_num_threads()
    try:
        torch.manual_seed(0)
        G, D, kind = _networks(frames, forward_only)

        def step():
            if forward_only:
                with torch.no_grad():
                    return G(1, frames)
            for p in G.parameters():
                p.grad = None
            F.softplus(-D(G(1, frames))).mean().backward()

        torch.set_num_threads(cores)
        med, reps = _median_time(step, budget_s, max_reps)
        torch.set_num_threads(1)
        med1, reps1 = _median_time(step, 8.0, 2)
        torch.set_num_threads(cores)

        def fwd128():
            with torch.no_grad():
                return G(1, 128)
        med128, reps128 = _median_time(fwd128, 8.0, 3)
        what = 'forward' if forward_only else 'forward + D forward + backward'
        impl = ("the reference's own networks and impl='ref' ops (imported from " + REFERENCE_ROOT + ')') if kind == 'reference' else \
               'NOT the reference\'s code (no reference checkout on this box): this repo\'s networks with the plain-PyTorch op definitions (the reference\'s CPU-fallback arithmetic, a different network decomposition)'
        return dict(value=round(frames / med, 3), unit='frames/s', cores=cores, kind=kind,
                    sample=f'{reps} x [G(1,{frames}) float32 {what}] on {cores} of {os.cpu_count()} host threads, median {med:.2f} s/step; {impl}',
                    one_thread=dict(value=round(frames / med1, 3), unit='frames/s', cores=1, sample=f'{reps1} x the same step, median {med1:.2f} s'),
                    forward_128_frames=dict(value=round(128 / med128, 3), unit='frames/s', cores=cores,
                                            sample=f'{reps128} x [G(1,128) float32 forward], median {med128:.2f} s'))
    finally:
        torch.set_num_threads(prev)


if __name__ == '__main__':
    import json
    print(json.dumps(run(forward_only='--forward-only' in sys.argv)))
