#!/usr/bin/env python3
"""bench.py -- frames/sec of the low-resolution generator hot path on MI355X.

One "step" = one generator update of train_lres on synthetic input: G forward (B clips of 128
frames, 36x64, bf16 activations) -> discriminator forward -> softplus(-logits).mean().backward()
-> data-parallel gradient exchange (RCCL all-reduce, no-op at N=1) -> Adam step. This is
BASELINE.json configs[1] ("generator_lres 128-frame 36x64 bf16 forward+backward on 1 MI355X"); for
N>1 every rank runs the same per-GPU batch (weak scaling) and the step includes the gradient
all-reduce of the 83.2 M generator parameters.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. `roofline` is measured live: every launch of the custom HIP ops in
the timed region is bracketed by HIP events on the launch stream, and the op with the largest
total time is reported against the HBM peak with its algorithmic bytes.
`cpu_baseline` (rank 0, N=1 only) times the CPU restatement of the same step (oracle/cpu_step.py)
on a bounded 16-frame sample on the host cores."""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'long-video-gan_amd'))
sys.path.insert(0, ROOT)

# MIOpen user find-db + kernel cache recorded on an MI355X for the convolution shapes of this benchmark
# (long-video-gan_amd/miopen_db, ~0.6 MB). A fresh box otherwise spends ~2 minutes searching/compiling the
# dense conv kernels before the first step; with the db the process starts in seconds. Missing entries
# (other batch sizes) are searched as usual and appended.
_MIOPEN_DB = os.path.join(ROOT, 'long-video-gan_amd', 'miopen_db')
if os.path.isdir(_MIOPEN_DB) and os.access(_MIOPEN_DB, os.W_OK):
    os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_MIOPEN_DB, 'db'))
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_MIOPEN_DB, 'cache'))

import torch
import torch.distributed as dist
import torch.nn.functional as F

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy ceiling)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / f16 MFMA peak (no sparsity)


class OpTimer:
    """Measures the custom HIP ops of one step with HIP events on the launch stream.

    Bracketing single launches inside an eager step over-counts short kernels whenever the stream is
    empty (the start event fires, then the GPU waits for the host to submit the kernel), so the
    launches of one step are RECORDED (arguments kept alive) and then each one is re-issued `reps`
    times back to back between one event pair: elapsed / reps is the kernel's own duration."""

    def __init__(self):
        self.calls = []       # (op name, fn, args, kwargs, algorithmic bytes)
        self.enabled = False

    def install(self):
        from torch_utils.ops import bias_act, modconv_epilogue, upfirdn2d

        def wrap(mod, name, op, bytes_fn):
            orig = getattr(mod, name)

            def recorded(*args, **kwargs):
                out = orig(*args, **kwargs)
                if self.enabled:
                    self.calls.append((op(args), orig, args, kwargs, bytes_fn(args, out, kwargs)))
                return out
            setattr(mod, name, recorded)

        # bias_act._launch(x, b, xref, yref, dy, grad, dim, act_id, alpha, gain, clamp): x, y (+ xref / yref / dy)
        def ba_bytes(args, out, kw):
            streams = 2 + sum(1 for t in args[2:5] if t is not None and t.numel() > 0)
            return out.numel() * out.element_size() * streams
        wrap(bias_act, '_launch', lambda a: 'bias_act_fwd' if a[5] == 0 else 'bias_act_bwd', ba_bytes)
        # bias_act._launch_grad_bias(dy, xref, yref, slots, ...): the backward launch that also leaves the bias gradient (dy, yref -> dx)
        wrap(bias_act, '_launch_grad_bias', lambda a: 'bias_act_bwd',
             lambda args, out, kw: out[0].numel() * out[0].element_size() * (2 + sum(1 for t in args[1:3] if t is not None and t.numel() > 0)))
        # upfirdn2d._launch(x, f, ...): (N_in + N_out) * s
        wrap(upfirdn2d, '_launch', lambda a: 'upfirdn2d', lambda args, out, kw: (args[0].numel() + out.numel()) * out.element_size())
        # modconv_epilogue: forward y -> out (2 streams), backward dout, y -> dy (3 streams)
        # (the dual form writes a second output / reads a second gradient: one stream more each way)
        wrap(modconv_epilogue, '_launch_fwd', lambda a: 'modconv_epilogue_fwd',
             lambda args, out, kw: (3 if kw.get('want_mid') else 2) * args[0].numel() * args[0].element_size())
        wrap(modconv_epilogue, '_launch_bwd', lambda a: 'modconv_epilogue_bwd',
             lambda args, out, kw: (4 if kw.get('dmid') is not None else 3) * args[0].numel() * args[0].element_size())
        # tapconv_epilogue (tap-stacked temporal conv, sum fused into the epilogue; lres binds the launch functions
        # by name, so they are wrapped there). forward: z (taps*N) -> out (+ saved sum) (+ residual); backward:
        # dout, saved sum (+ residual) -> dz (taps*N)
        from lvg.models import lres

        def tap_fwd_bytes(args, out, kw):
            z, res = args[0], args[3]
            streams = z.numel() + out[0].numel() + (out[1].numel() if out[1] is not None else 0) + (res.numel() if res is not None else 0)
            return streams * z.element_size()

        def tap_bwd_bytes(args, out, kw):
            dout, ysum, res = args[0], args[1], args[4]
            streams = dout.numel() + ysum.numel() + (res.numel() if res is not None else 0) + out[0].numel()
            return streams * ysum.element_size()
        wrap(lres, 'tap_gather_forward', lambda a: 'tapconv_epilogue_fwd', tap_fwd_bytes)
        wrap(lres, 'tap_gather_backward', lambda a: 'tapconv_epilogue_bwd', tap_bwd_bytes)
        # filtered_lrelu._fused(x, fu, fd, b, si, up, down, ...) -> (y, so, rc): (N_in + N_out) * s + mask bytes
        # (written in the forward of a training pass, read in its backward) -- SURVEY.md 8(d)
        from torch_utils.ops import filtered_lrelu

        def fl_bytes(args, out, kw):
            x, si = args[0], args[4]
            y, so, rc = out
            if rc < 0 or y is None:
                return 0
            mask = so.numel() if so is not None else (si.numel() if si is not None else 0)
            return (x.numel() + y.numel()) * x.element_size() + mask

        def fl_name(args):
            x, si, up, down = args[0], args[4], args[5], args[6]
            mode = 'bwd' if si is not None else 'fwd'
            return f'filtered_lrelu_u{up}d{down}_{mode}' if x.dtype != torch.float32 else f'filtered_lrelu_f32_u{up}d{down}_{mode}'
        wrap(filtered_lrelu, '_fused', fl_name, fl_bytes)
        # conv3d_frames.conv3d_frames_forward(x, weight, shift, ...): the hand-written implicit-GEMM convolution (forward
        # and, with the mirrored weight, data-gradient launches). MFMA-bound: its work is counted in FLOPs,
        # 2 * N_out * Cin * kt * kh * kw, not in bytes.
        from torch_utils.ops import conv3d_frames

        def conv_flops(args, out, kw):
            x, w = args[0], args[1]
            return 2 * x.shape[0] * x.shape[2] * x.shape[3] * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3] * w.shape[4]
        # 1 x 1 (skip) convolutions run on the same kernel but are a memory stream (K = Ci): counted apart from the dense contraction
        wrap(conv3d_frames, 'conv3d_frames_forward', lambda a: 'conv3d_igemm' if a[1].shape[2] * a[1].shape[3] * a[1].shape[4] > 1 else 'conv3d_igemm_1x1', conv_flops)
        # conv3d_frames_wgrad(x, dy, kt, kh, kw, shift): the hand-written weight gradient (its own kernel, csrc/conv3d_wgrad.hip)
        wrap(conv3d_frames, 'conv3d_frames_wgrad', lambda a: 'conv3d_wgrad',
             lambda args, out, kw: 2 * args[0].shape[0] * args[0].shape[2] * args[0].shape[3] * args[0].shape[1] * args[1].shape[1] * args[2] * args[3] * args[4])
        # conv2d_frames.conv2d_valid(x [N,Hi,Wi,Ci], wp [3,3,Co,Ci], ho, wo, ...): the 2-D implicit-GEMM convolution of the sres generator
        # (forward and data-gradient launches); conv2d_wgrad(x, dy): its weight gradient. ALGORITHMIC FLOPs (true channel counts and
        # output size, handed over by the caller as `alg_flops`; the launches themselves work on channels zero-padded to multiples of 64,
        # gradient frames padded to whole patches and -- float32 layers -- three partial products).
        from torch_utils.ops import conv2d_frames
        wrap(conv2d_frames, 'conv2d_valid', lambda a: 'conv2d_igemm',
             lambda args, out, kw: kw.get('alg_flops') or 2 * args[0].shape[0] * args[2] * args[3] * args[1].shape[2] * args[1].shape[3] * 9)
        # (round 6: the 16-bit layers' forward and data-gradient launches store NCHW planes themselves -- the same kernel, same family)
        wrap(conv2d_frames, 'conv2d_valid_planes', lambda a: 'conv2d_igemm',
             lambda args, out, kw: kw.get('alg_flops') or 2 * args[0].shape[0] * args[2] * args[3] * args[1].shape[2] * args[1].shape[3] * 9)
        wrap(conv2d_frames, 'conv2d_wgrad', lambda a: 'conv2d_wgrad',
             lambda args, out, kw: kw.get('alg_flops') or 2 * args[1].shape[0] * args[1].shape[1] * args[1].shape[2] * args[1].shape[3] * args[0].shape[3] * 9)
        # round 5: noise filter bank (float32 MFMA: FLOPs on the non-zero taps of the bank), the thin 1 x 1 kernels (streams over the wide tensor) and
        # the weight gradient of the wide 1 x 1 convolutions (a stream over x and dy)
        from torch_utils.ops import noise_bank, pointwise_thin
        wrap(noise_bank, 'noise_filter_bank', lambda a: 'noise_filter_bank',
             lambda args, out, kw: 2 * args[0].shape[0] * out.shape[2] * int((args[1] != 0).sum()))
        wrap(pointwise_thin, '_apply', lambda a: 'pointwise_thin', lambda args, out, kw: (args[0].numel() + out.numel()) * out.element_size())
        wrap(pointwise_thin, '_wgrad', lambda a: 'pointwise_thin', lambda args, out, kw: (args[0].numel() + args[1].numel()) * args[0].element_size())
        wrap(conv3d_frames, 'pointwise_wgrad', lambda a: 'pointwise_wgrad', lambda args, out, kw: (args[0].numel() + args[1].numel()) * args[0].element_size())
        self.flop_ops = {'conv3d_igemm', 'conv3d_igemm_1x1', 'conv3d_wgrad', 'conv2d_igemm', 'conv2d_wgrad', 'noise_filter_bank'}

    def measure(self, reps=3):
        """Time the recorded launches per op: ALL launches of that op from the step, once each and in step
        order, are captured into one hipGraph which is replayed `reps` times between one HIP event pair on
        the launch stream (a Python-issued launch costs ~20 us of host time, more than many of these
        kernels take, so eager back-to-back launches would measure the host). One replay streams several
        GB through HBM, far more than the 256 MiB Infinity Cache, so no launch finds its operands cached
        by its own previous run -- re-issuing a single launch back to back would (and measured 25 % fast)."""
        out = {}
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        by_op = {}
        for call in self.calls:
            if call[4] > 0:
                by_op.setdefault(call[0], []).append(call)
        for op, calls in by_op.items():
            g = torch.cuda.CUDAGraph()
            keep = []                                            # distinct output buffer per launch, as in the step
            with torch.cuda.graph(g, stream=side):
                for _, fn, args, kwargs, _ in calls:
                    keep.append(fn(*args, **kwargs))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay()                                           # first replay: upload
            torch.cuda.synchronize()
            a.record()
            for _ in range(reps):
                g.replay()
            b.record()
            b.synchronize()
            total_ms = a.elapsed_time(b) / reps
            nbytes = sum(c[4] for c in calls)
            out[op] = dict(launches=len(calls), total_ms=total_ms, bytes=nbytes,
                           gbps=nbytes / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0)
            del g, keep
            if os.environ.get('LVG_BENCH_VERBOSE'):
                self._dump_calls(op, calls, side)
        self.calls.clear()
        return out

    @staticmethod
    def _dump_calls(op, calls, side, reps=5):
        """Per-launch table (debug aid; each launch re-issued back to back, so cache-warm)."""
        for _, fn, args, kwargs, nbytes in calls:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(reps):
                    fn(*args, **kwargs)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay()
            a.record()
            g.replay()
            b.record()
            b.synchronize()
            us = a.elapsed_time(b) / reps * 1e3
            extra = [x for x in args[2:12] if isinstance(x, (int, float, bool))] if op == 'upfirdn2d' else []
            print(f'[op] {op:13s} x={tuple(args[0].shape)} strides={tuple(args[0].stride())} {args[0].dtype} {extra} {us:8.1f} us {nbytes / us / 1e3:8.1f} GB/s', file=sys.stderr)
            del g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch-per-gpu', type=int, default=8)
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--forward-only', action='store_true', help='time G forward alone (frames/sec/GPU lres-G forward)')
    ap.add_argument('--workload', default='update_G', choices=['update_G', 'train_lres'],
                    help='update_G: BASELINE configs[1] (default, the contract line). train_lres: the full iteration of train_lres.py:216-230 '
                         '(configs[2] body: update_G + update_D + R1 every 16th step + EMA, batch 32 / world, 2 micro-batches, 160-frame generator '
                         'clips cropped to 128, DiffAugment + temporal-scale augment) with backward-overlapped bucketed all-reduce, eager launches')
    ap.add_argument('--no-extra-legs', action='store_true', help='skip the forward-only / MFMA / super-resolution legs appended to the N=1 line')
    ap.add_argument('--selftest-launch', action='store_true',
                    help='only bring up the ranks (RCCL on GPUs, gloo without), all-reduce the rank ids and print one JSON line: checks the N>1 launch path')
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='replay the step from a captured hipGraph (removes ~2800 host launches per step)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # invoked plainly (`python bench.py --gpus N`): start the N ranks ourselves, one process per GPU, the way the
        # reference's launcher contract reads the environment (torch_utils/distributed.py:42-69: RANK / LOCAL_RANK /
        # WORLD_SIZE / MASTER_* with single-process defaults)
        sys.exit(_spawn_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.selftest_launch:
        return _launch_selftest(args, world, rank, local_rank)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get('LVG_FORCE_DIST'):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', init_method='env://')   # "nccl" = RCCL on ROCm
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from lvg import ddp
    from lvg.models import lres
    from lvg.models.lres import VideoDiscriminator, VideoGenerator

    dtype = dict(bf16=torch.bfloat16, fp32=torch.float32, fp16=torch.float16)[args.dtype]
    dev = torch.device('cuda', local_rank)
    if args.workload == 'train_lres':
        _train_lres_workload(args, world, rank, dev, dtype)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    torch.manual_seed(0)                       # same random-init weights on every rank
    G = VideoGenerator().to(dev).requires_grad_(True).train()
    D = VideoDiscriminator(seq_length=args.frames, max_edge=64).to(dev).requires_grad_(False).train()
    ddp.broadcast_module(G)
    ddp.broadcast_module(D)
    from lvg.optim import FlatAdam
    opt = FlatAdam(G.parameters(), lr=0.003, betas=(0.0, 0.99))       # one fused HIP launch over the flat parameter / moment / gradient buffers
    # Gradient exchange: ONE flat buffer (the .grad tensors are views into it), all-reduced after the
    # backward pass in 128 MB buckets. The compute part of the step is replayed from a hipGraph; the RCCL
    # collective and the optimizer stay outside the graph (an RCCL all-reduce inside a captured graph aborts
    # on this stack), which costs ~2-3 ms of non-overlapped all-reduce per step at N>1.
    sync = ddp.FlatGradSync(G.parameters(), overlap=False)
    torch.manual_seed(1 + rank)                # per-rank noise stream (train_lres.py:69)
    B, T = args.batch_per_gpu, args.frames

    timer = OpTimer()
    timer.install()

    # The generator pass tracks the input magnitude of every modulated layer (magnitude_ema_beta = 0.999; the
    # reference does this in the generator pass of update_D, video_gan_lres.py:140-144 -- timing it here makes
    # the measured pass a superset of update_G's). Across ranks the 21 per-convolution statistics are exchanged in ONE
    # all-reduce after the pass instead of 21 inside it (lres.MagnitudeEMA).
    ema_beta = 0.999
    pending_emas = []

    ASSIGN_GRADS = os.environ.get('LVG_BENCH_ASSIGN_GRADS', '1') != '0'      # (A/B: 0 = gradients added into the zeroed flat views)

    def compute():
        if args.forward_only:
            with torch.no_grad():
                return G(B, T, dtype=dtype)
        # one backward pass per step: autograd assigns the gradients (no add into zeroed views, ~100 launches), gather() copies them into
        # the flat buffer with a few multi-tensor launches -- inside the captured region, so every replay refills the flat buffer
        sync.zero(assign=ASSIGN_GRADS)
        with lres.deferred_magnitude_sync() as pending:
            video = G(B, T, magnitude_ema_beta=ema_beta, dtype=dtype)
        logits = D(video, dtype=dtype)
        F.softplus(-logits).mean().backward()
        if ASSIGN_GRADS:
            sync.gather()
        pending_emas[:] = [pending, lres.stack_pending(pending) if pending else None,
                           ddp.stat_sync_plan(pending) if pending and world > 1 else None]      # (built when the step is captured, reused by every replay)

    sync_events = []               # (start, end) HIP events around the gradient exchange of the timed steps

    def update():
        if not args.forward_only:
            if pending_emas and pending_emas[0]:
                lres.finish_magnitude_sync(*pending_emas)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sync.finish()          # all-reduce (mean) over ranks, nan_to_num -- no-op collective at N=1
            e1.record()
            sync_events.append((e0, e1))
            opt.step()

    def step():
        compute()
        update()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()

    graph = None
    if args.graph != 'off':
        # Capture one step into a hipGraph (all custom-op launches go to torch's current stream, which is
        # the capturing stream). Falls back to eager launches if capture is refused.
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # (more than one rank: the process group's watchdog thread must not invalidate this thread's capture)
            with torch.cuda.graph(graph, capture_error_mode='thread_local' if world > 1 else 'global'):
                compute()
            graph.replay()
            update()
            torch.cuda.synchronize()
        except Exception as err:  # pylint: disable=broad-except
            if args.graph == 'on':
                raise
            print(f'[bench] hipGraph capture refused ({type(err).__name__}: {err}); timing eager launches', file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    barrier()

    sync_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
            update()
        else:
            step()
    barrier()
    elapsed = time.perf_counter() - t0
    # the gradient exchange runs after the captured step, not under it: its device time is exposed time of the step
    exposed_sync_ms = sum(a.elapsed_time(b) for a, b in sync_events) / max(len(sync_events), 1) if sync_events else 0.0

    # Roofline of the custom ops: record the launches of ONE more step of the same workload, then time
    # each of them back to back with HIP events on the launch stream (see OpTimer).
    roofline_steps = 1
    ops = {}
    if not os.environ.get('LVG_BENCH_NO_ROOFLINE'):      # (set when tracing the timed region with rocprofv3)
        timer.enabled = True
        step()
        timer.enabled = False
        ops = timer.measure()

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    _flush_c_stdio()                       # every rank: library banners out before rank 0's JSON line
    if world > 1:
        dist.barrier()

    if rank == 0:
        frames = world * B * T * args.steps
        # Kernel families: the forward and backward launches of bias_act are instantiations of the same
        # streaming kernel (csrc/bias_act.hip), so they compete for "dominant" as one entry.
        fam = {}
        for k, v in ops.items():
            d = fam.setdefault(k.replace('_fwd', '').replace('_bwd', ''), dict(launches=0, total_ms=0.0, bytes=0))
            for key in d:
                d[key] += v[key]
        for d in fam.values():
            d['gbps'] = d['bytes'] / (d['total_ms'] * 1e-3) / 1e9 if d['total_ms'] > 0 else 0.0
        flop_ops = getattr(timer, 'flop_ops', set())

        def line(name):
            """roofline object of one kernel family: HBM-bound streams in GB/s, the convolution in TFLOP/s."""
            d = fam[name]
            common = dict(kernel=name, traffic=_pmc_traffic(name, 'lres'), traffic_source=_TRAFFIC_SOURCE.format(scope='lres'), launches=d['launches'], avg_launch_us=round(d['total_ms'] * 1e3 / d['launches'], 2),
                          measured_on='all launches of this kernel from one step, captured once each (step order) into a hipGraph replayed 3x between HIP events on the launch stream, right after the timed region')
            if name in flop_ops:
                tf = d['bytes'] / (d['total_ms'] * 1e-3) / 1e12          # the work unit of these entries is FLOPs
                return dict(bound='mfma', achieved=round(tf, 1), peak=MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=round(tf / MFMA_PEAK_TFLOPS, 4),
                            algorithmic_flops_per_launch=int(d['bytes'] / d['launches']), **common)
            return dict(bound='hbm', achieved=round(d['gbps'], 1), peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(d['gbps'] / HBM_PEAK_GBPS, 4),
                        algorithmic_bytes_per_launch=int(d['bytes'] / d['launches']), **common)
        dominant = max(fam, key=lambda k: fam[k]['total_ms']) if fam else None
        roofline = line(dominant) if dominant is not None else None
        # the dominant HBM-bound kernel as well (the convolution took over as the dominant hand-written kernel in round 2)
        hbm_fam = [k for k in fam if k not in flop_ops]
        roofline_hbm = line(max(hbm_fam, key=lambda k: fam[k]['total_ms'])) if hbm_fam else None
        result = {
            'metric': 'frames/sec lres-G 128x36x64 ' + ('forward' if args.forward_only else 'forward+backward (generator update)'),
            'value': round(frames / elapsed, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic', 'launch_mode': 'hipgraph' if graph is not None else 'eager',
            'config': {'workload': f'generator_lres {T}-frame 36x64 {args.dtype} ' + ('forward' if args.forward_only else 'forward+backward through discriminator_lres (magnitude EMA tracking on), Adam step') + f', batch {B}/GPU',
                       'global_batch': world * B, 'frames_per_clip': T, 'parallelism': f'dp{world}', 'params_G': 83215939},
            'roofline': roofline,
            'roofline_hbm': roofline_hbm,
            'ops': {k: (dict(launches=v['launches'], total_ms=round(v['total_ms'], 3), tflops=round(v['gbps'] / 1e3, 1)) if k in flop_ops else
                        dict(launches=v['launches'], total_ms=round(v['total_ms'], 3), gbps=round(v['gbps'], 1))) for k, v in ops.items()},
            'step_ms_in_custom_ops': round(sum(v['total_ms'] for v in ops.values()) / roofline_steps, 3),
            'grad_sync': {'bytes': int(sync.flat.numel() * 4), 'exposed_ms_per_step': round(exposed_sync_ms, 3), 'collective': 'RCCL all-reduce (mean) of the flat float32 gradient in 128 MB buckets + scale + nan_to_num, after the hipGraph replay (not overlapped)' if world > 1 else 'no collective at N = 1: scale + nan_to_num pass only',
                          'backend': dist.get_backend() if dist.is_initialized() else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = _cpu_baseline(forward_only=args.forward_only)
        if world == 1 and not args.no_extra_legs and not args.forward_only:
            # Same default invocation, further evidence: the headline forward-only rate, the
            # dense-contraction (MFMA) figure of the step, and BASELINE.json configs[3] (super-resolution pair)
            # with the filtered_lrelu roofline. Failures here never take the main line down.
            del graph
            legs = [('forward_only', lambda: _forward_only_leg(G, B, T, dtype, timer=timer)),
                    ('mfma', lambda: _mfma_leg(step, elapsed / args.steps)),
                    ('batch_sweep', lambda: _batch_sweep_leg(G, D, dtype, T)),
                    ('fp32', lambda: _fp32_leg(G, D, T)),
                    ('sres', lambda: _sres_leg(dev, timer))]
            if not os.environ.get('LVG_BENCH_NO_TRAIN_LEGS'):
                # BASELINE.json configs[2] and configs[4] at N = 1 (the driver's multi-GPU runs use --workload train_lres)
                # 16 timed iterations each: ONE R1 step (every 16th) falls inside the timed region, so `value` is the rate of the real
                # schedule, measured, not an amortisation done by hand
                legs += [('train_lres', lambda: _train_lres_run(1, 0, dev, dtype, T, steps=16, warmup=1, dtype_name=args.dtype)),
                         ('train_sres', lambda: _train_sres_leg(dev))]
            only = [n for n in os.environ.get('LVG_BENCH_LEGS', '').split(',') if n]      # (A/B measurements: run just these legs)
            for name, leg in legs:
                if only and name not in only:
                    continue
                try:
                    result[name] = leg()
                except Exception as err:  # pylint: disable=broad-except
                    result[name] = {'error': f'{type(err).__name__}: {err}'[:300]}
        _flush_c_stdio()
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def _flush_c_stdio():
    """RCCL prints a version banner through C stdio when a communicator comes up; on a pipe or file it is buffered until the process exits,
    i.e. it would land AFTER this script's JSON line. Flush it out now (every rank), so the JSON stays the last line of the output."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # pylint: disable=broad-except
        pass
    sys.stdout.flush()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _spawn_ranks(n):
    """Re-run this command line under torch.distributed.run: one process per GPU on this node, rendezvous on 127.0.0.1
    (the container hostname may not resolve). The ranks inherit stdout, so rank 0's JSON line is this process's output."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _launch_selftest(args, world, rank, local_rank):
    """The N>1 launch path without the workload: process group up (RCCL when GPUs are visible, gloo otherwise), one
    all-reduce of the rank ids, one JSON line on rank 0. Runs on a CPU-only host (tests/test_bench_launch.py)."""
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= max(world, 1)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    if use_gpu:
        torch.cuda.set_device(local_rank)
    dist.init_process_group('nccl' if use_gpu else 'gloo', init_method='env://')
    t = torch.tensor([float(rank)], device=torch.device('cuda', local_rank) if use_gpu else 'cpu')
    dist.all_reduce(t)
    dist.barrier()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    _flush_c_stdio()
    dist.barrier()
    if rank == 0:
        print(json.dumps({'launch_selftest': True, 'n_gpus': world, 'backend': dist.get_backend(), 'rank_sum': float(t.item())}), flush=True)
    dist.destroy_process_group()


def _train_lres_workload(args, world, rank, dev, dtype):
    """BASELINE.json configs[2] step body on synthetic video: LowResTrainer.train_step (reference train_lres.py:216-230,
    model/video_gan_lres.py:100-214) -- update_G, update_D, R1 on every 16th step, generator EMA -- total batch 32 split
    over the ranks, 2 micro-batches, G run at 160 frames and randomly cropped to 128, DiffAugment + temporal-scale
    augmentation in front of D. Gradients: FlatGradSync with overlap=True (each 128 MB bucket is all-reduced over
    RCCL from an autograd hook while the rest of the last backward still runs). Eager launches (the collectives
    cannot be captured into a hipGraph on this stack). One JSON line on rank 0, frames/s over all ranks."""
    res = _train_lres_run(world, rank, dev, dtype, args.frames, args.steps, args.warmup, dtype_name=args.dtype)
    _flush_c_stdio()
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(res), flush=True)


def _train_lres_run(world, rank, dev, dtype, frames_per_clip, steps, warmup, dtype_name='bf16', total_batch=32, accum=None):
    """-> the result dict (identical on every rank). Micro-batches: 2 per update as in the reference's 8-GPU recipe (batch 32 / 8 GPUs,
    grad_accum 2 => per-GPU micro-batch 2); at N = 1 the 32 clips are cut into micro-batches of 16."""
    from lvg.train_lres import LowResTrainer
    assert total_batch % world == 0
    B = total_batch // world
    if accum is None:
        # micro-batches of at most 16 clips (round 6; 8 before); ONE micro-batch whenever the rank's share is 16 clips or
        # fewer -- at 8 ranks that is 4 clips in one pass instead of the reference recipe's 2 x 2 (train_lres.py:65-69: its accumulation
        # exists to fit 32 clips into 8 x 32 GB; 288 GB of HBM do not need it, the gradient mean is the same, and the per-launch fixed cost
        # of the step is paid once instead of twice: VERDICT r03 item 4b). Measured at one rank, 32 clips, no R1, same call
        # (profiles/r06_train_lres_microbatch.log): 332.7 / 296.8 / 303.6 ms per iteration with 4 / 2 / 1 micro-batches (18 / 30 / 57 GiB).
        accum = max(1, B // int(os.environ.get('LVG_BENCH_LRES_MICRO', '16')))
    torch.manual_seed(0)
    # At every world size the compute of update_G / update_D is replayed from hipGraphs (LowResTrainer(use_graphs=True); the host draws of
    # the augmentations go through static buffers); the collectives stay outside the captured phases: the gradient exchange follows a
    # phase's replays, the generator's running statistics are exchanged in one all-reduce after the fake-generation replay
    # (lvg.phase_graphs). LVG_TRAIN_GRAPHS=0: eager launches with the bucketed exchange overlapped with backward (A/B).
    graphs = os.environ.get('LVG_TRAIN_GRAPHS', '1') != '0'
    tr = LowResTrainer(seq_length=frames_per_clip, device=dev, compute_dtype=dtype, G_grad_accum=accum, D_grad_accum=accum,
                       overlap_grad_sync=True, with_ema=True, use_graphs=graphs)
    torch.manual_seed(1 + rank)
    real = torch.rand(B, 3, frames_per_clip, 36, 64, device=dev) * 2 - 1
    tr.G_sync.exposed_events, tr.D_sync.exposed_events = [], []      # device time of the part of the exchange nothing hides (N > 1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    step_no = 1
    for _ in range(warmup):
        tr.train_step(step_no, real); step_no += 1
    r1_steps = sum(1 for k in range(warmup + 1, warmup + steps + 1) if k % 16 == 0)
    if r1_steps:
        tr.update_r1(real, gain=16)                                         # (first R1 pass: lazy initialisation outside the timed region)
    barrier()
    tr.G_sync.exposed_events.clear(); tr.D_sync.exposed_events.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(step_no, real); step_no += 1
    barrier()
    elapsed = time.perf_counter() - t0
    exposed_ms = sum(a.elapsed_time(b) for a, b in tr.G_sync.exposed_events + tr.D_sync.exposed_events) / steps
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # one R1 update on its own (warm), for the record; short runs that saw no R1 step (every 16th) also get the amortised rate
    tr.update_r1(real, gain=16)
    barrier()
    t1 = time.perf_counter()
    tr.update_r1(real, gain=16)
    barrier()
    r1_ms = (time.perf_counter() - t1) * 1e3
    extra = {'r1_update_ms': round(r1_ms, 2)}
    if r1_steps == 0:
        extra['value_with_r1_every_16'] = round(total_batch * frames_per_clip / (elapsed / steps + r1_ms * 1e-3 / 16), 2)
    del tr
    return {
        **extra,
        'metric': 'frames/sec train_lres iteration (update_G + update_D + R1/16 + EMA), 128-frame 36x64 clips',
        'value': round(total_batch * frames_per_clip * steps / elapsed, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': steps,
        'warmup': warmup, 'ms_per_step': round(elapsed / steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': dtype_name, 'data': 'synthetic',
        'launch_mode': 'hipgraph per phase (update_G, fake generation, update_D micro-batch, R1 micro-batch); optimizer and exchange eager' if graphs else 'eager',
        'config': {'workload': f'train_lres.py step body, total batch {total_batch} ({B}/GPU, {accum} micro-batches), G at {frames_per_clip + 32} frames cropped to {frames_per_clip}, '
                               f'DiffAugment + temporal-scale augment, R1 steps in the timed region: {r1_steps}', 'global_batch': total_batch,
                   'frames_per_clip': frames_per_clip, 'parallelism': f'dp{world}',
                   'grad_sync': 'FlatGradSync, one rank: no exchange' if world == 1 else
                                ('FlatGradSync, 128 MB buckets, exchange after the replayed phases' if graphs else 'FlatGradSync(overlap=True), 128 MB buckets')},
        'grad_sync': {'exposed_ms_per_step': round(exposed_ms, 3), 'note': 'device time of the all-reduces / waits of FlatGradSync.finish() per iteration (update_G + update_D [+ R1]); 0 at one rank'}}


def _train_sres_leg(dev, steps=16, warmup=1, total_batch=16):
    """BASELINE.json configs[4] at N = 1: the step body of train_sres.py:241-264 (SuperResTrainer.train_step: update_G, update_D, R1 on every
    16th step, ADA probability update on every 4th, generator EMA) on synthetic (low-resolution clip with context, high-resolution clip)
    pairs, total batch 16, ADA pipeline and conditioning augmentation on. Graph replay per phase (LVG_TRAIN_GRAPHS=0: eager).
    Micro-batches of at most 8 segments (round 6; 2 before = the per-GPU share of the 8-GPU recipe run eight times in a row): like the lres leg's
    rule, accumulation exists in the recipe to fit small memories, the gradient mean is the same, and a rank whose share is 16 segments pays the
    per-launch cost of ~20 000 launches per iteration once per 8 segments instead of once per 2 (452 / 397 / 374 / 366 ms per iteration at 2 / 4 / 8 / 16
    segments per micro-batch, profiles/r06_train_sres_microbatch.log; 32 GiB at 8). The discriminator's minibatch-std groups are
    min(4, micro-batch) segments as in the reference (discriminator_sres.py: group size 4), i.e. its default grouping from 4 segments up.
    LVG_BENCH_SRES_MICRO overrides."""
    from lvg.train_sres import SuperResTrainer
    torch.manual_seed(0)
    micro = max(1, min(total_batch, int(os.environ.get('LVG_BENCH_SRES_MICRO', '8'))))
    accum = max(1, total_batch // micro)
    graphs = os.environ.get('LVG_TRAIN_GRAPHS', '1') != '0'      # one rank: the compute of both updates replayed from hipGraphs (SuperResTrainer(use_graphs=True))
    tr = SuperResTrainer(device=dev, compute_dtype=torch.float16, G_grad_accum=accum, D_grad_accum=accum, augment_p_init=0.2,
                         overlap_grad_sync=True, with_ema=True, use_graphs=graphs)
    lr = torch.rand(total_batch, 3, tr.context_seq_length, 36, 64, device=dev) * 2 - 1
    hr = torch.rand(total_batch, 3, tr.seq_length, 144, 256, device=dev) * 2 - 1
    step_no = 1
    for _ in range(warmup):
        tr.train_step(step_no, lr, hr); step_no += 1
    if any(k % 16 == 0 for k in range(warmup + 1, warmup + steps + 1)):
        tr.update_r1(tr.crop_to_seq_length(lr), hr, gain=16)               # (first R1 pass: lazy initialisation outside the timed region)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(step_no, lr, hr); step_no += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    r1_steps = sum(1 for k in range(warmup + 1, warmup + steps + 1) if k % 16 == 0)
    ada_steps = sum(1 for k in range(warmup + 1, warmup + steps + 1) if k % 4 == 0)
    lr_c = tr.crop_to_seq_length(lr)
    tr.update_r1(lr_c, hr, gain=16)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tr.update_r1(lr_c, hr, gain=16)
    torch.cuda.synchronize()
    r1_ms = (time.perf_counter() - t1) * 1e3
    extra = {'r1_update_ms': round(r1_ms, 2)}
    if r1_steps == 0:
        extra['value_with_r1_every_16'] = round(total_batch * 8 / (dt + r1_ms * 1e-3 / 16), 2)
    del tr
    return {**extra, 'metric': 'frames/sec train_sres iteration (update_G + update_D + R1/16 + ADA/4 + EMA), 8-frame 144x256 segments', 'value': round(total_batch * 8 / dt, 2),
            'unit': 'frames/s', 'ms_per_step': round(dt * 1e3, 2), 'steps': steps, 'warmup': warmup, 'dtype': 'f16',
            'launch_mode': 'hipgraph per phase (update_G micro-batch, fake generation, update_D micro-batch, R1 micro-batch); optimizer, ADA update eager' if graphs else 'eager', 'n_gpus': 1,
            'config': {'workload': f'train_sres.py step body, total batch {total_batch} ({accum} micro-batches of {total_batch // accum} segments), ADA p = 0.2 + conditioning augmentation, '
                                   f'R1 steps in the timed region: {r1_steps}, ADA updates: {ada_steps}', 'global_batch': total_batch}}


def _batch_sweep_leg(G, D, dtype, T, batches=(1, 2, 4, 16), steps=6):
    """The main step (G forward, D forward, backward; hipGraph) at the per-GPU batch sizes of SURVEY.md 8(d) config 2 (b in {1, 2, 4}; the
    reference's 8-GPU recipe runs micro-batches of 2) and at 16 (the train_lres leg's micro-batch since round 6): frames/s and ms per step, without the optimizer
    (parameters stay as they are)."""
    from lvg.models import lres
    out = {}
    for b in batches:
        def compute():
            for p in G.parameters():
                p.grad = None
            video = G(b, T, magnitude_ema_beta=0.999, dtype=dtype)
            F.softplus(-D(video, dtype=dtype)).mean().backward()
        try:
            for _ in range(2):
                compute()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                compute()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                compute()
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            del g
            out[str(b)] = {'frames_per_s': round(b * T / dt, 1), 'ms_per_step': round(dt * 1e3, 3)}
        except Exception as err:  # pylint: disable=broad-except
            out[str(b)] = {'error': f'{type(err).__name__}: {err}'[:200]}
    return out




def _forward_only_leg(G, B, T, dtype, steps=6, timer=None):
    """BASELINE.json's headline metric: frames/sec of the low-resolution generator FORWARD at 128 x 36 x 64
    (inference: no gradient, no mask), replayed from a hipGraph like the main step. With `timer` (the main line's OpTimer) the
    leg carries its own `roofline`: the launches of one more forward pass, re-timed per kernel family like the main step's."""
    with torch.no_grad():
        for _ in range(2):
            G(B, T, dtype=dtype)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            G(B, T, dtype=dtype)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    del g
    out = {'metric': 'frames/sec/GPU lres-G forward 128x36x64', 'value': round(B * T / dt, 1), 'unit': 'frames/s',
           'ms_per_step': round(dt * 1e3, 3), 'steps': steps, 'batch': B, 'launch_mode': 'hipgraph'}
    if timer is not None and not os.environ.get('LVG_BENCH_NO_ROOFLINE'):
        timer.calls.clear()
        timer.enabled = True
        with torch.no_grad():
            G(B, T, dtype=dtype)
        timer.enabled = False
        ops = timer.measure()
        flop_ops = getattr(timer, 'flop_ops', set())
        if ops:
            name = max(ops, key=lambda k: ops[k]['total_ms'])
            d = ops[name]
            common = dict(kernel=name, launches=d['launches'], avg_launch_us=round(d['total_ms'] * 1e3 / d['launches'], 2), traffic=None,
                          measured_on='all launches of this kernel from one forward pass, captured once each (pass order) into a hipGraph replayed 3x between HIP events on the launch stream')
            if name in flop_ops:
                tf = d['bytes'] / (d['total_ms'] * 1e-3) / 1e12
                out['roofline'] = dict(bound='mfma', achieved=round(tf, 1), peak=MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=round(tf / MFMA_PEAK_TFLOPS, 4),
                                       algorithmic_flops_per_launch=int(d['bytes'] / d['launches']), **common)
            else:
                out['roofline'] = dict(bound='hbm', achieved=round(d['gbps'], 1), peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(d['gbps'] / HBM_PEAK_GBPS, 4),
                                       algorithmic_bytes_per_launch=int(d['bytes'] / d['launches']), **common)
            out['ms_in_custom_ops'] = round(sum(v['total_ms'] for v in ops.values()), 3)
            out['ops'] = {k: (dict(launches=v['launches'], total_ms=round(v['total_ms'], 3), tflops=round(v['gbps'] / 1e3, 1)) if k in flop_ops else
                              dict(launches=v['launches'], total_ms=round(v['total_ms'], 3), gbps=round(v['gbps'], 1))) for k, v in ops.items()}
    return out


def _fp32_leg(G, D, T, B=8, steps=3):
    """The same generator update in FLOAT32 (the reference trains the low-resolution networks in float32 with TF32 off,
    train_lres.py:268-269), at the main line's batch and launch mode: forward + backward through the discriminator on the
    float32 route of the hand-written kernels (16-bit operand splits on the matrix cores, float32 accumulation: DESIGN 4.13),
    replayed from a hipGraph, gradients discarded. Reported beside the bf16 main line, never instead of it.
    Its roofline: the step's dense-contraction FLOPs (the same count as the bf16 step) over the step time, against the float32
    MFMA peak (157 TFLOP/s: what exact-float32 matrix instructions would allow) and against a sixth of the 16-bit peak (the split
    route spends up to six 16-bit products per float32 product)."""
    from torch.utils.flop_counter import FlopCounterMode
    from lvg.models import lres
    from torch_utils.ops import conv3d_frames
    saved = [p.grad for p in G.parameters()]
    def one():
        for p in G.parameters():
            p.grad = None
        video = G(B, T, dtype=torch.float32)
        F.softplus(-D(video, dtype=torch.float32)).mean().backward()
    mode = 'eager'
    try:
        one()
        conv3d_frames.stats['flops'] = 0
        with FlopCounterMode(display=False) as fc:
            one()
        flops = float(fc.get_total_flops()) + float(conv3d_frames.stats['flops'])
        torch.cuda.synchronize()
        run = one
        if os.environ.get('LVG_FP32_GRAPH', '1') != '0':
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    one()
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    one()
                g.replay()
                run, mode = g.replay, 'hipgraph'
            except Exception:  # pylint: disable=broad-except
                run, mode = one, 'eager'
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        for p, g_ in zip(G.parameters(), saved):
            p.grad = g_
    tf = flops / dt / 1e12
    return {'metric': 'frames/sec lres-G 128x36x64 forward+backward (generator update)', 'dtype': 'fp32', 'value': round(B * T / dt, 1), 'unit': 'frames/s',
            'ms_per_step': round(dt * 1e3, 3), 'steps': steps, 'batch': B, 'launch_mode': mode,
            'route': 'split-operand float32 on the hand-written kernels' if lres.SPLIT_F32 else 'library float32',
            'roofline': {'bound': 'mfma', 'achieved': round(tf, 1), 'unit': 'TFLOP/s (float32-equivalent: the FLOPs of the float32 contraction / step time)',
                         'peak_f32_mfma': 157.3, 'frac_of_f32_mfma': round(tf / 157.3, 4),
                         'peak_split_route': round(MFMA_PEAK_TFLOPS / 6, 1), 'frac_of_split_route': round(tf / (MFMA_PEAK_TFLOPS / 6), 4),
                         'flops_per_step': int(flops), 'scope': 'all dense contractions of the step / whole step time (end to end)'}}


def _mfma_leg(step, sec_per_step):
    """Dense-contraction FLOPs of one step (convolutions, GEMMs; forward and backward, counted at the dispatcher by
    torch.utils.flop_counter) over the measured step time, against the dense 16-bit MFMA peak. This is the
    END-TO-END figure: the time includes everything that is not a contraction. Per-kernel MFMA time is in profiles/."""
    from torch.utils.flop_counter import FlopCounterMode
    from torch_utils.ops import conv3d_frames
    conv3d_frames.stats['flops'] = 0
    with FlopCounterMode(display=False) as fc:
        step()
    torch.cuda.synchronize()
    flops = float(fc.get_total_flops()) + float(conv3d_frames.stats['flops'])     # dispatcher-level ops + the hand-written convolution
    tf = flops / sec_per_step / 1e12
    return {'flops_per_step': int(flops), 'achieved_tflops': round(tf, 1), 'peak_tflops': MFMA_PEAK_TFLOPS, 'frac': round(tf / MFMA_PEAK_TFLOPS, 4),
            'scope': 'all dense contractions of the timed lres step / whole step time (end to end)'}


def _sres_leg(dev, timer, segments=2, steps=6, warmup=2):
    """BASELINE.json configs[3]: generator_sres + discriminator_sres on 8-frame 144x256 segments (+-4 context frames of
    36x64 input), one generator update = G forward -> D forward -> softplus(-logits).mean().backward() -> Adam, f16
    activations in the high-resolution layers as the reference (num_fp16_res = 4). The compute part is replayed from a hipGraph like
    the main step (falls back to eager launches if the capture is refused). The
    filtered_lrelu launches of one step are then re-timed per kernel family (OpTimer) for the roofline."""
    from torch.utils.flop_counter import FlopCounterMode
    from lvg.train_sres import SuperResTrainer
    torch.manual_seed(0)
    tr = SuperResTrainer(device=dev, compute_dtype=torch.float16, augment_real_sign_target=None, augment_p_init=0.0,
                         in_augment_strength=0.0, lr_cond_prob=1.0, overlap_grad_sync=False, with_ema=False)
    lr = (torch.rand(segments, 3, tr.context_seq_length, 36, 64, device=dev) * 2 - 1)

    def compute():
        # the compute part of SuperResTrainer.update_G (zero the flat gradient, G forward, D forward, backward)
        tr.G.requires_grad_(True)
        tr.G_sync.zero()
        logits = tr.run_D(tr.crop_to_seq_length(lr), tr.G(lr))
        F.softplus(-logits).mean().backward()
        tr.G.requires_grad_(False)

    def update():
        tr.G_sync.finish()          # gradient exchange (no-op collective at N = 1), then the fused Adam launch
        tr.G_opt.step()

    def step():
        compute()
        update()
    for _ in range(warmup):
        tr.update_G(lr)             # the trainer's own entry point: same launches as compute() + update()
    torch.cuda.synchronize()
    graph, mode = None, 'eager'
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            compute()
        graph.replay()
        update()
        torch.cuda.synchronize()
        mode = 'hipgraph'
    except Exception as err:  # pylint: disable=broad-except
        print(f'[bench] sres leg: hipGraph capture refused ({type(err).__name__}: {str(err)[:200]}); timing eager launches', file=sys.stderr)
        graph = None
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if graph is not None:
            graph.replay()
            update()
        else:
            step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del graph
    frames = segments * tr.seq_length
    from torch_utils.ops import conv2d_frames
    conv2d_frames.stats['flops'] = 0
    with FlopCounterMode(display=False) as fc:
        step()
    torch.cuda.synchronize()
    flops = float(fc.get_total_flops()) + float(conv2d_frames.stats['flops'])     # dispatcher-level ops + the hand-written 2-D convolution
    timer.calls.clear()
    timer.enabled = True
    step()
    timer.enabled = False
    measured = timer.measure()
    convs = {k: dict(launches=v['launches'], total_ms=round(v['total_ms'], 3), tflops=round(v['gbps'] / 1e3, 1), frac=round(v['gbps'] / 1e3 / MFMA_PEAK_TFLOPS, 4))
             for k, v in measured.items() if k.startswith('conv2d')}
    ops = {k: v for k, v in measured.items() if k.startswith('filtered_lrelu')}
    # the launches of the MFMA filtered_lrelu kernel (16-bit tensors, real resampling): the family the committed PMC traffic figure belongs to
    mf = [v for k, v in ops.items() if '_f32_' not in k and 'u1d1' not in k]
    mf_launches, mf_bytes = sum(v['launches'] for v in mf), sum(v['bytes'] for v in mf)
    tot_ms = sum(v['total_ms'] for v in ops.values())
    tot_b = sum(v['bytes'] for v in ops.values())
    n = sum(v['launches'] for v in ops.values())
    worst = min(ops.values(), key=lambda v: v['gbps']) if ops else None
    gb = tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
    out = {'metric': 'frames/sec sres G+D 8-frame 144x256 forward+backward (generator update)', 'value': round(frames / dt, 2), 'unit': 'frames/s',
           'ms_per_step': round(dt * 1e3, 3), 'steps': steps, 'warmup': warmup, 'dtype': 'f16', 'launch_mode': mode,
           'config': {'workload': f'generator_sres + discriminator_sres, {segments} segments x 8 frames 144x256 from 36x64 (+-4 context), Adam step', 'global_batch': segments},
           'roofline': {'bound': 'hbm', 'kernel': 'filtered_lrelu', 'achieved': round(gb, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(gb / HBM_PEAK_GBPS, 4),
                        'launches': n, 'avg_launch_us': round(tot_ms * 1e3 / max(n, 1), 2), 'algorithmic_bytes_per_launch': int(tot_b / max(n, 1)),
                        'traffic': _pmc_traffic('filtered_lrelu_fused16', 'sres') or _pmc_traffic('filtered_lrelu_wave', 'sres'), 'traffic_source': _TRAFFIC_SOURCE.format(scope='sres'),
                        'traffic_scope': 'per launch of the 16-bit fused kernels (strip + row-band + wave-per-tile), launch-weighted',
                        'algorithmic_bytes_per_launch_mfma_family': int(mf_bytes / max(mf_launches, 1)), 'launches_mfma_family': mf_launches,
                        'measured_on': 'all fused filtered_lrelu launches of one step (forward with mask write, backward with mask read), captured once each into a hipGraph per family, replayed 3x between HIP events',
                        'families': {k: dict(launches=v['launches'], total_ms=round(v['total_ms'], 3), gbps=round(v['gbps'], 1)) for k, v in ops.items()},
                        'slowest_family_gbps': round(worst['gbps'], 1) if worst else None},
           'step_ms_in_filtered_lrelu': round(tot_ms, 3),
           'conv2d': convs,
           'mfma': {'flops_per_step': int(flops), 'achieved_tflops': round(flops / dt / 1e12, 1), 'peak_tflops': MFMA_PEAK_TFLOPS,
                    'frac': round(flops / dt / 1e12 / MFMA_PEAK_TFLOPS, 4), 'scope': 'all dense contractions / whole step time (end to end)'}}
    return out


_TRAFFIC_SOURCE = ('profiles/traffic_{scope}.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this workload (tools/gpu_traffic.sh), '
                   'committed with the round; NOT collected in this invocation')


def _pmc_traffic(kernel, scope):
    """HBM bytes per launch of `kernel` in the `scope` workload ('lres': the main line's step, 'sres': the super-resolution leg) from the
    committed rocprofv3 --pmc run of that workload (profiles/traffic_<scope>.json), else None. Keyed by workload: the same kernel family
    moves different tensors in the two steps."""
    path = os.path.join(ROOT, 'profiles', f'traffic_{scope}.json')
    try:
        with open(path) as f:
            return json.load(f).get(kernel)
    except (OSError, ValueError):
        return None


def _cpu_baseline(forward_only):
    """CPU leg in a child process with a hard wall-clock limit, so a slow or oversubscribed host can
    never stretch the default run (the sample itself is bounded to ~20 s of CPU work)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'cpu_step.py')] + (['--forward-only'] if forward_only else [])
    env = dict(os.environ, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    # The reference's OWN CPU path cannot run on the driver's box (no reference checkout there): its figures measured in the build
    # container are BASELINE.md's, quoted beside whatever kind of baseline was timed here.
    quoted = {'source': 'BASELINE.md (reference networks on their upfirdn2d ref path, Xeon 2.1 GHz container, torch CPU)',
              'generator_lres_forward_16_frames': {'frames_per_s': 32.7, 'threads': 8, 'seconds': 0.489},
              'generator_lres_forward_16_frames_1_thread': {'frames_per_s': 6.5, 'threads': 1, 'seconds': 2.476},
              'generator_lres_forward_128_frames': {'frames_per_s': 78.8, 'threads': 8, 'seconds': 1.625},
              'generator_sres_forward_8_frames': {'frames_per_s': 0.23, 'threads': 8, 'seconds': 34.4}}
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=150, env=env)
        res = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as err:  # pylint: disable=broad-except
        res = {'value': None, 'unit': 'frames/s', 'cores': None, 'kind': 'port',
               'sample': f'not measured: {type(err).__name__} (limit 150 s)'}
    res['reference_cpu_measured_in_build_container'] = quoted
    return res


if __name__ == '__main__':
    main()
