"""Data-parallel gradient exchange: the reference's `utils.sync_grads` (utils.py:104-124) for one
process per GPU over RCCL/xGMI.

Reference behaviour (kept bit-for-bit in `sync_grads`): flatten all existing grads into one
float vector, all-reduce(SUM) it in equal shards of <= 2**23 elements, divide by world size,
multiply by `gain`, nan_to_num(nan=0, +-inf=+-1e5), scatter back.

MI355X version (`FlatGradSync`): the flat vector is allocated ONCE and every `param.grad` is a
view into it, so there is no pack (torch.cat) and no unpack copy per step; buckets are larger
(xGMI is per-link bound, few big ring collectives beat many small ones) and, with
`overlap=True`, each bucket's all-reduce is launched from an autograd hook as soon as its last
gradient of the final micro-batch has been accumulated, overlapping with the rest of backward.
Same mean / gain / nan_to_num semantics."""

import contextlib
import math
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# ---------------------------------------------------------------------------------------------
# Running statistics that are averaged over ranks inside a forward pass (the generators' input-magnitude EMAs, the mapping
# network's w_avg: reference model/generator_lres.py:298-312, model/generator_sres.py:116-123, 278-286). Each is a collective in
# the middle of the pass in the reference. Inside a `deferred_stat_sync()` scope a layer instead folds in its LOCAL statistic and
# records (buffer, local statistic, beta, previous buffer value); ONE all-reduce afterwards (`finish_stat_sync`) redoes every
# buffer with the mean over ranks -- the reference's arithmetic, bit-identical on every rank. The only difference is that the gain
# used in THIS pass saw the local statistic: a relative change of (1 - beta) * (local / global - 1) per layer -- measured on the full lres
# generator at two ranks of ONE clip each: 2.3e-4 of the emitted video's range (tests/test_ddp_gloo.py, gate 5e-4) -- nothing at world size 1.
# It is what lets a pass be captured into a hipGraph (an RCCL collective inside a capture aborts on this stack): the recorded
# tensors are static outputs of the graph, the all-reduce runs after the replay (lvg.phase_graphs).

_pending_stats = None       # list of (buffer, local, beta, previous, form) while a deferred scope is open

LERP_TOWARDS = 0            # buffer.lerp_(stat, 1 - beta)        (lres MagnitudeEMA)
LERP_FROM = 1               # buffer.copy_(stat.lerp(buffer, beta))  (sres w_avg / magnitude_ema)


@contextlib.contextmanager
def deferred_stat_sync():
    """Scope in which `ema_of_rank_mean` records local statistics instead of all-reducing them. Yields the list of records."""
    global _pending_stats
    assert _pending_stats is None, 'deferred_stat_sync scopes do not nest'
    _pending_stats = []
    try:
        yield _pending_stats
    finally:
        _pending_stats = None


def ema_of_rank_mean(buffer: torch.Tensor, local: torch.Tensor, beta: float, form: int = LERP_TOWARDS) -> None:
    """buffer <- exponential moving average step towards the mean over ranks of `local` (a detached float32 tensor of buffer's
    shape), written in the arithmetic form of the reference call site."""
    stat = local
    if _world() > 1:
        if _pending_stats is not None:
            _pending_stats.append((buffer, local, beta, buffer.clone(), form))
        else:
            stat = local.clone()
            dist.all_reduce(stat)
            stat = stat / _world()
    if form == LERP_TOWARDS:
        buffer.lerp_(stat.to(buffer.dtype), 1.0 - beta)
    else:
        buffer.copy_(stat.lerp(buffer, beta))


def stack_pending(pending):
    """(local statistics, previous buffer values) of a deferred scope as two flat vectors (cheap to keep in a captured graph)."""
    return (torch.cat([l.reshape(-1).float() for _, l, _, _, _ in pending]), torch.cat([p.reshape(-1).float() for _, _, _, p, _ in pending]))


def stat_sync_plan(pending, device=None):
    """What `finish_stat_sync` needs besides the statistics themselves, built ONCE per recorded scope (a captured phase keeps it next
    to its static tensors: rebuilding the weight vectors from Python lists after every replay cost two host-to-device copies per phase):
    (element counts, weights 1 - beta, weights beta). Every buffer may be recorded at most once per scope -- a second record would be
    redone from a 'previous' value that already holds the first local update."""
    seen = set()
    for buf, _, _, _, _ in pending:
        assert id(buf) not in seen, 'a running statistic was recorded twice in one deferred_stat_sync scope'
        seen.add(id(buf))
    sizes = [b.numel() for b, _, _, _, _ in pending]
    device = pending[0][1].device if device is None else device
    # (weights formed in Python floats and rounded once, like the scalar arguments of the eager calls)
    w_to = torch.tensor([1.0 - bt for (_, _, bt, _, _), n in zip(pending, sizes) for _ in range(n)], dtype=torch.float32, device=device)
    w_from = torch.tensor([bt for (_, _, bt, _, _), n in zip(pending, sizes) for _ in range(n)], dtype=torch.float32, device=device)
    return sizes, w_to, w_from


def finish_stat_sync(pending, stacked=None, plan=None) -> None:
    """One all-reduce for every statistic recorded in a deferred scope, then each buffer is REDONE from its previous value and the
    mean over ranks, in the form of its call site. `plan`: `stat_sync_plan(pending)` kept by the caller (replayed phases)."""
    world = _world()
    if not pending or world <= 1:
        return
    local, prev = stack_pending(pending) if stacked is None else stacked
    glob = local.clone()
    dist.all_reduce(glob)
    glob = glob / world
    sizes, w_to, w_from = stat_sync_plan(pending, glob.device) if plan is None else plan
    towards = torch.lerp(prev, glob, w_to)                # prev.lerp(stat, 1 - beta)
    frm = torch.lerp(glob, prev, w_from)                  # stat.lerp(prev, beta)
    o = 0
    for (buf, _, _, _, form), n in zip(pending, sizes):
        src = towards if form == LERP_TOWARDS else frm
        buf.copy_(src[o:o + n].view_as(buf))
        o += n


def sharded_all_mean(tensor: torch.Tensor, shard_size: int = 2 ** 23) -> torch.Tensor:
    assert tensor.dim() == 1
    if _world() > 1:
        for shard in tensor.tensor_split(max(1, math.ceil(tensor.numel() / shard_size))):
            dist.all_reduce(shard)
    return tensor / _world()


def sync_grads(network: nn.Module, gain: Optional[float] = None) -> None:
    """Drop-in for the reference's utils.sync_grads (same arithmetic, same result)."""
    params = [p for p in network.parameters() if p.grad is not None]
    if not params:
        return
    flat = torch.cat([p.grad.flatten() for p in params])
    flat = sharded_all_mean(flat)
    if gain is not None:
        flat = flat * gain
    torch.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)
    for p, g in zip(params, flat.split([p.numel() for p in params])):
        p.grad = g.reshape(p.size())


class FlatGradSync:
    """Persistent flat gradient buffer with bucketed (optionally backward-overlapped) all-reduce.

    Usage per optimizer step:
        sync.zero()                      # instead of opt.zero_grad(set_to_none=True)
        for micro-batch k: ... loss.backward()   (call sync.arm() before the LAST backward to overlap)
        sync.finish(gain)                # wait, mean, gain, nan_to_num (in place)
        opt.step()
    """

    def __init__(self, params: Iterable[nn.Parameter], bucket_numel: int = 1 << 25, overlap: bool = False):
        self.params: List[nn.Parameter] = [p for p in params]
        assert self.params, 'no parameters'
        dev, dtype = self.params[0].device, self.params[0].dtype
        sizes = [p.numel() for p in self.params]
        # every slice starts on a 16-byte boundary (4 floats): the same layout as lvg.optim.flatten_parameters, so the fused
        # optimizer kernel can treat a run of parameters and the matching run of gradients as two flat ranges
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += (n + 3) // 4 * 4
        self.flat = torch.zeros(o, device=dev, dtype=dtype)
        # Buckets follow REVERSE parameter order (roughly the order gradients become ready).
        self.views = [self.flat[o:o + n].view_as(p) for p, o, n in zip(self.params, offs, sizes)]
        self.buckets = []   # (start, end, [param indices])
        end, members, count = len(self.flat), [], 0
        for i in reversed(range(len(self.params))):
            members.append(i)
            count += sizes[i]
            if count >= bucket_numel or i == 0:
                self.buckets.append((offs[i], end, list(members)))
                end, members, count = offs[i], [], 0
        self.bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self.bucket_of[i] = b
        self.overlap = overlap
        self._armed = False
        self._pending = [0] * len(self.buckets)
        self._streams = [set() for _ in self.buckets]     # streams that accumulated gradients into each bucket since arm()
        self._launched = set()
        self._work = []
        self._hooks = []
        # Which parameters received a gradient since zero(): the reference starts every update from
        # zero_grad(set_to_none=True) and sync_grads / Adam skip parameters whose grad is None (utils.py:106);
        # finish() restores exactly that for parameters autograd never touched.
        self._fired = [False] * len(self.params)
        # measurement hook (bench.py): a list to which finish() appends (start, end) device events around the part of the exchange
        # that is not hidden behind backward (everything when nothing was armed), or None
        self.exposed_events = None
        for i, p in enumerate(self.params):
            # hooks can only be registered on tensors that require grad; the trainers keep their networks
            # frozen (requires_grad False) outside the update that trains them, so flip it for the call
            frozen = not p.requires_grad
            if frozen:
                p.requires_grad_(True)
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
            if frozen:
                p.requires_grad_(False)
        self.zero()

    def _make_hook(self, index: int):
        def hook(param):
            self._fired[index] = True
            if not self._armed:
                return
            b = self.bucket_of[index]
            if param.is_cuda:
                # Autograd accumulates a leaf's gradient on the stream the leaf was first USED on: parameters consumed on a side
                # stream (lres: the weight / style terms run ahead of the convolutions there) accumulate on that stream, others
                # on the main one. The hook runs right after the accumulation was enqueued, on that same stream: remember it.
                self._streams[b].add(torch.cuda.current_stream(param.device))
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b: int) -> None:
        self._launched.add(b)
        if dist.is_available() and dist.is_initialized():
            s, e, _ = self.buckets[b]
            if self.flat.is_cuda:
                # the collective is ordered after the CURRENT stream only: make it wait for every stream that accumulated into
                # this bucket (round-2 advisor finding: a bucket mixing side-stream and main-stream parameters could be reduced
                # before the other stream's accumulation kernels finished)
                cur = torch.cuda.current_stream(self.flat.device)
                for st in self._streams[b]:
                    if st != cur:
                        cur.wait_stream(st)
            self._work.append(dist.all_reduce(self.flat[s:e], async_op=True))

    def zero(self, assign: bool = False) -> None:
        """Zero the flat buffer and (re)attach the views as .grad of every parameter.

        `assign=True` (steps of ONE backward pass, not armed): the parameters' .grad are set to None instead, so autograd ASSIGNS each
        incoming gradient (no launch) where it would otherwise add it into the zeroed view (one launch per parameter, ~100 per generator
        update); `gather()` after the backward pass copies them into the flat buffer with a few multi-tensor launches and re-attaches the
        views. Inside a captured phase `gather()` belongs to the phase (the replay then refills the flat buffer)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = None if assign else v
        self._fired = [False] * len(self.params)

    def gather(self) -> None:
        """After a backward pass that followed `zero(assign=True)`: gradients -> flat buffer, views re-attached."""
        assert not self._armed, 'FlatGradSync.gather: not with the overlapped exchange (its buckets read the flat buffer from the hooks)'
        self._adopt_replaced_grads()

    def arm(self) -> None:
        """Call before the last backward of the step: buckets reduce as they fill."""
        assert self.overlap
        self._armed = True
        for b, (_, _, mem) in enumerate(self.buckets):
            self._pending[b] = sum(1 for i in mem if self.params[i].requires_grad)
        self._launched = set()
        self._streams = [set() for _ in self.buckets]
        self._work = []

    def _adopt_replaced_grads(self) -> None:
        """Autograd accumulates OUT of place when the backward itself is recorded (create_graph=True) or the
        view was otherwise not usable: `p.grad` is then a new tensor holding the full sum and the flat buffer
        is stale. Copy it in before anything is reduced; if that bucket has already been sent (overlap), the
        exchanged values are wrong and there is no way to repair them, so fail loudly."""
        dst, src = [], []
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            g = p.grad
            if g is None or g is v or (g.data_ptr() == v.data_ptr() and g.shape == v.shape):
                continue
            if self._armed and self.bucket_of[i] in self._launched:
                raise RuntimeError('FlatGradSync: autograd replaced the .grad view of a parameter whose bucket was already '
                                   'all-reduced (backward with create_graph=True under overlap=True); arm() only before a plain backward')
            dst.append(v)
            src.append(g.detach() if g.dtype == v.dtype else g.detach().to(v.dtype))
            self._fired[i] = True
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)                  # a few multi-tensor launches instead of one copy per parameter

    def finish(self, gain: Optional[float] = None, drop_unused: bool = True) -> None:
        """Complete the exchange: mean over ranks, * gain, nan_to_num. In place on the flat buffer.
        With `drop_unused`, parameters that received no gradient since zero() end with grad None (so the
        optimizer leaves their state alone, like the reference's zero_grad(set_to_none=True))."""
        self._adopt_replaced_grads()
        timed = self.exposed_events is not None and self.flat.is_cuda and _world() > 1
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._armed:
            # buckets whose hooks never fired (unused parameters) still have to be reduced
            for b in range(len(self.buckets)):
                if self._pending[b] > 0:
                    self._launch(b)
            for w in self._work:
                w.wait()
            self._work.clear()
            self._armed = False
        elif _world() > 1:
            for b in range(len(self.buckets)):
                s, e, _ = self.buckets[b]
                dist.all_reduce(self.flat[s:e])
        if timed:
            e1.record()
            self.exposed_events.append((e0, e1))
        scale = (1.0 / _world()) * (1.0 if gain is None else float(gain))
        if scale != 1.0:
            self.flat.mul_(scale)
        torch.nan_to_num(self.flat, nan=0, posinf=1e5, neginf=-1e5, out=self.flat)
        if drop_unused and any(self._fired):
            for p, fired in zip(self.params, self._fired):
                if not fired:
                    p.grad = None

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks.clear()


def broadcast_module(module: nn.Module, src: int = 0) -> None:
    """Initial weight sync (reference video_gan_lres.py:77-79) as ONE flat broadcast per dtype
    instead of one per tensor."""
    if _world() == 1:
        return
    tensors = [t for t in list(module.parameters()) + list(module.buffers())]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for group in by_dtype.values():
        flat = torch.cat([t.detach().reshape(-1) for t in group])
        dist.broadcast(flat, src=src)
        for t, piece in zip(group, flat.split([t.numel() for t in group])):
            t.detach().copy_(piece.view_as(t))
