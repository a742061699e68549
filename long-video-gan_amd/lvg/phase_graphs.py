"""Replay of a trainer phase (the compute of one micro-batch of update_G / update_D) from a hipGraph.

What a captured phase may contain: kernels on torch's streams, device-side random draws (the generator offsets of a captured graph are
advanced per replay), in-place updates of static tensors. What it may not: an RCCL collective (aborts inside a capture on this stack), host
reads, host-side random draws -- those stay outside and reach the graph through static tensors the caller fills before each replay.

More than one rank: the collectives a pass contains in the reference are taken OUT of it. The running statistics that are averaged over
ranks inside a generator pass (lvg.ddp.ema_of_rank_mean) are recorded per phase in a `deferred_stat_sync()` scope -- the recorded tensors
are static outputs of the graph -- and exchanged in one all-reduce after every replay; the gradient exchange runs after the phase's
replays (lvg.ddp.FlatGradSync.finish). Host-side effects of a phase happen only while it is captured, never on replay: the one the
trainers rely on -- FlatGradSync's note of which parameters received a gradient -- is recorded at capture and restored after each replay."""

from typing import Callable, Dict, Hashable, Iterable, Sequence

import torch

from . import ddp


class PhaseGraphs:
    def __init__(self, rollback: Callable[[], Iterable[torch.Tensor]], graphs: Dict[Hashable, object] = None,
                 syncs: Sequence['ddp.FlatGradSync'] = (), capture: bool = True):
        """`rollback()`: the tensors a phase updates in place (gradient buffers, running statistics): the eager warm-up run in front of a
        capture is undone on them, so the first call of a phase has the effect of exactly one execution. `syncs`: the gradient exchanges
        whose post-accumulate hooks a phase triggers. `capture=False`: every call runs the phase eagerly through the same protocol
        (deferred statistics, exchange afterwards) -- the segmentation without the graphs, on any device (tests, debugging)."""
        self.rollback = rollback
        self.graphs = {} if graphs is None else graphs
        self.syncs = list(syncs)
        self.capture = capture
        self.eager_keys = set()

    def _run(self, fn: Callable[[], None]):
        """The phase with its cross-rank statistics deferred -> (records, their stacked tensors or None)."""
        with ddp.deferred_stat_sync() as pending:
            fn()
            stacked = ddp.stack_pending(pending) if pending else None
        return pending, stacked

    def replay(self, key: Hashable, fn: Callable[[], None], optional: bool = False) -> None:
        """Run `fn` from its graph; first call for `key`: one eager run on a side stream (lazy initialisation, library plans), rolled back,
        then the capture. `fn` must read its inputs from static tensors and leave its outputs in static tensors. `optional`: a refused
        capture makes THIS phase eager from then on and leaves the other phases' graphs alone (the R1 phase: a double backward pass)."""
        if not self.capture or key in self.eager_keys:
            ddp.finish_stat_sync(*self._run(fn))
            return
        entry = self.graphs.get(key)
        if entry is None:
            keep = list(self.rollback())
            saved = [t.clone() for t in keep]
            fired_before = [list(s._fired) for s in self.syncs]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._run(fn)
            torch.cuda.current_stream().wait_stream(side)
            for t, v in zip(keep, saved):
                t.copy_(v)
            for s, f in zip(self.syncs, fired_before):
                s._fired = list(f)
            # More than one rank: the process group's watchdog thread issues event queries of its own; 'thread_local' keeps them from
            # invalidating this thread's capture. A capture that is refused anyway (never observed at one rank; an RCCL multi-GPU run has
            # not been available to this repository) turns the phase graphs off for good: the same protocol, launched eagerly.
            mode = 'thread_local' if ddp._world() > 1 else 'global'
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=mode):
                    pending, stacked = self._run(fn)
            except Exception as err:  # pylint: disable=broad-except
                import sys
                print(f'[phase_graphs] hipGraph capture of phase {key!r} refused ({type(err).__name__}: {err}); eager launches from here on', file=sys.stderr)
                torch.cuda.synchronize()
                for s, f in zip(self.syncs, fired_before):
                    s._fired = list(f)
                if optional:
                    self.eager_keys.add(key)
                else:
                    self.capture = False
                    self.graphs.clear()
                ddp.finish_stat_sync(*self._run(fn))
                return
            # which gradients this phase produces (their hooks ran during the capture and will not run again)
            fired = [[i for i, (now, was) in enumerate(zip(s._fired, f)) if now and not was] for s, f in zip(self.syncs, fired_before)]
            plan = ddp.stat_sync_plan(pending) if pending and ddp._world() > 1 else None
            entry = self.graphs[key] = (g, pending, stacked, fired, plan)
        g, pending, stacked, fired, plan = entry
        g.replay()
        for s, idx in zip(self.syncs, fired):
            for i in idx:
                s._fired[i] = True
        if pending:
            ddp.finish_stat_sync(pending, stacked, plan)
