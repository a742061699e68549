"""Replay of a trainer phase (the compute of one micro-batch of update_G / update_D) from a hipGraph.

What a captured phase may contain: kernels on torch's streams, device-side random draws (the generator offsets of a captured graph are
advanced per replay), in-place updates of static tensors. What it may not: an RCCL collective (aborts inside a capture on this stack), host
reads, host-side random draws -- those stay outside and reach the graph through static tensors the caller fills before each replay."""

from typing import Callable, Dict, Hashable, Iterable

import torch


class PhaseGraphs:
    def __init__(self, rollback: Callable[[], Iterable[torch.Tensor]], graphs: Dict[Hashable, object] = None):
        """`rollback()`: the tensors a phase updates in place (gradient buffers, running statistics): the eager warm-up run in front of a
        capture is undone on them, so the first call of a phase has the effect of exactly one execution."""
        self.rollback = rollback
        self.graphs = {} if graphs is None else graphs

    def replay(self, key: Hashable, fn: Callable[[], None]) -> None:
        """Run `fn` from its graph; first call for `key`: one eager run on a side stream (lazy initialisation, library plans), rolled back,
        then the capture. `fn` must read its inputs from static tensors and leave its outputs in static tensors."""
        g = self.graphs.get(key)
        if g is None:
            keep = list(self.rollback())
            saved = [t.clone() for t in keep]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            for t, v in zip(keep, saved):
                t.copy_(v)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            self.graphs[key] = g
        g.replay()
