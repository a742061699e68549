"""Minimal running-moment statistics (reference torch_utils/training_stats.py) -- logging is
outside the hot path; this shim keeps `report` / `report0` /
`Collector` working for GAN step code. One all_reduce per `Collector.update()` (:256-257)."""

import re

import numpy as np
import torch

import dnnlib

_MOMENTS = 3                      # count, sum, sum of squares
_ACC_DTYPE = torch.float64
_rank = 0
_sync_device = None
_sync_called = False
_pending = {}                     # name -> device -> running [count, sum, sumsq]
_totals = {}                      # name -> cumulative moments on CPU


def init_multiprocessing(rank, sync_device):
    global _rank, _sync_device
    assert not _sync_called
    _rank, _sync_device = rank, sync_device


def report(name, value):
    slot = _pending.setdefault(name, {})
    vals = torch.as_tensor(value)
    if vals.numel() == 0:
        return value
    vals = vals.detach().flatten().to(torch.float32)
    m = torch.stack([torch.ones_like(vals).sum(), vals.sum(), vals.square().sum()]).to(_ACC_DTYPE)
    if m.device not in slot:
        slot[m.device] = torch.zeros_like(m)
    slot[m.device].add_(m)
    return value


def report0(name, value):
    report(name, value if _rank == 0 else [])
    return value


def _sync(names):
    global _sync_called
    if not names:
        return []
    _sync_called = True
    device = _sync_device if _sync_device is not None else torch.device('cpu')
    rows = []
    for name in names:
        row = torch.zeros([_MOMENTS], dtype=_ACC_DTYPE, device=device)
        for counter in _pending[name].values():
            row.add_(counter.to(device))
            counter.zero_()
        rows.append(row)
    table = torch.stack(rows)
    if _sync_device is not None:
        torch.distributed.all_reduce(table)
    table = table.cpu()
    for row, name in zip(table, names):
        _totals.setdefault(name, torch.zeros([_MOMENTS], dtype=_ACC_DTYPE)).add_(row)
    return [(name, _totals[name]) for name in names]


class Collector:
    """Averages of reported scalars between consecutive `update()` calls."""

    def __init__(self, regex='.*', keep_previous=True):
        self._regex = re.compile(regex)
        self._keep_previous = keep_previous
        self._seen = {}
        self._window = {}
        self.update()
        self._window.clear()

    def names(self):
        return [n for n in _pending if self._regex.fullmatch(n)]

    def update(self):
        if not self._keep_previous:
            self._window.clear()
        for name, total in _sync(self.names()):
            prev = self._seen.setdefault(name, torch.zeros([_MOMENTS], dtype=_ACC_DTYPE))
            delta = total - prev
            prev.copy_(total)
            if float(delta[0]) != 0:
                self._window[name] = delta

    def _delta(self, name):
        assert self._regex.fullmatch(name)
        return self._window.setdefault(name, torch.zeros([_MOMENTS], dtype=_ACC_DTYPE))

    def num(self, name):
        return int(self._delta(name)[0])

    def mean(self, name):
        d = self._delta(name)
        return float(d[1] / d[0]) if int(d[0]) else float('nan')

    def std(self, name):
        d = self._delta(name)
        if int(d[0]) == 0 or not np.isfinite(float(d[1])):
            return float('nan')
        if int(d[0]) == 1:
            return 0.0
        mean = float(d[1] / d[0])
        return float(np.sqrt(max(float(d[2] / d[0]) - mean * mean, 0.0)))

    def as_dict(self):
        return dnnlib.EasyDict({n: dnnlib.EasyDict(num=self.num(n), mean=self.mean(n), std=self.std(n)) for n in self.names()})

    def __getitem__(self, name):
        return self.mean(name)


default_collector = Collector()
