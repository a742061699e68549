"""Small tensor/module helpers that reference model source and pickles import from
`torch_utils.misc` (reference torch_utils/misc.py). Re-written; same names and behaviour for:
constant (:26), nan_to_num (:46), suppress_tracer_warnings (:62), assert_shape (:74),
profiled_function (:92), InfiniteSampler (:103), params_and_buffers (:139),
named_params_and_buffers (:143), copy_params_and_buffers (:147), ddp_sync (:160),
check_ddp_consistency (:172)."""

import contextlib
import re
import warnings

import numpy as np
import torch

nan_to_num = torch.nan_to_num
symbolic_assert = torch._assert  # pylint: disable=protected-access

_constants = {}


def constant(value, shape=None, dtype=None, device=None, memory_format=None):
    """Cached constant tensor (saves a host->device copy per use)."""
    arr = np.asarray(value)
    shape = None if shape is None else tuple(shape)
    dtype = dtype or torch.get_default_dtype()
    device = torch.device('cpu') if device is None else device
    memory_format = memory_format or torch.contiguous_format
    key = (arr.shape, arr.dtype, arr.tobytes(), shape, dtype, device, memory_format)
    hit = _constants.get(key)
    if hit is None:
        hit = torch.as_tensor(arr.copy(), dtype=dtype, device=device)
        if shape is not None:
            hit = hit.expand(shape) if hit.ndim <= len(shape) else torch.broadcast_to(hit, shape)
        hit = hit.contiguous(memory_format=memory_format)
        _constants[key] = hit
    return hit


@contextlib.contextmanager
def suppress_tracer_warnings():
    """Mute torch.jit.TracerWarning inside the block (filter-list edit, as catch_warnings is
    not re-entrant across threads)."""
    entry = ('ignore', None, torch.jit.TracerWarning, None, 0)
    warnings.filters.insert(0, entry)
    try:
        yield
    finally:
        if entry in warnings.filters:
            warnings.filters.remove(entry)


def assert_shape(tensor, ref_shape):
    """Raise AssertionError unless tensor.shape matches ref_shape (None = any size)."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for dim, (got, want) in enumerate(zip(tensor.shape, ref_shape)):
        if want is None:
            continue
        if isinstance(want, torch.Tensor) or isinstance(got, torch.Tensor):
            with suppress_tracer_warnings():
                symbolic_assert(torch.equal(torch.as_tensor(got), torch.as_tensor(want)), f'Wrong size for dimension {dim}')
        elif got != want:
            raise AssertionError(f'Wrong size for dimension {dim}: got {got}, expected {want}')


def profiled_function(fn):
    """Wrap fn in an autograd-profiler range named after it (shows up in rocprofv3 marker
    traces via roctx when the profiler is active)."""
    def wrapper(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    wrapper.__name__ = fn.__name__
    wrapper.__doc__ = fn.__doc__
    return wrapper


class InfiniteSampler(torch.utils.data.Sampler):
    """Endless, optionally windowed-shuffling index stream, strided across replicas."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        super().__init__()
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        step = 0
        while True:
            pos = step % order.size
            if step % self.num_replicas == self.rank:
                yield order[pos]
            if window >= 2:
                other = (pos - rnd.randint(window)) % order.size
                order[pos], order[other] = order[other], order[pos]
            step += 1


def params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.parameters()) + list(module.buffers())


def named_params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.named_parameters()) + list(module.named_buffers())


def copy_params_and_buffers(src_module, dst_module, require_all=False):
    assert isinstance(src_module, torch.nn.Module) and isinstance(dst_module, torch.nn.Module)
    source = dict(named_params_and_buffers(src_module))
    with torch.no_grad():
        for name, dst in named_params_and_buffers(dst_module):
            if name not in source:
                assert not require_all, f'{name} missing from source module'
                continue
            dst.copy_(source[name].detach()).requires_grad_(dst.requires_grad)


@contextlib.contextmanager
def ddp_sync(module, sync):
    assert isinstance(module, torch.nn.Module)
    if sync or not isinstance(module, torch.nn.parallel.DistributedDataParallel):
        yield
    else:
        with module.no_sync():
            yield


def check_ddp_consistency(module, ignore_regex=None):
    """Assert that every param/buffer is bit-identical to rank 0's copy."""
    assert isinstance(module, torch.nn.Module)
    for name, tensor in named_params_and_buffers(module):
        fullname = f'{type(module).__name__}.{name}'
        if ignore_regex is not None and re.fullmatch(ignore_regex, fullname):
            continue
        mine = tensor.detach()
        if mine.is_floating_point():
            mine = nan_to_num(mine)
        theirs = mine.clone()
        torch.distributed.broadcast(tensor=theirs, src=0)
        assert (mine == theirs).all(), fullname
