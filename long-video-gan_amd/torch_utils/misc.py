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


def print_module_summary(module, inputs, max_nesting=3, skip_redundant=True, **input_kwargs):
    """Run `module(*inputs, **input_kwargs)` once and print one table row per sub-module (down to
    `max_nesting` levels): parameters and buffers it owns that no earlier row owned, output shape /
    dtype and output statistics. Returns the module's outputs (reference misc.py:196)."""
    assert isinstance(module, torch.nn.Module) and not isinstance(module, torch.jit.ScriptModule)
    assert isinstance(inputs, (tuple, list))

    depth = 0
    visited = []                                  # (module, [output tensors]) in completion order

    def enter(_mod, _args):
        nonlocal depth
        depth += 1

    def leave(mod, _args, result):
        nonlocal depth
        depth -= 1
        if depth <= max_nesting:
            seq = result if isinstance(result, (tuple, list)) else [result]
            visited.append((mod, [t for t in seq if isinstance(t, torch.Tensor)]))

    handles = []
    for sub in module.modules():
        handles.append(sub.register_forward_pre_hook(enter))
        handles.append(sub.register_forward_hook(leave))
    try:
        outputs = module(*inputs, **input_kwargs)
    finally:
        for h in handles:
            h.remove()

    names = {sub: name for name, sub in module.named_modules()}
    columns = ['Mean', 'Std', 'Min (abs)', 'Max (abs)']
    measure = [torch.mean, torch.std, lambda t: t.abs().min(), lambda t: t.abs().max()]
    table = [[type(module).__name__, 'Parameters', 'Buffers', 'Output shape', 'Datatype'] + [f'{c:<10}' for c in columns]]
    table.append(['---'] * len(table[0]))
    owned = set()
    totals = [0, 0]
    for sub, outs in visited:
        params = [t for t in sub.parameters() if id(t) not in owned]
        buffers = [t for t in sub.buffers() if id(t) not in owned]
        fresh_outs = [t for t in outs if id(t) not in owned]
        owned.update(id(t) for t in params + buffers + fresh_outs)
        if skip_redundant and not (params or buffers or fresh_outs):
            continue
        n_param, n_buf = sum(t.numel() for t in params), sum(t.numel() for t in buffers)
        totals[0] += n_param
        totals[1] += n_buf
        label = '<top-level>' if sub is module else names[sub]
        for idx in range(max(len(outs), 1)):
            if idx < len(outs):
                t = outs[idx]
                cells = [str(list(t.shape)), str(t.dtype).split('.')[-1]] + [f'{float(f(t.detach().float())):>10.3e}' for f in measure]
            else:
                cells = ['-'] * (2 + len(columns))
            head = label + (f':{idx}' if len(outs) >= 2 else '')
            counts = [str(n_param) if n_param else '-', str(n_buf) if n_buf else '-'] if idx == 0 else ['-', '-']
            table.append([head] + counts + cells)
    table.append(['---'] * len(table[0]))
    table.append(['Total', str(totals[0]), str(totals[1])] + ['-'] * (len(table[0]) - 3))

    widths = [max(len(row[i]) for row in table) for i in range(len(table[0]))]
    print()
    for row in table:
        print('  '.join(cell.ljust(w) for cell, w in zip(row, widths)))
    print()
    return outputs
