"""Pickle protocol v6 of the reference's `torch_utils.persistence` (persistence.py:27-208).

Wire format (what lives inside a reference `.pkl`, and what this module writes):
a persistent object reduces to

    torch_utils.persistence._reconstruct_persistent_obj(meta)

with `meta = dict(type='class', version=6, module_src=<source text of the defining module>,
class_name=<name>, state=<object state>)`. Loading execs `module_src` in a fresh module,
so pickles made by the reference load here (their source imports `torch_utils.ops.*`, which
resolve to the HIP-backed ops of this package) and pickles made here load under the reference.
"""

import copy
import inspect
import io
import pickle
import sys
import types
import uuid

import dnnlib

_version = 6            # must equal the reference's (persistence.py:27); checked on load
_decorators = set()     # every generated persistent subclass
_import_hooks = []      # callables meta -> meta, run on load
_module_to_src_dict = {}
_src_to_module_dict = {}


def _module_to_src(module):
    src = _module_to_src_dict.get(module)
    if src is None:
        src = inspect.getsource(module)
        _module_to_src_dict[module] = src
        _src_to_module_dict[src] = module
    return src


def _src_to_module(src):
    module = _src_to_module_dict.get(src)
    if module is None:
        module = types.ModuleType('_imported_module_' + uuid.uuid4().hex)
        sys.modules[module.__name__] = module
        _module_to_src_dict[module] = src
        _src_to_module_dict[src] = module
        exec(src, module.__dict__)  # pylint: disable=exec-used
    return module


def is_persistent(obj):
    """True for a persistent class or an instance of one."""
    try:
        if obj in _decorators:
            return True
    except TypeError:
        pass
    return type(obj) in _decorators


def import_hook(hook):
    """Register `hook(meta) -> meta`, called for every persistent object being unpickled
    (e.g. to patch `meta.module_src` of old pickles)."""
    assert callable(hook)
    _import_hooks.append(hook)


def _skeleton(obj):
    """Replace everything known to pickle by None so that pickling the remainder is cheap."""
    if isinstance(obj, (list, tuple, set)):
        return [_skeleton(v) for v in obj]
    if isinstance(obj, dict):
        return [[_skeleton(k), _skeleton(v)] for k, v in obj.items()]
    if isinstance(obj, (str, int, float, bool, bytes, bytearray)) or obj is None:
        return None
    if f'{type(obj).__module__}.{type(obj).__name__}' in ('numpy.ndarray', 'torch.Tensor', 'torch.nn.parameter.Parameter'):
        return None
    if is_persistent(obj):
        return None
    return obj


def _check_pickleable(obj):
    with io.BytesIO() as sink:
        pickle.dump(_skeleton(obj), sink)


def persistent_class(orig_class):
    """Class decorator: instances pickle together with the source of their defining module
    and remember their constructor arguments (`obj.init_args`, `obj.init_kwargs`)."""
    assert isinstance(orig_class, type)
    if is_persistent(orig_class):
        return orig_class
    assert orig_class.__module__ in sys.modules
    home = sys.modules[orig_class.__module__]
    home_src = _module_to_src(home)

    class Decorator(orig_class):
        _orig_module_src = home_src
        _orig_class_name = orig_class.__name__

        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            keep = getattr(self, '_record_init_args', True)
            self._init_args = copy.deepcopy(args) if keep else None
            self._init_kwargs = copy.deepcopy(kwargs) if keep else None
            assert orig_class.__name__ in home.__dict__
            _check_pickleable(self.__reduce__())

        @property
        def init_args(self):
            assert self._init_args is not None
            return copy.deepcopy(self._init_args)

        @property
        def init_kwargs(self):
            assert self._init_kwargs is not None
            return dnnlib.EasyDict(copy.deepcopy(self._init_kwargs))

        def __reduce__(self):
            parts = list(super().__reduce__())
            while len(parts) < 3:
                parts.append(None)
            if parts[0] is not _reconstruct_persistent_obj:
                meta = dict(type='class', version=_version, module_src=self._orig_module_src,
                            class_name=self._orig_class_name, state=parts[2])
                parts[0], parts[1], parts[2] = _reconstruct_persistent_obj, (meta,), None
            return tuple(parts)

    Decorator.__name__ = orig_class.__name__
    Decorator.__qualname__ = orig_class.__qualname__
    _decorators.add(Decorator)
    return Decorator


def _reconstruct_persistent_obj(meta):
    """Unpickle entry point named inside every persistent pickle."""
    meta = dnnlib.EasyDict(meta)
    meta.state = dnnlib.EasyDict(meta.state)
    for hook in _import_hooks:
        meta = hook(meta)
        assert meta is not None
    assert meta.version == _version, f'pickle protocol {meta.version}, this module speaks {_version}'
    assert meta.type == 'class'
    module = _src_to_module(meta.module_src)
    cls = persistent_class(module.__dict__[meta.class_name])
    obj = cls.__new__(cls)
    restore = getattr(obj, '__setstate__', None)
    if callable(restore):
        restore(meta.state)  # pylint: disable=not-callable
    else:
        obj.__dict__.update(meta.state)
    return obj
