"""Attribute-style dict and dotted-name resolution (behavioural mirror of the reference's
dnnlib/util.py:40-53 and :236-303; written from scratch)."""

import importlib
import os
from typing import Any


class EasyDict(dict):
    """dict whose items are also reachable as attributes (`d.key` <-> `d['key']`).

    Missing keys raise AttributeError on attribute access, so `getattr(d, k, default)`
    and `hasattr` behave the way pickled reference code expects."""

    def __getattr__(self, name: str) -> Any:
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def __delattr__(self, name: str) -> None:
        if name not in self:
            raise AttributeError(name)
        del self[name]


_NAME_ALIASES = {'np': 'numpy'}


def get_obj_by_name(name: str) -> Any:
    """Resolve 'pkg.module.attr.sub' to the Python object, importing the longest
    importable module prefix."""
    parts = name.split('.')
    parts[0] = _NAME_ALIASES.get(parts[0], parts[0])
    first_error = None
    for split in range(len(parts), 0, -1):
        module_name = '.'.join(parts[:split])
        try:
            obj = importlib.import_module(module_name)
        except ModuleNotFoundError as err:
            # Only "this prefix is not a module" is a wrong guess; a module that exists but
            # fails to import one of ITS dependencies is a real error.
            missing = err.name or ''
            if not (module_name == missing or module_name.startswith(missing + '.')):
                raise
            first_error = first_error or err
            continue
        try:
            for attr in parts[split:]:
                obj = getattr(obj, attr)
            return obj
        except AttributeError as err:
            first_error = first_error or err
    raise ImportError(f'cannot resolve object name {name!r}') from first_error


def call_func_by_name(*args, func_name: str = None, **kwargs) -> Any:
    assert func_name is not None
    fn = get_obj_by_name(func_name)
    assert callable(fn), f'{func_name} is not callable'
    return fn(*args, **kwargs)


def construct_class_by_name(*args, class_name: str = None, **kwargs) -> Any:
    return call_func_by_name(*args, func_name=class_name, **kwargs)


_cache_dir = None


def make_cache_dir_path(*paths: str) -> str:
    root = _cache_dir or os.environ.get('DNNLIB_CACHE_DIR') or os.path.join(os.path.expanduser('~'), '.cache', 'dnnlib')
    return os.path.join(root, *paths)


def format_time(seconds) -> str:
    s = int(round(seconds))
    if s < 60:
        return f'{s}s'
    if s < 3600:
        return f'{s // 60}m {s % 60:02d}s'
    if s < 86400:
        return f'{s // 3600}h {(s // 60) % 60:02d}m {s % 60:02d}s'
    return f'{s // 86400}d {(s // 3600) % 24:02d}h {(s // 60) % 60:02d}m'
