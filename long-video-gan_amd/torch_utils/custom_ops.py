"""Plugin loader boundary (reference torch_utils/custom_ops.py:59-157).

The reference JIT-compiles each op's CUDA sources with nvcc through
torch.utils.cpp_extension and imports a pybind module. Here there is ONE prebuilt C-ABI
library, `lib/liblvg_hip.so` (hipcc --offload-arch=gfx950, built by csrc/Makefile or
`__graft_entry__.build()`), opened with ctypes. `get_plugin` is kept so that code calling
it (train scripts call `<op>._init()`, which calls this) keeps working; it never compiles
anything and never falls back to another backend."""

import ctypes
import os

verbosity = 'brief'  # 'none' | 'brief' | 'full' -- same knob as the reference (custom_ops.py:24)

_LIB_NAME = 'liblvg_hip.so'
_lib = None
_cached_plugins = {}


class PluginUnavailable(RuntimeError):
    """liblvg_hip.so is missing or failed to load: GPU ops refuse to run (there is no
    silent PyTorch fallback on a GPU tensor)."""


def library_path() -> str:
    override = os.environ.get('LVG_HIP_LIB')
    if override:
        return override
    pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.join(pkg_root, 'lib', _LIB_NAME)


def load_library():
    """dlopen liblvg_hip.so once. torch is imported first so that the HIP runtime already
    mapped by PyTorch (same SONAME libamdhip64.so.7) is the one the kernels launch on --
    stream handles from torch.cuda.current_stream() are then valid inside the library."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
    path = library_path()
    if not os.path.isfile(path):
        raise PluginUnavailable(f'{path} not found -- build it with `make -C long-video-gan_amd/csrc` '
                                f'or `python -c "import __graft_entry__ as g; g.build()"`')
    try:
        lib = ctypes.CDLL(path)
    except OSError as err:
        raise PluginUnavailable(f'cannot load {path}: {err}') from err
    lib.lvg_last_error.restype = ctypes.c_char_p
    lib.lvg_abi_version.restype = ctypes.c_int
    _lib = lib
    return lib


def get_plugin(module_name, sources=None, headers=None, source_dir=None, **build_kwargs):  # pylint: disable=unused-argument
    """Reference-compatible entry point. Returns the ctypes handle of liblvg_hip.so for any
    of the reference's plugin names; `sources`/`headers`/build flags are ignored because
    nothing is compiled at run time."""
    if module_name not in _cached_plugins:
        if verbosity == 'full':
            print(f'Loading HIP plugin for "{module_name}" from {library_path()}')
        _cached_plugins[module_name] = load_library()
    return _cached_plugins[module_name]
