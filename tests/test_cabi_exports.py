"""The C-ABI library loads and exports every symbol include/lvg_ops.h declares (no compute)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'lvg_ops.h')
LIB = os.path.join(ROOT, 'long-video-gan_amd', 'lib', 'liblvg_hip.so')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lvg_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ('lvg_bias_act', 'lvg_upfirdn2d', 'lvg_filtered_lrelu', 'lvg_filtered_lrelu_act', 'lvg_last_error', 'lvg_abi_version'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(LIB):
        import __graft_entry__
        __graft_entry__.build()
    import torch  # noqa: F401  (maps libamdhip64 first, as the product loader does)
    lib = ctypes.CDLL(LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), f'{name} declared in include/lvg_ops.h but not exported by liblvg_hip.so'
    lib.lvg_abi_version.restype = ctypes.c_int
    assert lib.lvg_abi_version() == 1


def test_python_loader_sets_signatures():
    from torch_utils.ops import _hip
    lib = _hip.lib()
    for name in _hip._SIGNATURES:
        assert getattr(lib, name).argtypes is not None


def test_gpu_tensor_without_library_fails_loudly(monkeypatch):
    """No silent fallback: with the library path broken, _init() raises."""
    from torch_utils import custom_ops
    from torch_utils.ops import _hip
    monkeypatch.setattr(custom_ops, '_lib', None)
    monkeypatch.setattr(_hip, '_lib', None)
    monkeypatch.setattr(custom_ops, '_cached_plugins', {})
    monkeypatch.setenv('LVG_HIP_LIB', '/nonexistent/liblvg_hip.so')
    with pytest.raises(custom_ops.PluginUnavailable):
        _hip.lib()
