"""The C-ABI library loads and exports every symbol the headers under include/ declare (no compute): lvg_ops.h = the drop-in operator
ABI, lvg_test_hooks.h = the test / measurement controls kept OUT of it."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'lvg_ops.h')
HEADERS = sorted(os.path.join(ROOT, 'include', f) for f in os.listdir(os.path.join(ROOT, 'include')) if f.endswith('.h'))
LIB = os.path.join(ROOT, 'long-video-gan_amd', 'lib', 'liblvg_hip.so')


def declared_symbols(headers=None):
    syms = set()
    for h in (HEADERS if headers is None else headers):
        text = open(h).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        syms.update(re.findall(r'\b(lvg_[a-z0-9_]+)\s*\(', text))
    return sorted(syms)


def test_operator_header_holds_no_test_hooks():
    """Routing overrides are process-wide test controls: they are declared in lvg_test_hooks.h, not in the operator ABI."""
    ops = declared_symbols([HEADER])
    assert 'lvg_filtered_lrelu_set_impl' not in ops and 'lvg_conv3d_frames_set_plan' not in ops
    hooks = declared_symbols([os.path.join(ROOT, 'include', 'lvg_test_hooks.h')])
    assert hooks == ['lvg_conv3d_frames_set_plan', 'lvg_filtered_lrelu_set_impl']


def test_header_declares_the_hot_path():
    syms = declared_symbols([HEADER])
    for must in ('lvg_bias_act', 'lvg_upfirdn2d', 'lvg_filtered_lrelu', 'lvg_filtered_lrelu_act', 'lvg_last_error', 'lvg_abi_version'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(LIB):
        import __graft_entry__
        __graft_entry__.build()
    import torch  # noqa: F401  (maps libamdhip64 first, as the product loader does)
    lib = ctypes.CDLL(LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), f'{name} declared under include/ but not exported by liblvg_hip.so'
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


def test_planning_queries_without_a_gpu(monkeypatch):
    """The launch-planning queries of the C ABI are host arithmetic (no device work): tile / slot / split counts for the shapes of
    BASELINE.json configs[1], and 0 = "no kernel for this shape" where the callers keep the library route."""
    from torch_utils.ops import _hip
    for var in ('LVG_CONV_BM', 'LVG_CONV_BN', 'LVG_CONV_NB', 'LVG_CONV_PERSIST', 'LVG_WGRAD_SPLITS', 'LVG_WGRAD_TARGET'):
        monkeypatch.delenv(var, raising=False)
    lib = _hip.lib()
    wg = lib.lvg_conv3d_frames_workgroups
    # 640 frames of 9x16, 512 -> 512 channels, 3x3x3 taps: 128-pixel x 128-channel tiles
    assert wg(640, 9, 16, 512, 512, 3, 3, 3) == (640 * 9 * 16 // 128) * 4
    # 64 output channels: 64-channel tiles, 256 pixels each on wide frames with enough tiles (128 otherwise); 32 input channels / 7x7 taps: no kernel
    assert wg(1024, 36, 64, 64, 64, 1, 3, 3) == 1024 * 36 * 64 // 256
    assert wg(32, 36, 64, 64, 64, 1, 3, 3) == 32 * 36 * 64 // 128
    # the plan can be forced (A/B measurements, the bitwise-agreement tests) and released again
    assert lib.lvg_conv3d_frames_set_plan(128, 0, 0, 0) == 0 and wg(1024, 36, 64, 64, 64, 1, 3, 3) == 1024 * 36 * 64 // 128
    assert lib.lvg_conv3d_frames_set_plan(100, 0, 0, 0) != 0
    assert lib.lvg_conv3d_frames_set_plan(0, 0, 0, 0) == 0 and wg(1024, 36, 64, 64, 64, 1, 3, 3) == 1024 * 36 * 64 // 256
    assert wg(1024, 64, 64, 32, 64, 1, 3, 3) == 0 and wg(16, 8, 8, 64, 64, 1, 7, 7) == 0
    # weight gradient: one full round of workgroups, rounded down (2 per CU; 1 for 64-pixel-wide frames), at least 8 K-steps each
    sp = lib.lvg_conv3d_frames_wgrad_splits
    assert sp(640, 9, 16, 512, 512, 3, 3, 3) == 512 // (8 * 8 * 3)
    assert sp(1024, 18, 32, 128, 128, 1, 3, 3) == 512 // 4
    assert sp(1024, 36, 64, 64, 64, 1, 3, 3) == 256
    assert sp(1024, 36, 64, 64, 64, 1, 1, 1) == 0 and sp(1024, 3, 4, 512, 512, 3, 3, 3) == 0      # 1x1 taps / 4-pixel-wide frames: library route
    # fused bias gradient: channels-last streams whose channel vector count divides the block
    slots = lib.lvg_bias_act_grad_bias_slots
    n = 1024 * 64 * 36 * 64
    assert slots(n, 64, 2) == n // 8 // 1024 and slots(n, 24, 2) == 0 and slots(1024, 64, 2) == 0
    # epilogue partial-sum slots are positive for any shape
    assert lib.lvg_modconv_epilogue_slots(1024, 64, 36 * 64, 1, 2, 1) >= 1 and lib.lvg_tapconv_epilogue_slots(1024, 64, 36 * 64, 2) >= 1
