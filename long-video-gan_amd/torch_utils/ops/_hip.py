"""ctypes marshalling between torch tensors and the C ABI of liblvg_hip.so (include/lvg_ops.h).

The host side allocates every output with torch (caching allocator, current device) and hands
raw device pointers, element strides and the CURRENT torch stream to the library; nothing here
synchronises. All failures raise -- a GPU tensor never silently takes a PyTorch fallback."""

import ctypes

import torch

from .. import custom_ops

F32, F16, BF16, F64 = 0, 1, 2, 3
_DTYPE_CODES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16, torch.float64: F64}

SIGNS_NONE, SIGNS_WRITE, SIGNS_READ = 0, 1, 2
ERR_UNSUPPORTED = -2

_vp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
_i64x4 = ctypes.c_int64 * 4
_i64x2 = ctypes.c_int64 * 2

_SIGNATURES = {
    'lvg_bias_act': [_vp] * 6 + [_i64, _i64, _i64, _i32, _i32, _i32, _f32, _f32, _f32, _vp],
    'lvg_bias_act_grad_bias_slots': [_i64, _i32, _i32],
    'lvg_bias_act_grad_bias': [_vp] * 5 + [_i64, _i32, _i32, _i32, _f32, _f32, _f32, _vp],
    'lvg_upfirdn2d': [_vp] * 5 + [_i64x4] * 4 + [_i32, _i32, _i64, _i64] + [_i32] * 4 + [_i32, _i32, _i32, _f32, _i32, _vp],
    'lvg_filtered_lrelu': [_vp] * 6 + [_i64x4] * 4 + [_i32] * 6 + [_i64x2, _i32, _i32, _i32, _f32, _f32, _f32, _i32, _i32, _i32, _vp],
    'lvg_filtered_lrelu_supported': [_i32] * 5,
    'lvg_filtered_lrelu_set_impl': [_i32],
    'lvg_filtered_lrelu_act': [_vp, _vp, _i64x4, _i64x4, _i64x2, _i32, _i32, _f32, _f32, _f32, _i32, _i32, _vp],
    'lvg_modconv_epilogue': [_vp] * 6 + [_i64, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp],
    'lvg_modconv_epilogue_backward': [_vp] * 9 + [_i64, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp],
    'lvg_modconv_epilogue_dual': [_vp] * 7 + [_i64, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp],
    'lvg_modconv_epilogue_dual_backward': [_vp] * 10 + [_i64, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp],
    'lvg_modconv_epilogue_slots': [_i64, _i32, _i32, _i32, _i32, _i32],
    'lvg_tapconv_epilogue_slots': [_i64, _i32, _i32, _i32],
    'lvg_tapconv_epilogue': [_vp] * 8 + [_i64, _i32, _i32, _i32, _i64, _i32, _i32, _f32, _f32, _f32, _vp],
    'lvg_conv3d_frames': [_vp] * 9 + [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _i32, _f32, _f32, _f32, _vp],
    'lvg_conv3d_frames_workgroups': [_i64] + [_i32] * 7,
    'lvg_conv3d_frames_set_plan': [_i32] * 4,
    'lvg_conv3d_frames_workgroups_f32out': [_i64] + [_i32] * 7,
    'lvg_conv3d_frames_ex': [_vp] * 9 + [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _i32, _i32, _f32, _f32, _f32, _vp],
    'lvg_conv3d_frames_wgrad': [_vp] * 4 + [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _vp],
    'lvg_conv3d_frames_wgrad_splits': [_i64] + [_i32] * 7,
    'lvg_weight_prep': [_vp] * 4 + [_i32, _i32, _i32, _f32, _i32, _i32, _vp],
    'lvg_weight_prep_backward': [_vp, _vp, _vp, ctypes.c_int64 * 3, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _i32, _vp],
    'lvg_weight_dgrad_pack': [_vp, _vp, _i32, _i32, _i32, _vp],
    'lvg_style_prep': [_vp] * 5 + [_i32] * 4 + [_vp],
    'lvg_style_prep_backward': [_vp] * 10 + [_i32] * 4 + [_vp],
    'lvg_video_to_uint8': [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    'lvg_video_from_uint8': [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    'lvg_modconv2d_nchw_to_nhwc': [_vp] * 6 + [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    'lvg_modconv2d_nhwc_to_nchw': [_vp] * 6 + [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    'lvg_modconv2d_nchw_to_nhwc_padded': [_vp] * 6 + [_i64] + [_i32] * 12 + [_vp],
    'lvg_modconv2d_nchw_to_nhwc_padded_planar': [_vp] * 6 + [_i64] + [_i32] * 12 + [_vp],
    'lvg_conv2d_frames_workgroups': [_i64] + [_i32] * 8,
    'lvg_conv2d_frames': [_vp] * 4 + [_i64] + [_i32] * 10 + [_i64, _i64, _i32, _i32, _vp],
    'lvg_conv2d_frames_planes': [_vp] * 4 + [_i64] + [_i32] * 11 + [_i64, _i32, _vp],
    'lvg_conv2d_frames_planes_dot_rows': [_i32, _i32],
    'lvg_conv2d_frames_planes_dot': [_vp] * 7 + [_i32, _i32, _i64] + [_i32] * 11 + [_i64, _i32, _vp],
    'lvg_conv2d_frames_wgrad_splits': [_i64] + [_i32] * 8,
    'lvg_conv2d_frames_wgrad': [_vp] * 3 + [_i64] + [_i32] * 8 + [_i64, _i64, _i32, _i32, _vp],
    'lvg_ada_warp': [_vp] * 5 + [_i32] * 4 + [_vp],
    'lvg_ada_warp_adjoint': [_vp] * 6 + [_i32] * 4 + [_vp],
    'lvg_ada_colour': [_vp] * 6 + [_i32] * 5 + [_vp],
    'lvg_plane_sum': [_vp, _vp, _i64, _i64, _i32, _vp],
    'lvg_plane_sum_sq': [_vp, _vp, _i64, _i64, _i32, _vp],
    'lvg_plane_absmax': [_vp, _vp, _i64, _i64, _i32, _vp],
    'lvg_split16_frames': [_vp, _vp, _vp, _vp, _i64] + [_i32] * 10 + [_vp],
    'lvg_nhwc_f32_to_nchw': [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    'lvg_weight_prep2d': [_vp] * 5 + [_i32] * 5 + [_f32, _i32, _vp],
    'lvg_weight_prep2d_backward': [_vp] * 5 + [_i32] * 5 + [_f32, _vp],
    'lvg_adam_step': [_vp] * 5 + [_i64, _f32, _f32, _f32, _f32, _i64, _f32, _vp],
    'lvg_pointwise_thin_out': [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    'lvg_pointwise_thin_in': [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    'lvg_pointwise_thin_wgrad_blocks': [_i64, _i32],
    'lvg_pointwise_thin_wgrad': [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    'lvg_pointwise_wgrad_splits': [_i64, _i32, _i32],
    'lvg_pointwise_wgrad': [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i32, _i32, _vp],
    'lvg_split32_stack': [_vp, _vp, _i64, _i32, _i64, _i32, _i32, _vp],
    'lvg_noise_filter_bank': [_vp] * 5 + [_i32] * 7 + [_vp],
    'lvg_tapconv_epilogue_backward': [_vp] * 10 + [_i64, _i32, _i32, _i32, _i64, _i32, _i32, _f32, _f32, _f32, _vp],
}

_lib = None


def lib():
    """The loaded library with argtypes installed; raises custom_ops.PluginUnavailable if
    liblvg_hip.so is absent."""
    global _lib
    if _lib is None:
        handle = custom_ops.load_library()
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int64 if name in ('lvg_conv3d_frames_workgroups', 'lvg_conv3d_frames_workgroups_f32out', 'lvg_bias_act_grad_bias_slots', 'lvg_conv2d_frames_workgroups', 'lvg_conv2d_frames_planes_dot_rows') else ctypes.c_int
        _lib = handle
    return _lib


def dtype_code(dtype):
    code = _DTYPE_CODES.get(dtype)
    if code is None:
        raise TypeError(f'dtype {dtype} is not supported by the HIP ops (float32/float16/bfloat16/float64)')
    return code


def ptr(t):
    """Device pointer of a tensor, or NULL for None / empty tensors (the reference plugin
    convention: numel()==0 means "absent")."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def shape4(t):
    return _i64x4(*t.shape)


def stride4(t):
    return _i64x4(*t.stride())


def pair(a, b):
    return _i64x2(a, b)


def check(rc, what):
    """Raise for a negative return code (RuntimeError mirrors TORCH_CHECK in the reference)."""
    if rc < 0:
        msg = lib().lvg_last_error()
        raise RuntimeError(f'{what}: {msg.decode() if msg else "error"} (code {rc})')
    return rc
