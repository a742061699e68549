"""The two ends of the pixel path as single device passes (csrc/video_io.hip):

* `video_to_uint8`: network output [N, C, T, H, W] in [-1, 1] -> display bytes [N, T, H, W, C] uint8 -- the reference's
  `(segment * 127.5 + 128).clamp(0, 255).to(torch.uint8)` plus the channel-last rearrangement it does per frame before
  handing frames to the video writer (utils.py:163-171, :203-209);
* `video_from_uint8`: decoded frames [N, T, H, W, C] uint8 -> network input [N, C, T, H, W] = 2 * x / 255 - 1 with the
  optional horizontal flip per sample (dataset.py:81-83 read_frame, :93-94 x_flip) -- the reference does this per frame on
  the CPU inside the DataLoader workers; here bytes cross PCIe (4x less than float32) and the conversion is a device pass.

GPU tensors go through the C ABI (`lvg_video_to_uint8` / `lvg_video_from_uint8`) and fail loudly without the library;
CPU tensors take the reference's own tensor expressions. float32 results are bit-identical between the two."""

from typing import Optional

import torch

from torch_utils import custom_ops
from torch_utils.ops import _hip

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        custom_ops.get_plugin(module_name='video_io_plugin')
        _plugin = _hip.lib()
    return True


def video_to_uint8(video: torch.Tensor) -> torch.Tensor:
    """[N, C, T, H, W] float32 / float16 / bfloat16 -> [N, T, H, W, C] uint8 (16-bit inputs are widened to float32 first)."""
    assert video.ndim == 5
    n, c, t, h, w = video.shape
    if video.device.type == 'cuda' and _init():
        assert 1 <= c <= 4 and w % 4 == 0, 'video_to_uint8: 1..4 channels and a width that is a multiple of 4'
        video = video.contiguous()
        out = torch.empty((n, t, h, w, c), dtype=torch.uint8, device=video.device)
        with torch.cuda.device(video.device):
            rc = _hip.lib().lvg_video_to_uint8(video.data_ptr(), out.data_ptr(), n, c, t, h, w, _hip.dtype_code(video.dtype), _hip.stream(video.device))
        _hip.check(rc, 'video_to_uint8')
        return out
    return (video.float() * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 4, 1).contiguous()


def video_from_uint8(frames: torch.Tensor, flip: Optional[torch.Tensor] = None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """[N, T, H, W, C] uint8 -> [N, C, T, H, W] `dtype` in [-1, 1]; `flip` [N] (bool / uint8): samples to mirror in x."""
    assert frames.ndim == 5 and frames.dtype == torch.uint8
    n, t, h, w, c = frames.shape
    if frames.device.type == 'cuda' and _init():
        assert 1 <= c <= 4 and w % 4 == 0, 'video_from_uint8: 1..4 channels and a width that is a multiple of 4'
        frames = frames.contiguous()
        fl = None if flip is None else flip.to(device=frames.device, dtype=torch.uint8).contiguous()
        out = torch.empty((n, c, t, h, w), dtype=dtype, device=frames.device)
        with torch.cuda.device(frames.device):
            rc = _hip.lib().lvg_video_from_uint8(frames.data_ptr(), out.data_ptr(), _hip.ptr(fl), n, c, t, h, w, _hip.dtype_code(dtype), _hip.stream(frames.device))
        _hip.check(rc, 'video_from_uint8')
        return out
    video = (2 * frames.permute(0, 4, 1, 2, 3).to(torch.float32) / 255 - 1)
    if flip is not None:
        video = torch.where(flip.to(torch.bool).reshape(n, 1, 1, 1, 1), video.flip(dims=(-1,)), video)
    return video.to(dtype).contiguous()
