"""DiffAugment for videos ([N, C, T, H, W]): color, translation, cutout with one random draw per
sample shared by all frames (reference model/diff_augment.py:20-102). Written against the same
distributions; on-device, no host synchronisation."""

import math

import torch
import torch.nn.functional as F


def _per_sample(x: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    return torch.rand(x.size(0), 1, 1, 1, 1, dtype=x.dtype, device=x.device) * (hi - lo) + lo


def color(x: torch.Tensor) -> torch.Tensor:
    x = x + _per_sample(x, -0.5, 0.5)                                    # brightness
    mean_c = x.mean(dim=1, keepdim=True)
    x = (x - mean_c) * _per_sample(x, 0.0, 2.0) + mean_c                 # saturation
    mean_all = x.mean(dim=(1, 2, 3, 4), keepdim=True)
    return (x - mean_all) * _per_sample(x, 0.5, 1.5) + mean_all          # contrast


def translation(x: torch.Tensor, ratio: float = 0.25) -> torch.Tensor:
    n, c, t, h, w = x.shape
    shift = round(max(h, w) * ratio)
    dy = torch.randint(-shift, shift + 1, (n, 1), device=x.device)
    dx = torch.randint(-shift, shift + 1, (n, 1), device=x.device)
    iy = torch.arange(h, device=x.device).unsqueeze(0) + dy              # source row per output row
    ix = torch.arange(w, device=x.device).unsqueeze(0) + dx
    ok = ((iy >= 0) & (iy < h)).unsqueeze(2) & ((ix >= 0) & (ix < w)).unsqueeze(1)   # [n, h, w]
    iy, ix = iy.clamp(0, h - 1), ix.clamp(0, w - 1)
    flat = (iy.unsqueeze(2) * w + ix.unsqueeze(1)).reshape(n, 1, 1, h * w).expand(n, c, t, h * w)
    out = x.reshape(n, c, t, h * w).gather(3, flat).reshape(n, c, t, h, w)
    return out * ok.reshape(n, 1, 1, h, w).to(x.dtype)


def cutout(x: torch.Tensor, ratio: float = 0.5) -> torch.Tensor:
    n, c, t, h, w = x.shape
    ch, cw = int(h * ratio + 0.5), int(w * ratio + 0.5)
    oy = torch.randint(0, h + (1 - ch % 2), (n, 1, 1), device=x.device)
    ox = torch.randint(0, w + (1 - cw % 2), (n, 1, 1), device=x.device)
    yy = torch.arange(h, device=x.device).reshape(1, h, 1)
    xx = torch.arange(w, device=x.device).reshape(1, 1, w)
    # rows/cols the reference clamps into range are exactly those within the window clipped to the image
    y0, y1 = (oy - ch // 2).clamp(0, h - 1), (oy - ch // 2 + ch - 1).clamp(0, h - 1)
    x0, x1 = (ox - cw // 2).clamp(0, w - 1), (ox - cw // 2 + cw - 1).clamp(0, w - 1)
    hole = (yy >= y0) & (yy <= y1) & (xx >= x0) & (xx <= x1)
    return x * (~hole).reshape(n, 1, 1, h, w).to(x.dtype)


_POLICIES = {'color': color, 'translation': translation, 'cutout': cutout}


def diff_augment(x: torch.Tensor, policy: str = 'color,translation,cutout') -> torch.Tensor:
    for name in [p for p in policy.split(',') if p]:
        x = _POLICIES[name](x)
    return x.contiguous()


def temporal_scale_params(n: int, frames: int, seq_length: int, amount: float):
    """The host-side random draws of `temporal_scale_augment` for `n` clips of `frames` frames, in the reference's order (per sample: the
    stretch, the pad offset, the crop offset; video_gan_lres.py:242-263), turned into what the device needs: for every output frame the
    source frame below it, the interpolation weight of the next one and whether it lies inside the stretched clip (else zero padding).
    -> (i0 [n, seq_length] int64, frac [n, seq_length] float32, valid [n, seq_length] float32), CPU tensors.
    Parity is with the reference's interpolation as torch's CPU kernel computes it (the golden fixtures come from the reference on the
    CPU): source positions in float64, and the plain copy `upsample_bilinear2d` makes when the stretched length equals the clip's (scales
    in [1, 1 + 1/T)). The reference on a GPU would use float32 source indices and no such shortcut there: a difference below 1/T in one
    interpolation weight, for that range of scales only."""
    i0s, fracs, valids = [], [], []
    steps = torch.arange(seq_length, dtype=torch.float64)
    for _ in range(n):
        scale = float(2 ** torch.empty(()).uniform_(-amount, amount))
        length = int(math.floor(float(frames * scale)))                  # F.interpolate's output size for scale_factor = scale
        room = max(0, seq_length - length)
        p0 = int(torch.randint(room + 1, ()))
        c0 = int(torch.randint(length + room - seq_length + 1, ()))
        s = steps + (c0 - p0)                                            # index into the stretched clip
        src = ((s + 0.5) / scale - 0.5).clamp(min=0.0)                   # bilinear, align_corners=False, the GIVEN scale (recompute_scale_factor=False)
        if length == frames:                                             # upsample_bilinear2d copies when the size does not change, whatever the scale
            src = s.clamp(min=0.0)
        lo = src.floor().clamp(max=frames - 1)
        i0s.append(lo.long())
        fracs.append((src - lo).clamp(0.0, 1.0).float())
        valids.append(((s >= 0) & (s < length)).float())
    return torch.stack(i0s), torch.stack(fracs), torch.stack(valids)


def to_device_async(t: torch.Tensor, device) -> torch.Tensor:
    """A small host tensor of random draws -> `device` without stalling the host: from pageable memory a host-to-device copy blocks until the
    stream has drained (the trainers issue several per micro-batch); from pinned memory it is asynchronous, and torch's caching host
    allocator keeps the pinned block alive until the copy has run."""
    device = torch.device(device)
    if device.type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def temporal_scale_apply(video: torch.Tensor, i0: torch.Tensor, frac: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    """[N, C, T, H, W] -> [N, C, seq_length, H, W]: two gathers along time, one lerp, one mask -- for all samples at once (the reference
    form interpolates, pads, crops and stacks sample by sample: 24 slow launches of a bilinear kernel on a strided view per 8 clips)."""
    n, c, t, h, w = video.shape
    seq = i0.shape[1]
    i1 = (i0 + 1).clamp(max=t - 1)
    view = lambda v: v[:, None, :, None, None]
    g0 = video.gather(2, view(i0).expand(n, c, seq, h, w))
    g1 = video.gather(2, view(i1).expand(n, c, seq, h, w))
    return torch.lerp(g0, g1, view(frac).to(video.dtype)) * view(valid).to(video.dtype)


def temporal_scale_augment(video: torch.Tensor, seq_length: int, amount: float) -> torch.Tensor:
    """Per-sample random time stretch by 2**U(-amount, amount) (bilinear along T), then random
    pad/crop back to seq_length (reference video_gan_lres.py:242-263)."""
    if amount <= 0:
        return video
    i0, frac, valid = temporal_scale_params(video.size(0), video.size(2), seq_length, amount)
    dev = video.device
    return temporal_scale_apply(video, to_device_async(i0, dev), to_device_async(frac, dev), to_device_async(valid, dev))


def temporal_scale_augment_reference_form(video: torch.Tensor, seq_length: int, amount: float) -> torch.Tensor:
    """The sample-by-sample form of the reference (kept as the definition the vectorised form is tested against)."""
    if amount <= 0:
        return video
    out = []
    for v in video.permute(0, 1, 3, 4, 2):                       # [c, h, w, t] per sample
        scale = float(2 ** torch.empty(()).uniform_(-amount, amount))
        v = F.interpolate(v, mode='bilinear', align_corners=False, recompute_scale_factor=False, scale_factor=(1, scale))
        room = max(0, seq_length - v.size(-1))
        p0 = int(torch.randint(room + 1, ()))
        v = F.pad(v, (p0, room - p0))
        i0 = int(torch.randint(v.size(-1) - seq_length + 1, ()))
        out.append(v[..., i0:i0 + seq_length])
    return torch.stack(out).permute(0, 1, 4, 2, 3)


def crop_time(video: torch.Tensor, t0: torch.Tensor, seq_length: int) -> torch.Tensor:
    """video[i, :, t0[i] : t0[i] + seq_length] for every sample i as ONE gather (t0 on the video's device): no host-side indices, so the
    crop can live inside a captured graph."""
    n, c, _, h, w = video.shape
    idx = t0.reshape(n, 1) + torch.arange(seq_length, device=video.device).reshape(1, seq_length)
    return video.gather(2, idx[:, None, :, None, None].expand(n, c, seq_length, h, w))
