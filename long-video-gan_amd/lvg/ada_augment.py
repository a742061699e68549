"""Adaptive discriminator augmentation for videos (reference model/ada_augment.py: AugmentPipe :88,
forward :157, random_temporal_filter :437) on the MI355X op stack.

One random transform per SAMPLE, applied identically to every frame of the clip ([N, C, T, H, W] is
processed as N images of C*T channels): integer / fractional geometric transforms folded into one
inverse affine map and resampled through a x2 oversampled grid (`upfirdn2d` with the 12-tap sym6
low-pass up and down, `grid_sample` in between), a 4x4 colour matrix, a 4-band image-space filter
bank applied as per-sample separable grouped convs, additive noise and cutout.

Random numbers are drawn in the same order, shapes and distributions as the reference (value first,
then the Bernoulli gate, per transform), so a seeded run reproduces the reference's augmentations;
`debug_percentile` replaces every draw by a fixed quantile (that is what the golden fixtures use).
The code itself is organised differently: transforms are described by small samplers, matrices are
built by two helpers, and the three stages are separate methods."""

import math
import os
from typing import List, Optional, Sequence

import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F

from torch_utils.ops import ada_ops, conv2d_gradfix, grid_sample_gradfix, upfirdn2d

# Orthogonal wavelet low-pass prototypes (Daubechies least-asymmetric, 6 and 2 vanishing moments).
SYM6 = (0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
        0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
        0.0017677118642428036, -0.007800708325034148)
WARP_GRAD = os.environ.get('LVG_ADA_WARP_GRAD', 'adjoint')        # 'composed': clips that carry a gradient take the reference's composition of ops

SYM2 = (-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025)


_CONST = {}             # (values, shape, dtype, device) -> tensor


def _cached(values, device, dtype=torch.float32) -> torch.Tensor:
    """A constant tensor from python numbers, built ONCE per (values, device): building it from a list is a host-to-device copy, which a stream
    capture does not allow -- the pipeline must be replayable from a hipGraph (lvg.phase_graphs) after one eager pass. Read-only by contract."""
    arr = np.asarray(values, dtype=np.float64)
    device = torch.device('cpu' if device is None else device)
    key = (arr.tobytes(), arr.shape, dtype, device)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.tensor(arr, dtype=dtype, device=device)
    return t


def _const(value, like: torch.Tensor) -> torch.Tensor:
    return _cached(value, like.device)


_MAT_BASE = {}          # (constant pattern, device) -> the matrix with zeros where per-sample tensors go


def _mat(rows: Sequence[Sequence], device=None) -> torch.Tensor:
    """Matrix from a nested list whose entries are python numbers or [N] tensors -> [N, r, c] (or [r, c]).
    With per-sample entries: ONE copy of a cached constant pattern, then one strided assignment per tensor entry (it used to be a
    `full_like` per constant entry plus a stack: ~2 x the launches, and the pipeline is a chain of a hundred such tiny launches)."""
    tensors = [e for row in rows for e in row if isinstance(e, torch.Tensor)]
    if not tensors:
        return _cached(rows, device)
    ref = tensors[0]
    r, c = len(rows), len(rows[0])
    const = tuple(0.0 if isinstance(e, torch.Tensor) else float(e) for row in rows for e in row)
    key = (const, r, c, ref.device, ref.dtype)
    base = _MAT_BASE.get(key)
    if base is None:
        base = _MAT_BASE[key] = torch.tensor(const, dtype=ref.dtype, device=ref.device).reshape(r, c)
    out = base.expand(*ref.shape, r, c).clone()
    for i, row in enumerate(rows):
        for j, e in enumerate(row):
            if isinstance(e, torch.Tensor):
                out[..., i, j] = e
    return out


def shift2(tx, ty, **kw):
    return _mat([[1, 0, tx], [0, 1, ty], [0, 0, 1]], **kw)


def zoom2(sx, sy, **kw):
    return _mat([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], **kw)


def turn2(theta, **kw):
    c, s = torch.cos(theta), torch.sin(theta)
    return _mat([[c, -s, 0], [s, c, 0], [0, 0, 1]], **kw)


def shift3(tx, ty, tz):
    return _mat([[1, 0, 0, tx], [0, 1, 0, ty], [0, 0, 1, tz], [0, 0, 0, 1]])


def zoom3(s):
    return _mat([[s, 0, 0, 0], [0, s, 0, 0], [0, 0, s, 0], [0, 0, 0, 1]])


def turn3(axis: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """Rotation by theta [N] around the unit axis (first three entries of `axis`)."""
    x, y, z = axis[0], axis[1], axis[2]
    s, c = torch.sin(theta), torch.cos(theta)
    k = 1 - c
    return _mat([[x * x * k + c, x * y * k - z * s, x * z * k + y * s, 0],
                 [y * x * k + z * s, y * y * k + c, y * z * k - x * s, 0],
                 [z * x * k - y * s, z * y * k + x * s, z * z * k + c, 0],
                 [0, 0, 0, 1]])


def _band_filters() -> np.ndarray:
    """4 x taps bank of band-pass filters built from the sym2 half-band pair by the a-trous recursion."""
    lo = np.asarray(SYM2)
    hi = lo * ((-1) ** np.arange(lo.size))
    lo2 = np.convolve(lo, lo[::-1]) / 2
    hi2 = np.convolve(hi, hi[::-1]) / 2
    bank = np.eye(4, 1)
    for i in range(1, 4):
        bank = np.dstack([bank, np.zeros_like(bank)]).reshape(4, -1)[:, :-1]         # zero-stuff (dilate by 2)
        bank = scipy.signal.convolve(bank, [lo2])
        mid = bank.shape[1] // 2
        bank[i, mid - hi2.size // 2: mid - hi2.size // 2 + hi2.size] += hi2
    return bank


class AugmentPipe(torch.nn.Module):
    def __init__(self, xflip=0, rotate90=0, xint=0, xint_max=0.125,
                 scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2, rotate_max=1, aniso_std=0.2, xfrac_std=0.125,
                 brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, brightness_std=0.2, contrast_std=0.5,
                 hue_max=1, saturation_std=1, imgfilter=0, imgfilter_bands=(1, 1, 1, 1), imgfilter_std=1,
                 noise=0, cutout=0, noise_std=0.1, cutout_size=0.5):
        super().__init__()
        self.register_buffer('p', torch.ones(()))                 # overall probability multiplier (ADA state)
        for name, value in dict(xflip=xflip, rotate90=rotate90, xint=xint, xint_max=xint_max, scale=scale, rotate=rotate,
                                aniso=aniso, xfrac=xfrac, scale_std=scale_std, rotate_max=rotate_max, aniso_std=aniso_std,
                                xfrac_std=xfrac_std, brightness=brightness, contrast=contrast, lumaflip=lumaflip, hue=hue,
                                saturation=saturation, brightness_std=brightness_std, contrast_std=contrast_std, hue_max=hue_max,
                                saturation_std=saturation_std, imgfilter=imgfilter, imgfilter_std=imgfilter_std, noise=noise,
                                cutout=cutout, noise_std=noise_std, cutout_size=cutout_size).items():
            setattr(self, name, float(value))
        self.imgfilter_bands = list(imgfilter_bands)
        self.register_buffer('Hz_geom', upfirdn2d.setup_filter(list(SYM6)))
        self.register_buffer('Hz_fbank', torch.as_tensor(_band_filters(), dtype=torch.float32))

    # -- sampling helpers ---------------------------------------------------------------------------

    def _gate(self, prob: float, shape: List[int], value: torch.Tensor, neutral: float) -> torch.Tensor:
        """Keep `value` with probability prob * p (one uniform draw of `shape`), else `neutral`."""
        keep = torch.rand(shape, device=value.device) < prob * self.p
        return torch.where(keep, value, torch.full_like(value, neutral))

    @staticmethod
    def _gauss_q(q: torch.Tensor) -> torch.Tensor:
        return torch.erfinv(q * 2 - 1)

    # -- stage 1: geometry --------------------------------------------------------------------------

    def _inverse_affine(self, n: int, width: int, height: int, dev, q) -> Optional[torch.Tensor]:
        g = None                                                # None = identity so far

        def push(m):
            nonlocal g
            g = m if g is None else g @ m

        if self.xflip > 0:
            i = torch.floor(torch.rand([n], device=dev) * 2)
            i = self._gate(self.xflip, [n], i, 0)
            if q is not None:
                i = torch.full_like(i, float(torch.floor(q * 2)))
            push(zoom2(1 / (1 - 2 * i), 1))
        if self.rotate90 > 0:
            i = torch.floor(torch.rand([n], device=dev) * 4)
            i = self._gate(self.rotate90, [n], i, 0)
            if q is not None:
                i = torch.full_like(i, float(torch.floor(q * 4)))
            push(turn2(math.pi / 2 * i))
        if self.xint > 0:
            t = (torch.rand([n, 2], device=dev) * 2 - 1) * self.xint_max
            t = self._gate(self.xint, [n, 1], t, 0)
            if q is not None:
                t = torch.full_like(t, float((q * 2 - 1) * self.xint_max))
            push(shift2(-torch.round(t[:, 0] * width), -torch.round(t[:, 1] * height)))
        if self.scale > 0:
            s = torch.exp2(torch.randn([n], device=dev) * self.scale_std)
            s = self._gate(self.scale, [n], s, 1)
            if q is not None:
                s = torch.full_like(s, float(torch.exp2(self._gauss_q(q) * self.scale_std)))
            push(zoom2(1 / s, 1 / s))
        # two rotations (before / after the anisotropic scale) whose union has probability rotate * p
        p_rot = 1 - torch.sqrt((1 - self.rotate * self.p).clamp(0, 1))
        if self.rotate > 0:
            th = (torch.rand([n], device=dev) * 2 - 1) * math.pi * self.rotate_max
            th = torch.where(torch.rand([n], device=dev) < p_rot, th, torch.zeros_like(th))
            if q is not None:
                th = torch.full_like(th, float((q * 2 - 1) * math.pi * self.rotate_max))
            push(turn2(th))
        if self.aniso > 0:
            s = torch.exp2(torch.randn([n], device=dev) * self.aniso_std)
            s = self._gate(self.aniso, [n], s, 1)
            if q is not None:
                s = torch.full_like(s, float(torch.exp2(self._gauss_q(q) * self.aniso_std)))
            push(zoom2(1 / s, s))
        if self.rotate > 0:
            th = (torch.rand([n], device=dev) * 2 - 1) * math.pi * self.rotate_max
            th = torch.where(torch.rand([n], device=dev) < p_rot, th, torch.zeros_like(th))
            if q is not None:
                th = torch.zeros_like(th)
            push(turn2(th))
        if self.xfrac > 0:
            t = torch.randn([n, 2], device=dev) * self.xfrac_std
            t = self._gate(self.xfrac, [n, 1], t, 0)
            if q is not None:
                t = torch.full_like(t, float(self._gauss_q(q) * self.xfrac_std))
            push(shift2(-t[:, 0] * width, -t[:, 1] * height))
        return g

    def _warp_margins(self, g_inv: torch.Tensor, width: int, height: int) -> torch.Tensor:
        """Reflect-padding margins (mx0, my0, mx1, my1) of the geometric stage: the reach of the transformed corners (+ filter support),
        clamped to the image size -- an int32 tensor on g_inv's device (reference ada_augment.py:275-284; no host read here)."""
        dev = g_inv.device
        cx, cy = (width - 1) / 2, (height - 1) / 2
        corners = _cached([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dev)
        reach = (g_inv @ corners.t())[:, :2, :].permute(1, 0, 2).flatten(1)               # [xy, N*4]
        reach = torch.cat([-reach, reach]).max(dim=1).values                              # [x0, y0, x1, y1]
        pad_f = self.Hz_geom.shape[0] // 4
        reach = reach + _const([pad_f * 2 - cx, pad_f * 2 - cy] * 2, g_inv)
        reach = torch.nan_to_num(reach, nan=0.0)                 # (a non-finite map must not turn into an out-of-range margin on the device)
        reach = reach.max(_const([0, 0] * 2, g_inv)).min(_const([width - 1, height - 1] * 2, g_inv))
        return reach.ceil().to(torch.int32)

    def _warp_composed(self, x: torch.Tensor, g_inv: torch.Tensor, margins) -> torch.Tensor:
        """The geometric stage as the reference spells it (ada_augment.py:286-301): reflect pad, x2 up, affine_grid + grid_sample, x2 down.
        `margins`: four Python ints."""
        n, k, height, width = x.shape
        dev = x.device
        pad_f = self.Hz_geom.shape[0] // 4
        mx0, my0, mx1, my1 = margins
        # (constant matrices in g_inv's dtype: float32 in the pipeline, float64 in the oracle tests)
        shift = lambda tx, ty: torch.tensor([[1, 0, tx], [0, 1, ty], [0, 0, 1]], dtype=g_inv.dtype, device=dev)
        zoom = lambda sx, sy: torch.tensor([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], dtype=g_inv.dtype, device=dev)
        x = F.pad(x, [mx0, mx1, my0, my1], mode='reflect')
        g_inv = shift((mx0 - mx1) / 2, (my0 - my1) / 2) @ g_inv

        # x2 oversampling keeps the bilinear resampler away from the signal band
        x = upfirdn2d.upsample2d(x=x, f=self.Hz_geom, up=2)
        g_inv = zoom(2, 2) @ g_inv @ zoom(0.5, 0.5)
        g_inv = shift(-0.5, -0.5) @ g_inv @ shift(0.5, 0.5)

        out_shape = [n, k, (height + pad_f * 2) * 2, (width + pad_f * 2) * 2]
        g_inv = zoom(2 / x.shape[3], 2 / x.shape[2]) @ g_inv @ zoom(out_shape[3] / 2, out_shape[2] / 2)
        grid = F.affine_grid(theta=g_inv[:, :2, :].to(x.dtype), size=out_shape, align_corners=False)
        x = grid_sample_gradfix.grid_sample(x, grid)
        return upfirdn2d.downsample2d(x=x, f=self.Hz_geom, down=2, padding=-pad_f * 2, flip_filter=True)

    def _warp(self, x: torch.Tensor, g_inv: torch.Tensor) -> torch.Tensor:
        """x [N, K, H, W] resampled through the inverse map g_inv [N, 3, 3] (pixel units, centred). float32 GPU clips take the ONE fused
        launch (ada_ops.ada_warp: no padded / over-sampled intermediates in memory, no host read of the margins); under a gradient its
        backward is the gather-form adjoint (lvg_ada_warp_adjoint: 3 ms against 11 ms for the backward of the composition on 16 x 24
        planes of 144 x 256, profiles/r03_ada_bench.log), to any order. LVG_ADA_WARP_GRAD=composed restores the composition for clips
        that carry a gradient. Everything else (CPU, other dtypes) takes the composition above."""
        margins = self._warp_margins(g_inv, x.shape[3], x.shape[2])
        needs_grad = torch.is_grad_enabled() and x.requires_grad
        if ada_ops.warp_supported(x, self.Hz_geom) and (not needs_grad or WARP_GRAD == 'adjoint'):
            return ada_ops.ada_warp(x, g_inv, margins, self.Hz_geom)
        return self._warp_composed(x, g_inv, [int(v) for v in margins.tolist()])

    # -- stage 2: colour ----------------------------------------------------------------------------

    def _colour_matrix(self, n: int, channels: int, dev, q) -> Optional[torch.Tensor]:
        eye = torch.eye(4, device=dev)
        c_mat = None

        def push(m):
            nonlocal c_mat
            c_mat = m if c_mat is None else m @ c_mat

        luma = _cached([1 / math.sqrt(3)] * 3 + [0], dev)
        if self.brightness > 0:
            b = torch.randn([n], device=dev) * self.brightness_std
            b = self._gate(self.brightness, [n], b, 0)
            if q is not None:
                b = torch.full_like(b, float(self._gauss_q(q) * self.brightness_std))
            push(shift3(b, b, b))
        if self.contrast > 0:
            c = torch.exp2(torch.randn([n], device=dev) * self.contrast_std)
            c = self._gate(self.contrast, [n], c, 1)
            if q is not None:
                c = torch.full_like(c, float(torch.exp2(self._gauss_q(q) * self.contrast_std)))
            push(zoom3(c))
        if self.lumaflip > 0:
            i = torch.floor(torch.rand([n, 1, 1], device=dev) * 2)
            i = self._gate(self.lumaflip, [n, 1, 1], i, 0)
            if q is not None:
                i = torch.full_like(i, float(torch.floor(q * 2)))
            push(eye - 2 * torch.outer(luma, luma) * i)                        # Householder reflection about luma
        if self.hue > 0 and channels > 1:
            th = (torch.rand([n], device=dev) * 2 - 1) * math.pi * self.hue_max
            th = self._gate(self.hue, [n], th, 0)
            if q is not None:
                th = torch.full_like(th, float((q * 2 - 1) * math.pi * self.hue_max))
            push(turn3(luma, th))
        if self.saturation > 0 and channels > 1:
            s = torch.exp2(torch.randn([n, 1, 1], device=dev) * self.saturation_std)
            s = self._gate(self.saturation, [n, 1, 1], s, 1)
            if q is not None:
                s = torch.full_like(s, float(torch.exp2(self._gauss_q(q) * self.saturation_std)))
            ll = torch.outer(luma, luma)
            push(ll + (eye - ll) * s)
        return c_mat

    # -- stage 3: band filter -----------------------------------------------------------------------

    def _band_gains(self, n: int, dev, q) -> torch.Tensor:
        bands = self.Hz_fbank.shape[0]
        assert len(self.imgfilter_bands) == bands
        power = _cached([10 / 13, 1 / 13, 1 / 13, 1 / 13], dev)                         # expected 1/f power per band
        gains = torch.ones([n, bands], device=dev)
        for i, strength in enumerate(self.imgfilter_bands):
            t_i = torch.exp2(torch.randn([n], device=dev) * self.imgfilter_std)
            t_i = self._gate(self.imgfilter * strength, [n], t_i, 1)
            if q is not None:
                t_i = torch.full_like(t_i, float(torch.exp2(self._gauss_q(q) * self.imgfilter_std))) if strength > 0 else torch.ones_like(t_i)
            t = torch.ones([n, bands], device=dev)
            t[:, i] = t_i
            gains = gains * (t / (power * t.square()).sum(dim=-1, keepdim=True).sqrt())
        return gains

    # -----------------------------------------------------------------------------------------------

    def forward(self, videos: torch.Tensor, debug_percentile=None) -> torch.Tensor:
        assert isinstance(videos, torch.Tensor) and videos.ndim == 5
        n, channels, frames, height, width = videos.shape
        dev = videos.device
        q = None if debug_percentile is None else torch.as_tensor(debug_percentile, dtype=torch.float32, device=dev)

        g_inv = self._inverse_affine(n, width, height, dev, q)
        if g_inv is not None:
            videos = self._warp(videos.reshape(n, channels * frames, height, width), g_inv)

        c_mat = self._colour_matrix(n, channels, dev, q)
        # GPU float32 RGB clips: colour matrix, noise and cutout are applied together at the end by ONE pass over the pixels
        # (ada_ops.ada_colour); the random numbers are still drawn here, in the reference's order. With the band filter on, the colour
        # matrix has to be applied before it (the filter sits between the two in the reference).
        fused = ada_ops.colour_supported(videos.reshape(n, channels, frames, height, width))
        if c_mat is not None and not (fused and self.imgfilter == 0):
            flat = videos.reshape(n, channels, frames * height * width)
            if channels == 3:
                flat = c_mat[:, :3, :3] @ flat + c_mat[:, :3, 3:]
            elif channels == 1:
                row = c_mat[:, :3, :].mean(dim=1, keepdim=True)
                flat = flat * row[:, :, :3].sum(dim=2, keepdim=True) + row[:, :, 3:]
            else:
                raise ValueError('Image must be RGB (3 channels) or L (1 channel)')
            videos = flat
            c_mat = None

        if self.imgfilter > 0:
            taps = self._band_gains(n, dev, q) @ self.Hz_fbank                                  # [N, taps]
            taps = taps.unsqueeze(1).repeat(1, channels, 1).reshape(n * channels, 1, -1)
            half = self.Hz_fbank.shape[1] // 2
            # one image of N*C*T channels, depthwise: every (sample, colour) applies its own taps to its T frames.
            # (The reference passes N*C filters for N*C*T channels, :392-:393, which only type-checks for T = 1;
            # the shipped configurations leave imgfilter off.)
            x = videos.reshape(1, n * channels * frames, height, width)
            x = F.pad(x, [half, half, half, half], mode='reflect')
            x = conv2d_gradfix.conv2d(input=x, weight=self._per_frame(taps.unsqueeze(2), frames), groups=n * channels * frames)
            x = conv2d_gradfix.conv2d(input=x, weight=self._per_frame(taps.unsqueeze(3), frames), groups=n * channels * frames)
            videos = x

        videos = videos.reshape(n, channels * frames, height, width)

        sigma = noise = None
        if self.noise > 0:
            sigma = torch.randn([n, 1, 1, 1], device=dev).abs() * self.noise_std
            sigma = self._gate(self.noise, [n, 1, 1, 1], sigma, 0)
            if q is not None:
                sigma = torch.full_like(sigma, float(torch.erfinv(q) * self.noise_std))
            noise = torch.randn([n, channels * frames, height, width], device=dev)
            if not fused:
                videos = videos + noise * sigma

        size = centre = None
        if self.cutout > 0:
            size = torch.full([n, 2, 1, 1, 1], self.cutout_size, device=dev)
            size = self._gate(self.cutout, [n, 1, 1, 1, 1], size, 0)
            centre = torch.rand([n, 2, 1, 1, 1], device=dev)
            if q is not None:
                size = torch.full_like(size, self.cutout_size)
                centre = torch.full_like(centre, float(q))
            if not fused:
                xs = (torch.arange(width, device=dev).reshape(1, 1, 1, -1) + 0.5) / width
                ys = (torch.arange(height, device=dev).reshape(1, 1, -1, 1) + 0.5) / height
                keep = torch.logical_or((xs - centre[:, 0]).abs() >= size[:, 0] / 2, (ys - centre[:, 1]).abs() >= size[:, 1] / 2)
                videos = videos * keep.to(torch.float32)

        if fused and (c_mat is not None or noise is not None or size is not None):
            cut = None if size is None else torch.cat((centre.reshape(n, 2), size.reshape(n, 2)), dim=1)
            videos = ada_ops.ada_colour(videos.reshape(n, channels, frames, height, width), c_mat,
                                        None if noise is None else noise.reshape(n, channels, frames, height, width),
                                        None if sigma is None else sigma.reshape(n), cut)

        return videos.reshape(n, channels, frames, height, width)

    @staticmethod
    def _per_frame(weight: torch.Tensor, frames: int) -> torch.Tensor:
        """[N*C, 1, kh, kw] taps -> one copy per frame, [N*C*T, 1, kh, kw] (channel order (n c t))."""
        return weight.repeat_interleave(frames, dim=0)

    def random_temporal_filter(self, video: torch.Tensor, min_ksize: int = 2, max_ksize: int = 16, max_std: float = 1.0) -> torch.Tensor:
        """Random per-sample FIR along time: a box of random length plus zero-mean noise taps (so the
        filter sums to one); a sample takes the filtered clip where p < u, u uniform (as the reference).
        GPU clips always draw the random numbers and run the filter (no host read of p), so with p <= 0 the GPU path consumes random
        numbers where the CPU / reference path returns early: same result, different position in the random stream afterwards."""
        assert video.dim() == 5 and 2 <= min_ksize <= max_ksize
        if not video.is_cuda and self.p.item() <= 0:                 # (GPU clips: no host read of p -- the mask below covers p <= 0)
            return video
        n, dev = video.size(0), video.device
        ksize = torch.randint(2, max_ksize + 1, (n, 1, 1, 1, 1), device=dev)
        index = torch.arange(max_ksize, device=dev).reshape(1, 1, -1, 1, 1).expand(n, 1, -1, 1, 1)
        inside = (index >= (max_ksize - ksize).div(2)) * (index < (max_ksize + ksize).div(2))
        std = torch.rand(n, 1, 1, 1, 1, device=dev) * max_std
        taps = torch.randn(n, 1, max_ksize, 1, 1, device=dev) * std * inside
        taps = (1 / ksize) * inside + taps - taps.mean(dim=2, keepdim=True)
        padded = F.pad(video, (0, 0, 0, 0, max_ksize // 2, (max_ksize - 1) // 2), mode='reflect')
        filtered = F.conv3d(padded.transpose(0, 1), taps, groups=n).transpose(0, 1)
        use = (self.p < torch.rand(n, 1, 1, 1, 1, device=dev)) & (self.p > 0)
        return torch.where(use, filtered, video)
