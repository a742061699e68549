"""Temporal noise filter bank of the low-resolution generator on the float32 matrix cores (csrc/noise_bank.hip).

Replaces the grouped `F.conv1d` of the reference's `BlurredNoise.blur` (model/generator_lres.py:378-388) for float32 GPU tensors:
every noise row is correlated with each of the bank's right-aligned low-pass filters. The bank is a staircase (125 .. 5000 non-zero
taps in rows of 5000), so filters are processed in groups of 32 that share a tap count and the kernel walks only those taps; the
sliding-window operand is read straight out of the noise row (no [rows, frames, taps] window matrix in memory).

`pack_bank` builds the kernel's operand form from the bank AS IT IS (tap counts are read off its zeros), once per bank version."""

from typing import Optional, Tuple

import numpy as np
import torch

from . import _hip

_GROUP = 32          # filters per group (one MFMA tile column block)
_PAIR_ALIGN = 64     # pairs per group are split over four waves in double blocks of eight
_SPARE_LINES = 8     # zero lines behind the last group (the kernel prefetches one block ahead unconditionally)


def plan_groups(bank: np.ndarray) -> Tuple[np.ndarray, int]:
    """bank [F, K] -> (pairs per group [G], K): group g covers filters 32 g .. 32 g + 31 and needs the last 2 * pairs[g] taps."""
    f, k = bank.shape
    nz = bank != 0
    first = np.where(nz.any(axis=1), nz.argmax(axis=1), k)             # index of the first non-zero tap of every filter (k: an all-zero row)
    taps = k - first
    groups = (f + _GROUP - 1) // _GROUP
    pairs = np.zeros(groups, dtype=np.int64)
    for g in range(groups):
        longest = int(taps[g * _GROUP:(g + 1) * _GROUP].max())
        pairs[g] = max(_PAIR_ALIGN, -(-((longest + 1) // 2) // _PAIR_ALIGN) * _PAIR_ALIGN)
    return pairs, k


def pack_bank(bank: torch.Tensor):
    """bank [F, K] float32 (any device) -> (bankP [sum pairs + 8 spare zero lines, 64] float32, pairOff [G + 1] int32, max pairs) on the bank's device.
    bankP[(pairOff[g] + p), lane] = bank[32 g + lane % 32][K - 2 pairs[g] + 2 p + lane // 32], zero outside the bank. Reads the bank on the host."""
    host = bank.detach().to('cpu', torch.float32).numpy()
    f, k = host.shape
    pairs, _ = plan_groups(host)
    off = np.concatenate([[0], np.cumsum(pairs)]).astype(np.int32)
    packed = np.zeros((int(off[-1]) + _SPARE_LINES, 64), dtype=np.float32)
    lane = np.arange(64)
    for g, n in enumerate(pairs):
        filt = g * _GROUP + lane % _GROUP                                # [64]
        tap = (k - 2 * int(n)) + 2 * np.arange(int(n))[:, None] + (lane // _GROUP)[None, :]      # [n, 64]
        ok = (filt[None, :] < f) & (tap >= 0) & (tap < k)
        vals = host[np.minimum(filt, f - 1)[None, :].repeat(int(n), 0), np.clip(tap, 0, k - 1)]
        packed[off[g]:off[g + 1]] = np.where(ok, vals, 0.0)
    dev = bank.device
    return torch.from_numpy(packed).to(dev), torch.from_numpy(off).to(dev), int(pairs.max())


class PackedBank:
    """Cache of `pack_bank` keyed by the bank tensor's storage and version (a `load_state_dict` into the buffer bumps the version).

    The packed tensors are baked into any hipGraph that captured a forward pass, so they are NEVER replaced while their layout still fits: a repack
    after a version bump is copied INTO the existing tensors (trainers that capture phases restore every buffer with `copy_` after their eager
    warm-up pass, which bumps the version of this constant buffer -- replacing the tensors there freed memory an earlier capture still read:
    a GPU memory fault on its next replay). Tensors of a layout that no longer fits are retired, not freed. While a capture is running a version
    bump alone does not repack (packing reads the bank on the host)."""

    def __init__(self):
        self.where = None
        self.version = None
        self.value = None
        self.snapshot = None      # the bank values the packed tensors were built from (same device): a version bump with equal values costs one comparison, not a host repack
        self._retired = []

    def get(self, bank: torch.Tensor):
        where = (bank.data_ptr(), bank.device, tuple(bank.shape))
        capturing = bank.is_cuda and torch.cuda.is_current_stream_capturing()
        if self.value is not None and where == self.where and (bank._version == self.version or capturing):
            return self.value
        assert not capturing, 'noise_bank: the bank has to be packed (a host read) before the forward pass is captured into a graph: run one eager pass first'
        if self.value is not None and self.snapshot is not None and self.snapshot.device == bank.device and self.snapshot.shape == bank.shape \
                and self.value[0].device == bank.device and torch.equal(self.snapshot, bank):
            # in-place writes of the same values (the generator EMA lerps every buffer of G_ema towards G's equal constant each step, a
            # trainer's roll-back copies the buffer onto itself): nothing to repack
            self.where, self.version = where, bank._version
            return self.value
        new = pack_bank(bank)
        old = self.value
        if old is not None and old[0].device == new[0].device and old[0].shape == new[0].shape and old[1].shape == new[1].shape and old[2] == new[2]:
            old[0].copy_(new[0])
            old[1].copy_(new[1])
        else:
            if old is not None:
                self._retired.append(old)
            self.value = new
        self.where, self.version = where, bank._version
        self.snapshot = bank.detach().clone()
        return self.value


def supported(noise: torch.Tensor, bank: torch.Tensor) -> bool:
    return (noise.is_cuda and noise.dtype == torch.float32 and bank.dtype == torch.float32 and noise.dim() == 2 and bank.dim() == 2
            and not noise.requires_grad and not bank.requires_grad and noise.shape[1] >= bank.shape[1] and bank.shape[1] <= 8000)


def noise_filter_bank(noise: torch.Tensor, bank: torch.Tensor, packed, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """noise [R, L] float32 GPU, bank [F, K], packed = PackedBank.get(bank) / pack_bank(bank), scale [F] float32 or None ->
    [R, F, L - K + 1] float32: out[r, f, t] = scale[f] * sum_k noise[r, t + k] * bank[f, k]. Raises if the library is missing."""
    assert supported(noise, bank)
    bank_p, pair_off, max_pairs = packed
    rows, length = noise.shape
    filters, taps = bank.shape
    frames = length - taps + 1
    noise = noise.contiguous()
    out = torch.empty((rows, filters, frames), dtype=torch.float32, device=noise.device)
    sc = None if scale is None else scale.to(torch.float32).reshape(-1).contiguous()
    assert sc is None or sc.numel() == filters
    with torch.cuda.device(noise.device):                                    # (the launch goes to the CURRENT HIP device: make it the tensor's)
        rc = _hip.lib().lvg_noise_filter_bank(noise.data_ptr(), bank_p.data_ptr(), pair_off.data_ptr(), None if sc is None else sc.data_ptr(), out.data_ptr(),
                                              rows, length, frames, filters, taps, pair_off.numel() - 1, max_pairs, _hip.stream(noise.device))
    _hip.check(rc, 'lvg_noise_filter_bank')
    return out
