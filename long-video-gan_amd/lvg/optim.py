"""Flat-buffer Adam (+ weight EMA) for the GAN training loops: `torch.optim.Adam(betas=(b1, b2))` of the
reference (model/video_gan_lres.py:83-90, model/video_gan_sres.py) with the generator EMA of `update_G_ema`
(:208-214) folded into the same pass.

The parameters are re-pointed at slices of ONE flat float32 buffer (in the order of `lvg.ddp.FlatGradSync`,
whose flat gradient buffer is reused), the two moments are flat buffers too, so an update is one streaming HIP
launch (`lvg_adam_step`, csrc/optim.hip: 7 floats per element, 9 with the EMA) per run of parameters with the
same update count -- normally one launch per network. Parameters whose gradient is None are skipped and keep
their own update count, like torch.optim.Adam (the reference starts every update from
zero_grad(set_to_none=True)). CPU tensors take the same arithmetic spelled with torch ops (gloo tests).

Drop-in where the trainers used torch.optim.Adam: `zero_grad` is FlatGradSync's job; `state_dict()` /
`load_state_dict()` carry the moments and update counts per parameter index."""

from typing import Iterable, List, Optional

import torch
import torch.nn as nn


def flatten_parameters(params: List[nn.Parameter]) -> torch.Tensor:
    """Move the parameters into one flat buffer (each padded to a multiple of 4 elements so that every slice starts on a
    16-byte boundary) and re-point `.data` at its slices. Returns the flat buffer."""
    assert params and all(p.dtype == torch.float32 for p in params), 'float32 parameters only'
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
    with torch.no_grad():
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
    return flat


class FlatAdam:
    def __init__(self, params: Iterable[nn.Parameter], lr: float, betas=(0.0, 0.99), eps: float = 1e-8,
                 ema_params: Optional[Iterable[nn.Parameter]] = None):
        self.params = list(params)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.flat = flatten_parameters(self.params)
        self.offsets, o = [], 0
        for p in self.params:
            self.offsets.append(o)
            o += (p.numel() + 3) // 4 * 4
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.flat_grad = torch.zeros_like(self.flat)        # used when a gradient does not already live in a flat buffer
        self.steps = [0] * len(self.params)
        self.ema_flat = None
        if ema_params is not None:
            ema_params = list(ema_params)
            assert len(ema_params) == len(self.params) and all(a.shape == b.shape for a, b in zip(ema_params, self.params))
            self.ema_flat = flatten_parameters(ema_params)

    # --------------------------------------------------------------------------------------------
    def _range_update(self, lo: int, hi: int, grad: torch.Tensor, step: int, ema_weight: Optional[float]) -> None:
        p, m, v = self.flat[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi]
        ema = self.ema_flat[lo:hi] if (self.ema_flat is not None and ema_weight is not None) else None
        b1, b2 = self.betas
        if p.device.type == 'cuda':
            from torch_utils.ops import _hip
            with torch.cuda.device(p.device):
                rc = _hip.lib().lvg_adam_step(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), None if ema is None else ema.data_ptr(),
                                              hi - lo, self.lr, b1, b2, self.eps, step, 0.0 if ema_weight is None else float(ema_weight),
                                              _hip.stream(p.device))
            _hip.check(rc, 'adam_step')
            return
        m.lerp_(grad, 1 - b1)
        v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = v.sqrt().div_(bc2 ** 0.5).add_(self.eps)
        p.addcdiv_(m, denom, value=-self.lr / bc1)
        if ema is not None:
            ema.lerp_(p, float(ema_weight))

    @torch.no_grad()
    def step(self, ema_weight: Optional[float] = None) -> None:
        """One Adam update of every parameter that has a gradient; with `ema_weight` (= 1 - ema_beta) the EMA copy of EVERY
        parameter moves towards the current values: in the same pass for the updated ones, by a lerp of their (unchanged) slices for
        parameters without a gradient -- the reference's EMA lerps all parameters every step (video_gan_lres.py update_G_ema)."""
        # gather runs of consecutive parameters that (a) have a gradient, (b) share the update count, and (c) whose gradients
        # are consecutive slices of one flat buffer with the same spacing as the parameters
        runs, cur = [], None
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                cur = None
                if self.ema_flat is not None and ema_weight is not None:
                    lo = self.offsets[i]
                    self.ema_flat[lo:lo + p.numel()].lerp_(self.flat[lo:lo + p.numel()], float(ema_weight))
                continue
            assert g.dtype == torch.float32 and g.is_contiguous(), 'float32 contiguous gradients'
            self.steps[i] += 1
            lo, hi = self.offsets[i], self.offsets[i] + p.numel()
            gp = g.data_ptr()
            if cur is not None and cur['step'] == self.steps[i] and cur['gptr'] is not None and gp == cur['gptr'] + (lo - cur['lo']) * 4 \
                    and g.untyped_storage().data_ptr() == cur['storage']:
                cur['hi'] = hi
            else:
                cur = dict(lo=lo, hi=hi, step=self.steps[i], gptr=gp, storage=g.untyped_storage().data_ptr(), first=i)
                runs.append(cur)
        for r in runs:
            lo, hi = r['lo'], r['hi']
            first = self.params[r['first']]
            n = hi - lo
            g0 = first.grad
            # the run's gradients as ONE flat tensor of n elements starting at the first gradient
            if g0.storage_offset() + n <= g0.untyped_storage().nbytes() // 4 and (g0.data_ptr() % 16 == 0):
                grad = torch.as_strided(g0, (n,), (1,), g0.storage_offset())
            else:                                                   # scattered / unaligned: copy into the staging buffer
                grad = self.flat_grad[lo:hi]
                for i in range(r['first'], len(self.params)):
                    o = self.offsets[i]
                    if o >= hi:
                        break
                    if self.params[i].grad is not None:
                        grad[o - lo:o - lo + self.params[i].numel()].copy_(self.params[i].grad.reshape(-1))
            self._range_update(lo, hi, grad, r['step'], ema_weight)

    # --------------------------------------------------------------------------------------------
    def state_dict(self) -> dict:
        return dict(lr=self.lr, betas=self.betas, eps=self.eps, steps=list(self.steps),
                    exp_avg=[self.exp_avg[o:o + p.numel()].view(p.shape).clone() for p, o in zip(self.params, self.offsets)],
                    exp_avg_sq=[self.exp_avg_sq[o:o + p.numel()].view(p.shape).clone() for p, o in zip(self.params, self.offsets)])

    def load_state_dict(self, state: dict) -> None:
        self.lr, self.betas, self.eps = float(state['lr']), tuple(state['betas']), float(state['eps'])
        self.steps = list(state['steps'])
        with torch.no_grad():
            for p, o, m, v in zip(self.params, self.offsets, state['exp_avg'], state['exp_avg_sq']):
                self.exp_avg[o:o + p.numel()].copy_(m.reshape(-1))
                self.exp_avg_sq[o:o + p.numel()].copy_(v.reshape(-1))
