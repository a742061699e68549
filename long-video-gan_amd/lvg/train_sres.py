"""Training-step body of the super-resolution GAN on synthetic data (reference
model/video_gan_sres.py:126-294 + train_sres.py:236-264): the generator maps a low-resolution
clip with +-`temporal_context` extra frames to high-resolution frames; the discriminator sees the
(bilinearly upsampled) low-resolution clip stacked with the high-resolution one, both pushed
through the SAME ADA transform; R1 on the high-resolution input; the ADA probability follows the
sign of the real logits. One process per GPU; gradients are exchanged with `lvg.ddp` over RCCL.
The reference's dataset / W&B / checkpoint plumbing is out of scope."""

import copy
import math
import os
from typing import Optional

import torch
import torch.nn.functional as F

from torch_utils.ops import conv2d_gradfix, grid_sample_gradfix

from . import ddp
from .optim import FlatAdam
from .phase_graphs import PhaseGraphs
from .ada_augment import AugmentPipe
from .models import sres

R1_CLOSED_NODES = os.environ.get('LVG_SRES_R1_CLOSED_NODES', '1') != '0'
R1_GRAPH = os.environ.get('LVG_R1_GRAPH', '1') != '0'       # graph mode: the R1 pass replayed from a hipGraph too (0: eager, as before round 6)


class SuperResTrainer:
    def __init__(self, seq_length: int = 8, temporal_context: int = 4, lr_height: int = 36, lr_width: int = 64,
                 hr_height: int = 144, hr_width: int = 256, channels: int = 3, device='cuda',
                 compute_dtype: torch.dtype = torch.float16,
                 G_lrate: float = 0.003, G_beta2: float = 0.99, G_ema_beta: float = 0.99985, G_ema_warmup_steps: int = 25000,
                 G_magnitude_ema_beta: float = 0.999, G_grad_accum: int = 1,
                 D_lrate: float = 0.002, D_beta2: float = 0.99, D_grad_accum: int = 1,
                 r1_gamma: float = 1.0, lr_cond_prob: float = 0.1,
                 augment_p_init: float = 0.0, augment_p_max: float = 0.5, augment_p_update_rate: float = 0.000125,
                 augment_real_sign_target: Optional[float] = 0.6, augment_kwargs: Optional[dict] = None,
                 in_augment_p: float = 0.5, in_augment_strength: float = 8.0, overlap_grad_sync: bool = True,
                 G_kwargs: Optional[dict] = None, D_kwargs: Optional[dict] = None, with_ema: bool = True, use_graphs: bool = False,
                 G_warmup_steps: int = 0, D_warmup_steps: int = 0):
        self.G_lrate, self.D_lrate, self.G_warmup_steps, self.D_warmup_steps = G_lrate, D_lrate, G_warmup_steps, D_warmup_steps
        conv2d_gradfix.enabled = True            # as train_sres.py:81-82: R1 differentiates twice through
        grid_sample_gradfix.enabled = True       # the resampling convs and ADA's grid_sample
        self.seq_length, self.temporal_context, self.channels = seq_length, temporal_context, channels
        self.context_seq_length = seq_length + 2 * temporal_context
        self.lr_size, self.hr_size = (lr_height, lr_width), (hr_height, hr_width)
        self.device = torch.device(device)
        self.G_magnitude_ema_beta, self.G_ema_beta, self.G_ema_warmup_steps = G_magnitude_ema_beta, G_ema_beta, G_ema_warmup_steps
        self.G_grad_accum, self.D_grad_accum = G_grad_accum, D_grad_accum
        self.r1_gamma, self.lr_cond_prob = r1_gamma, lr_cond_prob
        self.augment_p_max, self.augment_p_update_rate, self.augment_real_sign_target = augment_p_max, augment_p_update_rate, augment_real_sign_target

        sizes = dict(hr_height=hr_height, hr_width=hr_width, lr_height=lr_height, lr_width=lr_width)
        self.G_init_kwargs = dict(temporal_context=temporal_context, compute_dtype=compute_dtype, **sizes, **(G_kwargs or {}))      # for save_G_ema / load_G
        self.G = sres.VideoGenerator(**self.G_init_kwargs)
        self.D = sres.VideoDiscriminator(channels=channels, seq_length=seq_length, compute_dtype=compute_dtype, **sizes, **(D_kwargs or {}))
        for net in (self.G, self.D):
            net.to(self.device).requires_grad_(False).train()
            ddp.broadcast_module(net, src=0)
        self.G_ema = copy.deepcopy(self.G).eval() if with_ema else None
        # One flat buffer per network for parameters, moments (FlatAdam) and gradients (FlatGradSync, same slice layout): an
        # optimizer step is one streaming launch, with the generator EMA of the parameters folded in (update_G_ema keeps the buffers).
        self.G_opt = FlatAdam(self.G.parameters(), lr=G_lrate, betas=(0.0, G_beta2),
                              ema_params=self.G_ema.parameters() if self.G_ema is not None else None)
        self.D_opt = FlatAdam(self.D.parameters(), lr=D_lrate, betas=(0.0, D_beta2))
        self._step = 0
        # use_graphs: the compute of a micro-batch of update_G / update_D and the fake generation are captured once per shape into hipGraphs
        # and replayed, at any world size (every random draw of this trainer is made on the device; the statistics a generator pass averages
        # over ranks -- w_avg, the layers' magnitude EMAs -- are exchanged in one all-reduce after its replay: lvg.phase_graphs); the gradient
        # exchange -- an RCCL collective cannot be captured -- follows a phase's replays, the optimizer steps, R1 (exchange overlapped with
        # its backward pass) and the ADA update stay eager. use_graphs='segmented': the same protocol without capturing (any device; tests).
        self.use_graphs = bool(use_graphs) and (self.device.type == 'cuda' or use_graphs == 'segmented')
        self._capture = self.device.type == 'cuda' and use_graphs != 'segmented'
        self._static = {}
        self.G_sync = ddp.FlatGradSync(self.G.parameters(), overlap=overlap_grad_sync)
        self.D_sync = ddp.FlatGradSync(self.D.parameters(), overlap=overlap_grad_sync)

        # discriminator-side ADA (probability adapted from the sign of the real logits)
        self.augment = None
        if augment_p_init > 0 or augment_real_sign_target is not None:
            self.augment = AugmentPipe(**(augment_kwargs or {})).to(self.device).requires_grad_(False).train()
            self.augment.p.fill_(augment_p_init)
        self._real_sign_sum = torch.zeros(2, device=self.device)          # [sum of signs, count] since the last update_ada
        # conditioning-side augmentation: mild geometric jitter + noise on the low-resolution input
        self.in_augment = None
        if in_augment_strength > 0 and in_augment_p > 0:
            k = in_augment_strength
            self.in_augment = AugmentPipe(scale=1, scale_std=0.01 * k, rotate=1, rotate_max=0.002 * k, aniso=1, aniso_std=0.01 * k,
                                          xfrac=1, xfrac_std=0.002 * k, noise=1, noise_std=0.01 * k)
            self.in_augment.to(self.device).requires_grad_(False).train()
            self.in_augment.p.fill_(in_augment_p)
        # what a phase may update in place: the gradient buffers, the sign statistic, every buffer of the networks and of the augmentation
        # pipelines (derived, not listed by hand: ADVICE r04)
        self._phase_graphs = PhaseGraphs(lambda: (self.G_sync.flat, self.D_sync.flat, self._real_sign_sum, *self.G.buffers(), *self.D.buffers(),
                                                  *(self.augment.buffers() if self.augment is not None else ()),
                                                  *(self.in_augment.buffers() if self.in_augment is not None else ())),
                                         syncs=(self.G_sync, self.D_sync), capture=self._capture)

    def _static_like(self, name: str, t: torch.Tensor) -> torch.Tensor:
        """A persistent input buffer of a captured phase, filled with `t`."""
        key = (name, tuple(t.shape), t.dtype)
        buf = self._static.get(key)
        if buf is None:
            buf = self._static[key] = torch.empty_like(t, memory_format=torch.contiguous_format)
        buf.copy_(t)
        return buf

    # ------------------------------------------------------------------------------------------
    def crop_to_seq_length(self, video: torch.Tensor) -> torch.Tensor:
        t0 = (video.size(2) - self.seq_length) // 2
        return video[:, :, t0:t0 + self.seq_length]

    def _jitter(self, lr_video: torch.Tensor) -> torch.Tensor:
        return lr_video if self.in_augment is None else self.in_augment(lr_video)

    def run_D(self, lr_video: torch.Tensor, hr_video: torch.Tensor) -> torch.Tensor:
        """lr [N, C, T, h, w] and hr [N, C, T, H, W] -> logits [N, 1]. Both clips go through ONE augmentation
        call (stacked along time) so they receive the same geometric / colour transform."""
        assert lr_video.shape[2:] == (self.seq_length, *self.lr_size) and hr_video.shape[2:] == (self.seq_length, *self.hr_size)
        pair = torch.cat((self.D.upsample(lr_video), hr_video), dim=2)
        if self.augment is not None:
            pair = self.augment(pair)
        lr_up, hr_video = pair.chunk(2, dim=2)
        if self.lr_cond_prob < 1:                                           # conditioning dropout, per sample
            keep = torch.rand(lr_up.size(0), 1, 1, 1, 1, device=lr_up.device) < self.lr_cond_prob
            lr_up = lr_up * keep.to(lr_up.dtype)
        return self.D(lr_up, hr_video)

    # ------------------------------------------------------------------------------------------
    def _ema_beta(self, step: int) -> float:
        halflife = math.log(self.G_ema_beta, 0.5) * (self.G_ema_warmup_steps + 1) / (step + 1)
        return min(0.5 ** halflife, self.G_ema_beta)

    def update_lrates(self, step: int) -> None:
        """Linear learning-rate warm-up of both optimizers (reference video_gan_sres.py:140-146)."""
        self.G_opt.lr = self.G_lrate * min((step + 1) / (self.G_warmup_steps + 1), 1.0)
        self.D_opt.lr = self.D_lrate * min((step + 1) / (self.D_warmup_steps + 1), 1.0)

    def update_G(self, lr_video: torch.Tensor, ema_step: Optional[int] = None) -> None:
        """`ema_step`: fold this iteration's generator-EMA update of the parameters into the optimizer pass."""
        assert lr_video.shape[1:] == (self.channels, self.context_seq_length, *self.lr_size)
        assert lr_video.size(0) % self.G_grad_accum == 0
        lr_video = self._jitter(lr_video)
        self.G.requires_grad_(True)
        self.G_sync.zero()
        chunks = lr_video.chunk(self.G_grad_accum)
        for k, lr in enumerate(chunks):
            if k == len(chunks) - 1 and self.G_sync.overlap and not self.use_graphs:
                self.G_sync.arm()                                          # (replayed phases run no hooks: their exchange follows them)
            if self.use_graphs:
                lr_in = self._static_like('G.lr', lr)
                self._phase_graphs.replay(('G', tuple(lr.shape)),
                                          lambda: F.softplus(-self.run_D(self.crop_to_seq_length(lr_in), self.G(lr_in))).mean().backward())
                continue
            logits = self.run_D(self.crop_to_seq_length(lr), self.G(lr))
            F.softplus(-logits).mean().backward()
        self.G.requires_grad_(False)
        self.G_sync.finish(gain=1 / self.G_grad_accum)
        fuse = ema_step is not None and self.G_ema is not None
        self.G_opt.step(ema_weight=(1.0 - self._ema_beta(ema_step)) if fuse else None)
        self._ema_fused_for = ema_step if fuse else None

    def update_D(self, fake_lr_video: torch.Tensor, real_lr_video: torch.Tensor, real_hr_video: torch.Tensor) -> None:
        assert fake_lr_video.size(0) == real_lr_video.size(0) == real_hr_video.size(0)
        assert fake_lr_video.size(0) % self.D_grad_accum == 0
        fake_lr_video, real_lr_video = self._jitter(fake_lr_video), self._jitter(real_lr_video)
        if self.use_graphs:
            return self._update_D_graphs(fake_lr_video, real_lr_video, real_hr_video)
        fake_hr_video = self.G(fake_lr_video, magnitude_ema_beta=self.G_magnitude_ema_beta)    # G frozen: no graph is built
        fake_lr_video, real_lr_video = self.crop_to_seq_length(fake_lr_video), self.crop_to_seq_length(real_lr_video)
        self.D.requires_grad_(True)
        self.D_sync.zero()
        parts = [t.chunk(self.D_grad_accum) for t in (fake_lr_video, fake_hr_video, real_lr_video, real_hr_video)]
        for k, (f_lr, f_hr, r_lr, r_hr) in enumerate(zip(*parts)):
            F.softplus(self.run_D(f_lr, f_hr)).mean().backward()
            real_logits = self.run_D(r_lr, r_hr)
            if k == self.D_grad_accum - 1 and self.D_sync.overlap:
                self.D_sync.arm()
            F.softplus(-real_logits).mean().backward()
            with torch.no_grad():
                self._real_sign_sum += torch.stack((real_logits.sign().sum(), torch.full((), float(real_logits.numel()), dtype=real_logits.dtype, device=real_logits.device)))
        self.D.requires_grad_(False)
        self.D_sync.finish(gain=1 / self.D_grad_accum)
        self.D_opt.step()

    def _update_D_graphs(self, fake_lr_video: torch.Tensor, real_lr_video: torch.Tensor, real_hr_video: torch.Tensor) -> None:
        """update_D (inputs already jittered) with the fake generation and every (fake, real) micro-batch replayed from graphs."""
        gen_in = self._static_like('D.gen_lr', fake_lr_video)
        out = self._static.setdefault(('D.gen_out', tuple(fake_lr_video.shape)), {})

        def generate():
            out['hr'] = self.G(gen_in, magnitude_ema_beta=self.G_magnitude_ema_beta)
        self._phase_graphs.replay(('Dgen', tuple(fake_lr_video.shape)), generate)
        fake_hr_video = out['hr']
        fake_lr_video, real_lr_video = self.crop_to_seq_length(fake_lr_video), self.crop_to_seq_length(real_lr_video)
        self.D.requires_grad_(True)
        self.D_sync.zero()
        parts = [t.chunk(self.D_grad_accum) for t in (fake_lr_video, fake_hr_video, real_lr_video, real_hr_video)]
        for f_lr, f_hr, r_lr, r_hr in zip(*parts):
            ins = [self._static_like(n, t) for n, t in (('D.f_lr', f_lr), ('D.f_hr', f_hr), ('D.r_lr', r_lr), ('D.r_hr', r_hr))]

            def passes():
                F.softplus(self.run_D(ins[0], ins[1])).mean().backward()
                real_logits = self.run_D(ins[2], ins[3])
                F.softplus(-real_logits).mean().backward()
                with torch.no_grad():
                    self._real_sign_sum += torch.stack((real_logits.sign().sum(), torch.full((), float(real_logits.numel()), dtype=real_logits.dtype, device=real_logits.device)))
            self._phase_graphs.replay(('D', tuple(f_lr.shape)), passes)
        self.D.requires_grad_(False)
        self.D_sync.finish(gain=1 / self.D_grad_accum)
        self.D_opt.step()

    def update_r1(self, lr_video: torch.Tensor, hr_video: torch.Tensor, gain: float = 1.0) -> None:
        assert lr_video.size(0) == hr_video.size(0) and lr_video.size(0) % self.D_grad_accum == 0
        lr_video = self._jitter(lr_video)
        self.D.requires_grad_(True)
        self.D_sync.zero()
        pairs = list(zip(lr_video.chunk(self.D_grad_accum), hr_video.chunk(self.D_grad_accum)))
        if self.use_graphs and R1_GRAPH:
            # round 6: ~1000 launches for ~10 ms of device time per micro-batch: replayed from a graph like the other phases
            for lr, hr in pairs:
                ins = [self._static_like('R1.lr', lr), self._static_like('R1.hr', hr)]

                def penalty_pass():
                    h = ins[1].detach().requires_grad_(True)
                    with conv2d_gradfix.closed_nodes(R1_CLOSED_NODES):
                        logits = self.run_D(ins[0], h)
                    (grad,) = torch.autograd.grad(outputs=[logits.sum()], inputs=[h], create_graph=True)
                    (grad.square().sum(dim=(1, 2, 3, 4)) * (self.r1_gamma / 2)).mean().backward()
                self._phase_graphs.replay(('R1', tuple(lr.shape)), penalty_pass, optional=True)
            pairs = []
        for k, (lr, hr) in enumerate(pairs):
            hr = hr.detach().requires_grad_(True)
            # the discriminator's dense convolutions as nodes closed under differentiation: the second-order pass then consists of ordinary
            # forward / backward-data / backward-weight calls (conv2d_gradfix.closed_nodes; LVG_SRES_R1_CLOSED_NODES=0: the library's own graph)
            with conv2d_gradfix.closed_nodes(R1_CLOSED_NODES):
                logits = self.run_D(lr, hr)
            (grad,) = torch.autograd.grad(outputs=[logits.sum()], inputs=[hr], create_graph=True)
            penalty = grad.square().sum(dim=(1, 2, 3, 4))
            if k == len(pairs) - 1 and self.D_sync.overlap:
                self.D_sync.arm()
            (penalty * (self.r1_gamma / 2)).mean().backward()
        self.D.requires_grad_(False)
        self.D_sync.finish(gain=gain / self.D_grad_accum)
        self.D_opt.step()

    @torch.no_grad()
    def update_ada(self, gain: float = 1.0) -> None:
        """p += rate * gain * sign(mean sign of the real logits since the last call - target), clamped to [0, p_max]."""
        if self.augment is None or self.augment_real_sign_target is None:
            return
        stats = ddp.sharded_all_mean(self._real_sign_sum.clone()) if torch.distributed.is_initialized() else self._real_sign_sum
        # on the device, no host read: step = copysign(rate, mean sign - target) * gain where anything was counted, else 0
        # (the reference reads the mean back through its statistics collector, video_gan_sres.py:256-262; copysign keeps its sign rule at 0)
        diff = stats[0] / stats[1].clamp(min=1) - self.augment_real_sign_target
        step = torch.copysign(torch.full_like(diff, self.augment_p_update_rate * gain), diff) * (stats[1] > 0)
        self.augment.p.add_(step.to(self.augment.p.dtype)).clamp_(0, self.augment_p_max)
        self._real_sign_sum.zero_()

    @torch.no_grad()
    def update_G_ema(self, step: int) -> None:
        if self.G_ema is None:
            return
        beta = self._ema_beta(step)
        src, dst = list(self.G.buffers()), list(self.G_ema.buffers())
        if getattr(self, '_ema_fused_for', None) != step:                 # parameters not already done inside update_G
            src, dst = list(self.G.parameters()) + src, list(self.G_ema.parameters()) + dst
        self._ema_fused_for = None
        torch._foreach_lerp_(dst, src, 1.0 - beta)

    def train_step(self, step: int, lr_video: torch.Tensor, hr_video: torch.Tensor, r1_interval: int = 16, ada_interval: int = 4) -> None:
        """One iteration of the reference loop (train_sres.py:241-264) on one (lr with context, hr) batch."""
        self.update_lrates(step)
        self.update_G(lr_video, ema_step=step)
        self.update_D(lr_video, lr_video, hr_video)
        if r1_interval > 0 and step % r1_interval == 0:
            self.update_r1(self.crop_to_seq_length(lr_video), hr_video, gain=r1_interval)
        if ada_interval > 0 and step % ada_interval == 0:
            self.update_ada(gain=ada_interval)
        self.update_G_ema(step)
