"""Training-step body of the low-resolution GAN on synthetic data (reference
model/video_gan_lres.py:100-214 + train_lres.py:216-230): non-saturating logistic loss, gradient
accumulation, R1 every `r1_interval` steps, generator EMA. One process per GPU; gradients are
exchanged with `lvg.ddp` over RCCL. Used by bench.py and the distributed tests -- the reference's
dataset / W&B / checkpoint plumbing is out of scope."""

import copy
import math
import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import ddp
from .optim import FlatAdam
from .phase_graphs import PhaseGraphs
from .augment import crop_time, diff_augment, temporal_scale_apply, temporal_scale_augment, temporal_scale_params, to_device_async
from .models import lres as lres_models
from .models.lres import VideoDiscriminator, VideoGenerator

R1_GRAPH = os.environ.get('LVG_R1_GRAPH', '1') != '0'       # graph mode: the R1 pass replayed from a hipGraph too (0: eager, as before round 6)


class LowResTrainer:
    def __init__(self, seq_length: int = 128, height: int = 36, width: int = 64, device='cuda',
                 compute_dtype: torch.dtype = torch.float32, G_lrate: float = 0.003, G_beta2: float = 0.99,
                 G_ema_beta: float = 0.99985, G_ema_warmup_steps: int = 25000, G_magnitude_ema_beta: float = 0.999,
                 G_grad_accum: int = 1, D_lrate: float = 0.002, D_beta2: float = 0.99, D_grad_accum: int = 1,
                 r1_gamma: float = 10.0, G_random_temp_translate: bool = True, temp_scale_augment: float = 1.0,
                 diffaug_policy: str = 'color,translation,cutout', overlap_grad_sync: bool = True,
                 G_kwargs: Optional[dict] = None, D_kwargs: Optional[dict] = None, with_ema: bool = True, use_graphs: bool = False,
                 G_warmup_steps: int = 0, D_warmup_steps: int = 0):
        self.seq_length, self.height, self.width = seq_length, height, width
        self.G_lrate, self.D_lrate, self.G_warmup_steps, self.D_warmup_steps = G_lrate, D_lrate, G_warmup_steps, D_warmup_steps
        self.device, self.dtype = torch.device(device), compute_dtype
        self.G_magnitude_ema_beta, self.G_ema_beta, self.G_ema_warmup_steps = G_magnitude_ema_beta, G_ema_beta, G_ema_warmup_steps
        self.G_grad_accum, self.D_grad_accum, self.r1_gamma = G_grad_accum, D_grad_accum, r1_gamma
        self.G_random_temp_translate, self.temp_scale_augment, self.diffaug_policy = G_random_temp_translate, temp_scale_augment, diffaug_policy
        self.G_init_kwargs = dict(out_height=height, out_width=width, **(G_kwargs or {}))       # what save_G_ema records for load_G
        self.G = VideoGenerator(**self.G_init_kwargs).to(self.device).requires_grad_(False).train()
        self.D = VideoDiscriminator(seq_length=seq_length, max_edge=max(height, width), **(D_kwargs or {})).to(self.device).requires_grad_(False).train()
        for net in (self.G, self.D):
            ddp.broadcast_module(net, src=0)
        self.G_ema = copy.deepcopy(self.G).eval() if with_ema else None
        # One flat buffer per network for parameters, moments (FlatAdam) and gradients (FlatGradSync, same slice layout): an
        # optimizer step is one streaming launch, with the generator EMA of the parameters folded in (update_G_ema keeps the buffers).
        self.G_opt = FlatAdam(self.G.parameters(), lr=G_lrate, betas=(0.0, G_beta2),
                              ema_params=self.G_ema.parameters() if self.G_ema is not None else None)
        self.D_opt = FlatAdam(self.D.parameters(), lr=D_lrate, betas=(0.0, D_beta2))
        self._step = 0
        # use_graphs: the compute of update_G / update_D (generator pass, augmentation, discriminator pass, backward) is captured ONCE per
        # micro-batch shape into hipGraphs and replayed, at ANY world size; what stays eager is what cannot be captured on this stack or
        # must see host values: every collective (an RCCL collective inside a captured graph aborts) -- the gradient exchange runs after a
        # phase's replays, the running statistics a generator pass averages over ranks are exchanged in one all-reduce after its replay
        # (lvg.phase_graphs) --, the optimizer steps, R1 (whose exchange still overlaps with its backward pass). The host-side random
        # draws of the reference (crop offsets, temporal stretch) are drawn in the same order as in eager mode and handed to the graphs
        # through static device buffers. use_graphs='segmented': the same protocol without capturing (any device; tests).
        self.use_graphs = bool(use_graphs) and (self.device.type == 'cuda' or use_graphs == 'segmented')
        self._graphs = {}
        self.G_sync = ddp.FlatGradSync(self.G.parameters(), overlap=overlap_grad_sync)
        self.D_sync = ddp.FlatGradSync(self.D.parameters(), overlap=overlap_grad_sync)
        self._phase_graphs = PhaseGraphs(lambda: (self.G_sync.flat, self.D_sync.flat, *self.G.buffers(), *self.D.buffers()), self._graphs,
                                         syncs=(self.G_sync, self.D_sync), capture=self.device.type == 'cuda' and use_graphs != 'segmented')

    # ------------------------------------------------------------------------------------------
    def _gen(self, batch: int, beta: float = 1.0, t0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`t0`: crop offsets on the device (graph mode); None: drawn here on the host like the reference (video_gan_lres.py:107-116)."""
        extra = self.G.total_temporal_scale if self.G_random_temp_translate else 0
        video = self.G(batch, self.seq_length + extra, magnitude_ema_beta=beta, dtype=self.dtype)
        if extra:
            if t0 is None:
                t0 = to_device_async(self._draw_crop(batch, video.size(2)), video.device)
            video = crop_time(video, t0, self.seq_length)
        return video

    def _draw_crop(self, batch: int, frames: int) -> torch.Tensor:
        return torch.randint(frames - self.seq_length, (batch,))

    def run_D(self, video: torch.Tensor, stretch=None) -> torch.Tensor:
        """`stretch`: (i0, frac, valid) of `temporal_scale_params` on the device (graph mode); None: drawn here on the host."""
        video = diff_augment(video, self.diffaug_policy)
        if stretch is None:
            video = temporal_scale_augment(video, self.seq_length, self.temp_scale_augment)
        elif self.temp_scale_augment > 0:
            video = temporal_scale_apply(video, *stretch)
        return self.D(video, dtype=self.dtype)

    # ---- graph mode ---------------------------------------------------------------------------
    def _static_draws(self, key, batch: int):
        """Static device buffers for the host-side draws of one `_gen` + `run_D` (+ a second `run_D`) of `batch` clips."""
        buf = self._graphs.get(('draws', key, batch))
        if buf is None:
            dev, T = self.device, self.seq_length
            buf = dict(t0=torch.zeros(batch, dtype=torch.int64, device=dev),
                       stretch=[(torch.zeros(batch, T, dtype=torch.int64, device=dev), torch.zeros(batch, T, device=dev), torch.ones(batch, T, device=dev))
                                for _ in range(2)])
            self._graphs[('draws', key, batch)] = buf
        return buf

    @staticmethod
    def _fill(dst: torch.Tensor, host: torch.Tensor) -> None:
        """A host-side draw into its static device buffer, through pinned memory (no host stall per copy)."""
        dst.copy_(host.pin_memory() if dst.is_cuda else host, non_blocking=True)

    def _fill_stretch(self, dst, batch: int) -> None:
        if self.temp_scale_augment > 0:
            for d, s in zip(dst, temporal_scale_params(batch, self.seq_length, self.seq_length, self.temp_scale_augment)):
                self._fill(d, s)

    def _replay(self, key, fn):
        """Run `fn` from its graph (lvg.phase_graphs: eager warm-up rolled back on the gradient buffers and the generator's running
        statistics, capture, replay)."""
        self._phase_graphs.replay(key, fn)

    # ------------------------------------------------------------------------------------------
    def _ema_beta(self, step: int) -> float:
        halflife = math.log(self.G_ema_beta, 0.5) * (self.G_ema_warmup_steps + 1) / (step + 1)
        return min(0.5 ** halflife, self.G_ema_beta)

    def update_lrates(self, step: int) -> None:
        """Linear learning-rate warm-up of both optimizers (reference video_gan_lres.py:89-96)."""
        self.G_opt.lr = self.G_lrate * min((step + 1) / (self.G_warmup_steps + 1), 1.0)
        self.D_opt.lr = self.D_lrate * min((step + 1) / (self.D_warmup_steps + 1), 1.0)

    def update_G(self, batch: int, ema_step: Optional[int] = None) -> None:
        """`ema_step`: fold this iteration's generator-EMA update of the PARAMETERS into the optimizer pass (they do not
        change again before `update_G_ema(ema_step)`, which then only has the buffers left)."""
        assert batch % self.G_grad_accum == 0
        self.G.requires_grad_(True)
        self.G_sync.zero()
        for k in range(self.G_grad_accum):
            if k == self.G_grad_accum - 1 and self.G_sync.overlap and not self.use_graphs:
                self.G_sync.arm()                                          # (replayed phases run no hooks: their exchange follows them)
            if self.use_graphs:
                b = batch // self.G_grad_accum
                draws = self._static_draws('G', b)
                if self.G_random_temp_translate:                           # host draws in eager order: crop, then the stretch
                    self._fill(draws['t0'], self._draw_crop(b, self.seq_length + self.G.total_temporal_scale))
                self._fill_stretch(draws['stretch'][0], b)
                self._replay(('G', b), lambda: F.softplus(-self.run_D(self._gen(b, t0=draws['t0']), stretch=draws['stretch'][0])).mean().backward())
                continue
            logits = self.run_D(self._gen(batch // self.G_grad_accum))
            F.softplus(-logits).mean().backward()
        self.G.requires_grad_(False)
        self.G_sync.finish(gain=1 / self.G_grad_accum)
        fuse = ema_step is not None and self.G_ema is not None
        self.G_opt.step(ema_weight=(1.0 - self._ema_beta(ema_step)) if fuse else None)
        self._ema_fused_for = ema_step if fuse else None

    def update_D(self, real_video: torch.Tensor) -> None:
        assert real_video.size(0) % self.D_grad_accum == 0
        if self.use_graphs:
            return self._update_D_graphs(real_video)
        with torch.no_grad():
            fake_video = self._gen(real_video.size(0), beta=self.G_magnitude_ema_beta)
        self.D.requires_grad_(True)
        self.D_sync.zero()
        chunks = list(zip(fake_video.chunk(self.D_grad_accum), real_video.chunk(self.D_grad_accum)))
        for k, (fake, real) in enumerate(chunks):
            F.softplus(self.run_D(fake)).mean().backward()
            if k == len(chunks) - 1 and self.D_sync.overlap:
                self.D_sync.arm()
            F.softplus(-self.run_D(real)).mean().backward()
        self.D.requires_grad_(False)
        self.D_sync.finish(gain=1 / self.D_grad_accum)
        self.D_opt.step()

    def _update_D_graphs(self, real_video: torch.Tensor) -> None:
        """update_D with the generator pass and every (fake, real) micro-batch replayed from graphs; same order of host draws as eager."""
        n = real_video.size(0)
        b = n // self.D_grad_accum
        gen = self._static_draws('Dgen', n)
        st = self._graphs.setdefault(('Dio', n, b), dict(fake=None, fake_in=torch.empty(b, *real_video.shape[1:], dtype=self.dtype, device=self.device),
                                                         real_in=torch.empty(b, *real_video.shape[1:], dtype=real_video.dtype, device=self.device)))
        if self.G_random_temp_translate:
            self._fill(gen['t0'], self._draw_crop(n, self.seq_length + self.G.total_temporal_scale))

        def generate():
            with torch.no_grad():
                st['fake'] = self._gen(n, beta=self.G_magnitude_ema_beta, t0=gen['t0'])
        self._replay(('Dgen', n), generate)
        self.D.requires_grad_(True)
        self.D_sync.zero()
        draws = self._static_draws('D', b)

        def passes():
            F.softplus(self.run_D(st['fake_in'], stretch=draws['stretch'][0])).mean().backward()
            F.softplus(-self.run_D(st['real_in'], stretch=draws['stretch'][1])).mean().backward()
        for fake, real in zip(st['fake'].chunk(self.D_grad_accum), real_video.chunk(self.D_grad_accum)):
            st['fake_in'].copy_(fake)
            st['real_in'].copy_(real)
            self._fill_stretch(draws['stretch'][0], b)
            self._fill_stretch(draws['stretch'][1], b)
            self._replay(('D', b), passes)
        self.D.requires_grad_(False)
        self.D_sync.finish(gain=1 / self.D_grad_accum)
        self.D_opt.step()

    def _r1_pass(self, chunk: torch.Tensor, stretch=None) -> None:
        """Gradient penalty of one micro-batch of reals, accumulated into the discriminator's gradients (reference video_gan_lres.py:180-197)."""
        chunk = chunk.detach().requires_grad_(True)
        with lres_models.second_order():                      # the gradient below is differentiated again
            logits = self.run_D(chunk, stretch=stretch)
        (grad,) = torch.autograd.grad(outputs=[logits.sum()], inputs=[chunk], create_graph=True)
        (grad.square().sum(dim=(1, 2, 3, 4)) * (self.r1_gamma / 2)).mean().backward()

    def update_r1(self, video: torch.Tensor, gain: float = 1.0) -> None:
        self.D.requires_grad_(True)
        self.D_sync.zero()
        chunks = video.chunk(self.D_grad_accum)
        if self.use_graphs and R1_GRAPH:
            # round 6: the pass is ~900 launches for 19 ms of device time per 8 clips -- launch-bound when eager (32 ms). Replayed from a graph
            # like the other phases; the exchange follows the replays (no overlap with the backward pass, as in update_D's graph mode).
            b = chunks[0].size(0)
            st = self._graphs.setdefault(('R1io', b), dict(real_in=torch.empty(b, *video.shape[1:], dtype=video.dtype, device=self.device)))
            draws = self._static_draws('R1', b)
            for chunk in chunks:
                st['real_in'].copy_(chunk)
                self._fill_stretch(draws['stretch'][0], b)
                self._phase_graphs.replay(('R1', b), lambda: self._r1_pass(st['real_in'], stretch=draws['stretch'][0]), optional=True)
            chunks = ()
        for k, chunk in enumerate(chunks):
            chunk = chunk.detach().requires_grad_(True)
            with lres_models.second_order():                  # the gradient below is differentiated again
                logits = self.run_D(chunk)
            (grad,) = torch.autograd.grad(outputs=[logits.sum()], inputs=[chunk], create_graph=True)
            penalty = grad.square().sum(dim=(1, 2, 3, 4))
            if k == len(chunks) - 1 and self.D_sync.overlap:
                self.D_sync.arm()
            (penalty * (self.r1_gamma / 2)).mean().backward()
        self.D.requires_grad_(False)
        self.D_sync.finish(gain=gain / self.D_grad_accum)
        self.D_opt.step()

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

    def train_step(self, step: int, real_video: torch.Tensor, r1_interval: int = 16) -> None:
        """One iteration of the reference loop (train_lres.py:216-230)."""
        self.update_lrates(step)
        self.update_G(real_video.size(0), ema_step=step)
        self.update_D(real_video)
        if r1_interval > 0 and step % r1_interval == 0:
            self.update_r1(real_video, gain=r1_interval)
        self.update_G_ema(step)
