"""Stand-in generator / discriminator for the trainer step-body parity test: the same two classes are driven by the REFERENCE's
LowResVideoGAN.update_G / update_D / update_r1 (tests/golden/make_golden_trainer_glue.py) and by lvg.train_lres.LowResTrainer. They have the call
signatures the trainers use, draw their noise from the default (CPU) generator so that the ORDER of all random draws of a step is part of what
is compared, are smooth (softplus, no kinks) and are initialised from a seeded generator of their own."""

import torch
import torch.nn as nn
import torch.nn.functional as F


class StubG(nn.Module):
    total_temporal_scale = 4

    def __init__(self, height: int = 6, width: int = 8, seed: int = 1):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.height, self.width = height, width
        self.mix = nn.Parameter(torch.randn(3, 5, generator=g) * 0.5)
        self.plane = nn.Parameter(torch.randn(5, height, width, generator=g) * 0.5)
        self.register_buffer('magnitude_ema', torch.ones([]))

    def forward(self, batch_size: int, seq_length: int, magnitude_ema_beta: float = 1.0, **_unused) -> torch.Tensor:
        z = torch.randn(batch_size, 5, seq_length)                                   # default generator: part of the draw order of a step
        feat = torch.einsum('nkt,khw->nkthw', z, self.plane)
        if magnitude_ema_beta < 1:
            with torch.no_grad():
                self.magnitude_ema.copy_(feat.square().mean().lerp(self.magnitude_ema, magnitude_ema_beta))
        return torch.tanh(torch.einsum('ck,nkthw->ncthw', self.mix, feat))


class StubD(nn.Module):
    def __init__(self, seq_length: int, seed: int = 2):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.conv = nn.Parameter(torch.randn(4, 3, 3, 3, 3, generator=g) * 0.2)
        self.bias = nn.Parameter(torch.randn(4, generator=g) * 0.1)
        self.head = nn.Parameter(torch.randn(4, seq_length, generator=g) * 0.3)

    def forward(self, video: torch.Tensor, **_unused) -> torch.Tensor:
        x = F.softplus(F.conv3d(video, self.conv, self.bias, padding=1))
        return torch.einsum('ncthw,ct->n', x, self.head).unsqueeze(1) / (x.size(3) * x.size(4))


class StubSresG(nn.Module):
    """Super-resolution stand-in: lr clip with temporal context [N, 3, T + 2c, h, w] -> hr clip [N, 3, T, 4h, 4w]; one latent per call from
    the default generator."""

    def __init__(self, temporal_context: int = 1, seed: int = 3):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.temporal_context = temporal_context
        self.mix = nn.Parameter(torch.randn(3, 3, 2 * temporal_context + 1, generator=g) * 0.4)
        self.style = nn.Parameter(torch.randn(3, 4, generator=g) * 0.3)
        self.register_buffer('magnitude_ema', torch.ones([]))

    def forward(self, lr_video: torch.Tensor, magnitude_ema_beta: float = 1.0, **_unused) -> torch.Tensor:
        z = torch.randn(lr_video.size(0), 4, device=lr_video.device)
        x = F.conv3d(lr_video, self.mix[:, :, :, None, None])                        # 'valid' in time: consumes the context
        x = x * (1 + torch.tanh(z @ self.style.t()))[:, :, None, None, None]
        if magnitude_ema_beta < 1:
            with torch.no_grad():
                self.magnitude_ema.copy_(x.square().mean().lerp(self.magnitude_ema, magnitude_ema_beta))
        n, c, t, h, w = x.shape
        up = F.interpolate(x.reshape(n, c * t, h, w), scale_factor=4, mode='bilinear', align_corners=False)
        return torch.tanh(up.reshape(n, c, t, 4 * h, 4 * w))


class StubSresD(nn.Module):
    def __init__(self, seq_length: int, seed: int = 4):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.conv = nn.Parameter(torch.randn(4, 6, 1, 3, 3, generator=g) * 0.2)
        self.bias = nn.Parameter(torch.randn(4, generator=g) * 0.1)
        self.head = nn.Parameter(torch.randn(4, seq_length, generator=g) * 0.3)

    def upsample(self, lr_video: torch.Tensor) -> torch.Tensor:
        n, c, t, h, w = lr_video.shape
        up = F.interpolate(lr_video.reshape(n, c * t, h, w), scale_factor=4, mode='bilinear', align_corners=False)
        return up.reshape(n, c, t, 4 * h, 4 * w)

    def forward(self, lr_video: torch.Tensor, hr_video: torch.Tensor, **_unused) -> torch.Tensor:
        x = F.softplus(F.conv3d(torch.cat((lr_video, hr_video), dim=1), self.conv, self.bias, padding=(0, 1, 1)))
        return torch.einsum('ncthw,ct->n', x, self.head).unsqueeze(1) / (x.size(3) * x.size(4))
