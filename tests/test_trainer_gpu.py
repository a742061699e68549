"""One full iteration of the low-resolution training step on the GPU (update_G, update_D, R1 with
double backward through the HIP ops and the frames-layout convolutions, G-EMA), on synthetic video."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_step_runs_and_updates_parameters():
    from lvg.train_lres import LowResTrainer
    torch.manual_seed(0)
    tr = LowResTrainer(seq_length=8, height=36, width=64, device='cuda', compute_dtype=torch.float32,
                       G_grad_accum=1, D_grad_accum=1, overlap_grad_sync=False, with_ema=True, temp_scale_augment=1.0)
    g_before = [p.detach().clone() for p in list(tr.G.parameters())[:4]]
    d_before = [p.detach().clone() for p in list(tr.D.parameters())[:4]]
    real = torch.rand(1, 3, 8, 36, 64, device='cuda') * 2 - 1
    tr.train_step(step=0, real_video=real, r1_interval=16)       # step 0 includes the R1 update
    torch.cuda.synchronize()
    for p in list(tr.G.parameters()) + list(tr.D.parameters()):
        assert torch.isfinite(p).all()
    assert any(not torch.equal(a, b) for a, b in zip(g_before, list(tr.G.parameters())[:4]))
    assert any(not torch.equal(a, b) for a, b in zip(d_before, list(tr.D.parameters())[:4]))
    ema_mag = [b for n, b in tr.G.named_buffers() if n.endswith('magnitude_ema')]
    assert any(float(b) != 1.0 for b in ema_mag)                  # update_D ran G with beta = 0.999
    # gradients live in the persistent flat buffers
    assert tr.G_sync.flat.numel() == sum(p.numel() for p in tr.G.parameters())
