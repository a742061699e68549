"""lvg.optim.FlatAdam (flat parameter / moment buffers, one fused launch per run, EMA folded in) vs torch.optim.Adam +
lerp: same numbers step after step, including parameters that receive no gradient in some steps (their moments and update
counts must stay untouched, reference: zero_grad(set_to_none=True) + torch.optim.Adam) and the generator EMA."""

import copy

import pytest
import torch

from lvg import ddp
from lvg.optim import FlatAdam


def _nets(device):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(13, 7), torch.nn.Tanh(), torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(device)
    return net, copy.deepcopy(net), copy.deepcopy(net), copy.deepcopy(net)


def _run(device, steps=6):
    net, ref, ema, ema_ref = _nets(device)
    opt = FlatAdam(net.parameters(), lr=0.01, betas=(0.0, 0.99), ema_params=ema.parameters())
    sync = ddp.FlatGradSync(net.parameters())
    opt_ref = torch.optim.Adam(ref.parameters(), lr=0.01, betas=(0.0, 0.99))
    g = torch.Generator().manual_seed(3)
    for k in range(steps):
        x = torch.randn(4, 13, generator=g).to(device)
        # every other step the last layer is bypassed: its parameters get no gradient
        def loss_of(m):
            h = m[2](m[1](m[0](x)))
            return (h if k % 2 else m[3](h)).square().sum()
        sync.zero()
        loss_of(net).backward()
        sync.finish()
        opt.step(ema_weight=0.1)
        opt_ref.zero_grad(set_to_none=True)
        loss_of(ref).backward()
        opt_ref.step()
        with torch.no_grad():
            for pe, p in zip(ema_ref.parameters(), ref.parameters()):
                pe.lerp_(p, 0.1)                  # the reference lerps EVERY parameter every step, also those without a gradient
        for a, b in zip(net.parameters(), ref.parameters()):
            torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7)
        for a, b in zip(ema.parameters(), ema_ref.parameters()):
            torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7)
    assert opt.steps[-1] == steps // 2 and opt.steps[0] == steps
    st = opt_ref.state[list(ref.parameters())[-1]]
    torch.testing.assert_close(opt.state_dict()['exp_avg_sq'][-1], st['exp_avg_sq'], rtol=2e-6, atol=1e-12)
    # parameters are views of one buffer, aligned to 16 bytes
    assert all(p.data_ptr() % 16 == 0 and p.untyped_storage().data_ptr() == opt.flat.untyped_storage().data_ptr() for p in net.parameters())


def test_flat_adam_matches_torch_adam_cpu():
    _run('cpu')


@pytest.mark.gpu
def test_flat_adam_matches_torch_adam_gpu():
    _run('cuda')


@pytest.mark.gpu
def test_one_launch_covers_the_whole_network_when_all_parameters_have_gradients():
    net, _, _, _ = _nets('cuda')
    opt = FlatAdam(net.parameters(), lr=0.01)
    sync = ddp.FlatGradSync(net.parameters())
    calls = []
    orig = opt._range_update
    opt._range_update = lambda lo, hi, grad, step, w: (calls.append((lo, hi)), orig(lo, hi, grad, step, w))
    sync.zero()
    net(torch.randn(2, 13, device='cuda')).sum().backward()
    sync.finish()
    opt.step()
    assert len(calls) == 1 and calls[0] == (0, opt.flat.numel() - (opt.flat.numel() - (opt.offsets[-1] + opt.params[-1].numel())))
