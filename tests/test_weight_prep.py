"""Weight side of the modulated convolution in one pass per direction (csrc/weight_prep.hip, torch_utils/ops/weight_prep.py)
against the tensor expressions of the reference (model/generator_lres.py:97-119) and their autograd."""

import math

import numpy as np
import pytest
import torch

from torch_utils.ops import weight_prep as wp


def _weights(seed, co, ci, taps_shape, ties=False):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(co, ci, *taps_shape, generator=g)
    if ties:                                       # two equal maxima (one negative) in every output channel
        flat = w.view(co, -1)
        flat[:, 3] = 7.5
        flat[:, 11] = -7.5
    return w


def test_definition_is_the_reference_chain_cpu():
    w = _weights(0, 4, 6, (3, 3, 3))
    scale = 1 / math.sqrt(6 * 27)
    out, w2 = wp.weight_prep(w, scale, True, torch.bfloat16)
    ref = w / w.abs().amax(dim=(1, 2, 3, 4), keepdim=True) / math.sqrt(np.prod(w.shape[1:]))       # generator_lres.py:98, :102-103
    assert torch.equal(out, ref.to(torch.bfloat16)) or float((out.float() - ref).abs().max()) < 1e-2
    np.testing.assert_allclose(w2.numpy(), ref.square().sum(dim=(2, 3, 4)).numpy(), rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape,normalize,ties', [((64, 64, 3, 3, 3), True, False), ((128, 512, 3, 3, 3), True, True), ((64, 128, 1, 3, 3), True, False),
                                                  ((32, 3, 1, 1, 1), False, False), ((128, 64, 5, 3, 3), False, False), ((3, 64, 1, 1, 1), True, False)])
def test_hip_weight_prep_forward_backward_gpu(shape, normalize, ties, dtype):
    w = _weights(1, shape[0], shape[1], shape[2:], ties).cuda()
    scale = 1 / math.sqrt(np.prod(shape[1:]))
    assert wp.supported(w, dtype)
    wa = w.clone().requires_grad_(True)
    wb = w.clone().requires_grad_(True)
    out, w2 = wp.weight_prep(wa, scale, normalize, dtype)
    ref, ref2 = wp._ref(wb, scale, normalize, dtype, True)
    assert out.shape == ref.shape and out.dtype == dtype
    assert out.permute(2, 3, 4, 0, 1).is_contiguous()                  # tap-major memory: packing for the convolution is free
    assert torch.equal(out, ref)                                        # same float32 operation order, same rounding
    np.testing.assert_allclose(w2.detach().cpu().numpy(), ref2.detach().cpu().numpy(), rtol=2e-6)
    g = torch.Generator().manual_seed(2)
    g_w = torch.randn(shape, generator=g).to(dtype).cuda()
    g_w2 = torch.randn(shape[0], shape[1], generator=g).cuda()
    # the gradient arrives either dense or as the permuted view of a tap-major tensor (what the weight-gradient kernel returns)
    for gv in (g_w, g_w.permute(2, 3, 4, 0, 1).contiguous().permute(3, 4, 0, 1, 2)):
        da, = torch.autograd.grad([out, w2], wa, [gv, g_w2], retain_graph=True)
        db, = torch.autograd.grad([ref, ref2], wb, [g_w, g_w2], retain_graph=True)
        err = float((da - db).abs().max() / db.abs().max())
        assert err < 2e-5, err
    again, = torch.autograd.grad([out, w2], wa, [g_w, g_w2], retain_graph=True)
    assert torch.equal(again, da)
    out_only, none = wp.weight_prep(wa, scale, normalize, dtype, want_w2=False)
    assert none is None and torch.equal(out_only, ref)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(64, 128, 3, 3, 3), (192, 64, 1, 3, 3), (128, 128, 5, 3, 3), (64, 64, 1, 1, 1)])
def test_data_gradient_packing_gpu(shape):
    """The weight packed for the data-gradient convolution (taps mirrored, channel roles swapped) is bit-identical to flip + transpose +
    pack of the prepared weight; weight_prep attaches it while gradients are recorded, and only then."""
    from torch_utils.ops import conv3d_frames as cf
    w = _weights(3, shape[0], shape[1], shape[2:]).cuda().requires_grad_(True)
    scale = 1 / math.sqrt(np.prod(shape[1:]))
    w16, _ = wp.weight_prep(w, scale, True, torch.bfloat16)
    wt = getattr(w16, '_lvg_dgrad', None)
    assert wt is not None and tuple(wt.shape) == tuple(shape[2:]) + (shape[1], shape[0]) and wt.is_contiguous()
    ref = cf.pack_weight(w16.detach().flip(2, 3, 4).transpose(0, 1))
    assert torch.equal(wt, ref)
    with torch.no_grad():
        w16n, _ = wp.weight_prep(w, scale, True, torch.bfloat16)
    assert getattr(w16n, '_lvg_dgrad', None) is None
