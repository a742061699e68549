"""Golden vectors for the dense contraction of the sres generator: the REFERENCE's `modulated_conv2d`
(model/generator_sres.py:28-67: per-sample weights, grouped convolution) on CPU, forward and -- through autograd -- the
gradients with respect to x, weight and style. Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_modconv2d.py [/root/reference]

Inputs are NOT stored: both sides draw them from torch.Generator seeds (tests/helpers/modconv2d_inputs.py); outputs are float32."""

import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

import torch  # noqa: E402
from model import generator_sres  # noqa: E402
from helpers.modconv2d_inputs import CASES, inputs  # noqa: E402

assert os.path.realpath(generator_sres.__file__).startswith(os.path.realpath(REF))

if __name__ == '__main__':
    out = {}
    for name, case in CASES.items():
        x, weight, style, gain, dy = inputs(name)
        x.requires_grad_(True); weight.requires_grad_(True); style.requires_grad_(True)
        y = generator_sres.modulated_conv2d(x, weight, style, demodulate=True, padding=case[6], input_gain=gain)
        gx, gw, gs = torch.autograd.grad(y, [x, weight, style], dy)
        out[name + '_y'] = y.detach().numpy()
        out[name + '_gx'] = gx.numpy()
        out[name + '_gw'] = gw.numpy()
        out[name + '_gs'] = gs.numpy()
        print(name, tuple(y.shape), float(y.abs().mean()), float(gw.abs().mean()))
    np.savez_compressed(os.path.join(HERE, 'modconv2d.npz'), **out)
