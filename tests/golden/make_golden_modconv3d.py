"""Golden vectors for the dense contraction of the lres generator: the REFERENCE's `temporal_modulated_conv3d`
(model/generator_lres.py:83-125) followed by its bias_act (:570, impl='ref'), on CPU. Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_modconv3d.py [/root/reference]

Inputs are NOT stored: both sides draw them from torch.Generator seeds (tests/helpers/modconv3d_inputs.py); outputs are float32."""

import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

import torch  # noqa: E402
from model import generator_lres  # noqa: E402
from torch_utils.ops import bias_act  # noqa: E402
from helpers.modconv3d_inputs import CASES, inputs  # noqa: E402

assert os.path.realpath(generator_lres.__file__).startswith(os.path.realpath(REF))

if __name__ == '__main__':
    out = {}
    for name in CASES:
        x, weight, style, bias, gain = inputs(name)
        kt, kh, kw = weight.shape[2:]
        with torch.no_grad():
            y = generator_lres.temporal_modulated_conv3d(x, weight, style, gain, padding=(kt // 2, kh // 2, kw // 2), demodulate=True)
            z = bias_act.bias_act(y, bias, act='lrelu', clamp=2.0, impl='ref')
        out[name + '_conv'] = y.numpy()
        out[name + '_act'] = z.numpy()
        print(name, tuple(y.shape), float(y.abs().mean()), float(z.abs().max()))
    np.savez_compressed(os.path.join(HERE, 'modconv3d.npz'), **out)
