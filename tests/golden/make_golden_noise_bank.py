"""Golden vectors for the temporal noise filter bank, produced by the REFERENCE code: BlurredNoise (model/generator_lres.py:323-388) built with
small sampling rates (filters of 32 .. 256 taps instead of 125 .. 5000) and its `blur` applied to seeded noise on CPU. Stored: the noise, the
module's own `blur_filters` / `output_scale` buffers, `normalize_per_filter`, and the features. Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_noise_bank.py [/root/reference]"""

import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import importlib  # noqa: E402
import types  # noqa: E402

import torch  # noqa: E402

for _ in range(16):
    try:
        generator_lres = importlib.import_module('model.generator_lres')
        break
    except ModuleNotFoundError as err:
        assert err.name and not err.name.startswith(('model', 'torch_utils', 'dnnlib', 'utils')), err
        sys.modules[err.name] = types.ModuleType(err.name)
assert os.path.realpath(generator_lres.__file__).startswith(os.path.realpath(REF))

out = {}
cases = [('norm1', dict(channels=80, blur_widths=40, min_sampling_rate=64, max_sampling_rate=512, normalize_per_filter=1.0), 3, 21),
         ('norm0', dict(channels=24, blur_widths=12, min_sampling_rate=64, max_sampling_rate=300, normalize_per_filter=0.0), 2, 16),
         ('linear_rates', dict(channels=66, blur_widths=33, min_sampling_rate=80, max_sampling_rate=400, sampling_rate_base=1.0, normalize_per_filter=0.5), 1, 35)]
for name, kw, batch, frames in cases:
    torch.manual_seed(7)
    mod = generator_lres.BlurredNoise(**kw)
    noise = torch.randn(batch, mod.noise_channels, frames + mod.kernel_size - 1)
    feat = mod.blur(noise)
    assert feat.shape == (batch, kw['channels'], frames)
    out[f'{name}/noise'] = noise.numpy()
    out[f'{name}/bank'] = mod.blur_filters[:, 0, :].numpy()
    out[f'{name}/scale'] = (mod.output_scale.reshape(-1).numpy() if kw['normalize_per_filter'] > 0 else np.zeros(0, np.float32))
    out[f'{name}/normalize_per_filter'] = np.float32(kw['normalize_per_filter'])
    out[f'{name}/features'] = feat.numpy()
    out[f'{name}/kwargs'] = np.array(repr(kw))
np.savez_compressed(os.path.join(HERE, 'noise_bank.npz'), **out)
print('wrote', os.path.join(HERE, 'noise_bank.npz'), {k: v.shape for k, v in out.items() if k.endswith('features')})
