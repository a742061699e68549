import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'long-video-gan_amd')
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# MIOpen find-db + compiled-kernel cache recorded on MI355X for the shapes the models use (a cold
# find on a fresh box costs minutes per network); same wiring as bench.py.
_MIOPEN_DB = os.path.join(PKG, 'miopen_db')
if os.path.isdir(_MIOPEN_DB) and os.access(_MIOPEN_DB, os.W_OK):
    os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(_MIOPEN_DB, 'db'))
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(_MIOPEN_DB, 'cache'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu through gpurun)')


def record_measured(name, **values):
    """Append measured parity errors to gpurun_out/parity_measured.json (best effort; merged back by gpurun) and print them,
    so that every gate in the suite can be read next to the number it gates."""
    import json
    vals = {k: (v if isinstance(v, (list, dict, str)) else float(v)) for k, v in values.items()}
    print(f'[measured] {name}: ' + ', '.join(f'{k}={v:.4g}' if isinstance(v, float) else f'{k}={v}' for k, v in vals.items()))
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, 'parity_measured.json')
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data[name] = vals
        with open(path, 'w') as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


def load_golden(name):
    """dict of arrays from tests/golden/<name>.npz; '*_spec' entries are parsed back to dicts."""
    raw = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    out = {}
    for k in raw.files:
        v = raw[k]
        if k.endswith('spec'):
            txt = str(v)
            txt = txt.replace('np.float64(', '(').replace('np.float32(', '(')
            out[k] = ast.literal_eval(txt)
        else:
            out[k] = v
    return out


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, so `pytest tests/` on the
    # CPU container stays green; `-m gpu` on the MI355X box runs them for real.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
