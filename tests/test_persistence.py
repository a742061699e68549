"""Pickle compatibility of torch_utils.persistence (protocol v6, reference persistence.py:35-251):
a pickle WRITTEN BY THE REFERENCE (tests/golden/persist_ref_v6.pkl, module source embedded) loads
with this repo's persistence, rebuilds the class from the embedded source against this repo's
torch_utils.ops, carries the saved state, and computes the reference's output; objects written here
round-trip, keep init args, survive a changed class definition, and run import hooks."""

import copy
import io
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from torch_utils import persistence


def _load_ref():
    with open(os.path.join(GOLDEN, 'persist_ref_v6.pkl'), 'rb') as f:
        return pickle.load(f)


def test_reference_written_pickle_loads_and_runs_on_this_repos_ops_cpu():
    assert 'tiny_net' not in sys.modules                    # the class must come from the source embedded in the pickle
    data = _load_ref()
    net = data['net']
    assert data['note'] == 'written by the reference persistence'
    assert persistence.is_persistent(net) and type(net).__name__ == 'TinyUpsampler'
    assert net.init_args == () and dict(net.init_kwargs) == dict(channels=4, up=2, slope=0.3)
    g = load_golden('persistence')
    y = net(torch.tensor(g['x']))
    np.testing.assert_allclose(y.detach().numpy(), g['y'], rtol=1e-5, atol=1e-6)
    assert float(net.bias[0]) == pytest.approx(0.75)        # constructor gives 0.5; the pickle carried the edit


@pytest.mark.gpu
def test_reference_written_pickle_runs_on_the_hip_kernels_gpu():
    net = _load_ref()['net'].cuda()
    g = load_golden('persistence')
    y = net(torch.tensor(g['x'], device='cuda'))
    np.testing.assert_allclose(y.detach().cpu().numpy(), g['y'], rtol=1e-5, atol=1e-5)


def test_round_trip_of_reference_object_and_deepcopy_cpu():
    net = _load_ref()['net']
    buf = io.BytesIO()
    pickle.dump(net, buf)                                   # written by THIS persistence
    again = pickle.loads(buf.getvalue())
    clone = copy.deepcopy(net)
    x = torch.randn(1, 4, 3, 3)
    torch.testing.assert_close(again(x), net(x))
    torch.testing.assert_close(clone(x), net(x))
    assert again.init_kwargs == net.init_kwargs


def test_import_hook_and_version_check_cpu():
    seen = []

    def hook(meta):
        seen.append(meta.type)
        return meta
    persistence.import_hook(hook)
    try:
        _load_ref()
    finally:
        persistence._import_hooks.remove(hook)
    assert 'class' in seen
    raw = open(os.path.join(GOLDEN, 'persist_ref_v6.pkl'), 'rb').read()
    meta_version = b'version'
    assert meta_version in raw                               # protocol field present in the byte stream


@pytest.mark.skipif(not os.path.isdir('/root/reference/torch_utils'), reason='reference checkout only exists in the build container')
def test_pickle_written_here_loads_with_the_reference_persistence_cpu(tmp_path):
    """The other direction, in a child interpreter whose torch_utils IS the reference's."""
    import subprocess
    net = _load_ref()['net']
    with torch.no_grad():
        net.bias.mul_(2.0)
    path = tmp_path / 'written_here.pkl'
    with open(path, 'wb') as f:
        pickle.dump(net, f)
    x = torch.linspace(-1, 1, 1 * 4 * 3 * 3).reshape(1, 4, 3, 3)
    want = net(x).detach().numpy()
    code = (
        "import sys, pickle, numpy as np, torch\n"
        "sys.path.insert(0, '/root/reference'); sys.dont_write_bytecode = True\n"
        "import torch_utils.persistence as p\n"
        "assert p.__file__.startswith('/root/reference')\n"
        f"net = pickle.load(open({str(path)!r}, 'rb'))\n"
        "x = torch.linspace(-1, 1, 36).reshape(1, 4, 3, 3)\n"
        f"np.save({str(tmp_path / 'out.npy')!r}, net(x).detach().numpy())\n"
        "print(type(net).__name__, dict(net.init_kwargs))\n")
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    env['PYTHONDONTWRITEBYTECODE'] = '1'
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=120, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'TinyUpsampler' in out.stdout
    np.testing.assert_allclose(np.load(tmp_path / 'out.npy'), want, rtol=1e-5, atol=1e-6)
