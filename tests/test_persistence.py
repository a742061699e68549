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


_REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(os.path.join(_REF, 'model')), reason='reference checkout only exists in the build container')
def test_reference_pickled_generator_loads_and_matches_on_this_repos_ops(tmp_path):
    """INTEGRATION.md option A end to end with the REAL network class: a child interpreter running the reference's own
    `torch_utils` + `model.generator_sres` pickles a (reduced-width, all 15 layers) super-resolution `Generator` with the
    reference persistence and stores its output; this process -- whose `torch_utils` is this repo's -- unpickles it
    (the class is rebuilt from the source embedded in the pickle, so its `from torch_utils.ops import ...` binds to this
    repo's ops), checks the persistence metadata and reproduces the output. The pickle embeds reference source text, so it
    is generated here on the fly and never committed (the GPU box has no reference checkout; the op layer under a
    reference-pickled module is covered there by persist_ref_v6.pkl)."""
    import subprocess
    helpers = os.path.join(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, pickle, numpy as np, torch\n"
        f"sys.path.insert(0, {_REF!r}); sys.path.insert(0, {helpers!r}); sys.dont_write_bytecode = True\n"
        "import torch_utils.persistence as p\n"
        f"assert p.__file__.startswith({_REF!r})\n"
        "from model import generator_sres\n"
        "from helpers.named_fill import fill_named\n"
        "from helpers.sres_cfg import SMALL_G, small_inputs\n"
        "torch.set_num_threads(4)\n"
        "G = generator_sres.Generator(**SMALL_G)\n"
        "fill_named(G)\n"
        "z, lr = small_inputs()\n"
        "with torch.no_grad():\n"
        "    y = G(z, lr)\n"
        f"pickle.dump(dict(G=G), open({str(tmp_path / 'ref_G.pkl')!r}, 'wb'))\n"
        f"np.save({str(tmp_path / 'ref_y.npy')!r}, y.numpy())\n"
        "print('ok', tuple(y.shape))\n")
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    env['PYTHONDONTWRITEBYTECODE'] = '1'
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    assert not any(m == 'model' or m.startswith('model.') for m in sys.modules), 'the class must come from the pickle, not from an import'
    with open(tmp_path / 'ref_G.pkl', 'rb') as f:
        G = pickle.load(f)['G']
    assert persistence.is_persistent(G) and type(G).__name__ == 'Generator'
    src_module = type(G).__mro__[1].__module__                            # (type(G) is the persistence wrapper subclass)
    assert src_module.startswith('_imported_module_')                     # rebuilt from the embedded source
    from helpers.sres_cfg import SMALL_G, small_inputs
    assert dict(G.init_kwargs) == SMALL_G
    ops_mod = sys.modules[src_module].filtered_lrelu                      # the embedded source's `from torch_utils.ops import filtered_lrelu`
    assert os.path.realpath(ops_mod.__file__).startswith(os.path.realpath(os.path.join(helpers, '..', 'long-video-gan_amd')))
    z, lr = small_inputs()
    torch.set_num_threads(4)
    with torch.no_grad():
        y = G(z, lr)
    want = np.load(tmp_path / 'ref_y.npy')
    assert y.shape == want.shape
    np.testing.assert_allclose(y.numpy(), want, rtol=0, atol=1e-4)
    # and it round-trips through THIS persistence
    buf = io.BytesIO()
    pickle.dump(G, buf)
    G2 = pickle.loads(buf.getvalue())
    with torch.no_grad():
        torch.testing.assert_close(G2(z, lr), y)
