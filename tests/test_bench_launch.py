"""`python bench.py --gpus N` invoked plainly must bring up N ranks itself (VERDICT r02 missing 4; the behaviour of the
reference's launcher contract, torch_utils/distributed.py:42-69: one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* from the environment). CPU: the launch path alone, on gloo, world size 2."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='', OMP_NUM_THREADS='1')
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_plain_invocation_spawns_the_ranks_gloo_world2():
    res = _run(['--gpus', '2', '--selftest-launch'])
    assert res == {'launch_selftest': True, 'n_gpus': 2, 'backend': 'gloo', 'rank_sum': 1.0}


def test_single_rank_needs_no_launcher():
    res = _run(['--gpus', '1', '--selftest-launch'])
    assert res['n_gpus'] == 1 and res['rank_sum'] == 0.0
