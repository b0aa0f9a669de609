"""RCCL executes on the GPU box: tests/_rccl_world1.py in a process of its own -- a world-1 "nccl"
process group, the data-parallel form of TrainStep / StaticTrainStep forced, every collective a real RCCL all-reduce."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_one_nccl_group_runs_the_data_parallel_step():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('CWN_FORCE_DP', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_rccl_world1.py')], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    tail = (r.stdout[-3000:] + '\n' + r.stderr[-3000:])
    assert r.returncode == 0, tail
    line = [l for l in r.stdout.splitlines() if l.startswith('RCCL_WORLD1 ')]
    assert line, tail
    out = json.loads(line[-1][len('RCCL_WORLD1 '):])
    print('[rccl world-1]', json.dumps(out))
    assert out['backend'] == 'nccl' and out['world'] == 1
    assert out['staged']['pieces'] > 1 and out['jumping_knowledge']['pieces'] == 1
    assert out['all_reduce_calls'] > 20 and out['all_reduce_bytes'] > 0
    assert out['static']['t_forced'] == 3
