"""The bench line the driver parses (round 5's 22 KB line was not parsed).  CPU tests feed a committed full
line through bench.slim_line; the GPU test runs `bench.py --gpus 2` through its own launcher (two ranks sharing the one GPU of
a test box over gloo: control flow only) and reads the line as the driver does."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline', 'cpu_baseline')
ROOFLINE = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')
CPU = ('value', 'unit', 'cores', 'kind', 'sample')


def _full_lines():
    return sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[56]_*_bench_zinc.json')))


@pytest.mark.parametrize('path', _full_lines(), ids=os.path.basename)
def test_slim_line_fits_the_driver(path):
    import bench
    with open(path) as fh:
        full = json.load(fh)
    if 'detail' in full and 'timing' in full and 'note' not in (full.get('timing') or {}):
        pytest.skip('already a slim line')
    slim = bench.slim_line(full, 'bench_detail.json')
    line = json.dumps(slim)
    assert len(line) < bench.LINE_BUDGET <= 6144, len(line)
    assert '\n' not in line
    for k in CONTRACT:
        assert k in slim, k
    for k in ROOFLINE:
        assert k in slim['roofline'], k
    for k in CPU:
        assert k in slim['cpu_baseline'], k
    assert slim['value'] == full['value'] and slim['ms_per_step'] == full['ms_per_step']
    assert slim['roofline']['frac'] == full['roofline']['frac']
    assert 'workload' in slim['config'] and 'model' not in slim['config']
    assert slim['roofline']['avg_launch_us_source'].startswith('hip events')
    # scalars only below `secondary.workloads`
    for w in (slim['secondary'].get('workloads') or {}).values():
        assert all(not isinstance(v, (dict, list)) for v in w.values()), w


def test_slim_line_survives_an_oversized_leg():
    import bench
    full = json.load(open(_full_lines()[-1]))
    full['secondary']['workloads'] = {f'w{i}': {'value': 1.0, 'ms_per_step': 1.0, 'roofline': {'frac': 0.5}} for i in range(400)}
    slim = bench.slim_line(full, None)
    assert len(json.dumps(slim)) < bench.LINE_BUDGET
    assert slim['roofline'] and slim['cpu_baseline'] and slim['value'] == full['value']


def test_gpus_flag_must_match_world_size(monkeypatch):
    import bench
    monkeypatch.setenv('WORLD_SIZE', '1')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'WORLD_SIZE=1' in str(e.value)


@pytest.mark.gpu
def test_bench_gpus_2_launches_two_ranks(tmp_path):
    env = dict(os.environ, CWN_BENCH_SHARE_GPU='1', CWN_BENCH_SKIP='eager,concurrent,collate,fresh,workloads,full,roofline',
               CWN_BENCH_DETAIL=str(tmp_path / 'detail.json'))
    env.pop('WORLD_SIZE', None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '8', '--warmup', '2', '--no-cpu'],
                        capture_output=True, text=True, timeout=600, env=env)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
    assert lines, pr.stdout[-500:]
    d = json.loads(lines[-1])
    assert len(lines[-1]) < 6144
    assert d['n_gpus'] == 2 and d['multi_gpu']['rccl_ranks'] == 2
    assert d['value'] > 0 and d['steps'] == 8 and d['warmup'] == 2
