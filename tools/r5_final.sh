#!/bin/bash
# The last check of round 5 at HEAD (a reduced tools/r5_regen.sh: no rocprofv3 passes): the whole -m gpu suite, smoke(), the default
# bench line.  $1 = tag -> gpurun_out/r5_<tag>_pytest_gpu.txt, r5_<tag>_bench_zinc.json
set -u
TAG=${1:-x}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[gate\]|^\[invariance\]|^\[static|^\[router\]|^\[rccl| passed| failed|^FAILED|^ERROR|max_ring" > "$OUT/r5_${TAG}_pytest_gpu.txt"; tail -1 "$OUT/r5_${TAG}_pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
SECONDS=0
timeout 1200 python bench.py > "$OUT/r5_${TAG}_bench_zinc.json" 2> "$OUT/r5_${TAG}_bench_zinc.err"; echo "bench wall ${SECONDS}s"
python - <<PY
import json
d = json.loads(open('$OUT/r5_${TAG}_bench_zinc.json').read().strip().splitlines()[-1])
s = d['secondary']
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
print('full_forward_ms', s['full_forward_ms'], s['forward_breakdown'], 'train', (s['train_step'] or {}).get('ms_per_step'), 'eager', s['eager_launches'])
for k, v in (s.get('workloads') or {}).items():
    print(k, v.get('value'), v.get('ms_per_step'), 'fwd', v.get('full_forward_ms'), 'train', v.get('train_step_ms'), (v.get('failed') or '')[:200])
PY
