#!/bin/bash
# Round-4 evidence in one call (about 5 GPU-minutes): the whole -m gpu suite, smoke(), the default bench line, the rocprofv3
# summaries of the propagate scope and of the training step.  $1 = tag -> gpurun_out/r4_<tag>_*
set -u
TAG=${1:-x}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[gate\]|^\[invariance\]|^\[static| passed| failed|^FAILED|^ERROR|max_ring" > "$OUT/r4_${TAG}_pytest_gpu.txt"; tail -1 "$OUT/r4_${TAG}_pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
SECONDS=0
timeout 1200 python bench.py > "$OUT/r4_${TAG}_bench_zinc.json" 2> "$OUT/r4_${TAG}_bench_zinc.err"; echo "bench wall ${SECONDS}s"
python - <<PY
import json
d = json.loads(open('$OUT/r4_${TAG}_bench_zinc.json').read().strip().splitlines()[-1])
s = d['secondary']
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
print('full_forward_ms', s['full_forward_ms'], 'train', (s['train_step'] or {}).get('ms_per_step'))
fb = s.get('fresh_batches') or {}
print('fresh', {k: fb.get(k) for k in ('propagate', 'forward', 'train', 'every_batch_within_capacity', 'device_error_word')})
for k, v in (s.get('workloads') or {}).items():
    print(k, v.get('value'), v.get('ms_per_step'), (v.get('failed') or '')[:200])
PY
ROOT=$PWD
cd /tmp
rm -rf /tmp/prof_scope
rocprofv3 --kernel-trace --stats -d /tmp/prof_scope -- python "$ROOT/bench.py" --only-primary > /dev/null 2>&1
cd "$ROOT"
python profiles/summarize_rocprof.py "$(ls /tmp/prof_scope/*/*results.db | head -1)" 30 > "$OUT/r4_${TAG}_propagate_scope.md"
head -6 "$OUT/r4_${TAG}_propagate_scope.md" | cut -c1-140
bash tools/prof_train.sh 128 200 > /dev/null 2>&1; cp gpurun_out/prof_train_128.md "$OUT/r4_${TAG}_train_step.md"
head -8 "$OUT/r4_${TAG}_train_step.md" | cut -c1-140
