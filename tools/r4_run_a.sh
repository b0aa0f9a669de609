#!/bin/bash
# round 4, first GPU call: the static-batch tests, the training step at the timed size, the whole GPU suite, the default bench
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_static.py -q -rP -x 2>&1 | tail -120 > "$OUT/r4_a_static.txt"; tail -5 "$OUT/r4_a_static.txt"
timeout 600 python -m pytest tests/test_gpu_train_full.py -q -rP 2>&1 | grep -E "^\[gate\]|passed|failed|Error|error|assert" | head -40 > "$OUT/r4_a_train_full.txt"; cat "$OUT/r4_a_train_full.txt"
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_static.py --deselect tests/test_gpu_train_full.py 2>&1 | tail -40 > "$OUT/r4_a_pytest_rest.txt"; tail -8 "$OUT/r4_a_pytest_rest.txt"
timeout 900 python bench.py > "$OUT/r4_a_bench.json" 2> "$OUT/r4_a_bench.err"; tail -c 1500 "$OUT/r4_a_bench.err"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r4_a_bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
    s = d['secondary']
    print('full_forward_ms', s['full_forward_ms'], 'train', (s['train_step'] or {}).get('ms_per_step'))
    print('fresh', json.dumps(s.get('fresh_batches'))[:1500])
except Exception as e:
    print('bench parse failed', e)
PY
