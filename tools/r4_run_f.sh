#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_static.py tests/test_gpu_train_full.py -q -x -k "live or train or static or dense_stage" 2>&1 | tail -15
for live in 0 1; do
CWN_LIVE_BN=$live CWN_BENCH_SKIP=eager,concurrent,collate,workloads timeout 900 python bench.py --no-cpu > "$OUT/r4_f_bench_live$live.json" 2> "$OUT/r4_f_bench_live$live.err"
python - <<PY
import json
d = json.loads(open('gpurun_out/r4_f_bench_live$live.json').read().strip().splitlines()[-1])
s = d['secondary']
print('live=$live train', (s.get('train_step') or {}).get('ms_per_step'), 'value', d['value'], 'fresh', {k: (v or {}).get('ms_per_step') if isinstance(v, dict) else v for k, v in (s.get('fresh_batches') or {}).items()})
PY
done
