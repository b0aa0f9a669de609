#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_static.py tests/test_gpu_train_full.py -q -x -k "live or train or static or dense_stage" 2>&1 | tail -15
for lb in 0 1; do
CWN_LIVE_BN_BWD=$lb CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh timeout 900 python bench.py --no-cpu > "$OUT/r4_g_bench_lb$lb.json" 2> "$OUT/r4_g_bench_lb$lb.err"
python - <<PY
import json
d = json.loads(open('gpurun_out/r4_g_bench_lb$lb.json').read().strip().splitlines()[-1])
s = d['secondary']
print('live_bwd=$lb train', (s.get('train_step') or {}).get('ms_per_step'), 'value', d['value'])
PY
done
