#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "live or dense_stage or fused_training" 2>&1 | tail -3
bash tools/prof_train.sh 128 140 > /dev/null 2>&1; cp gpurun_out/prof_train_128.md gpurun_out/r4_k_train_step.md
head -12 gpurun_out/r4_k_train_step.md | cut -c1-150
CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh timeout 900 python bench.py --no-cpu > "$OUT/r4_k_bench.json" 2> "$OUT/r4_k_bench.err"
python - <<PY
import json
d = json.loads(open('gpurun_out/r4_k_bench.json').read().strip().splitlines()[-1])
s = d['secondary']
print('train', (s.get('train_step') or {}).get('ms_per_step'), 'value', d['value'], 'full fwd', s.get('full_forward_ms'))
PY
