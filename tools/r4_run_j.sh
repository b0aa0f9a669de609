#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_static.py tests/test_gpu_train_full.py tests/test_gpu_ends.py -q -x -k "live or train or static or dense_stage or pack or adam or loss" 2>&1 | tail -5
bash tools/prof_train.sh 128 140 > /dev/null 2>&1; cp gpurun_out/prof_train_128.md gpurun_out/r4_j_train_step.md
head -34 gpurun_out/r4_j_train_step.md | cut -c1-150
for det in 0 1; do
CWN_DETERMINISTIC_TN=$det CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh timeout 900 python bench.py --no-cpu > "$OUT/r4_j_bench_det$det.json" 2> "$OUT/r4_j_bench_det$det.err"
python - <<PY
import json
d = json.loads(open('gpurun_out/r4_j_bench_det$det.json').read().strip().splitlines()[-1])
s = d['secondary']
print('det=$det train', (s.get('train_step') or {}).get('ms_per_step'), 'value', d['value'], 'full fwd', s.get('full_forward_ms'))
PY
done
