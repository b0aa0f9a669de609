#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ends.py tests/test_gpu_parity.py tests/test_gpu_static.py tests/test_gpu_train_full.py -q -x -k "front or embed or train or static" 2>&1 | tail -5
CWN_BENCH_SKIP=eager,concurrent,collate,workloads timeout 900 python bench.py --no-cpu > "$OUT/r4_n_bench.json" 2> "$OUT/r4_n_bench.err"
python - <<PY
import json
d = json.loads(open('gpurun_out/r4_n_bench.json').read().strip().splitlines()[-1])
s = d['secondary']
print('train', (s.get('train_step') or {}).get('ms_per_step'), 'fresh train', ((s.get('fresh_batches') or {}).get('train') or {}).get('ms_per_step'))
PY
