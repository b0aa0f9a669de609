#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
{
for det in 0 1; do for sp in 0 1; do
  echo "== det=$det split=$sp"; CWN_DETERMINISTIC_TN=$det CWN_TN_SPLIT=$sp python tools/ubench_tn24.py pro
done; done
for band in 192 256 320 384 512; do echo "== det split band $band"; CWN_DETERMINISTIC_TN=1 CWN_TN_BAND=$band python tools/ubench_tn24.py pro; done
} > "$OUT/r4_e_tn.txt" 2>&1
grep -v amdgpu.ids "$OUT/r4_e_tn.txt"
timeout 600 python -m pytest tests/test_gpu_static.py -q -x 2>&1 | tail -3
for det in 0 1; do
CWN_DETERMINISTIC_TN=$det CWN_BENCH_SKIP=eager,concurrent,collate,fresh timeout 900 python bench.py --no-cpu > "$OUT/r4_e_bench_det$det.json" 2> "$OUT/r4_e_bench_det$det.err"
python - <<PY
import json
d = json.loads(open('gpurun_out/r4_e_bench_det$det.json').read().strip().splitlines()[-1])
s = d['secondary']
print('det=$det train', (s.get('train_step') or {}).get('ms_per_step'), 'value', d['value'])
for k, v in (s.get('workloads') or {}).items():
    print(k, v.get('value'), v.get('ms_per_step'), (v.get('layer_kernel_form') or {}), (v.get('failed') or '')[:300])
PY
done
