#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_static.py -q -rP 2>&1 | tail -150 > "$OUT/r4_b_static.txt"; grep -E "passed|failed|static train" "$OUT/r4_b_static.txt" | tail -8
timeout 600 python -m pytest tests/test_gpu_train_full.py -q -rP 2>&1 | grep -E "^\[gate\]|passed|failed|Error|error|assert" | head -40 > "$OUT/r4_b_train_full.txt"; cat "$OUT/r4_b_train_full.txt"
CWN_BENCH_SKIP=full,eager,concurrent,collate,workloads timeout 900 python bench.py --no-cpu > "$OUT/r4_b_bench.json" 2> "$OUT/r4_b_bench.err"; tail -c 2500 "$OUT/r4_b_bench.err"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r4_b_bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
    s = d['secondary']
    print('full_forward_ms', s['full_forward_ms'], 'train', (s['train_step'] or {}).get('ms_per_step'))
    print('fresh', json.dumps(s.get('fresh_batches'), indent=1)[:3000])
except Exception as e:
    print('bench parse failed', e)
PY
ROOT=$PWD
cd /tmp && rm -rf /tmp/prof_fresh
CWN_BENCH_SKIP=full,eager,concurrent,train,collate,workloads,roofline CWN_BENCH_FRESH_EPOCHS=2 rocprofv3 --kernel-trace --stats -d /tmp/prof_fresh -- python $ROOT/bench.py --no-cpu > /tmp/prof_fresh.log 2>&1
cd $ROOT
python profiles/summarize_rocprof.py "$(ls /tmp/prof_fresh/*/*results.db | head -1)" 120 > "$OUT/r4_b_fresh_profile.md"
head -40 "$OUT/r4_b_fresh_profile.md" | cut -c1-170
