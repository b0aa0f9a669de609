#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/proto/tr_read_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe > "$OUT/r4_c_tr_probe.txt"; head -20 "$OUT/r4_c_tr_probe.txt"
timeout 600 python -m pytest tests/test_gpu_static.py tests/test_gpu_blocked.py -q -k "static or eighteen" -rP 2>&1 | tail -60 > "$OUT/r4_c_static.txt"; grep -E "passed|failed|static train|max_ring" "$OUT/r4_c_static.txt" | tail -20
SECONDS=0
timeout 1200 python bench.py > "$OUT/r4_c_bench.json" 2> "$OUT/r4_c_bench.err"; echo "bench wall ${SECONDS}s"; tail -c 800 "$OUT/r4_c_bench.err"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r4_c_bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
    s = d['secondary']
    print('full_forward_ms', s['full_forward_ms'], 'train', (s['train_step'] or {}).get('ms_per_step'))
    fb = s.get('fresh_batches') or {}
    print('fresh', {k: fb.get(k) for k in ('propagate', 'forward', 'train', 'every_batch_within_capacity', 'device_error_word')})
    for k, v in (s.get('workloads') or {}).items():
        print(k, v.get('value'), v.get('ms_per_step'), (v.get('layer_kernel_form') or {}), v.get('failed'))
    print('multi', json.dumps(d.get('multi_gpu'))[:400])
except Exception as e:
    print('bench parse failed', e)
PY
