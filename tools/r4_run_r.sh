#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_blocked.py tests/test_gpu_static.py -q -x 2>&1 | tail -4
timeout 600 python tools/fuzz_kernels.py 2>&1 | tail -3
for atoms in uniform zinc; do
CWN_BENCH_ATOMS=$atoms python bench.py --brief --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$atoms', d['value'], d['ms_per_step'], d['config'].get('layer_kernel_form'))"
CWN_BENCH_ATOMS=$atoms python bench.py --brief --batch 2048 --num-batches 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$atoms 2048', d['value'], d['ms_per_step'], d['config'].get('layer_kernel_form'))"
done
