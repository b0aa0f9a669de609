#!/bin/bash
set -u
export TMPDIR=/tmp
for S in 8 16 32; do
CWN_BENCH_FRESH_SLOTS=$S CWN_BENCH_SKIP=eager,concurrent,collate,workloads,train timeout 600 python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['secondary']['fresh_batches']; print('S=$S', f['propagate'], f['forward'])"
done
