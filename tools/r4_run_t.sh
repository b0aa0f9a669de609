#!/bin/bash
set -u
export TMPDIR=/tmp
CWN_TN_ROLES=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ends.py -q -x -k "tn or weight_grad or train" 2>&1 | tail -3
for r in 0 1; do echo "roles=$r: $(CWN_TN_ROLES=$r python tools/ubench_tn24.py pro 2>&1 | tail -1)"; echo "roles=$r det: $(CWN_DETERMINISTIC_TN=1 CWN_TN_ROLES=$r python tools/ubench_tn24.py pro 2>&1 | tail -1)"; done
for band in 256 320 384 448; do echo "roles=1 band $band: $(CWN_TN_BAND=$band CWN_TN_ROLES=1 python tools/ubench_tn24.py pro 2>&1 | tail -1)"; done
echo "roles=1 no output: $(CWN_TN_DBG=1 CWN_TN_ROLES=1 python tools/ubench_tn24.py pro 2>&1 | tail -1)"
for r in 0 1; do
CWN_TN_ROLES=$r CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh timeout 900 python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('roles=$r train', d['secondary']['train_step']['ms_per_step'])"
done
