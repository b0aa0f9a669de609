#!/bin/bash
set -u
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_full.py tests/test_gpu_ends.py -q -x -k "live or fused_training or train_full or training_step or front_backward or dense_stage" 2>&1 | tail -4
bash tools/prof_train.sh 128 10 > /dev/null 2>&1
grep -E 'dense_stage|norm_kernel<4, 0>|front_bwd' gpurun_out/prof_train_128.md | head -4 | cut -d'|' -f2,3,5
CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh timeout 900 python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['secondary']['train_step']['ms_per_step'])"
