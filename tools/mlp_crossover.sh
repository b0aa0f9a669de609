#!/bin/bash
# Fused update / combine launch (csrc/cwn_mlp.hip) vs the three grouped GEMM launches: full forward over batch sizes.
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['secondary']['full_forward_ms'], 'ms')"; }
for b in ${BATCHES:-128 256 512 1024 2048}; do
  for f in 1 0; do
    CWN_FUSED_UPDATE_MLP=$f CWN_BENCH_SKIP=eager,concurrent,train python bench.py --no-cpu --batch $b --num-batches 1 --steps 8 --warmup 2 --kernel-reps 4 2>/dev/null | line "batch $b fused=$f"
  done
done
