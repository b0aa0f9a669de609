#!/bin/bash
# Blocked layer kernel vs CSR path over batch sizes (ZINC-like, hidden 128): where BLOCKED_MAX_ITEMS belongs.
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']/1e6,1), 'M cells/s', d['ms_per_step'], 'ms')"; }
for b in ${BATCHES:-256 512 1024 2048 4096 8192}; do
  CWN_BLOCKED_MAX_ITEMS=1000000 python bench.py --only-primary --no-cpu --batch $b --num-batches 1 --steps 30 --warmup 5 2>/dev/null | line "batch $b blocked"
  CWN_BLOCKED_LAYER=0 python bench.py --only-primary --no-cpu --batch $b --num-batches 1 --steps 30 --warmup 5 2>/dev/null | line "batch $b csr    "
done
