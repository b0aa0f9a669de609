#!/bin/bash
# Batches with the molecule-size spread of the real ZINC subset (9 - 38 atoms): mixed launch / 16-wave form with big items / two-kernel path
mkdir -p gpurun_out
for n in ${1:-128 512 2048}; do
  for v in auto 0 csr; do
    if [ $v = csr ]; then env="CWN_BLOCKED_LAYER=0"; else env="CWN_LAYER_VARIANT=$v CWN_BLOCKED_MAX_ITEMS=100000"; fi
    nb=4; [ $n -ge 2048 ] && nb=1
    out=$(env CWN_BENCH_ATOMS=${2:-9,38} $env python bench.py --batch $n --num-batches $nb --steps 20 --warmup 3 --only-primary 2>/dev/null | tail -1)
    echo "atoms ${2:-9,38} batch $n $v $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config'].get('layer_kernel_form') or {}; print(round(d['value']/1e6,1), d['ms_per_step'], f.get('variant'), f.get('items_per_launch'), f.get('big_items'), d['config']['layer_kernel'][:16])")"
  done
done | tee -a gpurun_out/ab_mixed.txt
