#!/bin/bash
# A/B of the two forms of the blocked layer kernel over batch sizes (propagate scope, graph replay):
#   gpurun -- 'bash tools/ab_variant.sh "128 512 2048 8192" [zinc|molhiv] [variants]'
mkdir -p gpurun_out
WL=${2:-zinc}
for n in ${1:-128 512 2048 8192}; do
  for v in ${3:-0 1 csr}; do
    if [ $v = csr ]; then env="CWN_BLOCKED_LAYER=0"; else env="CWN_LAYER_VARIANT=$v CWN_BLOCKED_MAX_ITEMS=100000"; fi
    nb=4; [ $n -ge 2048 ] && nb=1
    out=$(env $env python bench.py --workload $WL --batch $n --num-batches $nb --steps 20 --warmup 3 --only-primary 2>/dev/null | tail -1)
    echo "$WL $n $v $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['config'].get('layer_kernel_form') or {}; print(round(d['value']/1e6,1), d['ms_per_step'], f.get('items_per_launch'), d['config']['layer_kernel'][:16])")"
  done
done | tee -a gpurun_out/ab_variant.txt
