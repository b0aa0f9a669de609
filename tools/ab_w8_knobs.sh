#!/bin/bash
# Knobs of the two-per-CU form of the layer kernel, one variant library each (make variant_w8 NAME=.. FLAGS=..):
#   gpurun -- 'bash tools/ab_w8_knobs.sh "default we0 we1 we3 we4 nr1 wbar0" "512 2048"'
mkdir -p gpurun_out
for n in ${2:-512 2048}; do
  for v in ${1:-default}; do
    lib=$PWD/cwn_amd/libcwn_hip_$v.so; [ $v = default ] && lib=$PWD/cwn_amd/libcwn_hip.so
    nb=4; [ $n -ge 2048 ] && nb=1
    out=$(CWN_HIP_LIB=$lib CWN_LAYER_VARIANT=1 CWN_BLOCKED_MAX_ITEMS=100000 python bench.py --batch $n --num-batches $nb --steps 20 --warmup 3 --only-primary 2>/dev/null | tail -1)
    echo "batch $n $v $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), d['ms_per_step'])")"
  done
done | tee gpurun_out/ab_w8_knobs.txt
