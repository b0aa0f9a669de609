#!/bin/bash
# The persistent form of the two-per-CU layer kernel against one workgroup per item, and its grid (workgroups per CU):
#   gpurun -- 'bash tools/ab_persist.sh "base p202" "2 4 16" "512 2048"'
mkdir -p gpurun_out
for n in ${3:-2048}; do
  for v in ${1:-default}; do
    for per in ${2:-2}; do
      lib=$PWD/cwn_amd/libcwn_hip_$v.so; [ $v = default ] && lib=$PWD/cwn_amd/libcwn_hip.so
      nb=4; [ $n -ge 2048 ] && nb=1
      out=$(CWN_LAYER_PERSIST_PER_CU=$per CWN_HIP_LIB=$lib CWN_LAYER_VARIANT=1 CWN_BLOCKED_MAX_ITEMS=100000 python bench.py --batch $n --num-batches $nb --steps 20 --warmup 3 --only-primary 2>/dev/null | tail -1)
      echo "batch $n $v per_cu=$per $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), d['ms_per_step'])")"
      [ $v = base ] && break
    done
  done
done | tee gpurun_out/ab_persist.txt
