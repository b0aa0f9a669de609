#!/bin/bash
# A/B of two builds of the library on the full forward and the training step of a workload (graph replay):
#   bash tools/ab_libs_forward.sh <variant A> <variant B> ["workload batch" ...]      (names as tools/ab_layer_libs.sh)
ROOT=$PWD
A=${1:-default}; B=${2:-base}; shift 2
[ $# -eq 0 ] && set -- "zinc 128" "molhiv 512"
for CFG in "$@"; do set -- $CFG
  for LIB in $A $B $A $B; do
    if [ $LIB = default ]; then unset CWN_HIP_LIB; else export CWN_HIP_LIB=$ROOT/cwn_amd/libcwn_hip_$LIB.so; fi
    CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh CWN_BENCH_DETAIL=/tmp/ab_detail.json python bench.py --workload $1 --batch $2 --no-cpu 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); s=j['secondary']; print('$1-$2 lib=$LIB', 'propagate', j['ms_per_step'], 'forward', s.get('full_forward_ms'), s.get('forward_breakdown_us'), 'train', s.get('train_step_ms'))"
  done
done
