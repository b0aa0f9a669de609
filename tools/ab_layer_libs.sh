#!/bin/bash
# A/B of two builds of the library on the propagate scope (graph replay: HIP events + rocprofv3 kernel averages):
#   bash tools/ab_layer_libs.sh <name of variant A> <name of variant B> ["workload batch atoms" ...]
# a name is `default` (cwn_amd/libcwn_hip.so) or X for cwn_amd/libcwn_hip_X.so (make -C cwn_amd/csrc variant NAME=X FLAGS=...,
# built BEFORE the gpurun call: in-tree .so files travel).
export TMPDIR=/tmp
ROOT=$PWD
A=${1:-default}; B=${2:-base}; shift 2
[ $# -eq 0 ] && set -- "zinc 128 18,30" "zinc 128 zinc" "zinc 512 zinc" "zinc 2048 18,30"
for CFG in "$@"; do set -- $CFG
  for LIB in $A $B $A $B; do
    if [ $LIB = default ]; then unset CWN_HIP_LIB; else export CWN_HIP_LIB=$ROOT/cwn_amd/libcwn_hip_$LIB.so; fi
    CWN_BENCH_ATOMS=$3 python bench.py --only-primary --workload $1 --batch $2 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$1-$2 atoms=$3 lib=$LIB', 'ms_per_step', j['ms_per_step'], 'cells/s', round(j['value']/1e6,1), 'M')"
  done
  for LIB in $A $B; do
    if [ $LIB = default ]; then unset CWN_HIP_LIB; else export CWN_HIP_LIB=$ROOT/cwn_amd/libcwn_hip_$LIB.so; fi
    cd /tmp; rm -rf /tmp/prof_ab
    CWN_BENCH_ATOMS=$3 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -- python $ROOT/bench.py --only-primary --workload $1 --batch $2 > /dev/null 2>&1
    cd $ROOT
    echo "rocprof lib=$LIB: $(python profiles/summarize_rocprof.py "$(ls /tmp/prof_ab/*/*results.db | head -1)" 4 | grep layer_kernel | cut -d'|' -f2-5 | sed 's/void (anonymous namespace):://; s/((anonymous namespace)::LayerArgs)//' | tr '\n' ';')"
  done
done
