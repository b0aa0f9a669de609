#!/bin/bash
# A/B of layer-kernel variants on one box: blocked-kernel tests on the default library, then the headline
# step and the phase stamps for each variant library given as arguments (names as in `make variant NAME=..`).
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
python -m pytest tests/test_gpu_blocked.py tests/test_gpu_invariance.py -x -q 2>&1 | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
python bench.py --only-primary --no-cpu 2>/dev/null | line default
for v in "$@"; do
  CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_$v.so python bench.py --only-primary --no-cpu 2>/dev/null | line "$v"
done
python bench.py --only-primary --no-cpu --workload molhiv 2>/dev/null | line default-molhiv
for v in "$@"; do
  CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_$v.so python bench.py --only-primary --no-cpu --workload molhiv 2>/dev/null | line "$v-molhiv"
done
for m in ${PHASE_MODES:-2 1}; do
  for lib in timing ${TIMING_LIBS:-}; do
    echo "== phases lib=$lib MODE=$m"
    MODE=$m CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_$lib.so python tools/time_layer_phases.py 128 128 2>&1 | tail -40
  done
done
