#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_timing.so python tools/time_layer_span.py 128 128 > "$OUT/r4_layer_span.md" 2> "$OUT/r4_layer_span.err"; cat "$OUT/r4_layer_span.md"; tail -3 "$OUT/r4_layer_span.err"
CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_timing.so MODE=2 python tools/time_layer_phases.py 128 128 > "$OUT/r4_layer_phases.txt" 2>&1; tail -40 "$OUT/r4_layer_phases.txt"
for lib in "" "$PWD/cwn_amd/libcwn_hip_preload.so"; do
  CWN_HIP_LIB=$lib python bench.py --only-primary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'))"
done
