#!/bin/bash
set -u
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" "$PWD/cwn_amd/libcwn_hip_preload.so"; do
  CWN_HIP_LIB=$lib python bench.py --only-primary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=[$lib]', d['value'], d['ms_per_step'])"
done
done
