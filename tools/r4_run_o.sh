#!/bin/bash
set -u
export TMPDIR=/tmp
for dbg in 0 8 1; do
CWN_STAGE_DBG=$dbg bash tools/prof_train.sh 128 10 > /dev/null 2>&1
echo "dbg=$dbg: $(grep -E 'dense_stage_kernel|norm_kernel<4, 0>' gpurun_out/prof_train_128.md | head -2 | cut -d'|' -f2,5 | tr '\n' ' ')"
done
