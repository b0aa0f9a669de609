#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ends.py -q -x -k "tn or weight_grad or train" 2>&1 | tail -3
{
echo "== default"; python tools/ubench_tn24.py pro
for band in 192 256 288 320 352 384 448 512; do echo "== band $band"; CWN_TN_BAND=$band python tools/ubench_tn24.py pro; done
} 2>&1 | grep -v amdgpu.ids > "$OUT/r4_i_tn.txt"
cat "$OUT/r4_i_tn.txt"
bash tools/prof_train.sh 128 120 > /dev/null 2>&1; cp gpurun_out/prof_train_128.md gpurun_out/r4_i_train_step.md
grep -n "gemm_tn" gpurun_out/r4_i_train_step.md | sed -n 3,8p | cut -c1-140
