#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ends.py -q -x -k "tn or weight_grad or train" 2>&1 | tail -4
{
echo "== default"; python tools/ubench_tn24.py pro
for band in 128 192 256 320 384 512 640 768 1024; do echo "== band $band"; CWN_TN_BAND=$band python tools/ubench_tn24.py pro; done
for dbg in 1 2 3; do echo "== dbg $dbg"; CWN_TN_DBG=$dbg python tools/ubench_tn24.py pro; done
echo "== det"; CWN_DETERMINISTIC_TN=1 python tools/ubench_tn24.py pro
} 2>&1 | grep -v amdgpu.ids > "$OUT/r4_h_tn.txt"
cat "$OUT/r4_h_tn.txt"
bash tools/prof_train.sh 128 120 > /dev/null 2>&1; cp gpurun_out/prof_train_128.md gpurun_out/r4_h_train_step.md
grep -n "gemm_tn" gpurun_out/r4_h_train_step.md | head -12 | cut -c1-140
