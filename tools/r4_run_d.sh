#!/bin/bash
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ends.py tests/test_gpu_static.py -q -x -k "tn or weight_grad or train or static" 2>&1 | tail -15
{
for sp in 0 1; do
  echo "== CWN_TN_SPLIT=$sp"
  CWN_TN_SPLIT=$sp python tools/ubench_tn24.py
  CWN_TN_SPLIT=$sp python tools/ubench_tn24.py pro
done
for band in 128 192 256 320 384 448 512 640; do
  echo "== split band $band"; CWN_TN_BAND=$band python tools/ubench_tn24.py pro
done
for dbg in 1 2 4 3; do echo "== split dbg $dbg"; CWN_TN_DBG=$dbg python tools/ubench_tn24.py pro; done
} > "$OUT/r4_d_tn.txt" 2>&1
cat "$OUT/r4_d_tn.txt"
