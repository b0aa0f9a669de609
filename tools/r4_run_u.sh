#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm or bnb or norm_apply" 2>&1 | tail -2
{
echo "A/B of the gemm_kernel launch bounds (CWN_GEMM_NOSPILL): training steps with CWN_STAGE_KERNEL=0, i.e. the update / combine"
echo "networks on cwn_gemm_f32 (prologue, transposed weight, BatchNorm-backward prologue) instead of cwn_dense_stage*_f32"
for lib in "" "$PWD/cwn_amd/libcwn_hip_gemmspill.so"; do
  tag=$([ -z "$lib" ] && echo "no spills (one workgroup per SIMD pair for the 7 instantiations)" || echo "round-3 bounds (6 - 51 VGPRs in scratch)")
  echo "== $tag"
  CWN_HIP_LIB=$lib CWN_STAGE_KERNEL=0 python tools/train_graph.py 512 30 molhiv 2>&1 | tail -1
  CWN_HIP_LIB=$lib CWN_STAGE_KERNEL=0 python tools/train_graph.py 128 30 2>&1 | tail -1
  CWN_HIP_LIB=$lib python bench.py --workload reddit --brief --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reddit-32 propagate scope', d['value'], d['ms_per_step'])"
done
} > "$OUT/r4_gemm_spills.txt" 2>&1
cat "$OUT/r4_gemm_spills.txt"
