#!/bin/bash
# Round-5 evidence in one call (about 8 GPU-minutes): the whole -m gpu suite, smoke(), the default bench line, the rocprofv3
# summaries of the propagate scope and of the training steps (ZINC-128; molhiv-512 with dropout 0.5), the PMC traffic of the layer
# kernel, the bench line with the data-parallel form forced over a world-1 RCCL group.  $1 = tag -> gpurun_out/r5_<tag>_*
set -u
TAG=${1:-x}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[gate\]|^\[invariance\]|^\[static|^\[router\]|^\[rccl| passed| failed|^FAILED|^ERROR|max_ring" > "$OUT/r5_${TAG}_pytest_gpu.txt"; tail -1 "$OUT/r5_${TAG}_pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
SECONDS=0
timeout 1200 python bench.py > "$OUT/r5_${TAG}_bench_zinc.json" 2> "$OUT/r5_${TAG}_bench_zinc.err"; echo "bench wall ${SECONDS}s"
python - <<PY
import json
d = json.loads(open('$OUT/r5_${TAG}_bench_zinc.json').read().strip().splitlines()[-1])
s = d['secondary']
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
print('full_forward_ms', s['full_forward_ms'], 'train', (s['train_step'] or {}).get('ms_per_step'))
fb = s.get('fresh_batches') or {}
print('fresh', {k: fb.get(k) for k in ('propagate', 'forward', 'train', 'fill', 'every_batch_within_capacity', 'device_error_word')})
for k, v in (s.get('workloads') or {}).items():
    print(k, v.get('value'), v.get('ms_per_step'), 'fwd', v.get('full_forward_ms'), 'train', v.get('train_step_ms'), (v.get('failed') or '')[:200])
PY
SECONDS=0
CWN_BENCH_FORCE_DP=1 CWN_BENCH_TRAIN_GRAPH=1 CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh timeout 600 python bench.py --no-cpu > "$OUT/r5_${TAG}_bench_forced_dp.json" 2> "$OUT/r5_${TAG}_bench_forced_dp.err"; echo "forced-dp bench wall ${SECONDS}s"
python - <<PY
import json
try:
    d = json.loads([l for l in open('$OUT/r5_${TAG}_bench_forced_dp.json').read().strip().splitlines() if l.startswith('{')][-1])
    print('multi_gpu', json.dumps(d['multi_gpu'])[:900])
except Exception as e:
    print('forced dp: no line', e)
PY
ROOT=$PWD
cd /tmp
rm -rf /tmp/prof_scope
rocprofv3 --kernel-trace --stats -d /tmp/prof_scope -- python "$ROOT/bench.py" --only-primary > /dev/null 2>&1
cd "$ROOT"
python profiles/summarize_rocprof.py "$(ls /tmp/prof_scope/*/*results.db | head -1)" 30 > "$OUT/r5_${TAG}_propagate_scope.md"
head -6 "$OUT/r5_${TAG}_propagate_scope.md" | cut -c1-140
for WL in molhiv reddit; do
  cd /tmp; rm -rf /tmp/prof_scope_$WL
  rocprofv3 --kernel-trace --stats -d /tmp/prof_scope_$WL -- python "$ROOT/bench.py" --workload $WL --only-primary --no-cpu > /dev/null 2>&1
  cd "$ROOT"
  python profiles/summarize_rocprof.py "$(ls /tmp/prof_scope_$WL/*/*results.db | head -1)" 12 > "$OUT/r5_${TAG}_propagate_scope_$WL.md"
  head -4 "$OUT/r5_${TAG}_propagate_scope_$WL.md" | cut -c1-140
done
bash tools/prof_forward_wl.sh reddit 50 > /dev/null 2>&1; cp gpurun_out/prof_fwd_reddit.md "$OUT/r5_${TAG}_forward_reddit.md"
bash tools/prof_train_wl.sh 32 reddit 0.0 reddit 120 > /dev/null 2>&1; cp gpurun_out/prof_train_reddit.md "$OUT/r5_${TAG}_train_step_reddit.md"
bash tools/prof_train.sh 128 200 > /dev/null 2>&1; cp gpurun_out/prof_train_128.md "$OUT/r5_${TAG}_train_step.md"
head -8 "$OUT/r5_${TAG}_train_step.md" | cut -c1-140
bash tools/prof_train_wl.sh 512 molhiv 0.5 molhiv_drop 160 > /dev/null 2>&1; cp gpurun_out/prof_train_molhiv_drop.md "$OUT/r5_${TAG}_train_step_molhiv_dropout.md"
grep -c "dropout\|Dropout" "$OUT/r5_${TAG}_train_step_molhiv_dropout.md" | sed 's/^/at::native dropout kernels in the molhiv step: /'
python tools/collect_traffic.py 128 2048 molhiv:512 reddit:32 > /dev/null 2>&1 && cp profiles/r5_traffic.json profiles/r5_pmc_fetch_write_raw.json "$OUT/" && cat profiles/r5_traffic.json | head -30
