#!/bin/bash
# Regenerates the files under profiles/ on the GPU box (run from the repo root, results land in
# gpurun_out/ which gpurun merges back; copy them into profiles/ afterwards):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/regen_profiles.sh g'
# $1 = tag of the state being measured (profiles/r3_<tag>_*).
set -u
TAG=${1:-x}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[gate\]|^\[invariance\]| passed| failed|^FAILED|^ERROR" > "$OUT/r3_${TAG}_pytest_gpu.txt"; tail -1 "$OUT/r3_${TAG}_pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > "$OUT/r3_${TAG}_bench_zinc.json" 2> "$OUT/bench_zinc.err"
python bench.py --workload molhiv > "$OUT/r3_${TAG}_bench_molhiv.json" 2> /dev/null
python bench.py --workload reddit > "$OUT/r3_${TAG}_bench_reddit.json" 2> /dev/null
python bench.py --batch 2048 --num-batches 1 --steps 20 --warmup 3 --brief > "$OUT/r3_${TAG}_bench_zinc_batch2048.json" 2> /dev/null
python bench.py --batch 8192 --num-batches 1 --steps 20 --warmup 3 --brief > "$OUT/r3_${TAG}_bench_zinc_batch8192.json" 2> /dev/null
for w in zinc molhiv reddit zinc_batch2048 zinc_batch8192; do tail -1 "$OUT/r3_${TAG}_bench_$w.json" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], (d.get('roofline_other') or {}).get('frac'), (d.get('roofline_step') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))"; done
# the same commands under rocprofv3 (kernel trace only)
ROOT=$PWD
cd /tmp
rm -rf /tmp/prof_full /tmp/prof_scope
rocprofv3 --kernel-trace --stats -d /tmp/prof_full -- python "$ROOT/bench.py" --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/prof_scope -- python "$ROOT/bench.py" --only-primary > /dev/null 2>&1
cd "$ROOT"
python profiles/summarize_rocprof.py "$(ls /tmp/prof_full/*/*results.db | head -1)" > "$OUT/r3_${TAG}_full_bench.md"
python profiles/summarize_rocprof.py "$(ls /tmp/prof_scope/*/*results.db | head -1)" 30 > "$OUT/r3_${TAG}_propagate_scope.md"
head -8 "$OUT/r3_${TAG}_propagate_scope.md"
