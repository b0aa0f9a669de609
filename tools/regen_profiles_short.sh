#!/bin/bash
# The reduced form of regen_profiles.sh (about 2 GPU-minutes): the default bench line, the same path
# at batch 8192, and the rocprofv3 summary of the propagate scope.  $1 = tag.
set -u
TAG=${1:-x}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[gate\]|^\[invariance\]| passed| failed|^FAILED|^ERROR" > "$OUT/r2_${TAG}_pytest_gpu.txt"; tail -1 "$OUT/r2_${TAG}_pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > "$OUT/r2_${TAG}_bench_zinc.json" 2> /dev/null
CWN_BENCH_SKIP=eager,concurrent,train python bench.py --batch 8192 --num-batches 1 --steps 20 --warmup 3 --no-cpu > "$OUT/r2_${TAG}_bench_zinc_batch8192.json" 2> /dev/null
for w in zinc zinc_batch8192; do tail -1 "$OUT/r2_${TAG}_bench_$w.json" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['roofline']['kernel'][:16], d['roofline']['frac'], (d.get('roofline_other') or {}).get('frac'), (d.get('roofline_step') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))"; done
ROOT=$PWD
cd /tmp
rm -rf /tmp/prof_scope
rocprofv3 --kernel-trace --stats -d /tmp/prof_scope -- python "$ROOT/bench.py" --only-primary > /dev/null 2>&1
cd "$ROOT"
python profiles/summarize_rocprof.py "$(ls /tmp/prof_scope/*/*results.db | head -1)" 30 > "$OUT/r2_${TAG}_propagate_scope.md"
head -7 "$OUT/r2_${TAG}_propagate_scope.md" | cut -c1-140
