#!/bin/bash
# Round-6 evidence in one call (about 6 GPU-minutes): the whole -m gpu suite, smoke(), the default bench line, the rocprofv3
# summaries of the propagate scope (ZINC-128, ZINC-2048, molhiv-512, REDDIT-32, CIN++) and of the training step, and the
# kernel averages bench.py quotes (profiles/r6_kernel_avg.json).  $1 = tag -> gpurun_out/r6_<tag>_*
set -u
TAG=${1:-x}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[gate\]|^\[invariance\]|^\[static| passed| failed|^FAILED|^ERROR|max_ring" > "$OUT/r6_${TAG}_pytest_gpu.txt"; tail -1 "$OUT/r6_${TAG}_pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
scope() {      # $1 = name, $2.. = bench args / env through `env`
  local name=$1; shift
  cd /tmp; rm -rf /tmp/prof_scope
  rocprofv3 --kernel-trace --stats -d /tmp/prof_scope -- "$@" > /dev/null 2>&1
  cd "$ROOT"
  python profiles/summarize_rocprof.py "$(ls /tmp/prof_scope/*/*results.db | head -1)" 16 > "$OUT/r6_${TAG}_propagate_scope${name}.md"
  head -4 "$OUT/r6_${TAG}_propagate_scope${name}.md" | cut -c1-150
}
scope "" python "$ROOT/bench.py" --only-primary
scope "_zinc2048" python "$ROOT/bench.py" --only-primary --batch 2048 --num-batches 1 --steps 40 --warmup 5
scope "_molhiv" python "$ROOT/bench.py" --only-primary --workload molhiv
scope "_reddit" python "$ROOT/bench.py" --only-primary --workload reddit
scope "_cinpp" env CWN_BENCH_MODEL=cinpp python "$ROOT/bench.py" --only-primary
python tools/collect_kernel_avg.py "zinc:128:128=$OUT/r6_${TAG}_propagate_scope.md" "zinc:2048:128=$OUT/r6_${TAG}_propagate_scope_zinc2048.md" \
  "molhiv:512:64=$OUT/r6_${TAG}_propagate_scope_molhiv.md" "reddit:32:64=$OUT/r6_${TAG}_propagate_scope_reddit.md" \
  "zinc_cinpp:128:128=$OUT/r6_${TAG}_propagate_scope_cinpp.md"
cp profiles/r6_kernel_avg.json "$OUT/r6_kernel_avg.json"
bash tools/prof_train.sh 128 200 > /dev/null 2>&1; cp gpurun_out/prof_train_128.md "$OUT/r6_${TAG}_train_step.md"
head -8 "$OUT/r6_${TAG}_train_step.md" | cut -c1-140
SECONDS=0
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r6_${TAG}_bench_line.json" 2> "$OUT/r6_${TAG}_bench_zinc.err"; echo "bench wall ${SECONDS}s, line $(wc -c < "$OUT/r6_${TAG}_bench_line.json") bytes"
cp bench_detail.json "$OUT/r6_${TAG}_bench_zinc.json"
python - <<PY
import json
d = json.loads(open('$OUT/r6_${TAG}_bench_line.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('frac', 'avg_launch_us', 'avg_launch_us_rocprof', 'frac_rocprof', 'traffic')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'legs', d.get('leg_seconds'))
print(json.dumps(d['secondary'])[:1500])
PY
