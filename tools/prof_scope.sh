#!/bin/bash
# rocprofv3 kernel trace of the propagate-scope step (bench.py --only-primary); $1 = tag, $2.. = bench args
set -u
TAG=${1:-x}; shift
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rm -rf /tmp/prof_scope
rocprofv3 --kernel-trace --stats -d /tmp/prof_scope -- python "$ROOT/bench.py" --only-primary "$@" > /dev/null 2>&1
cd "$ROOT"
python profiles/summarize_rocprof.py "$(ls /tmp/prof_scope/*/*results.db | head -1)" 16 > "$OUT/${TAG}_propagate_scope.md"
cat "$OUT/${TAG}_propagate_scope.md" | cut -c1-160
