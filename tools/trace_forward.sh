#!/bin/bash
# One eval forward of the ZINC-128 model under rocprofv3: per-kernel totals + the dispatch timeline of the LAST forward.
#   gpurun -- 'bash tools/trace_forward.sh [batch] [K dispatches]'
export TMPDIR=/tmp
ROOT=$PWD
N=${1:-128}
K=${2:-40}
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/prof_trace
rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -- python $ROOT/tools/profile_forward.py $N > /dev/null 2>&1
cd $ROOT
python profiles/summarize_rocprof.py "$(ls /tmp/prof_trace/*/*results.db | head -1)" $K > gpurun_out/trace_forward_$N.md
tail -$((K + 1)) gpurun_out/trace_forward_$N.md | cut -c1-160
