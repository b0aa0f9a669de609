#!/bin/bash
# Kernel timeline of the full eval forward of a workload:  bash tools/prof_forward_wl.sh <workload> [K dispatches]
export TMPDIR=/tmp
ROOT=$PWD
WL=${1:-reddit}; K=${2:-60}
cd /tmp && rm -rf /tmp/prof_fwd_$WL
CWN_BENCH_SKIP=eager,concurrent,train,collate,workloads,fresh,roofline rocprofv3 --kernel-trace --stats -d /tmp/prof_fwd_$WL -- python $ROOT/bench.py --workload $WL --no-cpu --steps 8 --warmup 2 --kernel-reps 4 > /dev/null 2>&1
cd $ROOT
python profiles/summarize_rocprof.py "$(ls /tmp/prof_fwd_$WL/*/*results.db | head -1)" $K > gpurun_out/prof_fwd_$WL.md
