#!/bin/bash
# Kernel mix of the never-seen-batch legs of bench.py (secondary.fresh_batches: fill launches + the scopes they feed):
#   gpurun -- 'bash tools/prof_fresh.sh'  ->  gpurun_out/prof_fresh.md
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/prof_fresh
CWN_BENCH_SKIP=full,eager,concurrent,train,collate,workloads,roofline CWN_BENCH_FRESH_EPOCHS=2 rocprofv3 --kernel-trace --stats -d /tmp/prof_fresh -- python $ROOT/bench.py --no-cpu "$@" > /tmp/prof_fresh.log 2>&1
cd $ROOT
python profiles/summarize_rocprof.py "$(ls /tmp/prof_fresh/*/*results.db | head -1)" 60 > gpurun_out/prof_fresh.md
head -14 gpurun_out/prof_fresh.md | cut -c1-150
