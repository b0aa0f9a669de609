#!/bin/bash
# One never-seen-batch leg alone under rocprofv3 (the fixed-batch primary cut to a handful of launches):
#   gpurun -- 'bash tools/prof_fresh_leg.sh propagate --workload molhiv'  ->  gpurun_out/prof_fresh_<leg>.md
export TMPDIR=/tmp
ROOT=$PWD
LEG=$1; shift
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/prof_fresh_leg
CWN_BENCH_FRESH_LEGS=$LEG CWN_BENCH_SKIP=full,eager,concurrent,train,collate,workloads,roofline CWN_BENCH_FRESH_EPOCHS=4 rocprofv3 --kernel-trace --stats -d /tmp/prof_fresh_leg -- python $ROOT/bench.py --no-cpu --steps 4 --warmup 1 "$@" > /tmp/prof_fresh_leg.log 2>&1
cd $ROOT
python profiles/summarize_rocprof.py "$(ls /tmp/prof_fresh_leg/*/*results.db | head -1)" ${PROF_K:-60} > gpurun_out/prof_fresh_$LEG.md
head -24 gpurun_out/prof_fresh_$LEG.md | cut -c1-160
tail -3 /tmp/prof_fresh_leg.log | cut -c1-600
