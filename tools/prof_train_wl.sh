#!/bin/bash
# Kernel mix of the graph-captured training step of a workload:  bash tools/prof_train_wl.sh <batch> <workload> <dropout> <tag> [K]
export TMPDIR=/tmp
ROOT=$PWD
N=${1:-512}; WL=${2:-molhiv}; DROP=${3:-0.5}; TAG=${4:-molhiv}; K=${5:-200}
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/prof_train_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_train_$TAG -- python $ROOT/tools/train_graph.py $N 30 $WL $DROP > /tmp/prof_train_$TAG.log 2>&1
cd $ROOT
tail -1 /tmp/prof_train_$TAG.log
python profiles/summarize_rocprof.py "$(ls /tmp/prof_train_$TAG/*/*results.db | head -1)" $K > gpurun_out/prof_train_$TAG.md
