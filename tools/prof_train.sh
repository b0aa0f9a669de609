#!/bin/bash
# Kernel mix + dispatch timeline of the graph-captured training step (cwn_amd.train.TrainStep, ZINC-128):
#   gpurun -- 'bash tools/prof_train.sh [batch] [K dispatches]'
export TMPDIR=/tmp
ROOT=$PWD
N=${1:-128}
K=${2:-330}
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/prof_train
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python $ROOT/tools/train_graph.py $N 30 > /tmp/prof_train.log 2>&1
cd $ROOT
tail -2 /tmp/prof_train.log
python profiles/summarize_rocprof.py "$(ls /tmp/prof_train/*/*results.db | head -1)" $K > gpurun_out/prof_train_$N.md
head -45 gpurun_out/prof_train_$N.md | cut -c1-150
