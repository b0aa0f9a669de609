export TMPDIR=/tmp
ROOT=$PWD
for d in 0 1; do
  cd /tmp && rm -rf /tmp/pm$d
  CWN_FUSED_FRONT_TRAINING=$d rocprofv3 --kernel-trace --stats -d /tmp/pm$d -- python $ROOT/tools/train_graph.py 512 20 molhiv > /tmp/pm$d.log 2>&1
  cd $ROOT
  python profiles/summarize_rocprof.py "$(ls /tmp/pm$d/*/*results.db | head -1)" 5 > gpurun_out/pm$d.md
done
