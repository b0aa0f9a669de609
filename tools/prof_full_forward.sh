export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && rm -rf /tmp/prof_fwd
CWN_BENCH_SKIP=eager,concurrent,train rocprofv3 --kernel-trace --stats -d /tmp/prof_fwd -- python $ROOT/bench.py --no-cpu --steps 8 --warmup 2 --kernel-reps 4 > /dev/null 2>&1
cd $ROOT
python profiles/summarize_rocprof.py "$(ls /tmp/prof_fwd/*/*results.db | head -1)" 60 > gpurun_out/prof_fwd.md
head -30 gpurun_out/prof_fwd.md | cut -c1-150
