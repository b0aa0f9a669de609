#!/bin/bash
# rocprofv3 kernel summary of the EmbedCINpp training step (tools/time_cinpp_model.py, the fused form alone) -> gpurun_out/r4_cinpp_step.md
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p "$ROOT/gpurun_out"
cd /tmp; rm -rf /tmp/prof_pp
CWN_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_pp -- python "$ROOT/tools/time_cinpp_model.py" "$@" > /tmp/pp.log 2>&1
cd "$ROOT"
python profiles/summarize_rocprof.py "$(ls /tmp/prof_pp/*/*results.db | head -1)" 260 > gpurun_out/r4_cinpp_step.md
tail -1 /tmp/pp.log
head -40 gpurun_out/r4_cinpp_step.md | cut -c1-150
