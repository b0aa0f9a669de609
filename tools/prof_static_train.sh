#!/bin/bash
# Kernel averages of the REDDIT-like training step, fixed batches against the csr-mode static batch over the SAME four batches:
#   gpurun -- 'bash tools/prof_static_train.sh'   ->  gpurun_out/pst_fixed.md, pst_static.md
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
for w in fixed static; do
  cd /tmp; rm -rf /tmp/pst
  rocprofv3 --kernel-trace --stats -d /tmp/pst -- python $R/tools/prof_static_train.py $w > /tmp/pst.log 2>&1
  cd $R
  python profiles/summarize_rocprof.py "$(ls /tmp/pst/*/*results.db | head -1)" 0 | head -24 | cut -c1-150 > gpurun_out/pst_$w.md
done
python - <<'PY'
import re
def load(p):
    d = {}
    for ln in open(p):
        m = re.match(r'\| (.+?) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', ln)
        if m:
            name = re.sub(r'\(anonymous namespace\)::|void ', '', m.group(1))[:60]
            d[name] = (int(m.group(2)), float(m.group(4)))
    return d
a, b = load('gpurun_out/pst_fixed.md'), load('gpurun_out/pst_static.md')
print(f'{"kernel":62s} fixed avg us   static avg us   ratio')
for k in a:
    if k in b:
        print(f'{k:62s} {a[k][1]:10.2f} {b[k][1]:14.2f} {b[k][1] / a[k][1]:10.2f}')
PY
