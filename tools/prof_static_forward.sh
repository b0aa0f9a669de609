#!/bin/bash
# Kernel averages of the full forward, fixed batches against a blocked static batch over the SAME batches:
#   gpurun -- 'bash tools/prof_static_forward.sh [molhiv|zinc]'
export TMPDIR=/tmp
R=$PWD
WL=${1:-molhiv}
mkdir -p gpurun_out
for w in fixed static; do
  cd /tmp; rm -rf /tmp/psf
  rocprofv3 --kernel-trace --stats -d /tmp/psf -- python $R/tools/prof_static_forward.py $w $WL > /tmp/psf.log 2>&1
  tail -1 /tmp/psf.log
  cd $R
  python profiles/summarize_rocprof.py "$(ls /tmp/psf/*/*results.db | head -1)" 0 2>/dev/null | head -24 | cut -c1-150 > gpurun_out/psf_$w.md
done
python - <<'PY'
import re
def load(p):
    d = {}
    for ln in open(p):
        m = re.match(r'\| (.+?) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', ln)
        if m:
            name = re.sub(r'\(anonymous namespace\)::|void ', '', m.group(1))[:64]
            d[name] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    return d
a, b = load('gpurun_out/psf_fixed.md'), load('gpurun_out/psf_static.md')
print(f'{"kernel":66s} fixed: calls, avg us     static: calls, avg us    total ratio')
for k in sorted(set(a) | set(b), key=lambda k: -(b.get(k, (0, 0, 0))[1])):
    fa, fb = a.get(k, (0, 0.0, 0.0)), b.get(k, (0, 0.0, 0.0))
    print(f'{k:66s} {fa[0]:6d} {fa[2]:9.2f}     {fb[0]:6d} {fb[2]:9.2f}     {(fb[1] / fa[1]) if fa[1] else float("nan"):6.2f}')
PY
