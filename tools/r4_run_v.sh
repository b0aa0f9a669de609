#!/bin/bash
set -u
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && rm -rf /tmp/pmc_tn
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_tn -- python $ROOT/tools/ubench_tn24.py pro > /tmp/pmc_tn.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('/tmp/pmc_tn/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        a = acc[k][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
for k, v in acc.items():
    if 'gemm_tn' in k:
        c, a = v['SQ_LDS_BANK_CONFLICT'], v['SQ_LDS_IDX_ACTIVE']
        print(k, 'conflict cycles/launch', c[0] / max(c[1], 1), 'active', a[0] / max(a[1], 1), 'ratio', (c[0] / max(c[1], 1)) / max(a[0] / max(a[1], 1), 1))
PY
tail -2 /tmp/pmc_tn.log
