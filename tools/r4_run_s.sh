#!/bin/bash
set -u
export TMPDIR=/tmp
for t in 128 171 256 512; do
CWN_LAYER_TARGET_ITEMS=$t python bench.py --workload molhiv --brief --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('molhiv target $t', d['value'], d['ms_per_step'], (d['config'].get('layer_kernel_form') or {}).get('items_per_launch'))"
done
for t in 128 256; do
CWN_LAYER_TARGET_ITEMS=$t python bench.py --brief --batch 256 --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zinc256 target $t', d['value'], d['ms_per_step'], (d['config'].get('layer_kernel_form') or {}).get('items_per_launch'))"
CWN_LAYER_TARGET_ITEMS=$t python bench.py --brief --batch 512 --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zinc512 target $t', d['value'], d['ms_per_step'], (d['config'].get('layer_kernel_form') or {}).get('items_per_launch'), (d['config'].get('layer_kernel_form') or {}).get('variant'))"
done
