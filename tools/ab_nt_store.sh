line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']; print('$1', d['value'], d['ms_per_step'], 'fwd', s['full_forward_ms'], 'train', (s.get('train_step') or {}).get('ms_per_step'), 'eager', (s.get('eager_launches') or {}).get('ms_per_step'))"; }
python bench.py --no-cpu 2>/dev/null | line nt
CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_nont.so python bench.py --no-cpu 2>/dev/null | line nont
python bench.py --no-cpu --batch 8192 --num-batches 1 --steps 20 --warmup 3 2>/dev/null | line nt8192
CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_nont.so python bench.py --no-cpu --batch 8192 --num-batches 1 --steps 20 --warmup 3 2>/dev/null | line nont8192
python bench.py --no-cpu --workload reddit 2>/dev/null | line ntreddit
CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_nont.so python bench.py --no-cpu --workload reddit 2>/dev/null | line nontreddit
