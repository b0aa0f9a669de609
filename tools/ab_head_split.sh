for P in auto 8 32 64; do
  CWN_HEAD_POOL_SPLIT=$P CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh,train python bench.py --workload reddit --batch 32 --num-batches 2 --no-cpu 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); s=j['secondary']; print('reddit P=$P', 'full_forward_ms', s['full_forward_ms'])"
done
