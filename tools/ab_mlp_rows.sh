for WL in reddit molhiv; do for R in 64 auto; do
  CWN_MLP_ROWS=$R CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh,train python bench.py --workload $WL --no-cpu 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); s=j['secondary']; print('$WL rows=$R', 'propagate', j['ms_per_step'], 'full_forward_ms', s['full_forward_ms'], s.get('forward_breakdown'))"
done; done
