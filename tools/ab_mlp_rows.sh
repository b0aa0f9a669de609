# A/B of the update / combine launch's schedule (CWN_MLP_FORM=5 alternating | auto: sequential, two per CU, beyond 256 workgroups)
for CFG in "zinc 2048" "reddit 32" "molhiv 512" "zinc 128"; do set -- $CFG; for R in 5 auto; do
  CWN_MLP_FORM=$R CWN_BENCH_SKIP=eager,concurrent,collate,workloads,fresh,train python bench.py --workload $1 --batch $2 --num-batches 2 --no-cpu 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); s=j['secondary']; print('$1-$2 form=$R', 'propagate', j['ms_per_step'], 'full_forward_ms', s['full_forward_ms'], s.get('forward_breakdown'))"
done; done
