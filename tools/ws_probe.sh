mkdir -p gpurun_out
for nb in 4 16; do
  python bench.py --workload molhiv --num-batches $nb --only-primary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fixed nb=$nb', d['value']/1e6, d['ms_per_step'])"
done
for s in 4 8 16; do
  CWN_BENCH_FRESH_SLOTS=$s CWN_BENCH_SKIP=eager,concurrent,collate,workloads,train,roofline CWN_BENCH_DETAIL=/tmp/d.json python bench.py --workload molhiv --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); f=d['secondary'].get('fresh_batches'); print('fresh slots=$s fixed', d['ms_per_step'], f)"
done
