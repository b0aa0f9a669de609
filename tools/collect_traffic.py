"""HBM traffic per launch of the hot kernels from the PMC counters, as MI355X_MICROARCH.md's HBM
section prescribes: separate rocprofv3 passes for FETCH_SIZE and WRITE_SIZE (with --kernel-trace
only), counters in KiB, gfx950 FETCH_SIZE doubled for wide coalesced streams, WRITE_SIZE as is.
Run ON the GPU box from the repo root:   python tools/collect_traffic.py [batch | workload:batch ...]
(a bare batch = the ZINC workload; `molhiv:512` / `reddit:32`: the dominant kernel of that workload's propagate scope)
Writes profiles/r6_pmc_fetch_write_raw.json and profiles/r6_traffic.json (the dominant kernel of the
propagate scope: layer_kernel<F, 2>, the variant that loads the per-item CSR; the first launch of a
step is layer_kernel<F, 1>)."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    m = re.search(r'(\w+_kernel(?:<[^>]*>)?)', name.replace('(anonymous namespace)::', ''))
    return m.group(1) if m else name[:60]


def one_pass(counter: str, batch: int, workload: str):
    out = f'/tmp/pmc_{counter}_{workload}_{batch}'
    subprocess.run(['rm', '-rf', out])
    cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', out, '--',
           sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--steps', '3', '--warmup', '1',
           '--batch', str(batch), '--num-batches', '1', '--only-primary', '--no-graph']
    env = dict(os.environ, TMPDIR='/tmp')
    subprocess.run(cmd, cwd='/tmp', env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    acc = {}
    for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                k = short(row['Kernel_Name'])
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(row['Counter_Value'])
                a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    args = sys.argv[1:] or ['128', '8192']
    batches = [int(a) for a in args if ':' not in a]
    others = [(a.split(':')[0], int(a.split(':')[1])) for a in args if ':' in a]
    raw, traffic = {}, {'_how': __doc__.split('Run ON')[0].strip().replace('\n', ' '),
                        'kernel': 'layer_kernel<128, 2>', 'hidden': 128, 'entries': {}}
    for b in batches:
        f, w = one_pass('FETCH_SIZE', b, 'zinc'), one_pass('WRITE_SIZE', b, 'zinc')
        raw[str(b)] = {k: {'FETCH_SIZE_KB_avg': round(f[k][0], 1), 'launches': f[k][1],
                           'WRITE_SIZE_KB_avg': round(w.get(k, (0, 0))[0], 1)} for k in f}
        cands = [n for n in f if n.startswith('layer_kernel<128, 2') and n in w] or \
            [n for n in f if n.startswith('aggregate_kernel<4') and n in w]
        if cands:
            k = max(cands, key=lambda n: f[n][1])      # the layer launches (most frequent variant)
            if not traffic['entries']:
                traffic['kernel'] = k                  # of the first batch given (the headline configuration)
            fb, wb = f[k][0] * 1024, w[k][0] * 1024
            traffic['entries'][str(b)] = {'kernel': k, 'fetch_bytes_raw': int(fb), 'write_bytes_raw': int(wb),
                                          'traffic_bytes': int(2 * fb + wb), 'launches_averaged': f[k][1]}
    # the other workloads: entries keyed `workload:batch` (bench.py looks its own up)
    for wl, b in others:
        f, w = one_pass('FETCH_SIZE', b, wl), one_pass('WRITE_SIZE', b, wl)
        raw[f'{wl}:{b}'] = {k: {'FETCH_SIZE_KB_avg': round(f[k][0], 1), 'launches': f[k][1],
                                 'WRITE_SIZE_KB_avg': round(w.get(k, (0, 0))[0], 1)} for k in f}
        cands = [n for n in f if (n.startswith('layer_kernel<') and ', 2' in n) and n in w] or \
            [n for n in f if n.startswith('aggregate_kernel<4') and n in w]
        if cands:
            k = max(cands, key=lambda n: f[n][1])
            fb, wb = f[k][0] * 1024, w[k][0] * 1024
            traffic['entries'][f'{wl}:{b}'] = {'kernel': k, 'fetch_bytes_raw': int(fb), 'write_bytes_raw': int(wb),
                                               'traffic_bytes': int(2 * fb + wb), 'launches_averaged': f[k][1]}
    # profiles/ is what bench.py reads; gpurun_out/ is what travels back from the GPU box
    for d in ('profiles', 'gpurun_out'):
        os.makedirs(os.path.join(ROOT, d), exist_ok=True)
        with open(os.path.join(ROOT, d, 'r6_pmc_fetch_write_raw.json'), 'w') as fh:
            json.dump(raw, fh, indent=1)
        with open(os.path.join(ROOT, d, 'r6_traffic.json'), 'w') as fh:
            json.dump(traffic, fh, indent=1)
    print(json.dumps(traffic['entries']))


if __name__ == '__main__':
    main()
