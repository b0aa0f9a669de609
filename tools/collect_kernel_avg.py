"""rocprofv3 --kernel-trace --stats averages of the roofline kernels -> profiles/r6_kernel_avg.json, which bench.py quotes next to
its own HIP-event figure (`roofline.avg_launch_us_rocprof`, `frac_rocprof`; VERDICT r5 weak #7: the line's frac must follow from
what is under profiles/).

    python tools/collect_kernel_avg.py <key>=<summary.md> ...      key = bench.py's config.key, e.g. zinc:128:128

The summaries are the tables profiles/summarize_rocprof.py writes (tools/prof_scope.sh).  Per key the dominant kernel of the
propagate scope is taken: layer_kernel<F, 2> (the form that loads the per-item CSR: L - 1 of the L launches of a step) where the
blocked launch runs, aggregate_kernel otherwise."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'profiles', 'r6_kernel_avg.json')


def rows(path):
    with open(path) as fh:
        for ln in fh:
            m = re.match(r'\|\s*(.+?)\s*\|\s*(\d+)\s*\|\s*([\d.]+)\s*\|\s*([\d.]+)\s*\|\s*([\d.]+)\s*\|', ln)
            if m:
                yield m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))


def main():
    data = {'_how': __doc__.split('\n\n')[0].replace('\n', ' '), 'entries': {}}
    if os.path.exists(OUT):
        with open(OUT) as fh:
            data = json.load(fh)
    for arg in sys.argv[1:]:
        key, path = arg.split('=', 1)
        table = list(rows(path))
        pick = [r for r in table if re.search(r'layer_kernel<\d+, 2>', r[0])] or [r for r in table if 'aggregate_kernel' in r[0]]
        if not pick:
            print(f'{key}: no roofline kernel in {path}', file=sys.stderr)
            continue
        name, calls, total, avg = max(pick, key=lambda r: r[2])
        short = re.search(r'(\w+_kernel<[^>]*>)', name)
        data['entries'][key] = {'kernel': short.group(1) if short else name[:60], 'avg_us': avg, 'calls': calls,
                                'source': os.path.relpath(os.path.abspath(path), ROOT)}
        print(key, data['entries'][key])
    with open(OUT, 'w') as fh:
        json.dump(data, fh, indent=1)


if __name__ == '__main__':
    main()
