"""Summarise a rocprofv3 results .db (--kernel-trace --stats) as markdown: per-kernel totals, and
optionally the dispatch timeline of the last K kernels (start offset / duration / gap, in us).
usage: python profiles/summarize_rocprof.py <results.db> [K]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    print('| kernel | calls | total us | avg us | % |')
    print('|---|---|---|---|---|')
    for n, c, t, a, p in rows:
        print(f'| {n[:100]} | {c} | {t:.1f} | {a:.2f} | {p:.2f} |')
    if len(sys.argv) > 2:
        k = int(sys.argv[2])
        disp = cur.execute('select name, start, end, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size '
                           'from kernels order by start').fetchall()[-k:]
        t0 = disp[0][1]
        print(f'\nlast {k} dispatches (us): start | dur | gap-before | grid | wg | vgpr | sgpr | lds | name')
        prev_end = None
        for n, s, e, gx, wx, vg, sg, lds in disp:
            gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
            print(f'{(s - t0) / 1e3:9.2f} | {(e - s) / 1e3:6.2f} | {gap:6.2f} | {gx:7d} | {wx:4d} | {vg:3d} | {sg:3d} | {lds:5d} | {n[:70]}')
            prev_end = e


if __name__ == '__main__':
    main()
