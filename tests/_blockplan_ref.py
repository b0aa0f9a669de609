"""The item-table builder as it was first written, in Python: the independent restatement that
cwn_layer_items_build (cwn_amd/csrc/cwn_blockplan.cpp) is checked against (tests/test_abi_and_host.py).
`self` is a cwn_amd.blockplan.BlockPlan (only its per-complex prefix sums are used)."""
from typing import List

import numpy as np

from cwn_amd import _ffi
from cwn_amd.blockplan import (ITEM_INTS, LDS_BYTES, MAX_ENTRIES, TARGET_ITEMS, TASK_ROWS, _IDX_BYTES, ItemTable, _pad4, _pad16,
                               gemm_rows_cap, lds_bytes)


def build(self, F: int, has_up):
    # one launch = one LDS size: the planes for the LARGEST staged block of any item plus the sources of
    # the item with the most of them (different items, in general: vertices + edges items want rows, edges +
    # rings items want sources).  A few splits of the LDS between the two are tried -- row cap from the top
    # down, the source cap = what is left -- and the one with the fewest items wins
    cap = gemm_rows_cap(F)
    step = max(16, cap // 8)
    best = None
    for row_cap in range(cap, step - 1, -step):
        src_cap = min(cap, (LDS_BYTES - 3 * row_cap * (F + 8) * 2 - _IDX_BYTES) // (F * 4) - 1)
        if src_cap < 16:
            continue
        t = _build_with(self, F, has_up, row_cap, src_cap)
        if t is None or lds_bytes(F, t.max_rows, t.max_src) > LDS_BYTES:
            continue
        if best is None or t.n_items < best.n_items:
            best = t
        elif t.n_items > best.n_items + best.n_items // 8:
            break                               # getting worse: smaller row caps only split more
        gmax = max(1, self.C // 128)
        if best is not None and best.n_items == len(self._sets(has_up)) * -(-self.C // gmax):
            break                               # one item per gmax complexes and set: nothing can have fewer
    return best

def _build_with(self, F: int, has_up, row_cap: int, src_cap: int):

    ng_round = int(_ffi.lib().cwn_layer_round_rows(F))   # rows per round of the kernel: the coface block starts at a multiple
    if ng_round <= 0:
        return None

    def first_coface_row(n_g: int, n_c: int) -> int:
        r1 = _pad16(n_g)
        return r1                               # (round 4: a multiple of 16, no longer of the rows per round)

    def staged(n_g: int, n_c: int) -> int:
        return first_coface_row(n_g, n_c) + _pad16(n_c) if n_c > 0 else _pad16(n_g)
    C = self.C
    if C == 0:
        return None
    for d in range(self.n_dims):
        if has_up[d] and (d + 1 >= self.n_dims or self.up_ptr[d] is None):
            return None
    gmax = max(1, C // TARGET_ITEMS)
    tables: List[np.ndarray] = []
    set_start = []
    max_rows = max_src = 0
    zero = np.zeros(C + 1, dtype=np.int64)
    cp = self.cell_ptr
    for set_id, (g, tasks) in enumerate(self._sets(has_up)):
        up = self.up_ptr[g] if g is not None else zero
        bps = [self.b_ptr[d] if (self.b_ptr[d] is not None and d > 0) else zero for d in tasks]
        d0 = tasks[0]
        recs: List[np.ndarray] = []
        c0 = 0
        while c0 < C:
            c1 = c0
            while c1 < C and c1 - c0 < gmax:
                nxt = c1 + 1
                rows = staged(int(cp[d0][nxt] - cp[d0][c0]),
                              int(cp[g + 1][nxt] - cp[g + 1][c0]) if g is not None else 0)
                # cells of dim d-1 the boundary streams read (staged in LDS), entries padded to 4
                src = sum(int(cp[d - 1][nxt] - cp[d - 1][c0]) for d, bp in zip(tasks, bps)
                          if d > 0 and bp[nxt] > bp[c0])
                ents = _pad4(int(up[nxt] - up[c0])) + sum(_pad4(int(bp[nxt] - bp[c0])) for bp in bps)
                ok = (rows <= row_cap and src <= src_cap and lds_bytes(F, rows, src) <= LDS_BYTES
                      and ents <= MAX_ENTRIES
                      and all(int(cp[d][nxt] - cp[d][c0]) <= TASK_ROWS for d in tasks))
                if not ok:
                    break
                c1 = nxt
            if c1 == c0:
                return None                 # a single complex exceeds the caps
            r = np.zeros(ITEM_INTS, dtype=np.int32)
            r[0] = set_id << 8
            n0 = int(cp[d0][c1] - cp[d0][c0])
            nc = une = 0
            live = tasks
            if g is not None:
                r[1] = g
                if n0 > 0:
                    nc, une = int(cp[g + 1][c1] - cp[g + 1][c0]), int(up[c1] - up[c0])
                    r[0] |= 1
                    r[2:8] = [cp[g][c0], n0, cp[g + 1][c0], nc, up[c0], une]
                else:
                    if any(int(cp[d][c1] - cp[d][c0]) > 0 for d in tasks[1:]):
                        return None             # cells of g + 1 without cells of g: not a cell complex
                    live = tasks[:1]
            max_rows = max(max_rows, staged(n0, nc))
            r[8] = len(live)
            src, bnes = 0, [0, 0]
            for t, d in enumerate(live):
                bp = bps[t]
                o = 9 + 7 * t
                bnes[t] = int(bp[c1] - bp[c0])
                r[o:o + 5] = [d, cp[d][c0], cp[d][c1] - cp[d][c0], bp[c0], bnes[t]]
                if d > 0 and bnes[t] > 0:       # boundary sources are staged only when read
                    r[o + 5] = cp[d - 1][c0]
                    r[o + 6] = cp[d - 1][c1] - cp[d - 1][c0]
                    src += int(r[o + 6])
            max_src = max(max_src, src)
            # derived fields (include/cwn_hip.h): the kernel reads them instead of re-deriving them
            b1 = _pad4(une)
            b2 = _pad4(b1 + bnes[0])
            r[23:28] = [first_coface_row(n0, nc), staged(n0, nc), b1, b2, _pad4(b2 + bnes[1])]
            recs.append(r)
            c0 = c1
        # heavy items first within the set: a workgroup with five row tiles should not start last
        tab = np.stack(recs)
        order = np.argsort(-(tab[:, 11].astype(np.int64) + tab[:, 5]), kind='stable')
        set_start.append(sum(t.shape[0] for t in tables))
        tables.append(tab[order])
    table = np.ascontiguousarray(np.concatenate(tables))
    cells_end = [int(cp[d][-1]) for d in range(self.n_dims)]
    up_end = [int(self.up_ptr[d][-1]) if (has_up[d] and self.up_ptr[d] is not None) else 0
              for d in range(self.n_dims)]
    b_end = [int(self.b_ptr[d][-1]) if (self.b_ptr[d] is not None and d > 0) else 0 for d in range(self.n_dims)]
    out = ItemTable(table, set_start, max(max_rows, 16), max_src, cells_end, up_end, b_end, self.device)
    if lds_bytes(F, out.max_rows, out.max_src) > LDS_BYTES:
        return out                          # the caller lowers the row cap and builds again
    rc = _ffi.lib().cwn_layer_items_check(table.ctypes.data, table.shape[0], F, out.c_plan(False))
    if rc != 0:
        raise _ffi.CwnError(f'item table failed cwn_layer_items_check ({rc})')
    return out
