"""Item table of the complex-blocked layer kernel (csrc/cwn_layer.hip, `cwn_layer_fused_f32`).

A batched complex is a disjoint union: `from_complex_list` offsets every index per complex
(data/complex.py:148-169), so adjacency is block-diagonal and the cells / index entries of one
complex are contiguous in every batched tensor.  The reference's collate records exactly where:
`ptr` (cells, data/complex.py:344, 432) and `__slices__` (entries per key, :349-394).  From those
host-side tables this module cuts a batch into ITEMS -- contiguous ranges of complexes for one
"GEMM dimension" (a dimension with an upper adjacency whose message needs the Y1 / Y2 products) --
one workgroup each; record layout in include/cwn_hip.h.  The table is a property of the batch
(like `batch` / `ptr`), built once on the host and kept on the device next to the index tensors;
every layer of every forward reuses it.  No device work, no sync.
"""
from typing import List, Optional, Sequence

import numpy as np
import torch

ITEM_INTS = 32                 # = CWN_LAYER_ITEM_INTS
TASK_ROWS = 192                # = CWN_LAYER_TASK_ROWS
MAX_ENTRIES = 1024             # = CWN_LAYER_MAX_ENTRIES
TARGET_ITEMS = 128             # per GEMM dimension: ~one workgroup per CU over the two sets of a 2-complex


def gemm_rows_cap(F: int) -> int:
    return 12288 // F          # = CWN_LAYER_GEMM_ROWS(F) = CWN_LAYER_SOURCE_ROWS(F)


LDS_BYTES = 160 * 1024
_IDX_BYTES = (MAX_ENTRIES * 4 + 5 * MAX_ENTRIES * 2 + 3 * (TASK_ROWS + 2) * 2 + 15) // 16 * 16


def lds_bytes(F: int, gemm_rows: int, source_rows: int) -> int:
    """= cwn_layer_fused_lds_bytes: bf16 planes of the GEMM rows (Y overwrites them) + fp32 boundary
    sources + index scratch."""
    return 3 * gemm_rows * (F + 8) * 2 + source_rows * F * 4 + _IDX_BYTES


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class BlockPlan:
    """Host-side description of a batch + the item tables cut from it (one per feature width)."""

    def __init__(self, cells: Sequence[Sequence[int]], up_ptr: Sequence[Optional[Sequence[int]]],
                 b_ptr: Sequence[Optional[Sequence[int]]], device=None):
        """cells[d][c]   number of d-cells of complex c
        up_ptr[d]     entry offsets (len C + 1) of upper_index of dim d, or None (no upper adjacency)
        b_ptr[d]      entry offsets of boundary_index of dim d, or None"""
        self.n_dims = len(cells)
        self.C = len(cells[0]) if self.n_dims else 0
        self.cells = [np.asarray(c, dtype=np.int64) for c in cells]
        self.cell_ptr = [np.concatenate([[0], np.cumsum(c)]) for c in self.cells]
        self.up_ptr = [None if p is None else np.asarray(p, dtype=np.int64) for p in up_ptr]
        self.b_ptr = [None if p is None else np.asarray(p, dtype=np.int64) for p in b_ptr]
        self.device = device
        self._tables = {}

    # ---- construction from a batch ----------------------------------------------------------------
    @classmethod
    def from_batch(cls, batch) -> Optional['BlockPlan']:
        """From a ComplexBatch built by `from_complex_list` (or the device collate): None when the
        per-complex tables are missing (a batch assembled by hand)."""
        dims = sorted(batch.cochains.keys())
        if dims != list(range(len(dims))) or len(dims) > 3:
            return None
        cells, up_ptr, b_ptr = [], [], []
        for d in dims:
            c = batch.cochains[d]
            n_list = getattr(c, '__num_cells_list__', None)
            sl = getattr(c, '__slices__', None)
            if n_list is None or sl is None:
                return None
            cells.append([n or 0 for n in n_list])
            up_ptr.append(sl.get('upper_index') if c.upper_index is not None else None)
            b_ptr.append(sl.get('boundary_index') if c.boundary_index is not None else None)
        if len({len(c) for c in cells}) != 1:
            return None
        dev = next((c.upper_index.device for c in batch.cochains.values() if c.upper_index is not None), None)
        return cls(cells, up_ptr, b_ptr, device=dev)

    # ---- items ------------------------------------------------------------------------------------
    def _sets(self, has_up: Sequence[bool]):
        """[(g or None, [task dims])]: every dimension with an upper adjacency is the GEMM dimension
        of a set; a dimension without one rides as the second task of the set below it (the top
        dimension of a 2-complex with the edges) or forms a set without GEMM."""
        sets, d = [], 0
        while d < self.n_dims:
            if has_up[d]:
                tasks = [d]
                if d + 1 < self.n_dims and not has_up[d + 1] and (d + 2 >= self.n_dims):
                    tasks.append(d + 1)
                sets.append((d, tasks))
                d += len(tasks)
            else:
                sets.append((None, [d]))
                d += 1
        return sets

    def items(self, F: int, has_up: Sequence[bool]):
        """(table int32 [n_items, ITEM_INTS] on the plan's device, max padded GEMM rows, max boundary
        source rows) for feature width F, or None when some complex does not fit one workgroup's LDS (hub complexes: the
        caller then runs the CSR path).  `has_up[d]`: dimension d reduces an upper adjacency with
        coboundary features (needs d + 1 < n_dims)."""
        key = (F, tuple(bool(h) for h in has_up))
        if key in self._tables:
            return self._tables[key]
        out = self._build(F, key[1])
        if out is not None:
            table, max_rows, max_src = out
            t = torch.from_numpy(table)
            if self.device is not None:
                t = t.to(self.device)
            out = (t, max_rows, max_src)
        self._tables[key] = out
        return out

    def _build(self, F: int, has_up):
        cap = gemm_rows_cap(F)
        ng_round = 2048 // F       # rows per round of the kernel: the coface block starts at a multiple

        def staged(n_g: int, n_c: int) -> int:
            r1 = _pad16(n_g)
            return (r1 + ng_round - 1) // ng_round * ng_round + _pad16(n_c) if n_c > 0 else r1
        C = self.C
        if C == 0:
            return None
        for d in range(self.n_dims):
            if has_up[d] and (d + 1 >= self.n_dims or self.up_ptr[d] is None):
                return None
        gmax = max(1, C // TARGET_ITEMS)
        recs: List[np.ndarray] = []
        max_rows = max_src = 0
        zero = np.zeros(C + 1, dtype=np.int64)
        for set_id, (g, tasks) in enumerate(self._sets(has_up)):
            n_g = self.cells[g] if g is not None else None
            n_c = self.cells[g + 1] if g is not None else None
            up = self.up_ptr[g] if g is not None else zero
            bps = [self.b_ptr[d] if (self.b_ptr[d] is not None and d > 0) else zero for d in tasks]
            c0 = 0
            while c0 < C:
                c1 = c0
                while c1 < C and c1 - c0 < gmax:
                    nxt = c1 + 1
                    if g is not None:
                        rows = staged(int(self.cell_ptr[g][nxt] - self.cell_ptr[g][c0]),
                                      int(self.cell_ptr[g + 1][nxt] - self.cell_ptr[g + 1][c0]))
                    else:
                        rows = staged(int(self.cell_ptr[tasks[0]][nxt] - self.cell_ptr[tasks[0]][c0]), 0)
                    # cells of dim d-1 the boundary streams read (staged in LDS), entries padded to 4
                    src = sum(int(self.cell_ptr[d - 1][nxt] - self.cell_ptr[d - 1][c0])
                              for d, bp in zip(tasks, bps) if d > 0 and bp[nxt] > bp[c0])
                    ents = _pad4(int(up[nxt] - up[c0])) + sum(_pad4(int(bp[nxt] - bp[c0])) for bp in bps)
                    ok = (rows <= cap and src <= cap and lds_bytes(F, rows, src) <= LDS_BYTES
                          and ents <= MAX_ENTRIES
                          and all(int(self.cell_ptr[d][nxt] - self.cell_ptr[d][c0]) <= TASK_ROWS for d in tasks))
                    if not ok:
                        break
                    c1 = nxt
                if c1 == c0:
                    return None                 # a single complex exceeds the caps
                r = np.zeros(ITEM_INTS, dtype=np.int32)
                r[0] = set_id << 8
                if g is not None:
                    ng = int(self.cell_ptr[g][c1] - self.cell_ptr[g][c0])
                    nc = int(self.cell_ptr[g + 1][c1] - self.cell_ptr[g + 1][c0])
                    r[0] |= 1 if ng > 0 else 0
                    r[1:8] = [g, self.cell_ptr[g][c0], ng, self.cell_ptr[g + 1][c0], nc, up[c0], up[c1] - up[c0]]
                    if ng > 0:
                        max_rows = max(max_rows, staged(ng, nc))
                else:
                    max_rows = max(max_rows, staged(int(self.cell_ptr[tasks[0]][c1] - self.cell_ptr[tasks[0]][c0]), 0))
                r[8] = len(tasks)
                src = 0
                for t, d in enumerate(tasks):
                    bp = bps[t]
                    o = 9 + 7 * t
                    r[o:o + 5] = [d, self.cell_ptr[d][c0], self.cell_ptr[d][c1] - self.cell_ptr[d][c0],
                                  bp[c0], bp[c1] - bp[c0]]
                    if d > 0:
                        r[o + 5] = self.cell_ptr[d - 1][c0]
                        r[o + 6] = self.cell_ptr[d - 1][c1] - self.cell_ptr[d - 1][c0]
                        if bp[c1] > bp[c0]:
                            src += int(r[o + 6])
                max_src = max(max_src, src)
                recs.append(r)
                c0 = c1
        # heavy items first: a workgroup with four row tiles should not start behind the short ones
        table = np.stack(recs)
        order = np.argsort(-(table[:, 3].astype(np.int64) + table[:, 5]) * (table[:, 0] & 1), kind='stable')
        return np.ascontiguousarray(table[order]), max(max_rows, 16), max_src
