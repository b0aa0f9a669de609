"""Item table of the complex-blocked layer kernel (csrc/cwn_layer.hip, `cwn_layer_fused_f32`).

A batched complex is a disjoint union: `from_complex_list` offsets every index per complex
(data/complex.py:148-169), so adjacency is block-diagonal and the cells / index entries of one
complex are contiguous in every batched tensor.  The reference's collate records exactly where:
`ptr` (cells, data/complex.py:344, 432) and `__slices__` (entries per key, :349-394).  From those
host-side tables this module cuts a batch into ITEMS -- contiguous ranges of complexes for one
"GEMM dimension" (a dimension with an upper adjacency whose message needs the Y1 / Y2 products) --
one workgroup each; record layout in include/cwn_hip.h.  The cut itself is `cwn_layer_items_build`
(csrc/cwn_blockplan.cpp, host C++: ~0.3 ms for a batch of 128 with this wrapper, where a first
Python version took 11 ms); this module gathers the prefix sums and holds the result.  The table is
a property of the batch (like `batch` / `ptr`), built once on the host and kept on the device next
to the index tensors; every layer of every forward reuses it.  No device work, no sync.
"""
from typing import List, Optional, Sequence

import numpy as np
import torch

ITEM_INTS = 32                 # = CWN_LAYER_ITEM_INTS
TASK_ROWS = 192                # = CWN_LAYER_TASK_ROWS
MAX_ENTRIES = 1024             # = CWN_LAYER_MAX_ENTRIES
CSR_SLOT_BYTES = 5264          # = CWN_LAYER_CSR_SLOT_BYTES
TARGET_ITEMS = 128             # per GEMM dimension: ~one workgroup per CU over the two sets of a 2-complex


def gemm_rows_cap(F: int) -> int:
    return 256 if F == 64 else 96          # = CWN_LAYER_GEMM_ROWS(F) = CWN_LAYER_SOURCE_ROWS(F)


LDS_BYTES = 160 * 1024
BWD_LDS_BYTES = 160 * 1024 - 2048      # csrc/cwn_layer_bwd_own.h kLdsCap: an item of the owner-form backward (the launch keeps 2 KiB for the BatchNorm sums)
_IDX_BYTES = ((3 * (TASK_ROWS + 2) * 4 + 15) // 16 * 16 + 3 * MAX_ENTRIES * 2 + 3 * (TASK_ROWS + 2) * 2 + 15) // 16 * 16


def lds_bytes(F: int, gemm_rows: int, source_rows: int) -> int:
    """= cwn_layer_fused_lds_bytes: bf16 planes of the GEMM rows (Y overwrites them) + fp32 boundary
    sources (+ one row of zeros) + index scratch."""
    return 3 * gemm_rows * (F + 8) * 2 + (source_rows + 1) * F * 4 + _IDX_BYTES


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class ItemTable:
    """One table (feature width F, which dimensions reduce an upper adjacency) + what the launcher
    needs to know about it (cwn_layer_plan) + the per-item CSR cache and its validity."""

    def __init__(self, table: np.ndarray, set_start: List[int], max_rows: int, max_src: int,
                 cells_end, up_end, b_end, device, variant: int = 0, lds_bytes: int = 0, n_big: int = 0):
        self.variant = int(variant)        # 0: one 16-wave workgroup per CU; 1: the two-per-CU form (include/cwn_hip.h)
        self.lds_bytes = int(lds_bytes)    # variant 1: dynamic LDS of the launch (the largest per-item need)
        self.n_big = int(n_big)            # BIG records: complexes a workgroup streams (include/cwn_hip.h)
        # (set, complex) of every BIG record and what the launch needs for them (ops.LayerLaunch fills it in)
        self.big_records = table[(table[:, 0] & 2) != 0].copy() if n_big else None
        self.big_ctx = None
        self.n_items = int(table.shape[0])
        self.items = torch.from_numpy(table)
        if device is not None:
            self.items = self.items.to(device)
        self.set_start, self.max_rows, self.max_src = set_start, int(max_rows), int(max_src)
        self.cells_end, self.up_end, self.b_end = cells_end, up_end, b_end
        self.device = device
        self.csr_cache: Optional[torch.Tensor] = None
        self.csr_key = None          # identity + version of the index tensors the cache was built from

    def c_plan(self, with_cache: bool):
        from . import _ffi
        if with_cache and self.csr_cache is None:
            self.csr_cache = torch.empty(self.n_items * CSR_SLOT_BYTES, dtype=torch.uint8, device=self.device)
        p = _ffi.LayerPlan(items=self.items.data_ptr(), csr_cache=_ffi.ptr(self.csr_cache) if with_cache else None,
                           n_items=self.n_items, max_gemm_rows=self.max_rows, max_source_rows=self.max_src,
                           variant=self.variant, lds_bytes=self.lds_bytes, n_big=self.n_big)
        for i, v in enumerate(self.set_start[:4]):
            p.set_start[i] = int(v)
        for d in range(len(self.cells_end)):
            p.cells_end[d], p.up_end[d], p.b_end[d] = int(self.cells_end[d]), int(self.up_end[d]), int(self.b_end[d])
        return p


class BwdItemTable:
    """Item table of the OWNER form of the backward launch (`cwn_layer_bwd_own_f32`, include/cwn_hip.h): one record per
    (range of complexes, dimension whose rows the workgroup owns), built by `cwn_layer_bwd_items_build`."""

    def __init__(self, table: np.ndarray, lds_bytes: int, cells_end, up_end, b_end, device):
        self.host = table
        self.n_items = int(table.shape[0])
        self.items = torch.from_numpy(table)
        if device is not None:
            self.items = self.items.to(device)
        self.lds_bytes = int(lds_bytes)
        self.cells_end, self.up_end, self.b_end = cells_end, up_end, b_end
        self.device = device

    def c_plan(self):
        from . import _ffi
        p = _ffi.LayerBwdPlan(items=self.items.data_ptr(), n_items=self.n_items, lds_bytes=self.lds_bytes)
        for d in range(len(self.cells_end)):
            p.cells_end[d], p.up_end[d], p.b_end[d] = int(self.cells_end[d]), int(self.up_end[d]), int(self.b_end[d])
        return p


class MixedTable:
    """Two item tables over complementary subsets of a batch's complexes -- `parts[0]` in the two-per-CU form (the
    complexes that fit its caps), `parts[1]` in the 16-wave form (the rest, BIG records included) -- served by two
    launches into the same outputs (ops.LayerLaunch).  Quacks like an ItemTable where the layer code looks at one."""
    variant = 'mixed'

    def __init__(self, parts: List[ItemTable], n_rest: int):
        self.parts = parts
        self.n_rest = int(n_rest)              # complexes in the 16-wave part
        self.n_items = sum(t.n_items for t in parts)
        self.n_big = sum(t.n_big for t in parts)
        self.big_records = parts[1].big_records
        self.device = parts[0].device

    @property
    def csr_key(self):
        keys = {t.csr_key for t in self.parts}
        return keys.pop() if len(keys) == 1 else None

    @csr_key.setter
    def csr_key(self, v):
        for t in self.parts:
            t.csr_key = v


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
        self.validated = False

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

    def cell_ptr_device(self, d: int, device) -> torch.Tensor:
        """`ptr` of dimension d (data/complex.py:344, 432: cells of complex c = rows ptr[c] .. ptr[c+1]) as a device
        int64 tensor, uploaded once per batch (the fused head reads it: cwn_head_f32)."""
        cache = self.__dict__.setdefault('_cell_ptr_dev', {})
        key = (d, str(device))
        if key not in cache:
            cache[key] = torch.from_numpy(np.ascontiguousarray(self.cell_ptr[d], dtype=np.int64)).to(device)
        return cache[key]

    def forget_csr(self) -> None:
        """Drop the cached per-item CSRs (the index tensors changed, or a caller wants a step that
        starts from the COO entries again)."""
        for t in self._tables.values():
            if t is not None:
                t.csr_key = None           # (a MixedTable passes it on to its parts)

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

    def at_least(self, F: int, has_up: Sequence[bool], has_b: Optional[Sequence[bool]] = None) -> int:
        """A lower bound of the number of items (staged rows of all complexes / the row cap, per set), in O(1):
        lets a caller with an item limit skip building the table of a very large batch."""
        cap = gemm_rows_cap(F)
        n = 0
        for g, tasks in self._sets(has_up):
            rows = int(self.cell_ptr[tasks[0]][-1]) + (int(self.cell_ptr[g + 1][-1]) if g is not None else 0)
            n += max(1, -(-rows // cap)) if self.C else 0
        return n

    def items(self, F: int, has_up: Sequence[bool], has_b: Optional[Sequence[bool]] = None,
              variant: int = 0, allow_big: bool = False) -> Optional[ItemTable]:
        """The item table for feature width F, or None when some complex does not fit one
        workgroup's LDS (hub complexes: the caller then runs the CSR path).  `has_up[d]`: dimension d
        reduces an upper adjacency with coboundary features (needs d + 1 < n_dims).  `has_b[d]`: the
        layer runs the boundary stream of dimension d (default: wherever the batch has a
        boundary_index) -- a layer without it (use_boundary_msg=False, include_boundary_features=False)
        gets a table whose records carry no boundary entries, so the launcher is never handed entry
        ranges of an index it was not given."""
        if has_b is None:
            has_b = [p is not None for p in self.b_ptr]
        key = (F, tuple(bool(h) for h in has_up), tuple(bool(h) and self.b_ptr[d] is not None for d, h in enumerate(has_b)),
               int(variant), bool(allow_big))
        if key not in self._tables:
            self._tables[key] = self._build(F, key[1], key[2], key[3], key[4])
        return self._tables[key]

    def items_mixed(self, F: int, has_up: Sequence[bool], has_b: Optional[Sequence[bool]] = None) -> Optional['MixedTable']:
        """A batch whose complexes do not ALL fit the two-per-CU form (its LDS holds ~30 atoms at width 128): the ones
        that fit are cut into a table of that form, the rest into a table of the 16-wave form (BIG records where even
        that is too small) -- two launches into the same outputs.  None when nothing fits the two-per-CU form or the rest
        has no table."""
        if has_b is None:
            has_b = [p is not None for p in self.b_ptr]
        key = ('mixed', F, tuple(bool(h) for h in has_up), tuple(bool(h) and self.b_ptr[d] is not None for d, h in enumerate(has_b)))
        if key not in self._tables:
            unfit = np.zeros(self.C, dtype=np.uint8)
            self._build(F, key[2], key[3], 1, False, unfit_out=unfit)                # which complexes do not fit (any set)
            res = None
            if unfit.any() and not unfit.all():
                t1 = self._build(F, key[2], key[3], 1, False, skip=unfit)
                t0 = self._build(F, key[2], key[3], 0, True, skip=(1 - unfit).astype(np.uint8))
                if t1 is not None and t0 is not None:
                    res = MixedTable([t1, t0], int(unfit.sum()))
            self._tables[key] = res
        return self._tables[key]

    def bwd_items(self, F: int, has_up: Sequence[bool], has_b: Optional[Sequence[bool]] = None) -> Optional[BwdItemTable]:
        """The item table of the owner form of the backward launch for feature width F (same meaning of `has_up` /
        `has_b` as `items`), or None when some complex is beyond a workgroup (the caller keeps the streaming backward)."""
        from . import _ffi
        if has_b is None:
            has_b = [p is not None for p in self.b_ptr]
        key = ('bwd', F, tuple(bool(h) for h in has_up), tuple(bool(h) and self.b_ptr[d] is not None for d, h in enumerate(has_b)))
        if key in self._tables:
            return self._tables[key]
        res = None
        C = self.C
        ok = C > 0 and all(not key[2][d] or (d + 1 < self.n_dims and self.up_ptr[d] is not None) for d in range(self.n_dims))
        if ok:
            sizes = _ffi.LayerSizes(n_complexes=C, n_dims=self.n_dims)
            keep = []
            for d in range(self.n_dims):
                sizes.has_up[d] = 1 if key[2][d] else 0
                for name, arr in (('cell_ptr', self.cell_ptr[d]), ('up_ptr', self.up_ptr[d]),
                                  ('b_ptr', self.b_ptr[d] if key[3][d] else None)):
                    if arr is not None:
                        a = np.ascontiguousarray(arr, dtype=np.int64)
                        keep.append(a)
                        getattr(sizes, name)[d] = a.ctypes.data
            cap_items = self.n_dims * C
            table = np.zeros((cap_items, _ffi.LAYER_BWD_ITEM_INTS), dtype=np.int32)
            plan = _ffi.LayerBwdPlan()
            n = int(_ffi.lib().cwn_layer_bwd_items_build(sizes, F, table.ctypes.data, cap_items, plan))
            if n < 0 and n != _ffi.LAYER_ITEMS_TOO_LARGE:
                raise _ffi.CwnError(f'cwn_layer_bwd_items_build failed ({n})')
            if n > 0:
                res = BwdItemTable(np.ascontiguousarray(table[:n]), int(plan.lds_bytes),
                                   [int(plan.cells_end[d]) for d in range(self.n_dims)],
                                   [int(plan.up_end[d]) for d in range(self.n_dims)],
                                   [int(plan.b_end[d]) for d in range(self.n_dims)], self.device)
        self._tables[key] = res
        return res

    def _build(self, F: int, has_up, has_b, variant: int = 0, allow_big: bool = False, skip=None, unfit_out=None) -> Optional[ItemTable]:
        """cwn_layer_items_build (csrc/cwn_blockplan.cpp, host C++): the greedy cut under the kernel's caps and
        the split of one launch's LDS between staged rows and boundary sources that gives the fewest items.  (A
        Python version of the same took 11 ms for a ZINC-like batch of 128 -- tests/_blockplan_ref.py keeps it as
        the independent restatement the C++ one is checked against.)"""
        from . import _ffi
        C = self.C
        if C == 0:
            return None
        for d in range(self.n_dims):
            if has_up[d] and (d + 1 >= self.n_dims or self.up_ptr[d] is None):
                return None
        sizes = _ffi.LayerSizes(n_complexes=C, n_dims=self.n_dims, allow_big=1 if (allow_big and variant == 0) else 0)
        keep = []
        if skip is not None:
            skip = np.ascontiguousarray(skip, dtype=np.uint8)
            keep.append(skip)
            sizes.skip = skip.ctypes.data
        if unfit_out is not None:
            assert unfit_out.dtype == np.uint8 and unfit_out.flags.c_contiguous and unfit_out.size == C
            sizes.unfit = unfit_out.ctypes.data
        for d in range(self.n_dims):
            sizes.has_up[d] = 1 if has_up[d] else 0
            for name, arr in (('cell_ptr', self.cell_ptr[d]), ('up_ptr', self.up_ptr[d]),
                              ('b_ptr', self.b_ptr[d] if has_b[d] else None)):
                if arr is not None:
                    a = np.ascontiguousarray(arr, dtype=np.int64)
                    keep.append(a)
                    getattr(sizes, name)[d] = a.ctypes.data
        cap_items = self.n_dims * C
        table = np.zeros((cap_items, ITEM_INTS), dtype=np.int32)
        plan = _ffi.LayerPlan(variant=variant)
        n = int(_ffi.lib().cwn_layer_items_build(sizes, F, table.ctypes.data, cap_items, plan))
        if n == _ffi.LAYER_ITEMS_TOO_LARGE or n == 0:
            return None                 # a single complex exceeds the caps (or nothing to do): the caller runs the CSR path
        if n < 0:
            raise _ffi.CwnError(f'cwn_layer_items_build failed ({n})')
        table = np.ascontiguousarray(table[:n])
        n_sets = len(self._sets(has_up))
        out = ItemTable(table, [int(plan.set_start[i]) for i in range(n_sets)], int(plan.max_gemm_rows),
                        int(plan.max_source_rows), [int(plan.cells_end[d]) for d in range(self.n_dims)],
                        [int(plan.up_end[d]) for d in range(self.n_dims)], [int(plan.b_end[d]) for d in range(self.n_dims)],
                        self.device, variant=variant, lds_bytes=int(plan.lds_bytes), n_big=int(plan.n_big))
        rc = _ffi.lib().cwn_layer_items_check(table.ctypes.data, n, F, out.c_plan(False))
        if rc != 0:
            raise _ffi.CwnError(f'item table failed cwn_layer_items_check ({rc})')
        return out


# ---- which single complexes a workgroup holds (numpy, vectorised over a dataset) ------------------------------------------
# The device-side table builders (csrc/cwn_items_dev.hip) give a complex that does not fit alone no record and set an error
# bit; a caller that feeds them fresh batches (cwn_amd/static_batch.py) checks BEFORE the launch, on the host, from the same
# per-complex sizes -- these functions restate the builders' `fits(c, c + 1)` for every complex of a dataset at once.
def _sets_of(n_dims: int, has_up: Sequence[bool]):
    sets, d = [], 0
    while d < n_dims:
        if has_up[d]:
            tasks = [d]
            if d + 1 < n_dims and not has_up[d + 1] and d + 2 >= n_dims:
                tasks.append(d + 1)
            sets.append((d, tasks))
            d += len(tasks)
        else:
            sets.append((None, [d]))
            d += 1
    return sets


def single_fit_forward(cells: Sequence[np.ndarray], up_len: Sequence[Optional[np.ndarray]], b_len: Sequence[Optional[np.ndarray]],
                       F: int, has_up: Sequence[bool], has_b: Sequence[bool], variant: int, row_cap: int, src_cap: int) -> np.ndarray:
    """bool per complex: does it fit one workgroup of cwn_layer_fused_f32 in EVERY set (cells[d], up_len[d], b_len[d]: per-complex
    counts; the caps: the launch's LDS split)?  = items_fwd_kernel's fits(c, c + 1)."""
    n_dims = len(cells)
    cells = [np.asarray(c, dtype=np.int64) for c in cells]
    z = np.zeros_like(cells[0])
    up = [np.asarray(u, dtype=np.int64) if (u is not None) else z for u in up_len]
    bl = [np.asarray(b, dtype=np.int64) if (b is not None and has_b[d]) else z for d, b in enumerate(b_len)]
    p16 = lambda n: (n + 15) // 16 * 16
    p4 = lambda n: (n + 3) // 4 * 4
    rr = (1024 if variant == 0 else 512) // (F // 4)                   # rows per round (cwn_layer_variant_round_rows)
    budget = LDS_BYTES if variant == 0 else 80 * 1024
    half = 0 if variant == 0 else (128 if F == 64 else 64)
    ok = np.ones(cells[0].shape, dtype=bool)
    for g, tasks in _sets_of(n_dims, has_up):
        d0 = tasks[0]
        n0 = cells[d0]
        nc = cells[g + 1] if g is not None else z
        first = p16(n0)                                                   # (round 4: a multiple of 16, not of `rr`)
        rows = np.where(nc > 0, first + p16(nc), p16(n0))
        src = np.zeros_like(n0)
        ents = p4(up[g]) if g is not None else np.zeros_like(n0)
        for t, d in enumerate(tasks):
            be = bl[d] if d > 0 else z
            if d > 0 and (variant == 0 or t == 0):
                src = src + np.where(be > 0, cells[d - 1], 0)
            ents = ents + p4(be)
            ok &= cells[d] <= TASK_ROWS
        lds = 3 * rows * (F + 8) * 2 + (src + 1) * F * 4 + _IDX_BYTES
        ok &= (rows <= row_cap) & (src <= src_cap) & (lds <= budget) & (ents <= MAX_ENTRIES)
        if half and g is not None:
            ok &= (p16(n0) <= half) & (p16(nc) <= half)
    return ok


def bwd_layout_total(F: int, flags: int, n_o, n_a, n_b, ne_a, ne_b, ne_bd):
    """cwn_bwd_own::layout(...).total (csrc/cwn_layer_bwd_own.h), vectorised."""
    pa, pb, top = bool(flags & 1), bool(flags & 2), bool(flags & 4)
    p16 = lambda n: (n + 15) // 16 * 16
    p4 = lambda n: (n + 3) // 4 * 4
    s_rows = (2 * n_o + n_a if pa else 0) + (n_o + 2 * n_b if pb else 0) + np.where(ne_bd > 0, n_a, 0)
    RO = p16(n_o)
    RT = p16(n_a) if top else 0 * n_o
    pl_rows = (RO if pa else 0) + (RO if pb else 0) + RT
    s_bytes, p_bytes = s_rows * (F + 4) * 4, 3 * pl_rows * (F + 8) * 2
    reg0 = (np.maximum(s_bytes, p_bytes) + 15) // 16 * 16
    ent_off = reg0 + (RO + RT) * (F + 4) * 4
    ea4 = p4(ne_a) if pa else 0 * n_o
    eb4 = p4(ne_b) if pb else 0 * n_o
    p_off = (ent_off + (6 * ea4 + 3 * eb4 + 2 * p4(ne_bd)) * 4 + 15) // 16 * 16
    rounds = 1024 // (F // 4)
    return p_off + (np.where(n_a > 0, rounds * (F + 4) * 4, 0) if top else 0)


def single_fit_backward(cells: Sequence[np.ndarray], up_len: Sequence[Optional[np.ndarray]], b_len: Sequence[Optional[np.ndarray]],
                        F: int, has_up: Sequence[bool], has_b: Sequence[bool], lds_cap: int = BWD_LDS_BYTES) -> np.ndarray:
    """bool per complex: does it fit one workgroup of cwn_layer_bwd_own_f32 in every set?  = items_bwd_kernel's classify >= 0."""
    n_dims = len(cells)
    cells = [np.asarray(c, dtype=np.int64) for c in cells]
    z = np.zeros_like(cells[0])
    up = [np.asarray(u, dtype=np.int64) if u is not None else z for u in up_len]
    bl = [np.asarray(b, dtype=np.int64) if (b is not None and has_b[d]) else z for d, b in enumerate(b_len)]
    own_cap = 256 if F == 64 else 96
    top_cap = 1024 // (F // 4)
    ok = np.ones(cells[0].shape, dtype=bool)
    for g, tasks in _sets_of(n_dims, has_up):
        d = tasks[0]
        top, pa, pb = len(tasks) == 2, bool(has_up[d]), d > 0 and bool(has_up[d - 1])
        above = d + 1 < n_dims
        n_o = cells[d]
        n_a = cells[d + 1] if above else z
        n_b = cells[d - 1] if pb else z
        ea = up[d] if pa else z
        eb = up[d - 1] if pb else z
        bd = bl[d + 1] if above else z
        flags = (1 if pa else 0) | (2 if pb else 0) | (4 if top else 0)
        need_a = pa or top
        na_eff = n_a if need_a else np.where(bd > 0, n_a, 0)
        total = bwd_layout_total(F, flags, n_o, na_eff, n_b, ea, eb, bd)
        ok &= (n_o <= own_cap) & ((n_a <= top_cap) if top else True) & (ea <= MAX_ENTRIES) & (eb <= MAX_ENTRIES) & (bd <= MAX_ENTRIES)
        ok &= (n_a <= 4096) & (n_b <= 4096) & (total <= lds_cap)
        ok &= ~((n_o == 0) & (((n_a > 0) if top else False) | (ea > 0) | (eb > 0) | (bd > 0)))
    return ok
