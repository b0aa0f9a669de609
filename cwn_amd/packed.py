"""Device-side batching: a dataset of complexes packed in HBM, batches built by ONE kernel.

The reference collates on the CPU: `ComplexBatch.from_complex_list` (data/complex.py:690-728) ->
`CochainBatch.from_cochain_list` (:323-458), a Python loop of per-complex tensor adds and one
`torch.cat` per key, followed by `batch.to(device)`.  Once propagate runs in tens of microseconds
that loop IS the step.  Here every complex's tensors are uploaded ONCE, concatenated per key
(288 GB of HBM holds any of the reference's datasets), and `collate(indices)`:
  * computes the per-array segment tables (lengths, source offsets, running cell offsets of
    data/complex.py:148-169) on the host from numpy metadata -- no device sync,
  * uploads them with ONE small H2D copy,
  * fills every array of the ComplexBatch with ONE launch of cwn_collate.
The integer layout is bit-exact against the reference's (tests/golden/batching.npz).
"""
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import _ffi
from .complex import CochainBatch, Complex, ComplexBatch

_INDEX_KEYS = ('upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries', 'boundary_index')
_ALL_KEYS = ('x', 'y') + _INDEX_KEYS
# The destination-sorted CSR of a complex's boundary adjacency and of its transpose, kept per complex in LOCAL numbers
# (`with_csr=True`).  A batch is block-diagonal and per-complex contiguous (data/complex.py:148-169), so the batch's CSR is
# the concatenation of its complexes' CSRs: the `col` arrays collate like an index row (+ the cell offset of the source
# dimension), the row pointers -- stored WITHOUT their leading zero, one number per cell -- collate to rowptr[1:] + the
# running entry count.  What cwn_csr_build does per batch (two launches per step for the front's plans and their transposes)
# is then part of the one collate launch.
_CSR_KEYS = ('b_rowptr', 'b_col', 'bt_rowptr', 'bt_col')


class _Packed:
    """One key of one dimension, all complexes concatenated along the last axis."""
    __slots__ = ('data', 'start', 'length', 'has', 'rows', 'width', 'op')

    def __init__(self, data, start, length, has, rows, width, op):
        self.data, self.start, self.length, self.has = data, start, length, has
        self.rows, self.width, self.op = rows, width, op


class PackedComplexes:
    """`complexes` (cwn_amd Complex objects with CPU tensors) packed on `device`."""

    def __init__(self, complexes: Sequence[Complex], device, max_dim: int = 2, with_csr: bool = False):
        self.device = torch.device(device)
        self.max_dim = max_dim
        self.with_csr = bool(with_csr)
        self.num = len(complexes)
        self.dims = np.array([min(c.dimension, max_dim) for c in complexes], dtype=np.int64)
        C = self.num
        D = max_dim + 1
        self.n_cells = np.zeros((D, C), dtype=np.int64)
        self.has_cells = np.zeros((D, C), dtype=bool)
        self.n_up = np.zeros((D, C), dtype=np.int64)
        self.n_down = np.zeros((D, C), dtype=np.int64)
        for ci, cx in enumerate(complexes):
            for d in range(D):
                if d in cx.cochains and d <= cx.dimension:
                    c = cx.cochains[d]
                    n = c.num_cells
                    self.has_cells[d, ci] = n is not None
                    self.n_cells[d, ci] = n or 0
                    self.n_up[d, ci] = c.num_cells_up or 0
                    self.n_down[d, ci] = (c.num_cells_down or 0) if d > 0 else 0
                elif d - 1 in cx.cochains:
                    # a complex without this dimension still shifts later boundary indices by its
                    # number of (d-1)-cells (data/complex.py:709-716)
                    self.n_down[d, ci] = cx.cochains[d - 1].num_cells or 0
        self.keys: List[Dict[str, _Packed]] = []
        for d in range(D):
            per_key = {}
            for key in _ALL_KEYS:
                items = [(cx.cochains[d][key] if (d in cx.cochains and d <= cx.dimension) else None)
                         for cx in complexes]
                if all(t is None for t in items):
                    continue
                per_key[key] = self._pack(items, key)
            self.keys.append(per_key)
        ys = [cx.y for cx in complexes]
        self.y = self._pack(ys, 'y') if all(t is not None for t in ys) else None
        self._finalise()

    @classmethod
    def from_arrays(cls, device, max_dim: int, dims, n_cells, has_cells, n_up, n_down, keys, y=None,
                    with_csr: bool = False) -> 'PackedComplexes':
        """A packed dataset from arrays that are ALREADY concatenated per key (cwn_amd.lifting.pack_graph_dataset_*:
        a dataset lifted by cwn_lift_many goes from graphs to HBM without per-complex Python objects).
        keys[d][name] = (data, lengths, has): `data` as the constructor would concatenate it (x: [rows, width] or flat;
        a two-row index: [2, total]; else 1-D), `lengths` per complex in the constructor's units (x: rows * width; an
        index: columns), `has` per complex."""
        self = cls.__new__(cls)
        self.device = torch.device(device)
        self.max_dim = max_dim
        self.with_csr = bool(with_csr)
        self.dims = np.minimum(np.asarray(dims, dtype=np.int64), max_dim)
        self.num = int(self.dims.size)
        D = max_dim + 1
        self.n_cells = np.asarray(n_cells, dtype=np.int64)[:D]
        self.has_cells = np.asarray(has_cells, dtype=bool)[:D]
        self.n_up = np.asarray(n_up, dtype=np.int64)[:D]
        self.n_down = np.asarray(n_down, dtype=np.int64)[:D]
        self.keys = []
        for d in range(D):
            per_key = {}
            for key in _ALL_KEYS:            # the constructor's key order
                if key in keys[d] and np.asarray(keys[d][key][2], dtype=bool).any():
                    per_key[key] = self._packed(*keys[d][key], key)
            self.keys.append(per_key)
        self.y = self._packed(*y, 'y') if y is not None else None
        self._finalise()
        return self

    def _packed(self, data: torch.Tensor, lengths, has, key) -> _Packed:
        lengths = np.asarray(lengths, dtype=np.int64)
        width = 1
        if key == 'x':
            width = int(data.size(1)) if data.dim() == 2 else 1
            data = data.reshape(-1)
        start = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
        return _Packed(data.contiguous().to(self.device), start, lengths, np.asarray(has, dtype=bool),
                       2 if key in ('upper_index', 'lower_index', 'boundary_index') else 1, width, self._op(data, key))

    @staticmethod
    def _op(data: torch.Tensor, key: str) -> int:
        if data.dtype == torch.float32 or data.dtype == torch.int32:
            return _ffi.COLLATE_COPY32
        if data.dtype == torch.int64:
            return _ffi.COLLATE_ADD64 if key in _INDEX_KEYS else _ffi.COLLATE_COPY64
        raise TypeError(f'{key}: unsupported dtype {data.dtype} (float32 / int32 / int64 only)')

    def _pack(self, items, key) -> _Packed:
        ref = next(t for t in items if t is not None)
        two_rows = key in ('upper_index', 'lower_index', 'boundary_index')
        width = 1
        if key == 'x':
            width = int(ref.size(1)) if ref.dim() == 2 else 1
        lengths = np.array([0 if t is None else (int(t.size(-1)) if key != 'x' else int(t.size(0)) * width)
                            for t in items], dtype=np.int64)
        has = np.array([t is not None for t in items], dtype=bool)
        start = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
        present = [t for t in items if t is not None]
        if key == 'x':
            data = torch.cat([t.reshape(-1) for t in present])
        else:
            data = torch.cat([t.unsqueeze(0) if t.dim() == 0 else t for t in present], dim=-1)
        return _Packed(data.contiguous().to(self.device), start, lengths, has, 2 if two_rows else 1, width,
                       self._op(data, key))

    # --------------------------------------------------------------------------------------------
    def _add_csr_keys(self) -> None:
        """Per complex, in local numbers: the destination-sorted CSR of boundary_index_d (rows = cells of d, col = boundary
        cell) and of its transpose (rows = cells of d - 1, col = cell), stable in entry order -- what cwn_csr_build
        produces for the batched index, cut at the complexes (module comment at _CSR_KEYS)."""
        D = self.max_dim + 1
        for d in range(1, D):
            pk = self.keys[d].get('boundary_index')
            if pk is None:
                continue
            idx = pk.data.detach().cpu().numpy()                       # [2, E]: row 0 boundary cell, row 1 cell (local)
            E = int(idx.shape[1])
            lengths = np.asarray(pk.length, dtype=np.int64)
            C = self.num
            cid = np.repeat(np.arange(C, dtype=np.int64), lengths)
            ent0 = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
            for name, rows_of, key_row, val_row in (('b', self.n_cells[d], 1, 0), ('bt', self.n_down[d], 0, 1)):
                rows_of = np.asarray(rows_of, dtype=np.int64)
                row0 = np.concatenate([[0], np.cumsum(rows_of)[:-1]]).astype(np.int64)
                key, val = idx[key_row].astype(np.int64), idx[val_row].astype(np.int64)
                if E and (key.min() < 0 or (key >= rows_of[cid]).any()):
                    raise IndexError(f'boundary_index of dimension {d}: a local index outside its complex')
                g = row0[cid] + key                                    # row number in packed order: sorted by complex already
                order = np.argsort(g, kind='stable')
                col = val[order].astype(np.int32)
                counts = np.bincount(g, minlength=int(rows_of.sum())).astype(np.int64)
                incl = np.cumsum(counts) - np.repeat(ent0, rows_of)    # local inclusive row pointers: rowptr[1:] of each complex
                dev = self.device
                self.keys[d][name + '_rowptr'] = _Packed(torch.from_numpy(incl.astype(np.int32)).to(dev), row0, rows_of,
                                                         rows_of > 0, 1, 1, _ffi.COLLATE_ADD32)
                self.keys[d][name + '_col'] = _Packed(torch.from_numpy(col).to(dev), ent0, lengths, lengths > 0, 1, 1,
                                                      _ffi.COLLATE_ADD32)

    def _finalise(self) -> None:
        """Per-complex metadata of every key stacked into matrices: a batch's tables are then a dozen numpy calls
        in all (the first form made ~100 small ones, 150 us of host time per batch of 128 -- four propagate
        steps)."""
        D = self.max_dim + 1
        if self.with_csr:
            self._add_csr_keys()
        self._klist = [(d, key, pk) for d in range(D) for key, pk in self.keys[d].items()]
        if self.y is not None:
            self._klist.append((-1, 'y', self.y))
        # one row per complex: [cells, down, up] of every dimension, then length / start / has of every key -- a
        # batch needs ONE gather of its rows
        cols = [np.stack([self.n_cells, self.n_down, self.n_up], axis=1).reshape(3 * D, self.num)]
        for f in ('length', 'start', 'has'):
            cols += [np.asarray(getattr(pk, f), dtype=np.int64)[None] for _, _, pk in self._klist]
        self._meta = np.ascontiguousarray(np.concatenate(cols, axis=0).T)                  # [num, 3D + 3K]
        self._meta_dev = None          # the same matrix in HBM (cwn_collate_tables: the tables of a batch built on the device)

    def meta_device(self) -> torch.Tensor:
        if self._meta_dev is None:
            self._meta_dev = torch.from_numpy(self._meta).to(self.device)
        return self._meta_dev

    def key_index(self, d: int, key: str) -> int:
        """Position of (dimension, key) in the table layout (the `k` of cwn_collate_tables), -1 when the dataset has no such array."""
        for k, (dd, kk, _) in enumerate(self._klist):
            if dd == d and kk == key:
                return k
        return -1

    # add-table of a key inside a dimension's block of five offset rows (here, here, down, here, up), in rows
    _ADD_ROW = {'upper_index': 0, 'lower_index': 0, 'shared_boundaries': 2, 'shared_coboundaries': 4, 'boundary_index': 2}

    def _prepare(self, idx):
        """Host half of collate: output tensors (uninitialised), ONE int64 array holding every segment table, and
        the launch plan -- entries (packed or None, out, offset of dst_start [B + 1], offset of src_start [B] or None,
        offset of the add rows or None, total elements per row)."""
        idx = np.asarray(idx, dtype=np.int64)
        B = int(idx.size)
        if B == 0:
            raise ValueError('collate of an empty index list')
        dimension = int(self.dims[idx].max())
        dev = self.device
        D, K = self.max_dim + 1, len(self._klist)
        m = self._meta[idx]                                               # [B, 3D + 3K]
        cs = np.cumsum(m[:, :3 * D + K], axis=0)
        # running cell offsets of data/complex.py:148-169, all dimensions at once: [D, (here, down, up), B]
        cnt = np.ascontiguousarray(m[:, :3 * D].T).reshape(D, 3, B)
        end = np.ascontiguousarray(cs[:, :3 * D].T).reshape(D, 3, B)
        off = end - cnt
        tot = end[:, :, -1].tolist()
        dst = np.zeros((K, B + 1), dtype=np.int64)
        dst[:, 1:] = cs[:, 3 * D:].T
        totals = dst[:, -1].tolist()
        present = m[:, 3 * D + 2 * K:].any(axis=0).tolist()
        seg = np.concatenate([off[:, 0, :], end[:, 0, -1:]], axis=1)                   # [D, B + 1]: batch-vector segments
        o_src = K * (B + 1)
        o_off = o_src + K * B
        o_seg = o_off + D * 5 * B
        tables = np.concatenate([dst.reshape(-1), m[:, 3 * D + K:3 * D + 2 * K].T.reshape(-1),
                                 off[:, (0, 0, 1, 0, 2), :].reshape(-1), seg.reshape(-1)])
        plan = []
        cochains = [CochainBatch(d) for d in range(dimension + 1)]
        for cb in cochains:
            cb.__slices__ = {}
        y = None
        for k, (d, key, pk) in enumerate(self._klist):
            if d > dimension:
                continue
            total = totals[k]
            if d < 0:                       # the complexes' labels
                y = torch.empty(total, dtype=pk.data.dtype, device=dev)
                plan.append((pk, y, k * (B + 1), o_src + k * B, None, total))
                continue
            if not present[k] or key in _CSR_KEYS:
                continue                    # (the per-complex CSRs serve the static path: cwn_amd/static_graph.py)
            if key == 'x':
                out = torch.empty(total // pk.width, pk.width, dtype=pk.data.dtype, device=dev)
            elif pk.rows == 2:
                out = torch.empty(2, total, dtype=pk.data.dtype, device=dev)
            else:
                out = torch.empty(total, dtype=pk.data.dtype, device=dev)
            row = self._ADD_ROW.get(key)
            plan.append((pk, out, k * (B + 1), o_src + k * B, None if row is None else o_off + (d * 5 + row) * B, total))
            cb = cochains[d]
            if key == 'x':
                cb._x = out
            else:
                cb.__slices__[key] = dst[k].tolist()
                setattr(cb, key, out)
        for d, cb in enumerate(cochains):
            n_sel = cnt[d, 0]
            # the per-complex tables the reference's collate keeps (data/complex.py:344-441): the blocked layer
            # kernel's item table is cut from them (cwn_amd/blockplan.py)
            cb.__num_cells_list__ = n_sel.tolist()
            has = self.has_cells[d, idx]
            if has.any():
                # batch vector: complexes that have cells of this dimension, numbered by position
                out = torch.empty(tot[d][0], dtype=torch.int64, device=dev)
                plan.append((None, out, o_seg + d * (B + 1), None, None, tot[d][0]))
                cb.batch = out
                cb.ptr = [0] + np.cumsum(n_sel[has]).tolist()
            cb.__num_cells__ = tot[d][0]
            cb.__num_cells_up__ = tot[d][2]
            if d > 0:
                cb.__num_cells_down__ = tot[d][1]
            cb.__num_cochains__ = B
        return cochains, y, tables, plan

    def collate(self, idx: Sequence[int]) -> ComplexBatch:
        """The ComplexBatch of complexes `idx` (in that order), on the device."""
        cochains, y, tables, plan = self._prepare(idx)
        B = cochains[0].__num_cochains__
        dev = self.device
        # one H2D copy for every table, one launch for every array
        tab = torch.from_numpy(tables).to(dev, non_blocking=True)
        base = tab.data_ptr()
        descs = []
        for pk, out, o_dst, o_src, o_add, total in plan:
            if total == 0:
                continue
            if pk is None:
                descs.append(_ffi.CollateDesc(src=None, dst=out.data_ptr(), dst_start=base + 8 * o_dst,
                                              src_start=None, add=None, src_row_stride=0, dst_row_stride=0,
                                              n_rows=1, op=_ffi.COLLATE_SEGID64))
                continue
            descs.append(_ffi.CollateDesc(
                src=pk.data.data_ptr(), dst=out.data_ptr(), dst_start=base + 8 * o_dst,
                src_start=base + 8 * o_src, add=None if o_add is None else base + 8 * o_add,
                src_row_stride=pk.data.size(-1) if pk.rows == 2 else 0,
                dst_row_stride=total if pk.rows == 2 else 0, n_rows=pk.rows, op=pk.op))
        L = _ffi.lib()
        s = _ffi.stream_ptr(dev)
        for i in range(0, len(descs), _ffi.MAX_COLLATE_DESCS):
            chunk = descs[i:i + _ffi.MAX_COLLATE_DESCS]
            arr = (_ffi.CollateDesc * len(chunk))(*chunk)
            _ffi.check(L.cwn_collate(arr, len(chunk), B, s), 'cwn_collate')
        tab.record_stream(torch.cuda.current_stream(dev))
        batch = ComplexBatch(*cochains, y=y, num_complexes=B, dimension=len(cochains) - 1)
        batch._collate_tables = tab     # keep the tables alive until the launch has consumed them
        return batch


class PackedLoader:
    """The reference's DataLoader (data/data_loading.py:84-111: `for batch in loader` of exp/train_utils.py:35)
    over an HBM-resident packed dataset: every batch is one `collate(indices)` launch on the device, no worker
    processes, no host collate.  `indices` selects a split (train / valid / test ids, data/datasets/dataset.py
    get_idx_split); `shuffle` draws a new permutation per epoch from `seed + epoch` (every rank draws the same one);
    with world > 1 rank r takes permutation entries r, r + world, ... of each GLOBAL batch -- the complexes of a
    batch shard across the ranks (cwn_amd/dist.py), and every rank sees the same number of batches (the tail that
    does not fill one complex per rank is dropped, as is an incomplete last batch with drop_last)."""

    def __init__(self, packed: PackedComplexes, batch_size: int = 1, shuffle: bool = False,
                 indices: Sequence[int] = None, drop_last: bool = False, seed: int = 0, rank: int = 0, world: int = 1):
        if batch_size < 1 or world < 1 or not (0 <= rank < world):
            raise ValueError('batch_size >= 1, world >= 1, 0 <= rank < world')
        self.packed, self.batch_size, self.shuffle, self.drop_last = packed, int(batch_size), shuffle, drop_last
        self.indices = np.arange(packed.num, dtype=np.int64) if indices is None else np.asarray(indices, dtype=np.int64)
        if self.indices.size and (self.indices.min() < 0 or self.indices.max() >= packed.num):
            raise IndexError('split index outside the dataset')
        self.seed, self.rank, self.world = int(seed), int(rank), int(world)
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def batches(self) -> List[np.ndarray]:
        """This rank's index lists for the current epoch."""
        order = self.indices
        if self.shuffle:
            order = order[np.random.default_rng(self.seed + self.epoch).permutation(order.size)]
        out = []
        for lo in range(0, order.size, self.batch_size):
            glob = order[lo:lo + self.batch_size]
            if glob.size < self.batch_size and self.drop_last:
                break
            if glob.size < self.world:          # not one complex per rank: a rank would sit out a collective
                break
            out.append(glob[self.rank::self.world])
        return out

    def __len__(self) -> int:
        n, b = self.indices.size, self.batch_size
        full, tail = divmod(n, b)
        return full + (1 if tail and not self.drop_last and tail >= self.world else 0) if b >= self.world else 0

    def __iter__(self):
        for idx in self.batches():
            yield self.packed.collate(idx)
        self.epoch += 1
