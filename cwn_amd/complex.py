"""Cochain / Complex containers: the input format of the hot path.

Host-side mirror of the reference's data/complex.py interface (Cochain :36-293, CochainBatch
:296-487, Complex :490-667, ComplexBatch :670-728) -- same constructor arguments, attribute names,
`get_cochain_params` / `get_all_cochain_params` / `set_xs` / `from_complex_list` semantics and the
same integer index layout (bit-exact against tests/golden/batching.npz) -- re-designed for the
engine:

  * batching is vectorised: one `torch.cat` per key plus one `repeat_interleave`d offset add,
    instead of a Python loop of per-complex adds (works on CPU tensors and on device tensors);
  * `up_attr` / `down_attr` are handed out LAZILY (`IndexedRows`) so the K3/K4 gathers of
    data/complex.py:579-588 never materialise `[E, F]` matrices for the fused layers
    (`lazy_attrs=False` restores the reference's dense tensors, produced by the HIP gather kernel);
  * `Complex.prepare()` converts every adjacency of the (batched) complex to int32 CSR with ONE
    batched call, once per batch; all layers and the backward pass reuse the plans.
"""
import copy
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

from .cell_mp import CochainMessagePassingParams, IndexedRows

INDEX_KEYS = ('upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries', 'boundary_index')


def _rows(src, index):
    from . import ops
    return ops.gather_rows(src, index)   # HIP gather kernel; GPU only, like everything else


class Cochain(object):
    """Vector-valued signal on the k-cells of a complex (data/complex.py:36-92).

    x [num_cells, F]; upper_index / lower_index [2, E] (int64); shared_coboundaries /
    shared_boundaries [E]; boundary_index [2, B] with row 0 = boundary cell (dim k-1), row 1 = cell."""

    def __init__(self, dim: int, x: Tensor = None, upper_index: Tensor = None,
                 lower_index: Tensor = None, shared_boundaries: Tensor = None,
                 shared_coboundaries: Tensor = None, mapping: Tensor = None,
                 boundary_index: Tensor = None, upper_orient=None, lower_orient=None, y=None,
                 **kwargs):
        if dim == 0:
            assert lower_index is None
            assert shared_boundaries is None
            assert boundary_index is None
        self.__dim__ = dim
        self._x = x
        self.upper_index = upper_index
        self.lower_index = lower_index
        self.boundary_index = boundary_index
        self.y = y
        self.shared_boundaries = shared_boundaries
        self.shared_coboundaries = shared_coboundaries
        self.upper_orient = upper_orient
        self.lower_orient = lower_orient
        self._mapping = mapping
        self.__num_cells__: Optional[int] = None
        self.__num_cells_up__: Optional[int] = None
        self.__num_cells_down__: Optional[int] = None
        for key, item in kwargs.items():
            if key == 'num_cells':
                self.__num_cells__ = item
            elif key == 'num_cells_down':
                self.__num_cells_down__ = item
            elif key == 'num_cells_up':
                self.__num_cells_up__ = item
            else:
                setattr(self, key, item)

    # ---- basic properties -------------------------------------------------------------------
    @property
    def dim(self):
        return self.__dim__

    @property
    def x(self):
        return self._x

    @x.setter
    def x(self, new_x):
        if new_x is not None:
            n = self.__num_cells__                    # (the common case without the property's detours: set_xs per layer)
            if n is None:
                n = self.num_cells
            assert n == (new_x.size(0) if isinstance(new_x, Tensor) else len(new_x))
        self._x = new_x

    @property
    def mapping(self):
        return self._mapping

    @property
    def keys(self):
        return [k for k in ('x', 'upper_index', 'lower_index', 'shared_boundaries',
                            'shared_coboundaries', 'boundary_index', 'upper_orient', 'lower_orient',
                            'y', 'batch', 'ptr') if getattr(self, k, None) is not None]

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys

    @property
    def num_cells(self):
        """data/complex.py:182-192."""
        if self.__num_cells__ is not None:
            return self.__num_cells__
        if self.x is not None:
            return self.x.size(0)
        if self.boundary_index is not None:
            return int(self.boundary_index[1, :].max()) + 1
        assert self.upper_index is None and self.lower_index is None
        return None

    @num_cells.setter
    def num_cells(self, n):
        self.__num_cells__ = n

    @property
    def num_cells_up(self):
        """data/complex.py:200-209."""
        if self.__num_cells_up__ is not None:
            return self.__num_cells_up__
        if self.shared_coboundaries is not None:
            assert self.upper_index is not None
            return int(self.shared_coboundaries.max()) + 1
        assert self.upper_index is None
        return 0

    @num_cells_up.setter
    def num_cells_up(self, n):
        self.__num_cells_up__ = n

    @property
    def num_cells_down(self):
        """data/complex.py:217-226."""
        if self.dim == 0:
            return None
        if self.__num_cells_down__ is not None:
            return self.__num_cells_down__
        if self.lower_index is None:
            return 0
        raise ValueError('Cannot infer the number of cells in the cochain below.')

    @num_cells_down.setter
    def num_cells_down(self, n):
        self.__num_cells_down__ = n

    @property
    def num_features(self):
        if self.x is None:
            return 0
        return 1 if self.x.dim() == 1 else self.x.size(1)

    # ---- tensor plumbing ------------------------------------------------------------------------
    def apply(self, func, *keys):
        for k in (keys or self.keys):
            v = getattr(self, k, None)
            if torch.is_tensor(v):
                if k == 'x':
                    self._x = func(v)
                else:
                    setattr(self, k, func(v))
        return self

    def contiguous(self, *keys):
        return self.apply(lambda t: t.contiguous(), *keys)

    def to(self, device, *keys, **kwargs):
        return self.apply(lambda t: t.to(device, **kwargs), *keys)

    def clone(self):
        new = copy.copy(self)
        for k in self.keys:
            v = getattr(self, k)
            if torch.is_tensor(v):
                if k == 'x':
                    new._x = v.clone()
                else:
                    setattr(new, k, v.clone())
        return new


def _offsets(counts: Sequence[int]) -> List[int]:
    out, run = [], 0
    for c in counts:
        out.append(run)
        run += c
    return out


def _cat_with_offsets(items: List[Tensor], offsets: List, two_rows: bool) -> Tensor:
    """cat(items, -1) + per-item offset, vectorised.  `offsets[i]` is an int, or a (row0, row1) pair
    for boundary_index."""
    flat = torch.cat(items, dim=-1)
    lengths = torch.tensor([t.size(-1) for t in items], dtype=torch.long)
    if two_rows:
        off = torch.tensor(offsets, dtype=flat.dtype).t()            # [2, n_items]
        off = torch.repeat_interleave(off, lengths, dim=1)
    else:
        off = torch.repeat_interleave(torch.tensor(offsets, dtype=flat.dtype), lengths)
    return (flat + off.to(flat.device)).contiguous()


class CochainBatch(Cochain):
    """A batch of cochains stored as one big cochain over disconnected cells
    (data/complex.py:296-458).  `batch[i]` is the position (in the list) of the complex cell i
    belongs to."""

    def __init__(self, dim, batch=None, ptr=None, **kwargs):
        super().__init__(dim, **kwargs)
        self.batch = batch
        self.ptr = ptr
        self.__num_cochains__ = None

    @classmethod
    def from_cochain_list(cls, data_list: List[Cochain], follow_batch=()):
        """Offsets per key follow data/complex.py:148-169: upper/lower_index by the cells of this
        dimension seen so far, shared_boundaries by the cells below, shared_coboundaries by the
        cells above, boundary_index by (cells below, cells here)."""
        dim = data_list[0].dim
        n_here = [c.num_cells for c in data_list]
        inc_here = [n or 0 for n in n_here]
        # (the reference asks for an increment only when the key it offsets is present, data/complex.py:344-369: a cochain of
        #  edges with adjacencies but no shared cells / boundaries -- mp/models.py EdgeOrient's input -- batches without them)
        def below(c):
            if dim == 0:
                return 0
            if c.shared_boundaries is None and c.boundary_index is None and c.__num_cells_down__ is None:
                return 0
            return c.num_cells_down or 0

        def above(c):
            if c.shared_coboundaries is None and c.__num_cells_up__ is None:
                return 0
            return c.num_cells_up or 0
        inc_down = [below(c) for c in data_list]
        inc_up = [above(c) for c in data_list]
        off_here, off_down, off_up = _offsets(inc_here), _offsets(inc_down), _offsets(inc_up)

        out = cls(dim)
        out.__num_cochains__ = len(data_list)
        key_offsets = {
            'upper_index': lambda i: off_here[i], 'lower_index': lambda i: off_here[i],
            'shared_boundaries': lambda i: off_down[i],
            'shared_coboundaries': lambda i: off_up[i],
            'boundary_index': lambda i: (off_down[i], off_here[i]),
        }
        slices = {}
        for key in ('x', 'y', 'upper_orient', 'lower_orient') + INDEX_KEYS:
            present = [(i, c[key]) for i, c in enumerate(data_list) if c[key] is not None]
            if key in INDEX_KEYS:
                # data/complex.py:349-394 `__slices__`: where each complex's entries start in the
                # batched index (a complex without the key contributes an empty range)
                run, offs = 0, [0]
                for c in data_list:
                    t = c[key]
                    run += 0 if t is None else int(t.size(-1))
                    offs.append(run)
                slices[key] = offs
            if not present:
                continue
            items = [t.unsqueeze(0) if t.dim() == 0 else t for _, t in present]
            if key in key_offsets:
                offs = [key_offsets[key](i) for i, _ in present]
                val = _cat_with_offsets(items, offs, two_rows=(key == 'boundary_index'))
            else:
                val = torch.cat(items, dim=0).contiguous()
            if key == 'x':
                out._x = val
            else:
                setattr(out, key, val)
        have = [(i, n) for i, n in enumerate(n_here) if n is not None]
        if have:
            ids = torch.tensor([i for i, _ in have], dtype=torch.long)
            cnt = torch.tensor([n for _, n in have], dtype=torch.long)
            out.batch = torch.repeat_interleave(ids, cnt)
            out.ptr = [0] + torch.cumsum(cnt, 0).tolist()
            dev = next((t.device for t in (out._x, out.upper_index, out.boundary_index)
                        if t is not None), None)
            if dev is not None:
                out.batch = out.batch.to(dev)
        out.__num_cells__ = sum(inc_here)
        out.__num_cells_up__ = sum(inc_up)
        if dim > 0:
            out.__num_cells_down__ = sum(inc_down)
        out.__num_cells_list__ = n_here
        out.__slices__ = slices
        return out

    @property
    def num_cochains(self) -> int:
        return self.__num_cochains__


class Complex(object):
    """A cochain complex: one Cochain per dimension (data/complex.py:490-667)."""

    lazy_attrs = True   # hand out up_attr / down_attr as IndexedRows (see module docstring)

    def __init__(self, *cochains: Cochain, y: Tensor = None, dimension: int = None):
        if len(cochains) == 0:
            raise ValueError('At least one cochain is required.')
        if dimension is None:
            dimension = len(cochains) - 1
        if len(cochains) < dimension + 1:
            raise ValueError(f'Not enough cochains passed, expected {dimension + 1}, '
                             f'received {len(cochains)}')
        self.dimension = dimension
        self.cochains: Dict[int, Cochain] = {i: cochains[i] for i in range(dimension + 1)}
        self.nodes = cochains[0]
        self.edges = cochains[1] if dimension >= 1 else None
        self.two_cells = cochains[2] if dimension >= 2 else None
        self.y = y
        self._consolidate()

    def _consolidate(self):
        """data/complex.py:518-537."""
        for dim in range(self.dimension + 1):
            c = self.cochains[dim]
            assert c.dim == dim
            if dim < self.dimension:
                n_up = self.cochains[dim + 1].num_cells
                assert n_up is not None
                c.num_cells_up = n_up
            if dim > 0:
                n_down = self.cochains[dim - 1].num_cells
                assert n_down is not None
                c.num_cells_down = n_down

    def to(self, device, **kwargs):
        for dim in range(self.dimension + 1):
            self.cochains[dim] = self.cochains[dim].to(device, **kwargs)
        if self.y is not None:
            self.y = self.y.to(device, **kwargs)
        return self

    # ---- engine extension: convert every adjacency once per batch -----------------------------
    def prepare(self, max_dim: int = 2, include_down: bool = False, backward: bool = False,
                overlap: bool = False, upper: bool = True):
        """Build the int32 CSR plans of all upper / boundary (and optionally lower) adjacencies with
        one batched call and register them in the plan cache `propagate` looks up.  Optional: a
        propagate call on an unprepared complex builds its plans on first use.  `overlap=True`
        runs the build on a side stream (see csr.build_many).  `upper=False`: the boundary adjacencies only -- for a caller
        that knows its layers take the complex-blocked launches, which read the int64 entries themselves (a training step over a
        fixed model: cwn_amd/train.py; a plan that is needed after all is still built on first use)."""
        from .csr import build_many, cached_adjacency
        todo = []
        for dim in range(min(max_dim, self.dimension) + 1):
            c = self.cochains[dim]
            n = c.num_cells
            specs = []
            if upper and (dim + 1) in self.cochains and c.upper_index is not None:
                specs.append((c.upper_index, n, n, c.shared_coboundaries, self.cochains[dim + 1].num_cells))
            if upper and include_down and c.lower_index is not None:
                specs.append((c.lower_index, n, n, c.shared_boundaries, self.cochains[dim - 1].num_cells))
            if c.boundary_index is not None and dim > 0:
                specs.append((c.boundary_index, n, self.cochains[dim - 1].num_cells, None, 0))
            for index, n_dst, n_src, aux, n_aux in specs:
                if not index.is_cuda:
                    raise RuntimeError('Complex.prepare() needs the complex on the GPU (.to(device))')
                adj = cached_adjacency(index, n_dst, n_src, aux, n_aux, build=False)
                todo.append(adj)
                if backward:
                    adj.transposes()
                    todo += [t for t in (adj._t_src, adj._t_aux) if t is not None]
        build_many(todo, overlap=overlap)
        return self

    def forget_plans(self):
        """Drop this complex's plans from the plan cache (the next prepare() / propagate rebuilds
        them): what a training loop does when a batch object is refilled, without touching the
        plans of other batches."""
        from . import csr
        # the per-complex tables are cut from THESE index tensors (`__slices__` / `ptr`): a refilled batch
        # gets a fresh plan, re-validated against its entries on the next launch (a stale table
        # would have the kernel drop entries outside their complex without any host-side error)
        old = getattr(self, '_block_plan', None)
        if old is not None:
            # (the device copies of `ptr` survive when the next plan has the same cells per complex: an upload is not
            # something a stream capture can hold, and a training step calls this inside its captured graph)
            # ... and so do the item tables (ranges per complex, cut from the SAME prefix sums: entry VALUES are checked by
            # the kernel on every launch) -- the training forward runs the blocked kernel inside its captured graph too
            if old[1] is not None and (old[1].__dict__.get('_cell_ptr_dev') or old[1]._tables):
                self._cell_ptr_keep = (old[1].cell_ptr, old[1].__dict__.get('_cell_ptr_dev'), old[1].up_ptr, old[1].b_ptr,
                                       old[1]._tables)
            self._block_plan = None
        for c in self.cochains.values():
            for index in (c.upper_index, c.lower_index, c.boundary_index):
                if index is not None:
                    csr._cache.pop(id(index), None)
                    flipped = getattr(index, '_cwn_flipped', None)
                    if flipped is not None:
                        csr._cache.pop(id(flipped), None)
            pooled = getattr(getattr(c, 'batch', None), '_cwn_index', None)     # the readout's plan
            if pooled is not None:
                csr._cache.pop(id(pooled), None)
        return self

    # ---- propagate arguments ------------------------------------------------------------------
    def get_cochain_params(self, dim: int, max_dim: int = 2, include_top_features=True,
                           include_down_features=True,
                           include_boundary_features=True, _plan=False) -> CochainMessagePassingParams:
        """data/complex.py:548-602."""
        if dim not in self.cochains:
            raise NotImplementedError(f'Dim {dim} is not present in the complex or not yet supported.')
        cochains = self.cochains
        cells = cochains[dim]
        x = cells._x
        lazy = self.lazy_attrs

        upper_index, upper_features = None, None
        up_c = cochains.get(dim + 1)
        if cells.upper_index is not None and up_c is not None:
            upper_index = cells.upper_index
            xu = up_c._x
            if xu is not None and (dim < max_dim or include_top_features):
                upper_features = IndexedRows(xu, cells.shared_coboundaries) if lazy else _rows(xu, cells.shared_coboundaries)
        lower_index, lower_features = None, None
        down_x = cochains[dim - 1]._x if dim > 0 else None
        if include_down_features and cells.lower_index is not None:
            lower_index = cells.lower_index
            if down_x is not None:
                lower_features = IndexedRows(down_x, cells.shared_boundaries) if lazy else _rows(down_x, cells.shared_boundaries)
        boundary_index, boundary_features = None, None
        if include_boundary_features and cells.boundary_index is not None:
            boundary_index = cells.boundary_index
            if down_x is not None:
                boundary_features = down_x
        # (= CochainMessagePassingParams(x, upper_index, lower_index, up_attr=..., down_attr=..., boundary_attr=...,
        #  boundary_index=...), mp/cell_mp.py:527-550, field by field: this runs once per dimension per layer of an eager forward)
        params = CochainMessagePassingParams.__new__(CochainMessagePassingParams)
        params.x, params.up_index, params.down_index = x, upper_index, lower_index
        params.kwargs = {'up_attr': upper_features, 'down_attr': lower_features, 'boundary_attr': boundary_features,
                         'boundary_index': boundary_index}
        params.boundary_index, params.boundary_attr = boundary_index, boundary_features
        # engine extension: what the co-boundary stream of this dimension reads -- the
        # next dimension's boundary_index (row 0 = a cell of THIS dimension, row 1 = its coface) and the
        # cofaces' features.  Plain attributes: the reference's kwargs are left as they are.
        params.coboundary_index = params.coboundary_attr = None
        if up_c is not None and up_c.boundary_index is not None:
            params.coboundary_index = up_c.boundary_index
            params.coboundary_attr = up_c._x
        n_cells = cells.__num_cells__
        params.num_cells = n_cells if n_cells is not None else cells.num_cells   # engine extension: sizes without a device sync
        params.block_plan = self.block_plan() if _plan is False else _plan  # engine extension: the batch's item table (or None)
        return params

    def block_plan(self):
        """The per-complex partition of this (batched) complex for the complex-blocked layer kernel
        (cwn_amd/blockplan.py), from the collate's `ptr` / `__slices__` tables; None for a complex
        that does not carry them or is not on the GPU."""
        x0 = next((c.upper_index if c.upper_index is not None else c.x for c in self.cochains.values()
                   if c.upper_index is not None or c.x is not None), None)
        if x0 is None or not x0.is_cuda:
            return None
        cached = getattr(self, '_block_plan', None)
        if cached is None or cached[0] != x0.device:
            from .blockplan import BlockPlan
            cached = (x0.device, BlockPlan.from_batch(self))
            keep = getattr(self, '_cell_ptr_keep', None)
            if keep is not None and cached[1] is not None:
                import numpy as np
                same = lambda xs, ys: len(xs) == len(ys) and all((a is None and b is None) or (
                    a is not None and b is not None and np.array_equal(a, b)) for a, b in zip(xs, ys))
                if same(keep[0], cached[1].cell_ptr):
                    if keep[1]:
                        cached[1].__dict__['_cell_ptr_dev'] = keep[1]
                    if same(keep[2], cached[1].up_ptr) and same(keep[3], cached[1].b_ptr):
                        cached[1]._tables = keep[4]
                        for t in keep[4].values():          # (their per-item CSR caches belong to the old index tensors' values)
                            if t is not None:
                                t.csr_key = None            # (a MixedTable passes it on to its parts)
                self._cell_ptr_keep = None
            self._block_plan = cached
        return cached[1]

    def get_all_cochain_params(self, max_dim: int = 2, include_top_features=True,
                               include_down_features=True,
                               include_boundary_features=True) -> List[CochainMessagePassingParams]:
        """data/complex.py:604-626."""
        plan = self.block_plan()
        return [self.get_cochain_params(d, max_dim=max_dim,
                                        include_top_features=include_top_features,
                                        include_down_features=include_down_features,
                                        include_boundary_features=include_boundary_features, _plan=plan)
                for d in range(min(max_dim, self.dimension) + 1)]

    def get_labels(self, dim=None):
        if dim is None:
            return self.y
        if dim in self.cochains:
            return self.cochains[dim].y
        raise NotImplementedError(f'Dim {dim} is not present in the complex or not yet supported.')

    def set_xs(self, xs: List[Tensor]):
        assert (self.dimension + 1) >= len(xs)
        for i, x in enumerate(xs):
            self.cochains[i].x = x

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith('_')]

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys


class ComplexBatch(Complex):
    """A batch of complexes stored as one complex of batched cochains (data/complex.py:670-728)."""

    def __init__(self, *cochains: CochainBatch, dimension: int, y: Tensor = None,
                 num_complexes: int = None):
        super().__init__(*cochains, y=y)
        self.num_complexes = num_complexes
        self.dimension = dimension

    @classmethod
    def from_complex_list(cls, data_list: List[Complex], follow_batch=(), max_dim: int = 2):
        dimension = min(max(d.dimension for d in data_list), max_dim)
        per_dim: List[List[Cochain]] = [[] for _ in range(dimension + 1)]
        labels = []
        for comp in data_list:
            for dim in range(dimension + 1):
                if dim in comp.cochains:
                    per_dim[dim].append(comp.cochains[dim])
                else:
                    # a complex without this dimension still shifts later boundary indices by its
                    # number of (dim-1)-cells (data/complex.py:709-716)
                    filler = Cochain(dim=dim)
                    if dim - 1 in comp.cochains:
                        filler.num_cells_down = comp.cochains[dim - 1].num_cells
                    per_dim[dim].append(filler)
            labels.append(comp.y)
        batched = [CochainBatch.from_cochain_list(lst) for lst in per_dim]
        y = torch.cat(labels, 0) if all(l is not None for l in labels) else None
        return cls(*batched, y=y, num_complexes=len(data_list), dimension=dimension)
