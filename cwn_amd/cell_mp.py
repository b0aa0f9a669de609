"""Drop-in `CochainMessagePassing` for MI355X.

Mirrors the call contract of the reference's mp/cell_mp.py (constructor :81-91, `propagate`
:357-363, overridable hooks :394-524, exceptions :153-193) so that layer / model code written
against it runs unchanged, while every gather / scatter of the path runs in the hand-written HIP
kernels behind the C ABI (include/cwn_hip.h):

  reference call site                         here
  ------------------------------------------  ---------------------------------------------------
  index_select in __lift__   (:198)           cwn_gather_rows_f32          (generic hooks only)
  torch_scatter.scatter      (:439,458,478)   cwn_aggregate_f32, CSR segmented reduce, no atomics
  torch.zeros in update      (:517-522)       zero rows written by the same kernel launch
  gather+message+scatter of an un-overridden  ONE fused cwn_aggregate_f32 launch for all present
  message hook                                and absent streams of the call

Three execution paths, chosen per adjacency:
  1. a subclass implements `message_and_aggregate_<adj>(adj_t, ...)` (the reference's own fusion
     extension point, :481-509; here `adj_t` is a `cwn_amd.csr.Adjacency`) -> it is called;
  2. neither `message_<adj>` nor `aggregate_<adj>` is overridden (base-class identity message)
     -> fused gather-reduce;
  3. anything else -> gather kernel -> the Python hook -> segmented-reduce kernel.  Always
     correct, materialises the messages like the reference does.
All three are differentiable.  GPU only: tensors on the CPU raise (the CPU checker is oracle/).
"""
import inspect
from typing import Dict, List, Optional, Set

import torch
from torch import Tensor

from . import ops
from .csr import Adjacency, cached_adjacency

ADJACENCIES = ('up', 'down', 'boundary')
_EMPTY = inspect.Parameter.empty


class IndexedRows:
    """A lazily gathered attribute matrix: rows `src[index]`.

    `Complex.get_cochain_params` hands `up_attr` / `down_attr` in this form instead of running the
    K3 / K4 gathers of data/complex.py:579-580, 587-588; fused layers read `src` through the
    adjacency's shared-cell index and never materialise the `[E, F]` matrix.  `tensor()` gives the
    reference's materialised form (differentiable) for code that wants it."""

    def __init__(self, src: Tensor, index: Tensor):
        self.src, self.index = src, index
        self._dense: Optional[Tensor] = None

    def tensor(self) -> Tensor:
        if self._dense is None:
            self._dense = ops.gather_rows(self.src, self.index)
        return self._dense

    def size(self, dim=None):
        shape = (self.index.numel(), self.src.size(1))
        return shape if dim is None else shape[dim]

    @property
    def shape(self):
        return self.size()

    @property
    def device(self):
        return self.src.device


def dense(t):
    return t.tensor() if isinstance(t, IndexedRows) else t


class CochainMessagePassing(torch.nn.Module):
    """See the module docstring; argument meaning as in mp/cell_mp.py:41-91."""

    special_args: Set[str] = {
        f'{a}_{s}' for a in ADJACENCIES
        for s in ('index', 'adj_t', 'index_i', 'index_j', 'size', 'size_i', 'size_j', 'ptr', 'dim_size')
    } | {f'agg_{a}_index' for a in ADJACENCIES}

    def __init__(self, up_msg_size, down_msg_size, aggr_up: Optional[str] = 'add',
                 aggr_down: Optional[str] = 'add', aggr_boundary: Optional[str] = 'add',
                 flow: str = 'source_to_target', node_dim: int = -2, boundary_msg_size=None,
                 use_down_msg=True, use_boundary_msg=True):
        super().__init__()
        self.up_msg_size = up_msg_size
        self.down_msg_size = down_msg_size
        self.boundary_msg_size = down_msg_size if boundary_msg_size is None else boundary_msg_size
        self.use_down_msg = use_down_msg
        self.use_boundary_msg = use_boundary_msg
        self.aggr_up, self.aggr_down, self.aggr_boundary = aggr_up, aggr_down, aggr_boundary
        assert self.aggr_up in ['add', 'mean', 'max', None]
        assert self.aggr_down in ['add', 'mean', 'max', None]
        self.flow = flow
        assert self.flow in ['source_to_target', 'target_to_source']
        self.node_dim = node_dim
        if node_dim not in (-2, 0):
            raise ValueError('the MI355X engine propagates along dim -2 of [cells, features] matrices')

        # hook signatures, recorded once (the reference does this with PyG's Inspector, :114-127)
        self._sig: Dict[str, List[inspect.Parameter]] = {}
        for a in ADJACENCIES:
            self._record(f'message_{a}', 0)
            self._record(f'aggregate_{a}', 1)
            self._record(f'message_and_aggregate_{a}', 1)
        self._record('update', 3)
        base = CochainMessagePassing
        cls = type(self)
        self._overrides = {name: getattr(cls, name) is not getattr(base, name) for name in self._sig}
        self.fuse_up = self._overrides['message_and_aggregate_up']
        self.fuse_down = self._overrides['message_and_aggregate_down']
        self.fuse_boundary = self._overrides['message_and_aggregate_boundary']
        self._live_adj: Dict[str, Optional[Adjacency]] = {}

    # ---- hook signature routing ---------------------------------------------------------
    def _record(self, name: str, skip: int):
        params = list(inspect.signature(getattr(self, name)).parameters.values())
        self._sig[name] = params[skip:]

    def _args_of(self, names) -> Set[str]:
        out = set()
        for n in names:
            out |= {p.name for p in self._sig[n]}
        return out

    def _distribute(self, name: str, pool: Dict):
        """Pick from `pool` what hook `name` declares; TypeError when a required one is missing
        (PyG Inspector.distribute semantics)."""
        out = {}
        for p in self._sig[name]:
            v = pool.get(p.name, _EMPTY)
            if v is _EMPTY:
                if p.default is _EMPTY:
                    raise TypeError(f'Required parameter {p.name} is empty.')
                v = p.default
            out[p.name] = v
        return out

    # ---- input checks (mp/cell_mp.py:146-193) -------------------------------------------
    def __check_input_together__(self, index_up, index_down, size_up, size_down):
        if (index_up is not None and index_down is not None
                and size_up is not None and size_down is not None):
            assert size_up[0] == size_down[0]
            assert size_up[1] == size_down[1]

    def __check_input_separately__(self, index, size) -> List[Optional[int]]:
        the_size: List[Optional[int]] = [None, None]
        if isinstance(index, Tensor):
            assert index.dtype == torch.long
            assert index.dim() == 2
            assert index.size(0) == 2
            if size is not None:
                the_size[0], the_size[1] = size[0], size[1]
            return the_size
        if index is None:
            return the_size
        raise ValueError('`MessagePassing.propagate` only supports `torch.LongTensor` of shape '
                         '`[2, num_messages]` for argument `edge_index` '
                         '(the SparseTensor branch of the reference is dead code, SURVEY.md §2.2).')

    def __set_size__(self, size: List[Optional[int]], dim: int, src: Tensor):
        the_size = size[dim]
        if the_size is None:
            size[dim] = src.size(0)
        elif the_size != src.size(0):
            raise ValueError(f'Encountered tensor with size {src.size(0)} in dimension '
                             f'{self.node_dim}, but expected size {the_size}.')

    # ---- per-adjacency plumbing ------------------------------------------------------------
    def _ij(self):
        return (1, 0) if self.flow == 'source_to_target' else (0, 1)

    def _source_of(self, adjacency: str, kwargs: Dict):
        """(matrix the `_j` rows are gathered from, matrix that fixes the row count, aux src)"""
        x = kwargs.get('x', _EMPTY)
        if adjacency == 'boundary':
            return kwargs.get('boundary_attr', _EMPTY), x
        return x, x

    def _adjacency(self, index: Tensor, adjacency: str, size, kwargs) -> Adjacency:
        """The CSR plan of `index` (cached per index tensor, built by the HIP kernels)."""
        i, j = self._ij()
        data, size_data = self._source_of(adjacency, kwargs)
        if isinstance(size_data, Tensor):
            self.__set_size__(size, 0, size_data)
        n_dst = size[1] or size[0]
        n_src = data.size(0) if isinstance(data, Tensor) else (size[0] or size[1])
        attr = kwargs.get({'up': 'up_attr', 'down': 'down_attr'}.get(adjacency, ''), None)
        aux, n_aux = (attr.index, attr.src.size(0)) if isinstance(attr, IndexedRows) else (None, 0)
        if i == 1:
            return cached_adjacency(index, n_dst, n_src, aux, n_aux)
        flipped = getattr(index, '_cwn_flipped', None)   # target_to_source: swap the roles once
        if flipped is None:
            flipped = index.flip(0).contiguous()
            index._cwn_flipped = flipped
        return cached_adjacency(flipped, n_dst, n_src, aux, n_aux)

    def _collect(self, args: Set[str], index, size, adjacency: str, kwargs: Dict,
                 adj: Optional[Adjacency], keep_lazy: bool = False) -> Dict:
        """mp/cell_mp.py:209-282: build the argument pool for the hooks of one adjacency."""
        i, j = self._ij()
        pre = adjacency + '_'
        out = {}
        for arg in args:
            if arg[-2:] not in ('_i', '_j'):
                v = kwargs.get(arg, _EMPTY)
                out[arg] = v if keep_lazy else dense(v)
            elif index is not None:
                if not arg.startswith(pre):
                    continue
                name = arg[len(pre):-2]
                want_j = arg.endswith('_j')
                if adjacency == 'boundary' and want_j:
                    data, size_data = kwargs.get('boundary_attr', _EMPTY), kwargs.get(name, _EMPTY)
                else:
                    data = kwargs.get(name, _EMPTY)
                    size_data = data
                if isinstance(data, (tuple, list)):
                    raise ValueError('This format is not supported for cellular message passing')
                data = dense(data)
                if isinstance(data, Tensor):
                    self.__set_size__(size, 0 if want_j else 1, dense(size_data))
                    row = j if want_j else i
                    provider = None
                    if adj is not None:
                        # backward of the gather = segmented sum over the CSR keyed on that row:
                        # `_i` rows are the plan's own key, `_j` rows the transposed plan's
                        provider = (lambda a=adj: a.t_src) if want_j else (lambda a=adj: a)
                    data = ops.gather_rows(data, index[row], provider)
                out[arg] = data
        if isinstance(index, Tensor):
            out[f'{adjacency}_adj_t'] = adj
            out[f'{adjacency}_ptr'] = None
            out[f'{adjacency}_index'] = index
            out[f'{adjacency}_index_i'] = index[i]
            out[f'{adjacency}_index_j'] = index[j]
            out[f'agg_{adjacency}_index'] = index[i]
        out[f'{adjacency}_size'] = size
        out[f'{adjacency}_size_i'] = size[1] or size[0]
        out[f'{adjacency}_size_j'] = size[0] or size[1]
        out[f'{adjacency}_dim_size'] = out[f'{adjacency}_size_i']
        return out

    def _identity_path(self, adjacency: str) -> bool:
        return not (self._overrides[f'message_{adjacency}'] or self._overrides[f'aggregate_{adjacency}']
                    or self._overrides[f'message_and_aggregate_{adjacency}'])

    def _require_hook_args(self, adjacency: str, kwargs: Dict):
        """The fused identity path never calls the message hook, but a caller that forgets one of
        its required arguments must still get the TypeError the reference raises."""
        pre = adjacency + '_'
        for p in self._sig[f'message_{adjacency}']:
            if p.default is not _EMPTY:
                continue
            name = p.name
            if name[-2:] in ('_i', '_j') and name.startswith(pre):
                name = 'boundary_attr' if (adjacency == 'boundary' and name.endswith('_j')) else name[len(pre):-2]
            if name not in kwargs:
                raise TypeError(f'Required parameter {p.name} is empty.')

    def _hook_path(self, index, adjacency: str, size, kwargs, adj: Adjacency) -> Tensor:
        """Fused user hook if implemented, else gather -> message hook -> aggregate hook."""
        self._live_adj[adjacency] = adj
        try:
            if self._overrides[f'message_and_aggregate_{adjacency}']:
                pool = self._collect(self._args_of([f'message_and_aggregate_{adjacency}']), index,
                                     size, adjacency, kwargs, adj, keep_lazy=True)
                fused = getattr(self, f'message_and_aggregate_{adjacency}')
                args = self._distribute(f'message_and_aggregate_{adjacency}', pool)
                args.pop(f'{adjacency}_adj_t', None)
                return fused(adj, **args)
            names = [f'message_{adjacency}', f'aggregate_{adjacency}']
            pool = self._collect(self._args_of(names), index, size, adjacency, kwargs, adj)
            msg = getattr(self, f'message_{adjacency}')(**self._distribute(names[0], pool))
            return getattr(self, f'aggregate_{adjacency}')(msg, **self._distribute(names[1], pool))
        finally:
            self._live_adj[adjacency] = None

    # ---- the hot path -------------------------------------------------------------------------
    def propagate(self, up_index: Optional[Tensor], down_index: Optional[Tensor],
                  boundary_index: Optional[Tensor], up_size=None, down_size=None,
                  boundary_size=None, **kwargs):
        """mp/cell_mp.py:357-392.  Returns (up_out, down_out, boundary_out), each `[N_d, msg_size]`."""
        up_size = self.__check_input_separately__(up_index, up_size)
        down_size = self.__check_input_separately__(down_index, down_size)
        boundary_size = self.__check_input_separately__(boundary_index, boundary_size)
        self.__check_input_together__(up_index, down_index, up_size, down_size)

        x = kwargs.get('x', None)
        active = {
            'up': (up_index, up_size) if up_index is not None else None,
            'down': (down_index, down_size) if (self.use_down_msg and down_index is not None) else None,
            'boundary': ((boundary_index, boundary_size)
                         if (self.use_boundary_msg and kwargs.get('boundary_attr', None) is not None)
                         else None),
        }
        if active['boundary'] is not None and boundary_index is None:
            # the reference would crash inside __collect__ on index[i] with index None
            raise TypeError("'NoneType' object is not subscriptable: boundary_attr given without "
                            "boundary_index")
        outs: Dict[str, Optional[Tensor]] = {a: None for a in ADJACENCIES}
        fused: List[ops.Stream] = []
        fused_names: List[str] = []
        for a in ADJACENCIES:
            if active[a] is None:
                continue
            index, size = active[a]
            adj = self._adjacency(index, a, size, kwargs)
            if self._identity_path(a):
                self._require_hook_args(a, kwargs)
                src = kwargs['boundary_attr'] if a == 'boundary' else x
                fused.append(ops.Stream(adj=adj, n_dst=size[1] or size[0], width=int(src.size(1)),
                                        A=src, reduce=getattr(self, f'aggr_{a}') or 'add'))
                fused_names.append(a)
            else:
                outs[a] = self._hook_path(index, a, size, kwargs, adj)

        # update's argument pool (mp/cell_mp.py:384-391)
        pool = {}
        upd_args = self._args_of(['update'])
        for a, index, size in (('up', up_index, up_size), ('down', down_index, down_size)):
            pool.update(self._collect(upd_args, index, size, a, kwargs, None))
        upd_kwargs = self._distribute('update', pool)

        # absent streams: when `update` is the base one, their zero rows come out of the same launch
        zero_in_kernel = not self._overrides['update'] and isinstance(x, Tensor) and x.is_cuda
        if zero_in_kernel:
            for a, width in (('up', self.up_msg_size), ('down', self.down_msg_size),
                             ('boundary', self.boundary_msg_size)):
                if outs[a] is None and a not in fused_names:
                    fused.append(ops.Stream(adj=None, n_dst=x.size(0), width=int(width)))
                    fused_names.append(a)
        if fused:
            if all(st.adj is None for st in fused):
                for a, st in zip(fused_names, fused):
                    outs[a] = ops.zeros_rows(st.n_dst, st.width, x.device)
            else:
                for a, o in zip(fused_names, ops.aggregate_many(fused)):
                    outs[a] = o
        return self.update(outs['up'], outs['down'], outs['boundary'], **upd_kwargs)

    # ---- co-boundary stream (engine extension) ------------------------------------
    def propagate_coboundary(self, boundary_index_up: Tensor, coboundary_attr: Tensor, n_cells: int,
                             reduce: str = 'add') -> Tensor:
        """out[i] = reduce over the cofaces c of cell i of message_coboundary(coboundary_attr[c]).

        The aggregation the reference leaves as a TODO (mp/cell_mp.py:44 "Add support for co-boundary
        adjacencies", README.md:178): cells of dimension d receive from the (d+1)-cells they bound.
        `boundary_index_up` is the NEXT dimension's `boundary_index` exactly as data/complex.py
        delivers it ([2, B], row 0 = cell of this dimension, row 1 = its coface); the stream runs
        over the transposed plan of that adjacency -- the structure the backward pass of the
        boundary stream already uses -- so nothing new is built for a batch that trains.
        It is the ADJOINT of the boundary stream of dimension d+1: <cob(v), w> = <v, bnd(w)>.
        No reference oracle exists: parity unpinned (property-tested: adjointness, exact integers)."""
        if boundary_index_up.dtype != torch.long or boundary_index_up.dim() != 2 or boundary_index_up.size(0) != 2:
            raise AssertionError('boundary_index_up must be a [2, B] LongTensor')
        if coboundary_attr.dim() != 2:
            raise ValueError('coboundary_attr must be [cells of dimension d+1, features]')
        n_up = int(coboundary_attr.size(0))
        adj = cached_adjacency(boundary_index_up, n_up, int(n_cells))      # keyed on the coface (row 1)
        t = adj.t_src                                                       # keyed on this dimension's cell
        msg = self.message_coboundary(coboundary_attr) if type(self).message_coboundary is not \
            CochainMessagePassing.message_coboundary else None
        if msg is None:       # identity message: fused gather-reduce through the transposed plan
            return ops.aggregate(t, int(n_cells), coboundary_attr, reduce=reduce)
        if msg.size(0) != n_up:
            raise ValueError('message_coboundary must return one row per (d+1)-cell')
        return ops.aggregate(t, int(n_cells), msg, reduce=reduce)

    def message_coboundary(self, coboundary_x_j: Tensor) -> Tensor:
        """Per COFACE (not per entry: the message of a coface is the same for every cell it sends to)."""
        return coboundary_x_j

    # ---- overridable hooks (same names / signatures as mp/cell_mp.py:394-524) -------------------
    def message_up(self, up_x_j: Tensor, up_attr: Tensor) -> Tensor:
        return up_x_j

    def message_down(self, down_x_j: Tensor, down_attr: Tensor) -> Tensor:
        return down_x_j

    def message_boundary(self, boundary_x_j: Tensor):
        return boundary_x_j

    def _aggregate(self, adjacency: str, inputs: Tensor, agg_index: Tensor, dim_size, aggr) -> Tensor:
        adj = self._live_adj.get(adjacency)   # set while propagate runs this adjacency's hooks
        if adj is None or adj.n_entries != agg_index.numel():
            n = int(dim_size) if dim_size is not None else int(agg_index.max()) + 1
            adj = Adjacency.from_index(torch.stack([agg_index, agg_index]), n, n)
        return ops.aggregate(adj, adj.n_dst, inputs, ia_mode='perm', reduce=aggr or 'add')

    def aggregate_up(self, inputs: Tensor, agg_up_index: Tensor, up_ptr: Optional[Tensor] = None,
                     up_dim_size: Optional[int] = None) -> Tensor:
        return self._aggregate('up', inputs, agg_up_index, up_dim_size, self.aggr_up)

    def aggregate_down(self, inputs: Tensor, agg_down_index: Tensor,
                       down_ptr: Optional[Tensor] = None,
                       down_dim_size: Optional[int] = None) -> Tensor:
        return self._aggregate('down', inputs, agg_down_index, down_dim_size, self.aggr_down)

    def aggregate_boundary(self, inputs: Tensor, agg_boundary_index: Tensor,
                           boundary_ptr: Optional[Tensor] = None,
                           boundary_dim_size: Optional[int] = None) -> Tensor:
        return self._aggregate('boundary', inputs, agg_boundary_index, boundary_dim_size,
                               self.aggr_boundary)

    def message_and_aggregate_up(self, up_adj_t) -> Tensor:
        raise NotImplementedError

    def message_and_aggregate_down(self, down_adj_t) -> Tensor:
        raise NotImplementedError

    def message_and_aggregate_boundary(self, boundary_adj_t) -> Tensor:
        raise NotImplementedError

    def update(self, up_inputs: Optional[Tensor], down_inputs: Optional[Tensor],
               boundary_inputs: Optional[Tensor], x: Tensor):
        """mp/cell_mp.py:511-524 (zeros are created on the device directly)."""
        if up_inputs is None:
            up_inputs = ops.zeros_rows(x.size(0), self.up_msg_size, x.device)
        if down_inputs is None:
            down_inputs = ops.zeros_rows(x.size(0), self.down_msg_size, x.device)
        if boundary_inputs is None:
            boundary_inputs = ops.zeros_rows(x.size(0), self.boundary_msg_size, x.device)
        return up_inputs, down_inputs, boundary_inputs


class CochainMessagePassingParams:
    """mp/cell_mp.py:527-550: plain holder of the per-dimension propagate arguments."""

    def __init__(self, x: Tensor, up_index: Tensor = None, down_index: Tensor = None, **kwargs):
        self.x = x
        self.up_index = up_index
        self.down_index = down_index
        self.kwargs = kwargs
        self.boundary_index = kwargs.get('boundary_index', None)
        self.boundary_attr = kwargs.get('boundary_attr', None)
