"""Graph -> cell complex lifting through the C-ABI host routines (`cwn_lift_*`, csrc/cwn_lift.cpp):
the native counterpart of the reference's `compute_ring_2complex` / `compute_clique_complex_with_gudhi`
(data/utils.py:224-272, 400-498), which need graph-tool and gudhi.  Produces `cwn_amd.complex.Complex`
objects with exactly the tensors of the Python restatement in `cwn_amd/synthetic.py`
(tests/test_lifting.py compares the two and the reference's expected tensors).  Host CPU code: this
is input preparation, not the GPU hot path."""
import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi
from .complex import Cochain, Complex

RING, CLIQUE = 0, 1
(EDGES, CELLS2_PTR, CELLS2_VERTS, UP0, COB0, UP1, COB1, DOWN1, BND1, DOWN2, BND2, BINDEX1,
 BINDEX2) = range(13)


class Lift:
    """RAII wrapper of a cwn_lift_t."""

    def __init__(self, kind: int, n: int, edges: Sequence[Tuple[int, int]], max_k: int = 6,
                 include_down: bool = False):
        L = _ffi.lib()
        e = np.ascontiguousarray(np.asarray(list(edges), dtype=np.int64).reshape(-1, 2))
        self._L = L
        self._h = L.cwn_lift_create(kind, n, e.ctypes.data_as(C.c_void_p), e.shape[0], max_k,
                                    int(include_down))
        if not self._h:
            raise ValueError('invalid graph for lifting (vertex out of range or self loop)')

    def array(self, which: int) -> torch.Tensor:
        n = self._L.cwn_lift_size(self._h, which)
        out = torch.empty(max(n, 0), dtype=torch.long)
        _ffi.check(self._L.cwn_lift_copy(self._h, which, out.data_ptr()), 'cwn_lift_copy')
        return out

    def index(self, which: int) -> Optional[torch.Tensor]:
        a = self.array(which)
        return a.view(2, -1) if a.numel() else None

    def vector(self, which: int) -> Optional[torch.Tensor]:
        a = self.array(which)
        return a if a.numel() else None

    def cells2(self):
        ptr, verts = self.array(CELLS2_PTR).tolist(), self.array(CELLS2_VERTS).tolist()
        return [tuple(verts[ptr[i]:ptr[i + 1]]) for i in range(len(ptr) - 1)]

    def __del__(self):
        if getattr(self, '_h', None):
            self._L.cwn_lift_destroy(self._h)
            self._h = None


def _complex(L: Lift, n: int, vx, ex, cx2, include_down_adj: bool, y) -> Complex:
    edges = L.array(EDGES).view(-1, 2)
    E, n2 = edges.size(0), L.array(CELLS2_PTR).numel() - 1
    dim = 2 if n2 else (1 if E else 0)
    cochains = [Cochain(dim=0, x=vx, upper_index=L.index(UP0), shared_coboundaries=L.vector(COB0),
                        num_cells_up=E if dim >= 1 else 0, num_cells=n)]
    if dim >= 1:
        cochains.append(Cochain(
            dim=1, x=ex, upper_index=L.index(UP1), shared_coboundaries=L.vector(COB1),
            lower_index=L.index(DOWN1) if include_down_adj else None,
            shared_boundaries=L.vector(BND1) if include_down_adj else None,
            boundary_index=L.index(BINDEX1), num_cells=E, num_cells_down=n,
            num_cells_up=n2 if dim >= 2 else 0))
    if dim >= 2:
        cochains.append(Cochain(
            dim=2, x=cx2, lower_index=L.index(DOWN2) if include_down_adj else None,
            shared_boundaries=L.vector(BND2) if include_down_adj else None,
            boundary_index=L.index(BINDEX2), num_cells=n2, num_cells_down=E, num_cells_up=0))
    return Complex(*cochains, y=y, dimension=dim)


def ring_lift(n: int, bonds: Sequence[Tuple[int, int]], vx: torch.Tensor,
              ex: Optional[torch.Tensor] = None, max_k: int = 6, include_down_adj: bool = False,
              y: Optional[torch.Tensor] = None, rx: Optional[torch.Tensor] = None) -> Complex:
    """data/utils.py:400-498.  `ex` rows follow the sorted (u < v) edge order."""
    return _complex(Lift(RING, n, bonds, max_k, include_down_adj), n, vx, ex, rx, include_down_adj, y)


def clique_lift(n: int, edges: Sequence[Tuple[int, int]], vx: torch.Tensor, max_dim: int = 2,
                init_method: str = 'sum', y: Optional[torch.Tensor] = None,
                include_down_adj: bool = False) -> Complex:
    """data/utils.py:224-272; higher-cell features = reduce of the vertices' features (construct_features, :141-156).
    Up to dimension 2 (every configuration of BASELINE.json) the native routine lifts; `max_dim > 2` takes
    `clique_lift_general` below."""
    if max_dim > 2:
        return clique_lift_general(n, edges, vx, max_dim, init_method, y, include_down_adj)
    L = Lift(CLIQUE, n, edges, 0, include_down_adj) if max_dim >= 2 else Lift(RING, n, edges, 0, include_down_adj)
    red = (lambda f: f.sum(1)) if init_method in ('sum', 'add') else (lambda f: f.mean(1))
    ev = L.array(EDGES).view(-1, 2)
    ex = red(vx[ev]) if ev.numel() else None
    tris = L.cells2()
    tx = red(vx[torch.tensor(tris, dtype=torch.long)]) if tris else None
    return _complex(L, n, vx, ex, tx, include_down_adj, y)


def clique_lift_general(n: int, edges: Sequence[Tuple[int, int]], vx: torch.Tensor, max_dim: int,
                        init_method: str = 'sum', y: Optional[torch.Tensor] = None,
                        include_down_adj: bool = False) -> Complex:
    """compute_clique_complex_with_gudhi (data/utils.py:224-272) for ANY expansion dimension, without gudhi: the k-cliques of
    the graph as the k-1-cells, dimension by dimension (a clique of k + 1 vertices = a clique of k plus a common neighbour
    larger than its last vertex).  Conventions, the ones the native routine follows up to dimension 2 (pinned there on the
    reference's expected tensors, tests/test_lifting.py; equal to this function at max_dim = 2, same file):
      * cells of a dimension are numbered in lexicographic order of their sorted vertex tuples (build_tables, :44-64: the
        simplex tree is walked depth first);
      * the boundaries of a cell in the order itertools.combinations drops a vertex (:39-41) = ascending face number;
      * upper adjacency of dimension k - 1: per k-cell in order, every pair of its faces in both directions, the k-cell as
        the shared coboundary (build_adj, :121-126); lower adjacency of dimension k + 1: per k-cell in order, every pair of
        its cofaces (ascending) in both directions, the k-cell as the shared boundary (:128-134);
      * features of a cell = sum / mean of its vertices' features (construct_features, :141-156); the complex's dimension
        is the largest one that has a cell (:247).
    Host Python (numpy): input preparation outside every configuration of BASELINE.json -- the reference's own is Python."""
    import itertools
    if n < 0:
        raise ValueError('invalid graph for lifting')
    es = set()
    for u, v in edges:
        u, v = int(u), int(v)
        if u < 0 or v < 0 or u >= n or v >= n or u == v:
            raise ValueError('invalid graph for lifting (vertex out of range or self loop)')
        es.add((min(u, v), max(u, v)))
    nbr = [set() for _ in range(n)]
    for u, v in es:
        nbr[u].add(v)
        nbr[v].add(u)
    cells = [[(v,) for v in range(n)], sorted(es)]
    while len(cells) <= max_dim and cells[-1]:
        nxt = []
        for c in cells[-1]:
            common = set.intersection(*(nbr[v] for v in c))
            nxt.extend(c + (w,) for w in sorted(common) if w > c[-1])
        cells.append(nxt)                      # (generated from a sorted list by appending ascending vertices: sorted)
    while len(cells) > 1 and not cells[-1]:
        cells.pop()
    dim = len(cells) - 1
    ids = [{c: i for i, c in enumerate(level)} for level in cells]
    faces = [None] + [[[ids[k - 1][f] for f in itertools.combinations(c, k)] for c in cells[k]] for k in range(1, dim + 1)]
    red = (lambda f: f.sum(1)) if init_method in ('sum', 'add') else (lambda f: f.mean(1))

    def pairs(groups):
        """[(shared, members)] -> index [2, L], shared [L]: every pair of members in both directions"""
        a, b, sh = [], [], []
        for g, members in groups:
            for i, j in itertools.combinations(members, 2):
                a += [i, j]
                b += [j, i]
                sh += [g, g]
        if not a:
            return None, None
        return torch.tensor([a, b], dtype=torch.long), torch.tensor(sh, dtype=torch.long)

    cochains = []
    for k in range(dim + 1):
        up_index = cob = low_index = bnd = b_index = None
        if k < dim:
            up_index, cob = pairs(enumerate(faces[k + 1]))
        if k > 0:
            b_index = torch.tensor([[f for fs in faces[k] for f in fs], [c for c, fs in enumerate(faces[k]) for _ in fs]],
                                   dtype=torch.long)
            if include_down_adj:
                cof = [[] for _ in cells[k - 1]]
                for c, fs in enumerate(faces[k]):
                    for f in fs:
                        cof[f].append(c)
                low_index, bnd = pairs(enumerate(cof))
        x = vx if k == 0 else red(vx[torch.tensor(cells[k], dtype=torch.long)])
        cochains.append(Cochain(dim=k, x=x, upper_index=up_index, shared_coboundaries=cob, lower_index=low_index,
                                shared_boundaries=bnd, boundary_index=b_index, num_cells=len(cells[k]),
                                num_cells_down=len(cells[k - 1]) if k > 0 else None,
                                num_cells_up=len(cells[k + 1]) if k < dim else 0))
    return Complex(*cochains, y=y, dimension=dim)


def induced_cycles(n: int, bonds: Sequence[Tuple[int, int]], max_k: int):
    """Chordless cycles with 3..max_k vertices, each in cyclic order from its smallest vertex."""
    return Lift(RING, n, bonds, max_k, False).cells2()


# ---- a whole dataset at once ---------------------------------------------------------------------------
class LiftSet:
    """RAII wrapper of a cwn_lift_set_t: the graphs of a dataset lifted by host threads, arrays concatenated."""

    def __init__(self, kind: int, n_vertices: np.ndarray, edge_ptr: np.ndarray, edges: np.ndarray, max_k: int = 6,
                 include_down: bool = False, n_threads: int = 0):
        L = _ffi.lib()
        self._L = L
        self.n_vertices = np.ascontiguousarray(n_vertices, dtype=np.int64)
        ptr = np.ascontiguousarray(edge_ptr, dtype=np.int64)
        e = np.ascontiguousarray(edges, dtype=np.int64).reshape(-1, 2)
        if ptr.size != self.n_vertices.size + 1 or (ptr.size and int(ptr[-1]) != e.shape[0]):
            raise ValueError('edge_ptr must have n_graphs + 1 entries and end at the number of edges')
        self.num = int(self.n_vertices.size)
        self._h = L.cwn_lift_many(kind, self.num, self.n_vertices.ctypes.data_as(C.c_void_p),
                                  ptr.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), max_k,
                                  int(include_down), int(n_threads))
        if not self._h:
            raise ValueError('invalid graph in the dataset (vertex out of range or self loop)')

    def lengths(self, which: int) -> np.ndarray:
        out = np.zeros(self.num, dtype=np.int64)
        _ffi.check(self._L.cwn_lift_many_lengths(self._h, which, out.ctypes.data_as(C.c_void_p)), 'cwn_lift_many_lengths')
        return out

    def array(self, which: int, two_rows: bool = False) -> torch.Tensor:
        n = int(self.lengths(which).sum()) * (2 if two_rows else 1)
        out = torch.empty(n, dtype=torch.long)
        _ffi.check(self._L.cwn_lift_many_copy(self._h, which, out.data_ptr()), 'cwn_lift_many_copy')
        return out.view(2, -1) if two_rows else out

    def __del__(self):
        if getattr(self, '_h', None):
            self._L.cwn_lift_many_destroy(self._h)
            self._h = None


def _field(g, name):
    return g[name] if isinstance(g, dict) else getattr(g, name, None)


def _reduce_rows(vx: torch.Tensor, members: torch.Tensor, seg: torch.Tensor, n_seg: int, init_method: str) -> torch.Tensor:
    """construct_features (data/utils.py:141-155): scatter of the member vertices' features onto their cells."""
    src = vx.index_select(0, members)
    out = torch.zeros((n_seg,) + tuple(vx.shape[1:]), dtype=vx.dtype).index_add_(0, seg, src)
    if init_method in ('sum', 'add'):
        return out
    if init_method == 'mean':
        cnt = torch.zeros(n_seg, dtype=torch.long).index_add_(0, seg, torch.ones_like(seg)).clamp_(min=1)
        shape = (n_seg,) + (1,) * (vx.dim() - 1)
        return out / cnt.view(shape) if vx.is_floating_point() else torch.div(out, cnt.view(shape), rounding_mode='floor')
    raise NotImplementedError(f"init_method {init_method!r} (sum / mean)")


def _pack_dataset(kind: int, graphs, max_k: int, include_down_adj: bool, init_method: str, init_edges: bool,
                  init_cells2: bool, n_threads: int, device, with_csr: bool = False):
    """Per-graph objects -> the collated form (one pass over the graphs, the only per-graph Python of the path)."""
    G = len(graphs)
    if G == 0:
        raise ValueError('empty dataset')
    xs = [torch.as_tensor(_field(g, 'x')) for g in graphs]
    n = np.array([int(_field(g, 'num_nodes')) if _field(g, 'num_nodes') is not None else int(x.size(0))
                  for g, x in zip(graphs, xs)], dtype=np.int64)
    eis = [torch.as_tensor(_field(g, 'edge_index')).view(2, -1) for g in graphs]
    m = np.array([e.size(1) for e in eis], dtype=np.int64)
    attrs = [_field(g, 'edge_attr') for g in graphs]
    with_attr = [a is not None for a, k in zip(attrs, m) if k]
    if any(with_attr) and not all(with_attr):
        raise ValueError('edge_attr must be given for every graph with edges, or for none')
    ea = torch.cat([torch.as_tensor(a).view(k, -1) for a, k in zip(attrs, m) if k], dim=0) if any(with_attr) else None
    ys = [None if _field(g, 'y') is None else torch.as_tensor(_field(g, 'y')) for g in graphs]
    # extract_labels (data/utils.py:158-174): first dimension 1 = a label of the graph, else one per vertex
    y_graph = y_vertex = None
    if all(y is not None and y.size(0) == 1 for y in ys):
        y_graph = torch.cat(ys, dim=0)
    elif any(y is not None and y.size(0) != 1 for y in ys):
        for y, k in zip(ys, n):
            if y is None or y.size(0) != k:
                raise ValueError('y must label the graph (first dimension 1) or every vertex, for every graph alike')
        y_vertex = torch.cat(ys, dim=0)
    return _pack_collated(kind, n, m, torch.cat(eis, dim=1), torch.cat(xs, dim=0), ea, y_graph, y_vertex, max_k,
                          include_down_adj, init_method, init_edges, init_cells2, n_threads, device, with_csr=with_csr)


def _pack_collated(kind: int, n: np.ndarray, m: np.ndarray, edge_index: torch.Tensor, vx: torch.Tensor,
                   ea: Optional[torch.Tensor], y_graph: Optional[torch.Tensor], y_vertex: Optional[torch.Tensor],
                   max_k: int, include_down_adj: bool, init_method: str, init_edges: bool, init_cells2: bool,
                   n_threads: int, device, with_csr: bool = False):
    """The dataset in collated form -- vertices per graph `n`, directed edge entries per graph `m`, edge_index
    [2, sum m] with vertex ids LOCAL to each graph (how a PyG InMemoryDataset stores itself), vx [sum n, F],
    ea [sum m, Fe] or None -- lifted by host threads and packed.  No per-graph Python."""
    from .packed import PackedComplexes
    n, m = np.asarray(n, dtype=np.int64), np.asarray(m, dtype=np.int64)
    G = int(n.size)
    if G == 0:
        raise ValueError('empty dataset')
    if int(n.sum()) != vx.size(0) or int(m.sum()) != edge_index.size(1):
        raise ValueError('x / edge_index do not match the per-graph counts')
    eptr = np.concatenate([[0], np.cumsum(m)]).astype(np.int64)
    e_all = np.ascontiguousarray(edge_index.numpy().T, dtype=np.int64)          # [M, 2]
    S = LiftSet(kind, n, eptr, e_all, max_k, include_down_adj, n_threads)
    v_off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    E = S.lengths(EDGES) // 2
    n2 = S.lengths(CELLS2_PTR) - 1
    dims = np.where(n2 > 0, 2, np.where(E > 0, 1, 0)).astype(np.int64)
    zeros = np.zeros(G, dtype=np.int64)

    # ---- features ---------------------------------------------------------------------------------------
    edges_local = S.array(EDGES).view(-1, 2)                                  # sorted (u < v) per graph
    g_of_edge = np.repeat(np.arange(G), E)
    edges_global = edges_local + torch.from_numpy(v_off[g_of_edge]).view(-1, 1)
    ex = None
    if init_edges and int(E.sum()):
        if ea is None:
            ex = _reduce_rows(vx, edges_global.reshape(-1), torch.arange(edges_global.size(0)).repeat_interleave(2),
                              edges_global.size(0), init_method)
        else:
            if ea.dim() == 1:
                ea = ea.view(-1, 1)
            if ea.size(0) != e_all.shape[0]:
                raise ValueError('edge_attr needs one row per edge_index entry')
            # undirected edge id of every directed entry: the lifted edges are sorted by (graph, u, v)
            g_of_entry = np.repeat(np.arange(G), m)
            lo, hi = np.minimum(e_all[:, 0], e_all[:, 1]), np.maximum(e_all[:, 0], e_all[:, 1])
            span = int(n.max()) + 1
            key_entry = (g_of_entry * span + lo) * span + hi
            el = edges_local.numpy()
            key_edge = (g_of_edge * span + el[:, 0]) * span + el[:, 1]
            eid = torch.from_numpy(np.searchsorted(key_edge, key_entry))
            ex = torch.zeros((key_edge.size,) + tuple(ea.shape[1:]), dtype=ea.dtype)
            ex[eid] = ea
            if not torch.equal(ex[eid], ea):          # data/utils.py:468: the two directions must agree
                raise ValueError('edge_attr differs between the two directions of an edge')
    cx2 = None
    if init_cells2 and int(n2.sum()):
        verts2, len2 = S.array(CELLS2_VERTS), S.lengths(CELLS2_VERTS)
        ptr_all = S.array(CELLS2_PTR).numpy()                                   # per graph: n2 + 1 local offsets
        # cell sizes: differences inside each graph's own run of offsets
        run_start = np.concatenate([[0], np.cumsum(n2 + 1)])[:-1]
        keep = np.ones(ptr_all.size, dtype=bool)
        keep[run_start] = False
        sizes = (ptr_all[1:] - ptr_all[:-1])[keep[1:]]
        seg = torch.from_numpy(np.repeat(np.arange(int(n2.sum())), sizes))
        members = verts2 + torch.from_numpy(np.repeat(v_off[:-1], len2))
        cx2 = _reduce_rows(vx, members, seg, int(n2.sum()), init_method)

    # ---- per-key arrays in the packed layout -----------------------------------------------------------------
    def idx2(which):
        L = S.lengths(which)
        return S.array(which, two_rows=True), L, L > 0

    def idx1(which):
        L = S.lengths(which)
        return S.array(which), L, L > 0

    def labels(y, count):
        """what PackedComplexes._pack makes of per-complex labels: concatenated along the LAST axis"""
        per = int(y[0].numel()) if y.dim() > 1 else 1
        data = y if y.dim() == 1 else y.reshape(1, -1)
        return data, count * per, np.ones(G, dtype=bool)

    w = lambda t: int(t.size(1)) if t.dim() == 2 else 1
    keys = [dict(), dict(), dict()]
    keys[0]['x'] = (vx, n * w(vx), np.ones(G, dtype=bool))
    keys[0]['upper_index'], keys[0]['shared_coboundaries'] = idx2(UP0), idx1(COB0)
    if y_vertex is not None:
        if y_vertex.dim() != 1 or y_vertex.size(0) != vx.size(0):
            raise ValueError('vertex labels: one scalar per vertex')
        keys[0]['y'] = labels(y_vertex, n)
    if ex is not None:
        keys[1]['x'] = (ex, E * w(ex), E > 0)
    keys[1]['upper_index'], keys[1]['shared_coboundaries'] = idx2(UP1), idx1(COB1)
    keys[1]['boundary_index'] = idx2(BINDEX1)
    if cx2 is not None:
        keys[2]['x'] = (cx2, n2 * w(cx2), n2 > 0)
    keys[2]['boundary_index'] = idx2(BINDEX2)
    if include_down_adj:
        keys[1]['lower_index'], keys[1]['shared_boundaries'] = idx2(DOWN1), idx1(BND1)
        keys[2]['lower_index'], keys[2]['shared_boundaries'] = idx2(DOWN2), idx1(BND2)
    # cell counts as the per-complex constructor reads them off the Cochain objects (num_cells_up / _down)
    n_cells = np.stack([n, E, n2])
    has_cells = np.stack([np.ones(G, dtype=bool), dims >= 1, dims >= 2])
    n_up = np.stack([np.where(dims >= 1, E, 0), np.where(dims >= 2, n2, 0), zeros])
    n_down = np.stack([zeros, n, np.where(dims >= 1, E, 0)])
    gy = None
    if y_graph is not None:
        if y_graph.size(0) != G:
            raise ValueError('graph labels: one row per graph')
        gy = labels(y_graph, np.ones(G, dtype=np.int64))
    packed = PackedComplexes.from_arrays(device, 2, dims, n_cells, has_cells, n_up, n_down, keys, gy, with_csr=with_csr)
    dimension = int(dims.max())
    feats = [w(vx), None if ex is None else w(ex), None if cx2 is None else w(cx2)][:dimension + 1]
    return packed, dimension, feats


def pack_collated_dataset_with_rings(x: torch.Tensor, edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor],
                                     y: Optional[torch.Tensor], slices: dict, max_ring_size: int = 7,
                                     include_down_adj: bool = False, init_method: str = 'sum', init_edges: bool = True,
                                     init_rings: bool = False, n_threads: int = 0, device='cuda'):
    """pack_graph_dataset_with_rings for a dataset that is ALREADY collated the way a PyG InMemoryDataset stores
    itself (`dataset.data` + `dataset.slices`): x [sum n, F], edge_index [2, sum m] with vertex ids local to each
    graph, edge_attr [sum m, Fe] or None, y with `slices['y']` (one row per graph, or one per vertex), and the
    offsets slices['x'], slices['edge_index'] (G + 1 entries each).  No Python per graph: 250 k ZINC molecules are a
    few seconds of host time."""
    px, pe = np.asarray(slices['x'], dtype=np.int64), np.asarray(slices['edge_index'], dtype=np.int64)
    if px.size != pe.size or px.size < 2:
        raise ValueError("slices['x'] and slices['edge_index'] need G + 1 entries each")
    n, m = np.diff(px), np.diff(pe)
    y_graph = y_vertex = None
    if y is not None:
        py = np.asarray(slices['y'], dtype=np.int64) if 'y' in slices else np.arange(px.size)
        per = np.diff(py)
        if (per == 1).all():
            y_graph = y
        elif np.array_equal(per, n):
            y_vertex = y
        else:
            raise ValueError('y must hold one row per graph or one per vertex')
    return _pack_collated(RING, n, m, edge_index, x, edge_attr, y_graph, y_vertex, max_ring_size, include_down_adj,
                          init_method, init_edges, init_rings, n_threads, device)


def pack_graph_dataset_with_rings(graphs, max_ring_size: int = 7, include_down_adj: bool = False,
                                  init_method: str = 'sum', init_edges: bool = True, init_rings: bool = False,
                                  n_threads: int = 0, device='cuda', with_csr: bool = False):
    """The dataset-level counterpart of convert_graph_dataset_with_rings (data/utils.py:501-544; same arguments,
    n_jobs -> n_threads): `graphs` are PyG-Data-like objects or dicts with x [n, F], edge_index [2, M] (both
    directions, as PyG stores them, or one), edge_attr [M, Fe] or None, y, num_nodes.  Returns
    (PackedComplexes on `device`, dimension, num_features): the lifted dataset resident in HBM, ready for
    `collate(indices)`, instead of a Python list of Complex objects -- the graphs are lifted by host threads
    (cwn_lift_many) and every tensor of the dataset is built once, concatenated.  Complex c of the result equals
    `ring_lift` of graph c (tests/test_lifting.py).  `with_csr`: also the per-complex CSRs of the boundary adjacencies a
    static batch collates (cwn_amd/static_batch.py)."""
    return _pack_dataset(RING, graphs, max_ring_size, include_down_adj, init_method, init_edges, init_rings,
                         n_threads, device, with_csr=with_csr)


def pack_graph_dataset_with_cliques(graphs, expansion_dim: int = 2, include_down_adj: bool = True,
                                    init_method: str = 'sum', n_threads: int = 0, device='cuda', with_csr: bool = False):
    """convert_graph_dataset_with_gudhi (data/utils.py:275-297) for expansion_dim <= 2, packed like
    pack_graph_dataset_with_rings; higher cells take the reduce of their vertices' features (:141-155).  (Beyond dimension 2:
    `clique_lift(..., max_dim=k)` per graph gives the list of Complex objects the reference's function returns; the packed
    form and the kernels behind it hold the three dimensions every configuration of BASELINE.json uses.)"""
    if expansion_dim > 2:
        raise NotImplementedError('the packed dataset holds dimensions 0 .. 2; lift per graph with clique_lift(max_dim=k) '
                                  'for a list of Complex objects of higher dimension')
    kind = CLIQUE if expansion_dim >= 2 else RING
    return _pack_dataset(kind, graphs, 0, include_down_adj, init_method, True, True, n_threads, device, with_csr=with_csr)


def convert_graph_dataset_with_cliques(graphs, expansion_dim: int, include_down_adj: bool = True, init_method: str = 'sum'):
    """convert_graph_dataset_with_gudhi (data/utils.py:275-297) as the reference returns it -- (list of Complex, dimension,
    num_features per dimension) -- for ANY expansion dimension: `clique_lift` per graph (native up to dimension 2,
    `clique_lift_general` beyond).  `graphs`: PyG-Data-like objects or dicts with x, edge_index, y, num_nodes.  A vertex-level
    label (one row per vertex) goes to the vertices' cochain, a graph-level one to the complex (extract_labels, :159-175)."""
    dimension, complexes = -1, []
    num_features = [None] * (expansion_dim + 1)
    for g in graphs:
        x = torch.as_tensor(_field(g, 'x'))
        n = _field(g, 'num_nodes')
        n = int(n) if n is not None else int(x.size(0))
        ei = torch.as_tensor(_field(g, 'edge_index')).view(2, -1)
        y = _field(g, 'y')
        v_y = complex_y = None
        if y is not None:
            y = torch.as_tensor(y)
            if y.size(0) == 1:
                complex_y = y
            else:
                if y.size(0) != n:
                    raise ValueError('y is one row (the graph) or one row per vertex')
                v_y = y
        cx = clique_lift(n, ei.t().tolist(), x, max_dim=expansion_dim, init_method=init_method, y=complex_y,
                         include_down_adj=include_down_adj)
        if v_y is not None:
            cx.cochains[0].y = v_y
        dimension = max(dimension, cx.dimension)
        for d in range(cx.dimension + 1):
            f = cx.cochains[d].num_features
            if num_features[d] is None:
                num_features[d] = f
            elif num_features[d] != f:
                raise ValueError(f'graphs disagree on the number of features in dimension {d}')
        complexes.append(cx)
    return complexes, dimension, num_features[:dimension + 1]
