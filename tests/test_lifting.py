"""The graph -> complex lifts restated in cwn_amd/synthetic.py (inputs of the hot path; SURVEY.md
§8f rank 3) against the expected tensors the reference's own tests hold for the house graph
(data/test_utils.py) and against an independent brute-force ring enumerator
(the approach of data/helper_test.py:68-99).  CPU only."""
import itertools

import networkx as nx
import numpy as np
import torch

from cwn_amd.complex import Complex
from cwn_amd.synthetic import (clique_lift, induced_cycles, random_molecule, ring_lift,
                               zinc_like_complexes, reddit_like_complexes)

HOUSE = [(0, 1), (0, 3), (1, 2), (2, 3), (2, 4), (3, 4)]   # data/test_utils.py:32-35
X = torch.arange(0, 5, dtype=torch.float).view(5, 1)


def dense(a):
    return a.src.index_select(0, a.index)


def test_clique_lift_house_expected_tensors():
    """data/test_utils.py:40-124."""
    Complex.lazy_attrs = True
    cx = clique_lift(5, HOUSE, X, include_down_adj=True, y=torch.tensor([1]))
    assert cx.nodes.num_cells_down is None and cx.nodes.num_cells_up == 6
    assert cx.edges.num_cells_down == 5 and cx.edges.num_cells_up == 1
    assert cx.two_cells.num_cells_down == 6 and cx.two_cells.num_cells_up == 0
    v = cx.get_cochain_params(dim=0)
    assert v.up_index.tolist() == [[0, 1, 0, 3, 1, 2, 2, 3, 2, 4, 3, 4], [1, 0, 3, 0, 2, 1, 3, 2, 4, 2, 4, 3]]
    assert dense(v.kwargs['up_attr']).flatten().tolist() == [1, 1, 3, 3, 3, 3, 5, 5, 6, 6, 7, 7]
    assert v.down_index is None and v.kwargs['boundary_attr'] is None
    e = cx.get_cochain_params(dim=1)
    assert e.x.flatten().tolist() == [1, 3, 3, 5, 6, 7]
    assert e.up_index.tolist() == [[3, 4, 3, 5, 4, 5], [4, 3, 5, 3, 5, 4]]
    assert dense(e.kwargs['up_attr']).flatten().tolist() == [9] * 6
    assert e.down_index.tolist() == [[0, 1, 0, 2, 2, 3, 2, 4, 3, 4, 1, 3, 1, 5, 3, 5, 4, 5],
                                     [1, 0, 2, 0, 3, 2, 4, 2, 4, 3, 3, 1, 5, 1, 5, 3, 5, 4]]
    assert dense(e.kwargs['down_attr']).flatten().tolist() == [0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4]
    assert torch.equal(e.kwargs['boundary_attr'], X)
    assert e.kwargs['boundary_index'].tolist() == [[0, 1, 0, 3, 1, 2, 2, 3, 2, 4, 3, 4],
                                                   [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5]]
    t = cx.get_cochain_params(dim=2)
    assert t.x.flatten().tolist() == [9] and t.down_index is None and t.up_index is None
    assert t.kwargs['boundary_index'].tolist() == [[3, 4, 5], [0, 0, 0]]


def test_ring_lift_house_expected_tensors():
    """data/test_utils.py:215-289 (max_k = 4: the square and the triangle are 2-cells)."""
    ex = torch.tensor([[1.], [3.], [3.], [5.], [6.], [7.]])
    rx = torch.tensor([[6.], [9.]])
    cx = ring_lift(5, HOUSE, X, ex=ex, rx=rx, max_k=4, include_down_adj=True, y=torch.tensor([1]))
    assert cx.nodes.num_cells_up == 6 and cx.edges.num_cells_up == 2
    assert cx.cochains[2].num_cells == 2 and cx.cochains[2].num_cells_down == 6
    v = cx.get_cochain_params(dim=0)
    assert v.up_index.tolist() == [[0, 1, 0, 3, 1, 2, 2, 3, 2, 4, 3, 4], [1, 0, 3, 0, 2, 1, 3, 2, 4, 2, 4, 3]]
    e = cx.get_cochain_params(dim=1)
    assert e.up_index.tolist() == [[0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3, 3, 4, 3, 5, 4, 5],
                                   [1, 0, 2, 0, 3, 0, 2, 1, 3, 1, 3, 2, 4, 3, 5, 3, 5, 4]]
    assert dense(e.kwargs['up_attr']).flatten().tolist() == [6] * 12 + [9] * 6
    assert e.down_index.tolist() == [[0, 1, 0, 2, 2, 3, 2, 4, 3, 4, 1, 3, 1, 5, 3, 5, 4, 5],
                                     [1, 0, 2, 0, 3, 2, 4, 2, 4, 3, 3, 1, 5, 1, 5, 3, 5, 4]]
    t = cx.get_cochain_params(dim=2)
    assert t.down_index.tolist() == [[0, 1], [1, 0]]
    assert dense(t.kwargs['down_attr']).flatten().tolist() == [5, 5]
    assert t.up_index is None
    assert t.kwargs['boundary_index'].tolist() == [[0, 1, 2, 3, 3, 4, 5], [0, 0, 0, 0, 1, 1, 1]]
    # larger / smaller k (data/test_utils.py:409-552): k=3 keeps only the triangle
    assert induced_cycles(5, HOUSE, 3) == [(2, 3, 4)]
    assert induced_cycles(5, HOUSE, 4) == [(0, 1, 2, 3), (2, 3, 4)]
    assert induced_cycles(5, HOUSE, 7) == [(0, 1, 2, 3), (2, 3, 4)]


def _brute_force_rings(n, bonds, max_k):
    """Independent enumerator in the style of data/helper_test.py:68-99."""
    g = nx.Graph()
    g.add_nodes_from(range(n))
    g.add_edges_from(bonds)
    rings = set()
    for cyc in nx.simple_cycles(g.to_directed()):
        if len(cyc) <= 2 or len(cyc) > max_k:
            continue
        chordless = True
        for (i1, v1), (i2, v2) in itertools.combinations(enumerate(cyc), 2):
            adjacent_in_cycle = (i2 == i1 + 1) or (i1 == 0 and i2 == len(cyc) - 1)
            if not adjacent_in_cycle and g.has_edge(v1, v2):
                chordless = False
                break
        if chordless:
            rings.add(tuple(sorted(cyc)))
    return rings


def test_ring_finder_matches_brute_force_on_random_molecules():
    rng = np.random.default_rng(7)
    for _ in range(25):
        n, bonds = random_molecule(rng, 8, 16)
        for k in (5, 6):
            got = {tuple(sorted(r)) for r in induced_cycles(n, bonds, k)}
            assert got == _brute_force_rings(n, bonds, k)


def test_generators_are_deterministic_and_consistent():
    a, b = zinc_like_complexes(4, seed=5), zinc_like_complexes(4, seed=5)
    for ca, cb in zip(a, b):
        for d in range(ca.dimension + 1):
            for k in ('upper_index', 'boundary_index', 'shared_coboundaries'):
                ta, tb = ca.cochains[d][k], cb.cochains[d][k]
                assert (ta is None) == (tb is None) and (ta is None or torch.equal(ta, tb))
    for cx in a + reddit_like_complexes(2, seed=2, n_lo=40, n_hi=60):
        for d in range(cx.dimension + 1):
            c = cx.cochains[d]
            if c.upper_index is not None:      # symmetric pair lists with a shared coface each
                assert c.upper_index.size(1) == c.shared_coboundaries.numel()
                assert int(c.upper_index.max()) < c.num_cells
                assert int(c.shared_coboundaries.max()) < cx.cochains[d + 1].num_cells
            if c.boundary_index is not None:
                assert int(c.boundary_index[0].max()) < cx.cochains[d - 1].num_cells
                assert torch.all(c.boundary_index[1][1:] >= c.boundary_index[1][:-1])


# ------------------------------------------------------------------------------------------------
# the native (C ABI, csrc/cwn_lift.cpp) lifts: same tensors as the restatement above, which is
# pinned on the reference's expected tensors
# ------------------------------------------------------------------------------------------------
KEYS = ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries', 'boundary_index')


def _same_complex(a: Complex, b: Complex):
    assert a.dimension == b.dimension
    for d in range(a.dimension + 1):
        ca, cb = a.cochains[d], b.cochains[d]
        assert (ca.num_cells, ca.num_cells_up, ca.num_cells_down) == (cb.num_cells, cb.num_cells_up, cb.num_cells_down)
        for k in KEYS:
            ta, tb = ca[k], cb[k]
            assert (ta is None) == (tb is None), (d, k)
            if ta is not None:
                assert ta.dtype == tb.dtype and torch.equal(ta, tb), (d, k)


def test_native_lifts_house_expected_tensors():
    """data/test_utils.py:40-124, 215-289 through the C ABI."""
    from cwn_amd import lifting
    Complex.lazy_attrs = True
    cx = lifting.clique_lift(5, HOUSE, X, include_down_adj=True, y=torch.tensor([1]))
    e = cx.get_cochain_params(dim=1)
    assert e.x.flatten().tolist() == [1, 3, 3, 5, 6, 7]
    assert e.up_index.tolist() == [[3, 4, 3, 5, 4, 5], [4, 3, 5, 3, 5, 4]]
    assert e.down_index.tolist() == [[0, 1, 0, 2, 2, 3, 2, 4, 3, 4, 1, 3, 1, 5, 3, 5, 4, 5],
                                     [1, 0, 2, 0, 3, 2, 4, 2, 4, 3, 3, 1, 5, 1, 5, 3, 5, 4]]
    assert cx.get_cochain_params(dim=2).kwargs['boundary_index'].tolist() == [[3, 4, 5], [0, 0, 0]]
    ex = torch.tensor([[1.], [3.], [3.], [5.], [6.], [7.]])
    rx = torch.tensor([[6.], [9.]])
    cr = lifting.ring_lift(5, HOUSE, X, ex=ex, rx=rx, max_k=4, include_down_adj=True, y=torch.tensor([1]))
    e = cr.get_cochain_params(dim=1)
    assert e.up_index.tolist() == [[0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3, 3, 4, 3, 5, 4, 5],
                                   [1, 0, 2, 0, 3, 0, 2, 1, 3, 1, 3, 2, 4, 3, 5, 3, 5, 4]]
    t = cr.get_cochain_params(dim=2)
    assert t.down_index.tolist() == [[0, 1], [1, 0]]
    assert t.kwargs['boundary_index'].tolist() == [[0, 1, 2, 3, 3, 4, 5], [0, 0, 0, 0, 1, 1, 1]]
    assert lifting.induced_cycles(5, HOUSE, 3) == [(2, 3, 4)]
    assert lifting.induced_cycles(5, HOUSE, 4) == [(0, 1, 2, 3), (2, 3, 4)]
    assert lifting.induced_cycles(5, HOUSE, 7) == [(0, 1, 2, 3), (2, 3, 4)]


def test_native_ring_lift_equals_restatement_on_random_molecules():
    from cwn_amd import lifting
    rng = np.random.default_rng(11)
    for i in range(40):
        n, bonds = random_molecule(rng, 6, 40)
        vx = torch.from_numpy(rng.integers(0, 28, size=(n, 1))).float()
        ex = torch.from_numpy(rng.integers(0, 4, size=(len(bonds), 1))).float()
        for k in (3, 5, 6, 8):
            assert lifting.induced_cycles(n, bonds, k) == induced_cycles(n, bonds, k)
        down = bool(i % 2)
        # shuffled, flipped edge list: the lift sorts it (features follow the sorted order)
        sh = [(v, u) if j % 3 == 0 else (u, v) for j, (u, v) in enumerate(bonds)]
        order = rng.permutation(len(sh))
        sh = [sh[j] for j in order]
        _same_complex(lifting.ring_lift(n, sh, vx, ex, max_k=6, include_down_adj=down),
                      ring_lift(n, bonds, vx, ex, max_k=6, include_down_adj=down))


def test_native_clique_lift_equals_restatement_on_hub_graphs():
    from cwn_amd import lifting
    from cwn_amd.synthetic import preferential_attachment_graph
    rng = np.random.default_rng(3)
    for i in range(6):
        n = int(rng.integers(30, 120))
        edges = preferential_attachment_graph(rng, n)
        if isinstance(edges, tuple):
            edges = edges[-1]
        vx = torch.ones(n, 1)
        for init in ('sum', 'mean'):
            _same_complex(lifting.clique_lift(n, edges, vx, init_method=init, include_down_adj=bool(i % 2)),
                          clique_lift(n, edges, vx, init_method=init, include_down_adj=bool(i % 2)))


def test_native_lift_rejects_bad_graphs():
    from cwn_amd import lifting
    for bad in ([(0, 5)], [(1, 1)], [(-1, 0)]):
        try:
            lifting.ring_lift(3, bad, torch.zeros(3, 1))
        except ValueError:
            continue
        raise AssertionError(f'{bad} accepted')
    cx = lifting.ring_lift(4, [], torch.zeros(4, 1))          # no edges: a 0-complex
    assert cx.dimension == 0 and cx.cochains[0].upper_index is None


# ---- a whole dataset at once (cwn_lift_many -> PackedComplexes.from_arrays) --------------------------------
def _pyg_like(rng, n, bonds, with_attr=True, y_kind='graph', long_x=False):
    """A PyG-Data-like dict: edge_index with BOTH directions in shuffled order, edge_attr per directed entry."""
    und = np.asarray(bonds, dtype=np.int64).reshape(-1, 2)
    attr_und = rng.integers(0, 4, size=(und.shape[0], 1))
    ei = np.concatenate([und, und[:, ::-1]], axis=0)
    ea = np.concatenate([attr_und, attr_und], axis=0)
    order = rng.permutation(ei.shape[0])
    x = torch.from_numpy(rng.integers(0, 28, size=(n, 1)))
    g = dict(x=x if long_x else x.float(), edge_index=torch.from_numpy(ei[order].T.copy()), num_nodes=n,
             edge_attr=torch.from_numpy(ea[order]).float() if with_attr else None,
             y={'graph': torch.tensor([float(n)]), 'vertex': torch.arange(n).float() + 0.5, 'none': None}[y_kind])
    srt = np.lexsort((und.max(1), und.min(1)))                # features of the sorted (u < v) edges
    return g, torch.from_numpy(attr_und[srt]).float()


def _same_packed(a, b):
    assert a.num == b.num and a.max_dim == b.max_dim
    for name in ('dims', 'n_cells', 'has_cells', 'n_up', 'n_down'):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert (a.y is None) == (b.y is None)
    pairs = [(a.y, b.y, 'y')] if a.y is not None else []
    for d in range(a.max_dim + 1):
        assert list(a.keys[d].keys()) == list(b.keys[d].keys()), (d, list(a.keys[d]), list(b.keys[d]))
        pairs += [(a.keys[d][k], b.keys[d][k], f'{d}/{k}') for k in a.keys[d]]
    for p, q, what in pairs:
        assert p.data.dtype == q.data.dtype and torch.equal(p.data, q.data), what
        for f in ('start', 'length', 'has'):
            assert np.array_equal(getattr(p, f), getattr(q, f)), (what, f)
        assert (p.rows, p.width, p.op) == (q.rows, q.width, q.op), what


def test_dataset_ring_lift_packs_what_the_per_graph_lift_packs():
    """pack_graph_dataset_with_rings (graphs -> host threads -> concatenated arrays -> packed dataset) against
    PackedComplexes over per-graph `ring_lift` complexes: every array, table and flag of the packed dataset --
    molecules with and without rings, a graph without edges, both edge directions in shuffled order."""
    from cwn_amd import lifting
    from cwn_amd.packed import PackedComplexes
    rng = np.random.default_rng(21)
    for trial, (down, init_rings, with_attr, long_x) in enumerate(
            [(False, False, True, False), (True, True, True, False), (False, True, False, False), (True, False, True, True)]):
        graphs, ref = [], []
        for i in range(70):
            n, bonds = random_molecule(rng, 6, 40)
            if i % 9 == 4:                                # a tree: no ring, dimension 1
                bonds = [(j, j + 1) for j in range(n - 1)]
            if i % 17 == 6:                               # isolated atoms: dimension 0
                bonds = []
            g, ex_sorted = _pyg_like(rng, n, bonds, with_attr=with_attr, long_x=long_x)
            graphs.append(g)
            vx = g['x']
            pairs = sorted({(min(u, v), max(u, v)) for u, v in bonds})
            if not with_attr:
                ex_sorted = (vx[torch.tensor(pairs, dtype=torch.long)].sum(1) if pairs else None)
            cx = lifting.ring_lift(n, bonds, vx, ex_sorted if pairs else None, max_k=6, include_down_adj=down, y=g['y'])
            if init_rings and cx.dimension == 2:
                rings = lifting.induced_cycles(n, bonds, 6)
                cx.cochains[2].x = torch.stack([vx[list(r)].sum(0) for r in rings])
            ref.append(cx)
        want = PackedComplexes(ref, 'cpu', max_dim=2)
        for threads in (1, 3):
            got, dimension, feats = lifting.pack_graph_dataset_with_rings(
                graphs, max_ring_size=6, include_down_adj=down, init_method='sum', init_edges=True,
                init_rings=init_rings, n_threads=threads, device='cpu')
            _same_packed(got, want)
            assert dimension == 2 and feats[0] == 1
        # ... and with the per-complex CSRs of the boundary adjacencies a static batch collates (round 4)
        got_csr, _, _ = lifting.pack_graph_dataset_with_rings(
            graphs, max_ring_size=6, include_down_adj=down, init_method='sum', init_edges=True, init_rings=init_rings,
            n_threads=2, device='cpu', with_csr=True)
        _same_packed(got_csr, PackedComplexes(ref, 'cpu', max_dim=2, with_csr=True))
        # the same dataset handed over the way a PyG InMemoryDataset stores itself (data + slices)
        sl = dict(x=np.concatenate([[0], np.cumsum([g['num_nodes'] for g in graphs])]),
                  edge_index=np.concatenate([[0], np.cumsum([g['edge_index'].size(1) for g in graphs])]),
                  y=np.arange(len(graphs) + 1))
        got, _, _ = lifting.pack_collated_dataset_with_rings(
            torch.cat([g['x'] for g in graphs]), torch.cat([g['edge_index'] for g in graphs], dim=1),
            torch.cat([g['edge_attr'] for g in graphs if g['edge_attr'] is not None]) if with_attr else None,
            torch.cat([g['y'] for g in graphs]), sl, max_ring_size=6, include_down_adj=down, init_rings=init_rings,
            n_threads=2, device='cpu')
        _same_packed(got, want)
    # vertex labels ride with the vertices (extract_labels, data/utils.py:158-174); no graph label then
    gs = [_pyg_like(rng, *random_molecule(rng, 6, 20), y_kind='vertex')[0] for _ in range(5)]
    p, _, _ = lifting.pack_graph_dataset_with_rings(gs, max_ring_size=6, device='cpu')
    assert p.y is None and 'y' in p.keys[0] and int(p.keys[0]['y'].length.sum()) == int(p.n_cells[0].sum())
    # the two directions of an edge must carry the same features (data/utils.py:468)
    bad = _pyg_like(rng, 3, [(0, 1), (1, 2)])[0]
    bad['edge_attr'][0] += 1
    try:
        lifting.pack_graph_dataset_with_rings([bad], device='cpu')
    except ValueError:
        pass
    else:
        raise AssertionError('asymmetric edge_attr accepted')
    try:
        lifting.pack_graph_dataset_with_rings([dict(x=torch.zeros(2, 1), edge_index=torch.tensor([[0], [5]]), num_nodes=2,
                                                    edge_attr=None, y=None)], device='cpu')
    except ValueError:
        pass
    else:
        raise AssertionError('vertex out of range accepted')


def test_dataset_clique_lift_packs_what_the_per_graph_lift_packs():
    from cwn_amd import lifting
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.synthetic import preferential_attachment_graph
    rng = np.random.default_rng(5)
    graphs, ref = [], []
    for i in range(12):
        n = int(rng.integers(20, 90))
        edges = preferential_attachment_graph(rng, n)
        if isinstance(edges, tuple):
            edges = edges[-1]
        vx = torch.from_numpy(rng.integers(1, 5, size=(n, 2))).float()
        und = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        graphs.append(dict(x=vx, edge_index=torch.from_numpy(np.concatenate([und, und[:, ::-1]]).T.copy()), num_nodes=n,
                           edge_attr=None, y=torch.tensor([i % 2])))
        ref.append(lifting.clique_lift(n, edges, vx, init_method='sum', include_down_adj=True, y=torch.tensor([i % 2])))
    got, dimension, feats = lifting.pack_graph_dataset_with_cliques(graphs, expansion_dim=2, include_down_adj=True,
                                                                    n_threads=2, device='cpu')
    _same_packed(got, PackedComplexes(ref, 'cpu', max_dim=2))
    assert dimension == 2 and feats == [2, 2, 2]


# ------------------------------------------------------------------------------------------------
# clique lift beyond dimension 2 (VERDICT r5 missing #5; data/utils.py:224-297 takes any expansion_dim)
# ------------------------------------------------------------------------------------------------
def test_general_clique_lift_equals_the_native_one_up_to_dimension_two():
    """clique_lift_general restated at max_dim = 2 gives every tensor of the native lift (which is pinned on the reference's
    expected tensors above): hub graphs, both reduces, with and without the lower adjacencies."""
    from cwn_amd import lifting
    from cwn_amd.synthetic import preferential_attachment_graph
    rng = np.random.default_rng(13)
    for i in range(6):
        n = int(rng.integers(20, 90))
        edges = preferential_attachment_graph(rng, n)
        if isinstance(edges, tuple):
            edges = edges[-1]
        vx = torch.from_numpy(rng.integers(0, 5, size=(n, 2))).float()
        for init in ('sum', 'mean'):
            for down in (False, True):
                _same_complex(lifting.clique_lift_general(n, edges, vx, 2, init, None, down),
                              lifting.clique_lift(n, edges, vx, max_dim=2, init_method=init, include_down_adj=down))
    _same_complex(lifting.clique_lift_general(5, HOUSE, X, 1, 'sum', None, True),
                  lifting.clique_lift(5, HOUSE, X, max_dim=1, include_down_adj=True))


def test_clique_lift_of_complete_graphs_to_their_full_dimension():
    """K_m expanded to dimension m - 1: C(m, k + 1) cells in dimension k, every k-cell has k + 1 faces, a face of K_m's
    k-cells has m - k - 1 cofaces; index tensors against their definitions (build_adj, data/utils.py:103-138)."""
    import itertools
    from math import comb
    from cwn_amd import lifting
    for m in (4, 5, 6):
        edges = list(itertools.combinations(range(m), 2))
        vx = torch.arange(m, dtype=torch.float).view(m, 1) + 1
        cx = lifting.clique_lift(m, edges, vx, max_dim=m + 2, include_down_adj=True, y=torch.tensor([1]))
        assert cx.dimension == m - 1
        cells = [list(itertools.combinations(range(m), k + 1)) for k in range(m)]
        for k in range(m):
            c = cx.cochains[k]
            assert c.num_cells == comb(m, k + 1)
            assert torch.equal(c.x, torch.tensor([[float(sum(v + 1 for v in cell))] for cell in cells[k]]))
            if k > 0:
                # boundary_index: face ids ascending per cell, cells ascending (generate_cochain, :204-211)
                want0 = [cells[k - 1].index(f) for cell in cells[k] for f in itertools.combinations(cell, k)]
                want1 = [i for i, cell in enumerate(cells[k]) for _ in range(k + 1)]
                assert c.boundary_index.tolist() == [want0, want1]
                # lower adjacency: per (k-1)-cell, every pair of its cofaces both ways
                n_low = comb(m, k) * (m - k) * (m - k - 1)
                if n_low == 0:                   # the one top cell: every face has a single coface
                    assert c.lower_index is None and c.shared_boundaries is None
                else:
                    assert c.lower_index.size(1) == n_low
                    lo = set(zip(*c.lower_index.tolist()))
                    for (i, a), (j, b) in itertools.combinations(enumerate(cells[k]), 2):
                        assert ((i, j) in lo) == (len(set(a) & set(b)) == k)
                    for col, sh in zip(zip(*c.lower_index.tolist()), c.shared_boundaries.tolist()):
                        assert set(cells[k - 1][sh]) == set(cells[k][col[0]]) & set(cells[k][col[1]])
            else:
                assert c.boundary_index is None and c.lower_index is None
            if k < m - 1:
                assert c.upper_index.size(1) == comb(m, k + 2) * (k + 2) * (k + 1) and c.num_cells_up == comb(m, k + 2)
                for col, sh in zip(zip(*c.upper_index.tolist()), c.shared_coboundaries.tolist()):
                    assert set(cells[k + 1][sh]) == set(cells[k][col[0]]) | set(cells[k][col[1]])
            else:
                assert c.upper_index is None and c.num_cells_up == 0
        # 'mean' reduce and batching of four-dimensional complexes
        cm = lifting.clique_lift(m, edges, vx, max_dim=m, init_method='mean')
        assert torch.allclose(cm.cochains[m - 1].x, vx.mean(0, keepdim=True))
    from cwn_amd.complex import ComplexBatch
    k5 = lifting.clique_lift(5, list(itertools.combinations(range(5), 2)), torch.ones(5, 1), max_dim=4, include_down_adj=True)
    k4 = lifting.clique_lift(4, list(itertools.combinations(range(4), 2)), torch.ones(4, 1), max_dim=4, include_down_adj=True)
    b = ComplexBatch.from_complex_list([k5, k4, k5], max_dim=4)
    assert b.dimension == 4 and [b.cochains[d].num_cells for d in range(5)] == [14, 26, 24, 11, 2]
    assert b.cochains[3].boundary_index.size(1) == 4 * 11 and int(b.cochains[3].boundary_index[0].max()) == 23


def test_dataset_conversion_with_cliques_as_the_reference_tests_it():
    """data/test_utils.py:127-213 (expansion_dim = 3 on the house graph: dimension 2 comes out) through
    convert_graph_dataset_with_cliques: same counts, labels, and the batch of the three complexes."""
    from cwn_amd import lifting
    from cwn_amd.complex import ComplexBatch
    ei = torch.tensor([[u for u, v in HOUSE] + [v for u, v in HOUSE], [v for u, v in HOUSE] + [u for u, v in HOUSE]])
    data = [dict(edge_index=ei, x=torch.arange(0, 5, dtype=torch.float).view(5, 1), y=torch.tensor([1]), num_nodes=5)
            for _ in range(3)]
    for down in (True, False):
        complexes, dim, num_features = lifting.convert_graph_dataset_with_cliques(data, expansion_dim=3, include_down_adj=down)
        assert dim == 2 and num_features == [1, 1, 1] and len(complexes) == 3
        for cx in complexes:
            assert cx.dimension == 2 and cx.cochains[0].boundary_index is None
            assert list(cx.cochains[1].boundary_index.size()) == [2, 2 * 6]
            assert list(cx.cochains[2].boundary_index.size()) == [2, 3 * 1]
            assert (cx.cochains[1].lower_index.size(1) == 18) if down else (cx.cochains[1].lower_index is None)
            assert torch.equal(cx.cochains[0].x, data[0]['x']) and torch.equal(cx.y, data[0]['y'])
        batch = ComplexBatch.from_complex_list(complexes)
        assert batch.dimension == 2
        assert list(batch.cochains[1].boundary_index.size()) == [2, 3 * 2 * 6]
        assert list(batch.cochains[2].boundary_index.size()) == [2, 1 * 3 * 3]
        assert (batch.cochains[1].lower_index.size(1) == 18 * 3) if down else (batch.cochains[1].lower_index is None)
    # vertex-level labels go to the vertices' cochain (extract_labels)
    data[0]['y'] = torch.arange(5)
    complexes, _, _ = lifting.convert_graph_dataset_with_cliques(data[:1], expansion_dim=4)
    assert complexes[0].y is None and torch.equal(complexes[0].cochains[0].y, torch.arange(5))
