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
