"""Invariance / property tests of the reference's test-suite, run on the CPU checker
and the host-side lifting (no GPU):

  * batched forward == concatenation of per-complex forwards, every batch size, per-layer partial
    outputs included (mp/test_models.py:139-185, mp/test_molec_models.py:11-68);
  * SparseCIN(use_coboundaries=True) on ring-lifted molecules is invariant to a relabelling of the
    vertices (mp/test_permutation.py:9-36) -- here the ring lift is csrc/cwn_lift.cpp, so this also
    pins the lifting's cell / adjacency construction under permutation;
  * ZINC ring lifting finds exactly the induced cycles: the C++ enumerator against the brute-force
    Python one (data/datasets/test_zinc.py:12-57 with data/helper_test.py:68-99).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cwn_oracle as O                      # noqa: E402  (tests may import the checker)
from cwn_amd import lifting, synthetic                  # noqa: E402
from cwn_amd.complex import ComplexBatch                # noqa: E402
from cwn_amd.models import EmbedSparseCIN               # noqa: E402

KEYS = ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries', 'boundary_index', 'y',
        'batch')


def oracle_cx(b):
    return {'dimension': b.dimension, 'y': None, 'num_complexes': b.num_complexes,
            'cochains': [{k: b.cochains[d][k] for k in KEYS} for d in range(b.dimension + 1)]}


def model_state(hidden=16, layers=3, seed=0, use_coboundaries=True):
    torch.manual_seed(seed)
    m = EmbedSparseCIN(28, 4, 3, layers, hidden, dropout_rate=0.0, max_dim=2, embed_edge=True,
                       use_coboundaries=use_coboundaries, graph_norm='bn').eval()
    with torch.no_grad():                                # non-trivial BatchNorm statistics
        for name, buf in m.named_buffers():
            if name.endswith('running_mean'):
                buf.copy_(torch.randn_like(buf) * 0.1)
            elif name.endswith('running_var'):
                buf.copy_(torch.rand_like(buf) + 0.5)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def forward(state, complexes, layers=3, use_coboundaries=True):
    b = ComplexBatch.from_complex_list(complexes, max_dim=2)
    return O.embed_sparse_cin_forward(state, oracle_cx(b), layers, max_dim=2, use_coboundaries=use_coboundaries)


@pytest.mark.parametrize('use_coboundaries', [True, False])
def test_batched_forward_equals_per_complex_forwards(use_coboundaries):
    complexes = synthetic.zinc_like_complexes(7, seed=3, n_lo=8, n_hi=16)
    # a molecule without rings and one that is a single bond: dimension < 2 inside a dimension-2 batch
    complexes.append(lifting.ring_lift(2, [(0, 1)], torch.tensor([[3.], [5.]]), torch.tensor([[1.]]), max_k=6,
                                       y=torch.zeros(1)))
    complexes.append(lifting.ring_lift(4, [(0, 1), (1, 2), (1, 3)], torch.tensor([[1.], [2.], [3.], [4.]]),
                                       torch.tensor([[0.], [1.], [2.]]), max_k=6, y=torch.zeros(1)))
    state = model_state(use_coboundaries=use_coboundaries)
    single = [forward(state, [c], use_coboundaries=use_coboundaries) for c in complexes]
    for bs in range(2, len(complexes) + 1):
        outs, parts = [], {}
        for i in range(0, len(complexes), bs):
            out, res = forward(state, complexes[i:i + bs], use_coboundaries=use_coboundaries)
            outs.append(out)
            for k, v in res.items():
                parts.setdefault(k, []).append(v)
        torch.testing.assert_close(torch.cat(outs), torch.cat([s[0] for s in single]), rtol=1e-5, atol=1e-5)
        for k in parts:
            want = [s[1][k] for s in single if k in s[1]]
            # a complex without 2-cells contributes no rows of dimension 2 either way
            torch.testing.assert_close(torch.cat(parts[k]), torch.cat(want), rtol=1e-5, atol=1e-5)


def test_ring_lifted_model_is_invariant_to_vertex_relabelling():
    rng = np.random.default_rng(11)
    state = model_state(hidden=32, seed=5)
    for _ in range(6):
        n, bonds = synthetic.random_molecule(rng, 8, 18)
        vx = torch.from_numpy(rng.integers(0, 28, size=(n, 1))).float()
        etype = {b: float(rng.integers(0, 4)) for b in bonds}

        def lifted(perm):
            """Relabel vertex v as perm[v]; edge features follow the sorted (u < v) edge order."""
            pb = sorted((min(perm[u], perm[v]), max(perm[u], perm[v])) for u, v in bonds)
            inv = {(min(perm[u], perm[v]), max(perm[u], perm[v])): etype[(u, v)] for u, v in bonds}
            pvx = torch.empty_like(vx)
            pvx[torch.tensor(perm)] = vx
            ex = torch.tensor([[inv[b]] for b in pb])
            return lifting.ring_lift(n, pb, pvx, ex, max_k=6, y=torch.zeros(1))

        ref, _ = forward(state, [lifted(list(range(n)))])
        for _ in range(4):
            perm = rng.permutation(n).tolist()
            out, _ = forward(state, [lifted(perm)])
            torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)


def test_ring_lift_finds_exactly_the_induced_cycles():
    rng = np.random.default_rng(2)
    for _ in range(40):
        n, bonds = synthetic.random_molecule(rng, 6, 24)
        # extra chords make non-induced cycles that the lift must NOT report
        for _ in range(int(rng.integers(0, 3))):
            u, v = (int(t) for t in rng.integers(0, n, 2))
            if u != v:
                bonds = sorted(set(bonds) | {(min(u, v), max(u, v))})
        for max_k in (3, 5, 6, 8):
            fast = {tuple(sorted(c)) for c in lifting.induced_cycles(n, bonds, max_k)}
            slow = {tuple(sorted(c)) for c in synthetic.induced_cycles(n, bonds, max_k)}
            assert fast == slow
