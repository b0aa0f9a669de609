"""The reference's invariance tests on the HIP path (the product, not the checker):

  * batched forward == concatenation of per-complex forwards, batch sizes 2 .. 9 of the list, per-layer
    partial outputs included, atol 1e-6 as in the reference (mp/test_models.py:139-185,
    mp/test_molec_models.py:11-68) -- with the complex-blocked layer kernel and with the CSR path;
  * the ring-lifted model is invariant to a relabelling of the vertices, through csrc/cwn_lift.cpp
    (mp/test_permutation.py:9-36);
  * batch composition does not leak: a complex's rows are bit-identical whatever else is in the batch
    (the per-complex kernels sum in entry order inside the complex).
"""
import numpy as np
import pytest
import torch

from cwn_amd import layers, lifting, synthetic
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import EmbedSparseCIN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _model(hidden=64, n_layers=3, seed=0, use_coboundaries=True):
    torch.manual_seed(seed)
    m = EmbedSparseCIN(28, 4, 3, n_layers, hidden, dropout_rate=0.0, max_dim=2, embed_edge=True,
                       use_coboundaries=use_coboundaries, graph_norm='bn').eval()
    with torch.no_grad():                                # non-trivial BatchNorm statistics
        for name, buf in m.named_buffers():
            if name.endswith('running_mean'):
                buf.copy_(torch.randn_like(buf) * 0.1)
            elif name.endswith('running_var'):
                buf.copy_(torch.rand_like(buf) + 0.5)
    return m.to(DEV)


def _forward(model, complexes):
    b = ComplexBatch.from_complex_list(complexes, max_dim=2).to(DEV)
    with torch.no_grad():
        y, res = model(b, include_partial=True)
    return y, res


def _complexes():
    cs = synthetic.zinc_like_complexes(7, seed=3, n_lo=8, n_hi=16)
    # a molecule without rings and one that is a single bond: dimension < 2 inside a dimension-2 batch
    cs.append(lifting.ring_lift(2, [(0, 1)], torch.tensor([[3.], [5.]]), torch.tensor([[1.]]), max_k=6,
                                y=torch.zeros(1)))
    cs.append(lifting.ring_lift(4, [(0, 1), (1, 2), (1, 3)], torch.tensor([[1.], [2.], [3.], [4.]]),
                                torch.tensor([[0.], [1.], [2.]]), max_k=6, y=torch.zeros(1)))
    return cs


@pytest.mark.parametrize('blocked', [True, False])
@pytest.mark.parametrize('use_coboundaries', [True, False])
def test_batched_forward_equals_per_complex_forwards_on_the_gpu(use_coboundaries, blocked):
    prev = layers.BLOCKED_LAYER
    layers.BLOCKED_LAYER = blocked
    try:
        complexes = _complexes()
        model = _model(use_coboundaries=use_coboundaries)
        single = [_forward(model, [c]) for c in complexes]
        worst = 0.0
        for bs in range(2, len(complexes) + 1):
            outs, parts = [], {}
            for i in range(0, len(complexes), bs):
                out, res = _forward(model, complexes[i:i + bs])
                outs.append(out)
                for k, v in res.items():
                    parts.setdefault(k, []).append(v)
            want = torch.cat([s[0] for s in single])
            worst = max(worst, float((torch.cat(outs) - want).abs().max()))
            torch.testing.assert_close(torch.cat(outs), want, rtol=0, atol=1e-6)
            for k in parts:
                wk = [s[1][k] for s in single if k in s[1]]
                # a complex without 2-cells contributes no rows of dimension 2 either way
                torch.testing.assert_close(torch.cat(parts[k]), torch.cat(wk), rtol=0, atol=1e-6)
        print(f'[invariance] batched vs per-complex (cob={use_coboundaries}, blocked={blocked}): max|delta| = {worst:.3e}')
        if blocked and use_coboundaries:
            assert model.convs[0].blocked_reason is None, model.convs[0].blocked_reason
    finally:
        layers.BLOCKED_LAYER = prev


def test_ring_lifted_model_is_invariant_to_vertex_relabelling_on_the_gpu():
    rng = np.random.default_rng(11)
    model = _model(hidden=64, seed=5)
    worst = 0.0
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

        ref, _ = _forward(model, [lifted(list(range(n)))])
        for _ in range(4):
            perm = rng.permutation(n).tolist()
            out, _ = _forward(model, [lifted(perm)])
            worst = max(worst, float((out - ref).abs().max()))
            torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    print(f'[invariance] vertex relabelling: max|delta| = {worst:.3e}')


def test_a_complex_is_computed_the_same_in_any_batch():
    """Propagate scope (blocked kernel): complex 0's output rows do not depend on its batch mates --
    bit for bit (its own workgroup, its own entry order)."""
    from cwn_amd.layers import SparseCINConv
    torch.manual_seed(1)
    F = 128
    conv = SparseCINConv(F, F, F, None, None, None, None, max_dim=2, hidden=F, act_module=torch.nn.ReLU,
                         layer_dim=F, use_coboundaries=True).to(DEV).eval()
    cs = synthetic.zinc_like_complexes(40, seed=9)
    g = torch.Generator().manual_seed(0)
    feats = [[torch.randn(c.cochains[d].num_cells, F, generator=g) if d in c.cochains else None for d in range(3)]
             for c in cs]

    def run(idx):
        b = ComplexBatch.from_complex_list([cs[i] for i in idx], max_dim=2).to(DEV)
        xs = [torch.cat([feats[i][d] for i in idx if feats[i][d] is not None]).to(DEV) for d in range(3)]
        b.set_xs(xs)
        with torch.no_grad():
            plans, outs = conv.propagate_all(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
        assert plans[0] == 'blocked'
        return outs, [cs[idx[0]].cochains[d].num_cells if d in cs[idx[0]].cochains else 0 for d in range(3)]

    a, n0 = run([0])
    for idx in ([0, 5, 7], list(range(40)), [0] + list(range(39, 20, -1))):
        o, _ = run(idx)
        for d in range(3):
            assert torch.equal(o[2 * d][:n0[d]], a[2 * d][:n0[d]]) and torch.equal(o[2 * d + 1][:n0[d]], a[2 * d + 1][:n0[d]])
