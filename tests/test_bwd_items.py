"""The item table of the OWNER form of the blocked backward launch (cwn_layer_bwd_items_build, host C++) and the
algorithm its records drive (csrc/cwn_layer_bwd_own.hip), on the CPU: a numpy restatement of the kernel -- per record:
stage by local row number, gather per owned row over the three entry lists, multiply by the transposed weights, store
the owned rows ONCE -- executes the table and must reproduce float64 autograd of the plain propagate step
(mp/layers.py:184-192, 290-295), write every row of dx / gY1 / gY2 exactly once and stay inside the record's ranges."""
import numpy as np
import pytest
import torch

R_FLAGS, R_DIM, R_OWN_R0, R_OWN_N, R_ABOVE_R0, R_ABOVE_N, R_BELOW_R0, R_BELOW_N = range(8)
R_UPA_E0, R_UPA_NE, R_UPB_E0, R_UPB_NE, R_BND_E0, R_BND_NE, R_LDS = range(8, 15)
PA, PB, TOP = 1, 2, 4


def _batch(n, seed, kind='zinc', **kw):
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes, molhiv_like_complexes
    gen = zinc_like_complexes if kind == 'zinc' else molhiv_like_complexes
    return ComplexBatch.from_complex_list(gen(n, seed, 6, **kw), max_dim=2)


def _problem(b, F, seed):
    """Random features / weights / output gradients for the batch's dimensions (float64)."""
    g = torch.Generator().manual_seed(seed)
    P = {'x': [], 'W': [], 'bias': [], 'eps1': [], 'eps2': [], 'gU': [], 'gB': [], 'up': [], 'sh': [], 'bi': []}
    nd = len(b.cochains)
    for d in range(nd):
        c = b.cochains[d]
        n = c.num_cells
        P['x'].append(torch.randn(n, F, generator=g, dtype=torch.float64))
        P['W'].append(torch.randn(F, 2 * F, generator=g, dtype=torch.float64) / (2 * F) ** 0.5)
        P['bias'].append(torch.randn(F, generator=g, dtype=torch.float64) * 0.1)
        P['eps1'].append(0.25 * (d + 1))
        P['eps2'].append(-0.1 * (d + 1))
        P['gU'].append(torch.randn(n, F, generator=g, dtype=torch.float64))
        P['gB'].append(torch.randn(n, F, generator=g, dtype=torch.float64))
        up = c.upper_index if (d < nd - 1 and c.upper_index is not None and c.upper_index.size(1)) else None
        P['up'].append(up)
        P['sh'].append(c.shared_coboundaries if up is not None else None)
        bi = c.boundary_index if (d > 0 and c.boundary_index is not None and c.boundary_index.size(1)) else None
        P['bi'].append(bi)
    return P


def _autograd(P, F):
    xs = [x.clone().requires_grad_() for x in P['x']]
    nd = len(xs)
    Y1, Y2, loss = [None] * nd, [None] * nd, 0.0
    for d in range(nd):
        n = xs[d].size(0)
        out_up = (1 + P['eps1'][d]) * xs[d]
        if P['up'][d] is not None:
            W = P['W'][d]
            Y1[d] = xs[d] @ W[:, :F].t() + P['bias'][d]
            Y2[d + 1] = xs[d + 1] @ W[:, F:].t()
            Y1[d].retain_grad()
            Y2[d + 1].retain_grad()
            msg = torch.relu(Y1[d][P['up'][d][0]] + Y2[d + 1][P['sh'][d]])
            out_up = out_up + torch.zeros(n, F, dtype=torch.float64).index_add(0, P['up'][d][1], msg)
        out_b = (1 + P['eps2'][d]) * xs[d]
        if P['bi'][d] is not None:
            out_b = out_b + torch.zeros(n, F, dtype=torch.float64).index_add(0, P['bi'][d][1], xs[d - 1][P['bi'][d][0]])
        loss = loss + (out_up * P['gU'][d]).sum() + (out_b * P['gB'][d]).sum()
    loss.backward()
    det = lambda y: None if y is None else y.detach()
    return ([x.grad for x in xs], [None if y is None else y.grad for y in Y1], [None if y is None else y.grad for y in Y2],
            [det(y) for y in Y1], [det(y) for y in Y2])


def _execute(tab, P, Y1, Y2, F):
    """The kernel, record by record, in numpy float64.  Returns dx, gy1, gy2 and the per-row write counts."""
    n = [int(x.size(0)) for x in P['x']]
    dx = [np.full((m, F), np.nan) for m in n]
    gy1 = [np.full((m, F), np.nan) for m in n]
    gy2 = [np.full((m, F), np.nan) for m in n]
    cnt = {k: [np.zeros(m, dtype=np.int64) for m in n] for k in ('dx', 'gy1', 'gy2')}
    npy = lambda t: None if t is None else t.numpy()
    gU, gB = [npy(t) for t in P['gU']], [npy(t) for t in P['gB']]
    y1, y2 = [npy(t) for t in Y1], [npy(t) for t in Y2]
    W = [npy(t) for t in P['W']]
    up, sh, bi = [npy(t) for t in P['up']], [npy(t) for t in P['sh']], [npy(t) for t in P['bi']]
    for r in tab:
        fl, d = int(r[R_FLAGS]) & 7, int(r[R_DIM])
        o0, no, a0, na, b0, nb = (int(r[k]) for k in (R_OWN_R0, R_OWN_N, R_ABOVE_R0, R_ABOVE_N, R_BELOW_R0, R_BELOW_N))
        ea0, nea, eb0, neb, bd0, nbd = (int(r[k]) for k in (R_UPA_E0, R_UPA_NE, R_UPB_E0, R_UPB_NE, R_BND_E0, R_BND_NE))
        assert no > 0
        O = (1 + P['eps1'][d]) * gU[d][o0:o0 + no] + (1 + P['eps2'][d]) * gB[d][o0:o0 + no]
        g1 = np.zeros((no, F))
        g2 = np.zeros((no, F))
        g3 = np.zeros((na, F))
        if fl & PA:
            ej = up[d][0, ea0:ea0 + nea] - o0
            ei = up[d][1, ea0:ea0 + nea] - o0
            ec = sh[d][ea0:ea0 + nea] - a0
            assert ((0 <= ej) & (ej < no) & (0 <= ei) & (ei < no) & (0 <= ec) & (ec < na)).all()
            y1o, guo, y2a = y1[d][o0:o0 + no], gU[d][o0:o0 + no], y2[d + 1][a0:a0 + na]
            for q in range(nea):                      # (entry order per row = ascending q)
                m = guo[ei[q]] * ((y1o[ej[q]] + y2a[ec[q]]) > 0)
                g1[ej[q]] += m
                if fl & TOP:
                    g3[ec[q]] += m
        else:
            assert nea == 0
        if fl & PB:
            fj = up[d - 1][0, eb0:eb0 + neb] - b0
            fi = up[d - 1][1, eb0:eb0 + neb] - b0
            fc = sh[d - 1][eb0:eb0 + neb] - o0
            assert ((0 <= fj) & (fj < nb) & (0 <= fi) & (fi < nb) & (0 <= fc) & (fc < no)).all()
            y1b, gub, y2o = y1[d - 1][b0:b0 + nb], gU[d - 1][b0:b0 + nb], y2[d][o0:o0 + no]
            for q in range(neb):
                g2[fc[q]] += gub[fi[q]] * ((y1b[fj[q]] + y2o[fc[q]]) > 0)
        else:
            assert neb == 0 and nb == 0
        if nbd:
            bb = bi[d + 1][0, bd0:bd0 + nbd] - o0
            ii = bi[d + 1][1, bd0:bd0 + nbd] - a0
            assert ((0 <= bb) & (bb < no) & (0 <= ii) & (ii < na)).all()
            gba = gB[d + 1][a0:a0 + na]
            for q in range(nbd):
                O[bb[q]] += gba[ii[q]]
        if fl & PA:
            O = O + g1 @ W[d][:, :F]
            gy1[d][o0:o0 + no] = g1
            cnt['gy1'][d][o0:o0 + no] += 1
        if fl & PB:
            O = O + g2 @ W[d - 1][:, F:]
            gy2[d][o0:o0 + no] = g2
            cnt['gy2'][d][o0:o0 + no] += 1
        dx[d][o0:o0 + no] = O
        cnt['dx'][d][o0:o0 + no] += 1
        if fl & TOP and na:
            Ot = (1 + P['eps1'][d + 1]) * gU[d + 1][a0:a0 + na] + (1 + P['eps2'][d + 1]) * gB[d + 1][a0:a0 + na] + g3 @ W[d][:, F:]
            dx[d + 1][a0:a0 + na] = Ot
            gy2[d + 1][a0:a0 + na] = g3
            cnt['dx'][d + 1][a0:a0 + na] += 1
            cnt['gy2'][d + 1][a0:a0 + na] += 1
    return dx, gy1, gy2, cnt


@pytest.mark.parametrize('n,F,kind', [(128, 128, 'zinc'), (7, 128, 'zinc'), (600, 128, 'zinc'), (300, 64, 'molhiv'), (40, 64, 'zinc')])
def test_owner_table_executed_on_the_cpu_reproduces_autograd(n, F, kind):
    from cwn_amd.blockplan import BlockPlan
    b = _batch(n, seed=n + F, kind=kind)
    plan = BlockPlan.from_batch(b)
    has_up = [True, True, False]
    t = plan.bwd_items(F, has_up)
    assert t is not None
    tab = t.host
    assert tab.shape[1] == 16 and tab.dtype == np.int32
    assert 0 < t.lds_bytes <= 160 * 1024 and int(tab[:, R_LDS].max()) == t.lds_bytes
    # two sets: vertices own dimension 0; edges own dimension 1 and ride the rings (TOP); flags as the header says
    fl = tab[:, R_FLAGS]
    assert set((fl >> 8).tolist()) == {0, 1}
    assert ((fl[(fl >> 8) == 0] & 7) == PA).all() and ((fl[(fl >> 8) == 1] & 7) == (PA | PB | TOP)).all()
    # limits of the kernel
    cap = 256 if F == 64 else 96
    assert (tab[:, R_OWN_N] <= cap).all() and (tab[:, R_OWN_N] > 0).all()
    top = (fl & TOP) != 0
    assert (tab[top, R_ABOVE_N] <= 1024 // (F // 4)).all()
    assert (tab[:, [R_UPA_NE, R_UPB_NE, R_BND_NE]] <= 1024).all()
    # the chip is filled where the batch allows it: about one item per complex and set up to 128 per set
    assert t.n_items >= min(2 * n, 200) or n < 100
    P = _problem(b, F, seed=3)
    dx_ref, gy1_ref, gy2_ref, Y1, Y2 = _autograd(P, F)
    dx, gy1, gy2, cnt = _execute(tab, P, Y1, Y2, F)
    for d in range(len(dx)):
        assert (cnt['dx'][d] == 1).all(), f'dx[{d}]: a row without exactly one owner'
        np.testing.assert_allclose(dx[d], dx_ref[d].numpy(), rtol=1e-10, atol=1e-10)
        if gy1_ref[d] is not None:
            assert (cnt['gy1'][d] == 1).all()
            np.testing.assert_allclose(gy1[d], gy1_ref[d].numpy(), rtol=1e-10, atol=1e-10)
        if gy2_ref[d] is not None:
            assert (cnt['gy2'][d] == 1).all()
            np.testing.assert_allclose(gy2[d], gy2_ref[d].numpy(), rtol=1e-10, atol=1e-10)


def test_owner_table_limits_and_bad_arguments():
    """A complex beyond a workgroup gives no table (the caller keeps the streaming backward); argument checks."""
    from cwn_amd import _ffi
    from cwn_amd.blockplan import BlockPlan
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    L = _ffi.lib()
    assert L.cwn_layer_bwd_items_build(None, 128, None, 0, None) == _ffi.LAYER_ITEMS_BAD_ARG
    cxs = zinc_like_complexes(10, 0, 6) + zinc_like_complexes(1, 1, 6, n_lo=120, n_hi=130)
    plan = BlockPlan.from_batch(ComplexBatch.from_complex_list(cxs, max_dim=2))
    assert plan.bwd_items(128, [True, True, False]) is None
    assert plan.bwd_items(128, [True, True, False]) is None            # (cached)
    small = BlockPlan.from_batch(ComplexBatch.from_complex_list(cxs[:10], max_dim=2))
    t = small.bwd_items(128, [True, True, False])
    assert t is not None and t.n_items == 20
    # a layer without the boundary stream: no record carries boundary entries
    t2 = small.bwd_items(128, [True, True, False], [False, False, False])
    assert t2 is not None and not t2.host[:, [R_BND_E0, R_BND_NE]].any()
    # the launcher refuses a plan that names more than the tensors hold, without touching the GPU
    dims = (_ffi.LayerBwdDim * 3)()
    p = t.c_plan()
    assert L.cwn_layer_bwd_own_f32(dims, 3, 128, p, None, None) == 1          # CWN_ERR_BAD_ARG
    assert L.cwn_layer_bwd_own_f32(dims, 3, 96, p, None, None) == 1          # CWN_ERR_BAD_ARG
    assert L.cwn_layer_bwd_own_f32(None, 3, 128, p, None, None) == 1          # CWN_ERR_BAD_ARG


def test_owner_table_of_a_one_dimensional_batch_and_of_complexes_without_rings():
    """Graphs lifted to 1-complexes (vertices own their rows, the edges ride as the TOP rows of the same items: first and
    third product, no second) and a batch in which some molecules have no ring at all (items whose TOP block is empty)."""
    from cwn_amd.blockplan import BlockPlan
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    F = 128
    small = [c for c in zinc_like_complexes(60, 3, 6) if c.cochains[1].num_cells <= 32]      # (TOP rows: one round of lane groups)
    b1 = ComplexBatch.from_complex_list(small, max_dim=1)
    t1 = BlockPlan.from_batch(b1).bwd_items(F, [True, False])
    assert t1 is not None and ((t1.host[:, R_FLAGS] & 7) == (PA | TOP)).all()
    # ... and one molecule with more edges than that gives no table: the caller keeps the streaming backward
    assert BlockPlan.from_batch(ComplexBatch.from_complex_list(zinc_like_complexes(60, 3, 6), max_dim=1)).bwd_items(F, [True, False]) is None
    P = _problem(b1, F, seed=5)
    dx_ref, gy1_ref, gy2_ref, Y1, Y2 = _autograd(P, F)
    dx, gy1, gy2, cnt = _execute(t1.host, P, Y1, Y2, F)
    for d in range(2):
        assert (cnt['dx'][d] == 1).all()
        np.testing.assert_allclose(dx[d], dx_ref[d].numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gy1[0], gy1_ref[0].numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gy2[1], gy2_ref[1].numpy(), rtol=1e-10, atol=1e-10)
    # trees next to ring molecules: rings (and the edges' upper adjacency) absent for some complexes
    cxs = zinc_like_complexes(30, 7, 6)
    trees = [c for c in zinc_like_complexes(60, 11, 2) if 2 not in c.cochains or c.cochains[2].num_cells == 0][:6]
    if trees:
        mixed = cxs[:10] + trees + cxs[10:]
        b2 = ComplexBatch.from_complex_list(mixed, max_dim=2)
        t2 = BlockPlan.from_batch(b2).bwd_items(F, [True, True, False])
        assert t2 is not None
        P = _problem(b2, F, seed=6)
        dx_ref, gy1_ref, gy2_ref, Y1, Y2 = _autograd(P, F)
        dx, gy1, gy2, cnt = _execute(t2.host, P, Y1, Y2, F)
        for d in range(3):
            assert (cnt['dx'][d] == 1).all()
            np.testing.assert_allclose(dx[d], dx_ref[d].numpy(), rtol=1e-10, atol=1e-10)
