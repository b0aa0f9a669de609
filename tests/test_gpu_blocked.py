"""The complex-blocked layer kernel (csrc/cwn_layer.hip, `cwn_layer_fused_f32`) through the C ABI:

  * against the CPU oracle's propagate (mp/cell_mp.py:357-392 restated) evaluated in FLOAT64 with the
    reference's coboundary message ReLU(Linear(cat(x_j, up_attr))) (mp/layers.py:290-295) and the
    self terms of mp/layers.py:191-192 -- gate: max|delta| <= 1e-5 * max(1, |ref|_inf) (north star);
  * bit-for-bit against the two-kernel path (grouped split GEMM + CSR aggregation) at F = 128: same
    split, same MFMA order, same epilogue arithmetic, same entry order;
  * exactly (torch.equal) against the oracle on integer-valued features and weights;
  * its failure modes: an index that crosses complexes, a table that is not the batch's."""
import numpy as np
import pytest
import torch

from oracle import cwn_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def cpu(t):
    return None if t is None else t.detach().cpu()


def _conv(F, seed=0, eps=0.0, train_eps=False):
    from cwn_amd.layers import SparseCINConv
    torch.manual_seed(seed)
    conv = SparseCINConv(F, F, F, None, None, None, None, max_dim=2, hidden=F, eps=eps, train_eps=train_eps,
                         act_module=torch.nn.ReLU, layer_dim=F, use_coboundaries=True)
    return conv.to(DEV).eval()


def _batch(kind, n, F, seed=0, integer=False):
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes, molhiv_like_complexes
    gen = zinc_like_complexes if kind in ('zinc', 'zinctrees', 'zincwide') else molhiv_like_complexes
    # 'zincwide': molecules of 26 .. 40 atoms -- items of 33 .. 48 cells (one round of the reduce passes' lane groups plus a few)
    cxs = gen(n, seed, 6, n_lo=26, n_hi=40) if kind == 'zincwide' else gen(n, seed, 6)
    if kind == 'zinctrees':            # some molecules without a ring (no 2-cells, no upper adjacency of their edges)
        trees = [c for c in zinc_like_complexes(3 * n, seed + 1, 2) if 2 not in c.cochains or c.cochains[2].num_cells == 0][:max(1, n // 4)]
        cxs = cxs[: n // 2] + trees + cxs[n // 2:]
    b = ComplexBatch.from_complex_list(cxs, max_dim=2).to(DEV)
    g = torch.Generator().manual_seed(seed + 17)
    for d in range(3):
        n_d = b.cochains[d].num_cells
        x = (torch.randint(-3, 4, (n_d, F), generator=g).float() if integer
             else torch.randn(n_d, F, generator=g))
        b.cochains[d].x = x.to(DEV)
    return b


def _oracle_scope(conv, b, dtype=torch.float64):
    """[(out_up, out_b)] per dimension: the oracle's propagate + self terms, in `dtype`."""
    outs = []
    for d in range(3):
        c = b.cochains[d]
        lvl = conv.mp_levels[d]
        x = cpu(c.x).to(dtype)
        W = cpu(lvl.msg_up_nn[1].weight).to(dtype)
        bias = cpu(lvl.msg_up_nn[1].bias).to(dtype)
        up_index = cpu(c.upper_index) if (d + 1) in b.cochains else None
        up_attr = None
        if up_index is not None:
            up_attr = cpu(b.cochains[d + 1].x).to(dtype)[cpu(c.shared_coboundaries)]
        b_attr = cpu(b.cochains[d - 1].x).to(dtype) if d > 0 else None
        msg = lambda xj, a: torch.relu(torch.cat([xj, a], -1) @ W.t() + bias)
        F = x.size(1)
        up, _, bnd = O.propagate(x, up_index, None, cpu(c.boundary_index) if d > 0 else None,
                                 up_attr=up_attr, boundary_attr=b_attr, message_up=msg,
                                 use_down_msg=False, up_msg_size=F, down_msg_size=F, boundary_msg_size=F)
        up, bnd = up.to(dtype), bnd.to(dtype)
        e1, e2 = float(lvl.eps1), float(lvl.eps2)
        outs.append((up + (1 + e1) * x, bnd + (1 + e2) * x))
    return outs


def _run(conv, b, blocked):
    from cwn_amd import layers
    prev = layers.BLOCKED_LAYER
    layers.BLOCKED_LAYER = blocked
    try:
        with torch.no_grad():
            params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
            plans, outs = conv.propagate_all(*params)
        assert (plans[0] == 'blocked') == blocked
    finally:
        layers.BLOCKED_LAYER = prev
    return outs


def _gate(got, ref64, what):
    ref = ref64
    err = (cpu(got).double() - ref).abs().max().item() if ref.numel() else 0.0
    bound = 1e-5 * max(1.0, ref.abs().max().item() if ref.numel() else 1.0)
    print(f'{what}: max|delta| = {err:.3e}  (gate {bound:.3e}, |ref|_inf = {bound / 1e-5:.3f})')
    assert err <= bound, (what, err, bound)


@pytest.mark.parametrize('kind,n,F', [('zinc', 128, 128), ('zinc', 7, 128), ('molhiv', 512, 64), ('zinc', 37, 64),
                                      ('zinc', 300, 128)])
def test_blocked_layer_vs_float64_oracle(kind, n, F):
    b = _batch(kind, n, F, seed=1)
    conv = _conv(F, seed=2, eps=0.25)
    outs = _run(conv, b, blocked=True)
    ref = _oracle_scope(conv, b)
    for d in range(3):
        _gate(outs[2 * d], ref[d][0], f'{kind}-{n} F={F} out_up[{d}]')
        _gate(outs[2 * d + 1], ref[d][1], f'{kind}-{n} F={F} out_b[{d}]')


@pytest.mark.parametrize('kind,n', [('zinc', 128), ('zinc', 300), ('zinc', 1)])
def test_blocked_layer_bit_identical_to_two_kernel_path(kind, n):
    """F = 128: the grouped GEMM runs the same bf16-split arithmetic, the CSR aggregation the same
    sequential entry order -> every output bit must agree."""
    from cwn_amd import csr
    b = _batch(kind, n, 128, seed=3)
    conv = _conv(128, seed=4, eps=0.5)
    fused = _run(conv, b, blocked=True)
    csr._cache.clear()
    plain = _run(conv, b, blocked=False)
    for i, (f, p) in enumerate(zip(fused, plain)):
        assert torch.equal(f, p), (i, (f - p).abs().max().item())


def test_blocked_layer_items_one_round_plus_a_few_cells():
    """Round 6: at F = 128 a round of the reduce passes is 32 lane groups; an item of 33 .. 48 edges (or 33 .. 64 cells in the
    upper reduce) no longer takes a second round -- its overflow cells ride as the second chain of the lane groups that loaded
    them (csrc/cwn_layer.hip, CWN_LAYER_MERGE2).  The batch holds such items, items beyond them (49+ edges: the generic two
    rounds) and ordinary ones; every output bit equals the two-kernel path's, in sort / store and load mode."""
    from cwn_amd import csr, layers
    b = _batch('zincwide', 96, 128, seed=71)
    n_v = torch.bincount(cpu(b.cochains[0].batch))
    n_e = torch.bincount(cpu(b.cochains[1].batch))
    assert int(((n_e > 32) & (n_e <= 48)).sum()) >= 10 and int((n_v > 32).sum()) >= 5 and int((n_e <= 32).sum()) >= 5, (n_v, n_e)
    conv = _conv(128, seed=72, eps=0.375)
    keep, layers.LAYER_VARIANT = layers.LAYER_VARIANT, '0'           # the 16-wave form (the two-per-CU form always ran two chains)
    try:
        layers._BLOCKED_CACHE.clear()
        b.block_plan().forget_csr()
        first = _run(conv, b, blocked=True)
        second = _run(conv, b, blocked=True)
    finally:
        layers.LAYER_VARIANT = keep
        layers._BLOCKED_CACHE.clear()
    csr._cache.clear()
    plain = _run(conv, b, blocked=False)
    for i, (f, s2, p) in enumerate(zip(first, second, plain)):
        assert torch.equal(f, s2), (i, 'store vs load')
        assert torch.equal(f, p), (i, (f - p).abs().max().item())
    ref = _oracle_scope(conv, b)
    for d in range(3):
        _gate(first[2 * d], ref[d][0], f'zincwide out_up[{d}]')
        _gate(first[2 * d + 1], ref[d][1], f'zincwide out_b[{d}]')


@pytest.mark.parametrize('F', [64, 128])
def test_blocked_layer_exact_on_integers(F):
    """Integer features and weights: every bf16 piece, product and partial sum is exact, so the
    kernel must reproduce the oracle's fp32 result exactly."""
    b = _batch('zinc', 50, F, seed=5, integer=True)
    conv = _conv(F, seed=6)
    with torch.no_grad():
        for lvl in conv.mp_levels:
            lin = lvl.msg_up_nn[1]
            lin.weight.copy_(torch.randint(-2, 3, lin.weight.shape).float())
            lin.bias.copy_(torch.randint(-2, 3, lin.bias.shape).float())
    outs = _run(conv, b, blocked=True)
    ref = _oracle_scope(conv, b, dtype=torch.float32)
    for d in range(3):
        assert torch.equal(cpu(outs[2 * d]), ref[d][0]), d
        assert torch.equal(cpu(outs[2 * d + 1]), ref[d][1]), d


def test_blocked_layer_entry_order_and_empty_parts():
    """Shuffled COO entries inside each complex give the same integers; complexes without rings
    (no 2-cells, empty upper adjacency of the edges) and a batch of one are handled."""
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    cs = zinc_like_complexes(9, 11, 6)
    b = ComplexBatch.from_complex_list(cs, max_dim=2).to(DEV)
    g = torch.Generator().manual_seed(0)
    xs = [torch.randint(-3, 4, (b.cochains[d].num_cells, 128), generator=g).float().to(DEV) for d in range(3)]
    b.set_xs(xs)
    conv = _conv(128, seed=8)
    with torch.no_grad():
        for lvl in conv.mp_levels:
            lin = lvl.msg_up_nn[1]
            lin.weight.copy_(torch.randint(-1, 2, lin.weight.shape).float())
            lin.bias.copy_(torch.randint(-2, 3, lin.bias.shape).float())
    ref = _run(conv, b, blocked=True)
    # shuffle entries within each complex's slice of every index
    b2 = ComplexBatch.from_complex_list(cs, max_dim=2).to(DEV)
    b2.set_xs(xs)
    rng = np.random.default_rng(0)
    for d in range(3):
        c = b2.cochains[d]
        for key, extra in (('upper_index', 'shared_coboundaries'), ('boundary_index', None)):
            idx = c[key]
            if idx is None:
                continue
            sl = c.__slices__[key]
            perm = np.concatenate([s + rng.permutation(e - s) for s, e in zip(sl[:-1], sl[1:])]) if len(sl) > 1 else []
            perm = torch.as_tensor(perm, dtype=torch.long, device=DEV)
            setattr(c, key, idx[:, perm].contiguous())
            if extra is not None:
                setattr(c, extra, c[extra][perm].contiguous())
    got = _run(conv, b2, blocked=True)
    for r, o in zip(ref, got):
        assert torch.equal(r, o)
    one = ComplexBatch.from_complex_list(cs[:1], max_dim=2).to(DEV)
    one.set_xs([x[:one.cochains[d].num_cells] for d, x in enumerate(xs)])
    o1 = _run(conv, one, blocked=True)
    for d in range(3):
        n = one.cochains[d].num_cells
        assert torch.equal(o1[2 * d], ref[2 * d][:n]) and torch.equal(o1[2 * d + 1], ref[2 * d + 1][:n])


def test_blocked_layer_reports_indices_that_leave_their_complex():
    from cwn_amd import csr
    b = _batch('zinc', 12, 128, seed=9)
    conv = _conv(128, seed=10)
    _run(conv, b, blocked=True)           # fine as built
    c = b.cochains[0]
    bad = c.upper_index.clone()
    bad[0, 3] = c.num_cells - 1           # source vertex of complex 0's entry -> last complex
    c.upper_index = bad
    b._block_plan = None                  # fresh plan object (unvalidated) for the edited batch
    with pytest.raises(IndexError, match='outside its complex'):
        _run(conv, b, blocked=True)
    csr.check_errors(DEV)                 # the sticky word was cleared by the raise


def test_blocked_layer_c_abi_argument_checks():
    from cwn_amd import _ffi
    L = _ffi.lib()
    assert 0 < L.cwn_layer_fused_lds_bytes(128, 96, 96) <= 160 * 1024
    assert 0 < L.cwn_layer_fused_lds_bytes(64, 192, 192) <= 160 * 1024
    assert L.cwn_layer_fused_lds_bytes(128, 112, 0) == 0 and L.cwn_layer_fused_lds_bytes(32, 16, 0) == 0
    arr = (_ffi.LayerDim * 1)()
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    items = torch.zeros(1, 32, dtype=torch.int32, device=DEV)
    s = _ffi.stream_ptr(torch.device(DEV))
    plan = _ffi.LayerPlan(items=items.data_ptr(), n_items=1, max_gemm_rows=16, max_source_rows=0)
    assert L.cwn_layer_fused_f32(arr, 1, 96, plan, 0, err.data_ptr(), s) == 1       # F
    assert L.cwn_layer_fused_f32(arr, 4, 128, plan, 0, err.data_ptr(), s) == 1      # n_dims
    assert L.cwn_layer_fused_f32(arr, 1, 128, plan, 3, err.data_ptr(), s) == 1      # store AND load
    assert L.cwn_layer_fused_f32(arr, 1, 128, plan, 1, err.data_ptr(), s) == 1      # store without a cache
    plan.cells_end[0] = 5                                                           # table names 5 cells, tensor has 0
    assert L.cwn_layer_fused_f32(arr, 1, 128, plan, 0, err.data_ptr(), s) == 1
    empty = _ffi.LayerPlan(n_items=0)
    assert L.cwn_layer_fused_f32(arr, 1, 128, empty, 0, err.data_ptr(), s) == 0     # nothing to do


def test_blocked_layer_csr_cache_store_and_load_are_bit_identical():
    """Layer 0 of a forward stores every item's sorted adjacency, later layers load it: same bits as
    sorting again; the cache is dropped when the index tensors change."""
    from cwn_amd import layers
    b = _batch('zinc', 77, 128, seed=21)
    conv = _conv(128, seed=22, eps=0.1)
    prev = layers.CSR_REUSE
    try:
        layers.CSR_REUSE = False
        ref = _run(conv, b, blocked=True)
        layers.CSR_REUSE = True
        first = _run(conv, b, blocked=True)       # stores
        table = next(t for t in b.block_plan()._tables.values() if t is not None)
        assert table.csr_key is not None
        again = _run(conv, b, blocked=True)       # loads
        for r, f, a in zip(ref, first, again):
            assert torch.equal(r, f) and torch.equal(r, a)
        # new index tensors (same values): the key changes, the launch sorts and stores again
        c = b.cochains[1]
        c.boundary_index = c.boundary_index.clone()
        key_before = table.csr_key
        third = _run(conv, b, blocked=True)
        assert table.csr_key != key_before
        for r, t in zip(ref, third):
            assert torch.equal(r, t)
    finally:
        layers.CSR_REUSE = prev


def test_blocked_layer_falls_back_when_a_complex_exceeds_one_workgroup():
    """molhiv-like complexes of up to 60 atoms do not fit the F = 128 row cap: the layer must take
    the CSR path (and still match the oracle)."""
    b = _batch('molhiv', 40, 128, seed=12)
    conv = _conv(128, seed=13)
    with torch.no_grad():
        plans, outs = conv.propagate_all(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
    assert plans[0] != 'blocked'
    ref = _oracle_scope(conv, b)
    for d in range(3):
        _gate(outs[2 * d], ref[d][0], f'fallback out_up[{d}]')
        _gate(outs[2 * d + 1], ref[d][1], f'fallback out_b[{d}]')


def test_one_captured_graph_serves_batches_of_different_shapes():
    """cwn_amd/static_graph.py: the propagate scope of a 3-layer stack captured ONCE over capacity-sized
    buffers, replayed for batches of 100, 128, 77 and 5 complexes (different cell, entry and item counts)
    -- every output against the float64 oracle, and bit-identical to the eager per-batch launch."""
    from cwn_amd import csr
    from cwn_amd.static_graph import StaticPropagate
    F, L = 128, 3
    convs = [_conv(F, seed=30 + l, eps=0.1 * l) for l in range(L)]
    batches = [_batch('zinc', n, F, seed=40 + i) for i, n in enumerate((100, 128, 77, 5))]
    g = torch.Generator().manual_seed(5)
    feats = [[[torch.randn(b.cochains[d].num_cells, F, generator=g).to(DEV) for d in range(3)] for _ in range(L)]
             for b in batches]
    from cwn_amd.synthetic import batch_stats
    st = [batch_stats(b) for b in batches]
    sp = StaticPropagate(convs, F, cap_cells=[max(s[f'N{d}'] for s in st) + 7 for d in range(3)],
                         cap_up=[max(s[f'E_up{d}'] for s in st) + 3 for d in range(2)],
                         cap_b=[max(s[f'B{d}'] for s in st) + 5 for d in (1, 2)], cap_items=[130, 131], device=DEV)
    graphs_seen = set()
    for order in ((0, 1, 2, 3), (3, 1)):
        for i in order:
            b = batches[i]
            sp.load(b, feats[i])
            outs = sp.replay()
            graphs_seen.add(id(sp.graph))
            for l in range(L):
                b.set_xs(feats[i][l])
                ref = _oracle_scope(convs[l], b)
                eager = _run(convs[l], b, blocked=True)
                for d in range(3):
                    _gate(outs[l][2 * d], ref[d][0], f'static graph batch {i} layer {l} out_up[{d}]')
                    _gate(outs[l][2 * d + 1], ref[d][1], f'static graph batch {i} layer {l} out_b[{d}]')
                    assert torch.equal(outs[l][2 * d], eager[2 * d]) and torch.equal(outs[l][2 * d + 1], eager[2 * d + 1])
    csr.check_errors(DEV)
    assert len(graphs_seen) == 1                          # captured once
    with pytest.raises(ValueError, match='exceed the capacity'):
        sp.load(_batch('zinc', 140, F, seed=50), feats[0])


def test_layers_with_prepared_launches_still_deepcopy_and_pickle():
    import copy
    import pickle
    b = _batch('zinc', 6, 128, seed=70)
    conv = _conv(128, seed=71)
    ref = _run(conv, b, blocked=True)
    twin = copy.deepcopy(conv)
    pickle.loads(pickle.dumps(conv.state_dict()))
    for r, o in zip(ref, _run(twin, b, blocked=True)):
        assert torch.equal(r, o)


def test_blocked_kernel_refuses_records_that_do_not_fit_its_lds():
    """The derived record fields come from the table builder and are validated on the HOST
    (cwn_layer_items_check); what the KERNEL still checks is what keeps a workgroup inside its LDS.  A device
    table tampered with after the host check (staged rows beyond the launch's cap; entries beyond the
    scratch; a first coface row behind the staged rows) must set the sticky error bit, leave the item's
    rows unwritten and not fault -- and the other items are still correct."""
    from cwn_amd import _ffi, csr, ops
    b = _batch('zinc', 6, 128, seed=21)
    conv = _conv(128, seed=22)
    good = [o.clone() for o in _run(conv, b, blocked=True)]
    params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
    with torch.no_grad():
        dims, plan, table, key = conv._blocked_args(params, 0)
    keep = table.items.clone()
    for col, val in ((24, 4096), (27, 2048), (23, 4000)):
        table.items.copy_(keep)
        table.items[1, col] = val                      # item 1 only
        table.csr_key = None
        with torch.no_grad():
            outs = ops.layer_fused(dims, table, 0)
        with pytest.raises(IndexError):
            csr.check_errors(DEV)
        it = keep[1].tolist()
        r0, n = it[10], it[11]                          # task 0 of item 1: its rows are not trusted
        mask = torch.ones(outs[0].size(0), dtype=torch.bool, device=DEV)
        if it[0] >> 8 == 0:
            mask[r0:r0 + n] = False
            assert torch.equal(outs[0][mask], good[0][mask]) and torch.equal(outs[1][mask], good[1][mask])
        for k in range(2, 6):                           # the other dimensions' items are untouched
            assert torch.equal(outs[k], good[k])
    table.items.copy_(keep)
    table.csr_key = None
    with torch.no_grad():
        outs = ops.layer_fused(dims, table, 0)
    csr.check_errors(DEV)
    for o, g in zip(outs, good):
        assert torch.equal(o, g)


@pytest.mark.parametrize('seed', range(6))
def test_blocked_layer_random_structures_vs_oracle(seed):
    """Differential test over structures the molecular generator does not produce: small clique complexes
    (triangles as 2-cells: every edge of a triangle is upper-adjacent to the two others through it),
    ring complexes of odd sizes, single bonds, isolated vertices, and DUPLICATED index entries (summed
    with multiplicity, mp/test_cell_mp.py:179-269) -- integer-valued features and weights, so the blocked
    kernel must reproduce the oracle exactly, in both CSR modes."""
    from cwn_amd import lifting
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import random_molecule
    rng = np.random.default_rng(1000 + seed)
    cs = []
    for _ in range(int(rng.integers(3, 9))):
        kind = int(rng.integers(0, 4))
        if kind == 0:                                         # random graph, clique lift (triangles)
            n = int(rng.integers(3, 14))
            p = float(rng.uniform(0.2, 0.6))
            edges = [(i, j) for i in range(n) for j in range(i + 1, n) if rng.random() < p] or [(0, 1)]
            cs.append(lifting.clique_lift(n, edges, torch.zeros(n, 1), max_dim=2, y=torch.zeros(1)))
        elif kind == 1:                                       # molecule, rings up to 8
            n, bonds = random_molecule(rng, 5, 20)
            cs.append(lifting.ring_lift(n, bonds, torch.zeros(n, 1), torch.zeros(len(bonds), 1), max_k=8,
                                        y=torch.zeros(1)))
        elif kind == 2:                                       # a path (no 2-cells), possibly one bond
            n = int(rng.integers(2, 7))
            bonds = [(i, i + 1) for i in range(n - 1)]
            cs.append(lifting.ring_lift(n, bonds, torch.zeros(n, 1), torch.zeros(len(bonds), 1), max_k=6,
                                        y=torch.zeros(1)))
        else:                                                 # one big ring
            n = int(rng.integers(3, 9))
            bonds = sorted((min(i, (i + 1) % n), max(i, (i + 1) % n)) for i in range(n))
            cs.append(lifting.ring_lift(n, bonds, torch.zeros(n, 1), torch.zeros(n, 1), max_k=8, y=torch.zeros(1)))
    if all(c.dimension < 2 for c in cs):
        n, bonds = 3, [(0, 1), (0, 2), (1, 2)]
        cs.append(lifting.ring_lift(n, bonds, torch.zeros(n, 1), torch.zeros(3, 1), max_k=6, y=torch.zeros(1)))
    b = ComplexBatch.from_complex_list(cs, max_dim=2)
    # duplicate a few entries inside their complex's slice (kept grouped: the table is per complex)
    for d in range(2):
        c = b.cochains[d]
        if c.upper_index is None or c.upper_index.size(1) == 0:
            continue
        sl = list(c.__slices__['upper_index'])
        cols, new_sl = [], [0]
        for s0, e0 in zip(sl[:-1], sl[1:]):
            idx = list(range(s0, e0))
            if idx and rng.random() < 0.5:
                idx += [int(rng.choice(idx))] * int(rng.integers(1, 3))
            cols += idx
            new_sl.append(len(cols))
        perm = torch.tensor(cols, dtype=torch.long)
        c.upper_index = c.upper_index[:, perm].contiguous()
        c.shared_coboundaries = c.shared_coboundaries[perm].contiguous()
        c.__slices__['upper_index'] = new_sl
        if 'shared_coboundaries' in c.__slices__:
            c.__slices__['shared_coboundaries'] = new_sl
    b = b.to(DEV)
    F = 128 if seed % 2 == 0 else 64
    g = torch.Generator().manual_seed(seed)
    for d in range(3):
        b.cochains[d].x = torch.randint(-3, 4, (b.cochains[d].num_cells, F), generator=g).float().to(DEV)
    conv = _conv(F, seed=seed, eps=0.0)
    with torch.no_grad():
        for lvl in conv.mp_levels:
            lin = lvl.msg_up_nn[1]
            lin.weight.copy_(torch.randint(-1, 2, lin.weight.shape).float())
            lin.bias.copy_(torch.randint(-2, 3, lin.bias.shape).float())
    ref = _oracle_scope(conv, b, dtype=torch.float32)
    for _ in range(2):                                        # first call sorts and stores, second loads
        outs = _run(conv, b, blocked=True)
        for d in range(3):
            assert torch.equal(cpu(outs[2 * d]), ref[d][0]), (seed, d, 'up')
            assert torch.equal(cpu(outs[2 * d + 1]), ref[d][1]), (seed, d, 'boundary')
    assert conv.blocked_reason is None


@pytest.mark.parametrize('off', ['params', 'level'])
def test_blocked_layer_without_the_boundary_stream(off):
    """a legal reference configuration -- get_all_cochain_params(include_boundary_features=False), or a
    level built with use_boundary_msg=False -- runs the blocked kernel with a table that carries no boundary
    entries (it used to raise CWN_ERR_BAD_ARG from the launcher), bit-identical to the CSR path; out_b is the
    self term alone (mp/cell_mp.py:380-382, :517-522)."""
    from cwn_amd import csr, layers
    b = _batch('zinc', 40, 128, seed=11)
    conv = _conv(128, seed=12, eps=0.125)
    if off == 'level':
        for lvl in conv.mp_levels:
            lvl.use_boundary_msg = False
    kw = dict(max_dim=2, include_down_features=False, include_boundary_features=(off != 'params'))
    res = {}
    for blocked in (True, False):
        prev, layers.BLOCKED_LAYER = layers.BLOCKED_LAYER, blocked
        try:
            csr._cache.clear()
            with torch.no_grad():
                plans, outs = conv.propagate_all(*b.get_all_cochain_params(**kw))
            assert (plans[0] == 'blocked') == blocked, conv.blocked_reason
        finally:
            layers.BLOCKED_LAYER = prev
        res[blocked] = outs
    for i, (f, p) in enumerate(zip(res[True], res[False])):
        assert torch.equal(f, p), (i, (f - p).abs().max().item())
    for d in range(3):
        x = b.cochains[d].x
        assert torch.equal(res[True][2 * d + 1], (1 + float(conv.mp_levels[d].eps2)) * x)


def test_forget_plans_drops_the_block_plan():
    """a batch object refilled with new index tensors gets a fresh per-complex table (validated again)."""
    b = _batch('zinc', 8, 128, seed=13)
    p0 = b.block_plan()
    p0.validated = True
    b.forget_plans()
    p1 = b.block_plan()
    assert p1 is not p0 and not p1.validated


def test_static_graph_at_width_64():
    """StaticPropagate at F = 64 (it asked for 186 KB of LDS and failed at the first replay): one graph
    over two batches, bit-identical to per-batch launches."""
    from cwn_amd.static_graph import StaticPropagate
    F = 64
    conv = _conv(F, seed=21, eps=0.5)
    bs = [_batch('zinc', n, F, seed=22 + n) for n in (40, 25)]
    caps = [max(b.cochains[d].num_cells for b in bs) + 8 for d in range(3)]
    sp = StaticPropagate([conv], F, caps, [max(b.cochains[d].upper_index.size(1) for b in bs) + 8 for d in range(2)],
                         [max(b.cochains[d].boundary_index.size(1) for b in bs) + 8 for d in (1, 2)], [48, 48], DEV)
    for b in bs:
        sp.load(b, [[b.cochains[d].x for d in range(3)]])
        got = sp.replay()[0]
        want = _run(conv, b, blocked=True)
        for i, (g, w) in enumerate(zip(got, want)):
            assert torch.equal(g, w), (i, (g - w).abs().max().item())


@pytest.mark.parametrize('kind,n,F', [('zinc', 128, 128), ('zinc', 300, 128), ('zinc', 1, 128), ('zinc', 64, 64), ('zinc', 700, 64)])
def test_two_per_cu_form_bit_identical_to_the_16_wave_form(kind, n, F):
    """cwn_layer_plan.variant = 1 (8 waves, <= 128 VGPRs, <= 80 KiB LDS: two workgroups per CU; a wave multiplies both
    products one after the other through one register set, the rings' boundary sources are read out of the bf16
    planes, every item lays out its own rows) against variant 0: same split, same MFMA order per tile, same entry
    order -> every output bit agrees; sort, sort + store and load modes."""
    from cwn_amd import layers
    b = _batch(kind, n, F, seed=31)
    conv = _conv(F, seed=32, eps=0.375)
    outs = {}
    prev = layers.LAYER_VARIANT
    try:
        for v in ('0', '1'):
            layers.LAYER_VARIANT = v
            layers._BLOCKED_CACHE.clear()
            b.block_plan().forget_csr()
            first = _run(conv, b, blocked=True)          # sorts (and stores the per-item CSR)
            second = _run(conv, b, blocked=True)         # loads it back
            with torch.no_grad():
                prm = b.get_all_cochain_params(max_dim=2, include_down_features=False)
                table = conv._blocked_args(prm, 0)[2]
            assert table.variant == int(v), (v, table.variant)
            outs[v] = (first, second)
    finally:
        layers.LAYER_VARIANT = prev
        layers._BLOCKED_CACHE.clear()
    for i in range(6):
        assert torch.equal(outs['0'][0][i], outs['1'][0][i]), (i, (outs['0'][0][i] - outs['1'][0][i]).abs().max().item())
        assert torch.equal(outs['1'][0][i], outs['1'][1][i]), i


def test_auto_variant_takes_the_two_per_cu_form_beyond_one_item_per_cu():
    from cwn_amd import layers
    conv = _conv(128, seed=33)
    assert layers.LAYER_VARIANT == 'auto'
    small, big = _batch('zinc', 100, 128, seed=34), _batch('zinc', 400, 128, seed=35)
    with torch.no_grad():
        t_small = conv._blocked_args(small.get_all_cochain_params(max_dim=2, include_down_features=False), 0)[2]
        t_big = conv._blocked_args(big.get_all_cochain_params(max_dim=2, include_down_features=False), 0)[2]
    assert t_small.variant == 0 and t_small.n_items <= layers.TWO_PER_CU_MIN_ITEMS
    assert t_big.variant == 1 and t_big.n_items > layers.TWO_PER_CU_MIN_ITEMS
    ref = _oracle_scope(conv, big)
    outs = _run(conv, big, blocked=True)
    for d in range(3):
        _gate(outs[2 * d], ref[d][0], f'two-per-CU out_up[{d}]')
        _gate(outs[2 * d + 1], ref[d][1], f'two-per-CU out_b[{d}]')


@pytest.mark.parametrize('F,n_small,big_sizes', [(128, 30, [(70, 90)]), (128, 100, [(50, 60), (150, 200)]), (64, 40, [(130, 200)])])
def test_big_complexes_are_streamed_inside_the_blocked_launch(F, n_small, big_sizes):
    """a complex beyond a workgroup's LDS no longer sends its whole batch to the two-kernel path --
    its workgroup streams it (BIG records, include/cwn_hip.h) while the rest stays blocked, in the same launch:
    bit-identical to the two-kernel path on every row of the mixed batch, and inside the gate against the oracle."""
    from cwn_amd import csr, layers
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    cxs = zinc_like_complexes(n_small, 41, 6)
    for k, (lo, hi) in enumerate(big_sizes):
        giants = zinc_like_complexes(2, 42 + k, 6, n_lo=lo, n_hi=hi)
        cxs = cxs[:7 * (k + 1)] + giants[:1] + cxs[7 * (k + 1):] + giants[1:]        # in the middle and at the end
    b = ComplexBatch.from_complex_list(cxs, max_dim=2).to(DEV)
    g = torch.Generator().manual_seed(43)
    for d in range(3):
        b.cochains[d].x = torch.randn(b.cochains[d].num_cells, F, generator=g).to(DEV)
    conv = _conv(F, seed=44, eps=0.3)
    keep_policy, layers.BIG_ITEMS = layers.BIG_ITEMS, 'always'      # (the cost model would send the largest ones of these
    try:                                                            # small batches to the two-kernel path: _streaming_pays)
        with torch.no_grad():
            table = conv._blocked_args(b.get_all_cochain_params(max_dim=2, include_down_features=False), 0)[2]
        assert table.variant == 0 and 2 <= table.n_big <= 4 * len(big_sizes), (table.n_big, table.n_items)
        first = _run(conv, b, blocked=True)
        second = _run(conv, b, blocked=True)                  # the layers after the first load the cached per-item CSR
    finally:
        layers.BIG_ITEMS = keep_policy
    csr._cache.clear()
    plain = _run(conv, b, blocked=False)
    for i, (f, s2, p) in enumerate(zip(first, second, plain)):
        assert torch.equal(s2, f), i
        if F == 128:      # (at width 64 the two-kernel path multiplies on fp32 MFMA, not through the bf16 split: not bit-comparable)
            assert torch.equal(f, p), (i, (f - p).abs().max().item())
    ref = _oracle_scope(conv, b)
    for d in range(3):
        _gate(first[2 * d], ref[d][0], f'big items F={F} out_up[{d}]')
        _gate(first[2 * d + 1], ref[d][1], f'big items F={F} out_b[{d}]')
    # the policy: a giant of 30 us in a one-round launch of 9 us does not pay, the same giant in a larger batch does
    big_rows = max(int(r[11]) + int(r[5]) for r in table.big_records)
    assert layers._streaming_pays(table, F) == ((0.07 if F == 128 else 0.035) * big_rows < 1.9 * 8.7 * max(1, -(-(table.n_items - table.n_big) // 256)))
    # switched off: the whole batch takes the two-kernel path, as before
    prev, layers.BIG_ITEMS = layers.BIG_ITEMS, False
    try:
        layers._BLOCKED_CACHE.clear()
        with torch.no_grad():
            assert isinstance(conv._blocked_args(b.get_all_cochain_params(max_dim=2, include_down_features=False), 0), str)
    finally:
        layers.BIG_ITEMS = prev
        layers._BLOCKED_CACHE.clear()


def test_mixed_launch_two_per_cu_form_plus_16_wave_form():
    """A batch with more items than CUs in which some molecules are too large for the two-per-CU form (its 80 KiB hold
    ~30 atoms at width 128): that form serves the complexes that fit, the 16-wave form (BIG records for the ones beyond
    ITS caps) the rest -- two launches over complementary complexes into the same outputs (blockplan.MixedTable),
    bit-identical to the two-kernel path."""
    from cwn_amd import csr, layers
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    F = 128
    small, large = zinc_like_complexes(300, 51, 6), zinc_like_complexes(24, 52, 6, n_lo=28, n_hi=44)
    cxs = []
    for i, c in enumerate(small):
        cxs.append(c)
        if i % 13 == 5 and large:
            cxs.append(large.pop())
    cxs += large
    b = ComplexBatch.from_complex_list(cxs, max_dim=2).to(DEV)
    g = torch.Generator().manual_seed(53)
    for d in range(3):
        b.cochains[d].x = torch.randn(b.cochains[d].num_cells, F, generator=g).to(DEV)
    conv = _conv(F, seed=54, eps=0.2)
    prev_v, prev_b = layers.LAYER_VARIANT, layers.BIG_ITEMS
    layers.LAYER_VARIANT, layers.BIG_ITEMS = 'mixed', 'always'
    try:
        layers._BLOCKED_CACHE.clear()
        with torch.no_grad():
            table = conv._blocked_args(b.get_all_cochain_params(max_dim=2, include_down_features=False), 0)[2]
        assert table.variant == 'mixed' and [t.variant for t in table.parts] == [1, 0]
        assert table.parts[0].n_items > 256 and 0 < table.n_rest < 24 + 10 and table.parts[1].n_items == 2 * table.n_rest
        first = _run(conv, b, blocked=True)
        second = _run(conv, b, blocked=True)
    finally:
        layers.LAYER_VARIANT, layers.BIG_ITEMS = prev_v, prev_b
        layers._BLOCKED_CACHE.clear()
    csr._cache.clear()
    plain = _run(conv, b, blocked=False)
    for i, (f, s2, p) in enumerate(zip(first, second, plain)):
        assert torch.equal(f, p), (i, (f - p).abs().max().item())
        assert torch.equal(s2, p), i


@pytest.mark.parametrize('kind,n,F,blocked_bwd', [('zinc', 64, 128, 0), ('zinc', 300, 128, 0), ('molhiv', 96, 64, 0),
                                                  ('zinc', 64, 128, 1), ('zinc', 20, 64, 1),
                                                  ('zinc', 64, 128, 2), ('zinc', 300, 128, 2), ('molhiv', 96, 64, 2)])
def test_training_forward_through_the_blocked_kernel(kind, n, F, blocked_bwd):
    """CWN_LAYER_STORE_Y: with autograd on, the propagate step runs as the blocked launch, leaves Y1 / Y2 for the backward
    pass (ops._GemmAggregate) and gives the gradients of the two-kernel path.  At F = 128 outputs and the stored products
    are bit-identical to that path (same split, same MFMA order); at F = 64 the two-kernel path multiplies on fp32 MFMA."""
    from cwn_amd import layers, ops
    b = _batch(kind, n, F, seed=5)
    conv = _conv(F, seed=6, eps=0.5).train()
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(b.cochains[d].num_cells, F, generator=g).to(DEV) for d in range(3)]
    captured = {}
    orig = ops.gemm_aggregate

    def spy(specs, make_streams, precomputed=None):
        captured['pre'] = precomputed
        return orig(specs, make_streams, precomputed=precomputed)

    prev_bwd = ops.BLOCKED_BACKWARD

    def run(flag):
        layers.BLOCKED_TRAIN_FORWARD = flag
        ops.BLOCKED_BACKWARD = blocked_bwd if flag else 0      # (the one-launch backward: 1 atomic form, 2 owner form)
        ops.gemm_aggregate = spy
        if blocked_bwd and flag:
            ops.pack_layer_weights_many([conv.mp_levels[d].msg_up_nn[1].weight for d in range(2)], transposed=True)
        try:
            conv.zero_grad(set_to_none=True)
            xin = [b.cochains[d].x.detach().clone().requires_grad_() for d in range(3)]
            b.set_xs(xin)
            plans, outs = conv.propagate_all(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
            sum((o * torch.cat([w, w])[:o.size(0)]).sum() for o, w in zip(outs, [w for w in ws for _ in range(2)])).backward()
            return outs, xin, {k: v.grad.clone() for k, v in conv.named_parameters() if v.grad is not None}, captured.get('pre')
        finally:
            layers.BLOCKED_TRAIN_FORWARD = True
            ops.BLOCKED_BACKWARD = prev_bwd
            ops.gemm_aggregate = orig

    before = list(ops.BLOCKED_BACKWARD_LAUNCHES)
    outs1, x1, g1, pre1 = run(True)
    if blocked_bwd:
        assert ops.BLOCKED_BACKWARD_LAUNCHES[blocked_bwd - 1] == before[blocked_bwd - 1] + 1, 'the blocked backward did not run'
    outs0, x0, g0, pre0 = run(False)
    assert pre1 is not None and pre0 is None, 'the training forward did not take the blocked kernel'
    for a, c in zip(outs1, outs0):
        if F == 128:
            assert torch.equal(a, c)
        else:
            torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-5 * max(1.0, float(c.abs().max())))
    for a, c in zip(x1, x0):
        torch.testing.assert_close(a.grad, c.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(c.grad.abs().max())))
    assert g1.keys() == g0.keys() and g1
    for k in g0:
        torch.testing.assert_close(g1[k], g0[k], rtol=1e-4, atol=1e-4 * max(1.0, float(g0[k].abs().max())), msg=k)
    if F == 128:      # the stored products are the grouped GEMM's, bit for bit
        ys = pre1[0]
        lin0 = conv.mp_levels[0].msg_up_nn[1]
        ref, = ops.run_gemm([ops.Gemm(X=b.cochains[0].x.detach(), W=lin0.weight.detach(), w_col0=0, bias=lin0.bias.detach())], DEV)
        assert torch.equal(ys[0], ref)


def test_item_tables_survive_forget_plans_when_the_structure_is_the_same():
    """A training step calls forget_plans() inside its captured graph and then runs the blocked forward: the item tables
    (ranges per complex) are carried over to the fresh plan, their per-item CSR caches are not."""
    b = _batch('zinc', 16, 128, seed=21)
    conv = _conv(128, seed=1)
    _run(conv, b, blocked=True)
    p0 = b.block_plan()
    t0 = dict(p0._tables)
    assert t0 and all(t is None or t.csr_key is not None for t in t0.values())
    b.forget_plans()
    p1 = b.block_plan()
    assert p1 is not p0 and p1._tables.keys() == t0.keys() and all(p1._tables[k] is t0[k] for k in t0)
    assert all(t is None or t.csr_key is None for t in p1._tables.values()) and not p1.validated
    out1 = _run(conv, b, blocked=True)
    ref = _oracle_scope(conv, b)
    for d in range(3):
        _gate(out1[2 * d], ref[d][0], f'out_up[{d}]')


def test_packing_many_layer_weights_at_once_equals_one_by_one():
    from cwn_amd import ops
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(F, 2 * F, device=DEV)) for F in (128, 64, 128, 128, 64)]
    ops.pack_layer_weights_many(ws)
    many = [ops.pack_layer_weight(w, fresh=True).clone() for w in ws]           # fresh entries of the latest batch: cache hits
    one = [ops.pack_layer_weight(torch.nn.Parameter(w.detach().clone())) for w in ws]
    for a, c in zip(many, one):
        assert torch.equal(a, c)
    ops.pack_layer_weights_many(ws[:1])                                         # a later batch: the other entries are stale
    again = ops.pack_layer_weight(ws[1], fresh=True)
    assert torch.equal(again, one[1])


def _propagate_reference_backward(conv, b, gs, F):
    """float64 autograd through a plain restatement of the propagate step (mp/layers.py:184-192, 290-295): per dimension
    the gradients of x, Y1, Y2 for output gradients gs = [gU_0, gB_0, gU_1, ...]."""
    xs = [cpu(b.cochains[d].x).double().requires_grad_() for d in range(3)]
    Y1, Y2, loss = [None] * 3, [None] * 3, 0.0
    for d in range(3):
        c = b.cochains[d]
        lvl = conv.mp_levels[d]
        n = xs[d].size(0)
        e1, e2 = float(lvl.eps1), float(lvl.eps2)
        out_up = (1 + e1) * xs[d]
        if d + 1 < 3 and c.upper_index is not None and c.upper_index.size(1):
            W = cpu(lvl.msg_up_nn[1].weight).double()
            bias = cpu(lvl.msg_up_nn[1].bias).double()
            Y1[d] = xs[d] @ W[:, :F].t() + bias
            Y2[d + 1] = xs[d + 1] @ W[:, F:].t()
            Y1[d].retain_grad()
            Y2[d + 1].retain_grad()
            ui, sh = cpu(c.upper_index), cpu(c.shared_coboundaries)
            msg = torch.relu(Y1[d][ui[0]] + Y2[d + 1][sh])
            out_up = out_up + torch.zeros(n, F, dtype=torch.float64).index_add(0, ui[1], msg)
        out_b = (1 + e2) * xs[d]
        if d > 0 and c.boundary_index is not None and c.boundary_index.size(1):
            bi = cpu(c.boundary_index)
            out_b = out_b + torch.zeros(n, F, dtype=torch.float64).index_add(0, bi[1], xs[d - 1][bi[0]])
        loss = loss + (out_up * cpu(gs[2 * d]).double()).sum() + (out_b * cpu(gs[2 * d + 1]).double()).sum()
    loss.backward()
    return [x.grad for x in xs], [None if y is None else y.grad for y in Y1], [None if y is None else y.grad for y in Y2]


@pytest.mark.parametrize('form', ['atomic', 'own'])
@pytest.mark.parametrize('kind,n,F,eps', [('zinc', 64, 128, 0.0), ('zinc', 128, 128, 0.3), ('zinc', 9, 128, 0.0), ('zinc', 40, 64, 0.2),
                                          ('zinc', 300, 128, 0.1), ('molhiv', 200, 64, 0.0), ('zinctrees', 48, 128, 0.2)])
def test_blocked_backward_launch_vs_float64_autograd(kind, n, F, eps, form):
    """cwn_layer_bwd_f32 (ops.layer_backward) over the item table of the forward launch, and cwn_layer_bwd_own_f32 over
    the owner table (one writer per row; dx is handed over UNINITIALISED -- filled with NaN here): dx of every dimension
    and the gradients of the stored products against float64 autograd of the plain restatement."""
    from cwn_amd import layers, ops, _ffi
    b = _batch(kind, n, F, seed=31)
    conv = _conv(F, seed=32, eps=eps).train()
    params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
    args = conv._blocked_args(params, 0, training=True)
    assert not isinstance(args, str), args
    dims, plan, table, key = args
    if table.variant == 'mixed' or table.n_big:
        pytest.skip('the forward of this batch is not one plain launch')
    if form == 'atomic' and table.variant != 0:
        pytest.skip('the atomic form runs over the 16-wave table only')
    bwd_table = None
    if form == 'own':
        bwd_table = plan.bwd_items(F, [D.up_index is not None and D.up_index.size(1) > 0 for D in dims], [D.b_index is not None for D in dims])
        assert bwd_table is not None
    rows = [int(D.x.size(0)) for D in dims]
    ys_of = [[None, None] for _ in range(3)]
    for d in range(2):
        if dims[d].up_index is not None and dims[d].up_index.size(1):
            ys_of[d][0] = torch.empty(rows[d], F, device=DEV)
            ys_of[d + 1][1] = torch.empty(rows[d + 1], F, device=DEV)
    outs = ops.LayerLaunch(dims, table).run([D.x for D in dims], 0, ys=[tuple(p) for p in ys_of])
    g = torch.Generator().manual_seed(5)
    gs = [torch.randn(o.shape, generator=g).to(DEV) for o in outs]
    gs[3] = None if n == 9 else gs[3]                                   # an output nobody differentiates: a NULL gradient
    ws = [conv.mp_levels[d].msg_up_nn[1].weight for d in range(2)]
    ops.pack_layer_weights_many(ws, transposed=True)
    wt_of = [ops.packed_layer_weight_t(ws[0]), ops.packed_layer_weight_t(ws[1]), None]
    if form == 'own':                  # whatever the allocator hands out next: NaN (the launch must write every row)
        junk = torch.full((sum(rows) * 3, F), float('nan'), device=DEV)
        del junk
    got = ops.layer_backward(dims, table, [tuple(p) for p in ys_of], [(gs[2 * d], gs[2 * d + 1]) for d in range(3)], wt_of,
                             bwd_table=bwd_table)
    assert got is not None
    from cwn_amd import csr
    csr.check_errors(DEV)
    dxs, gys = got
    ref_gs = [torch.zeros_like(outs[k]) if x is None else x for k, x in enumerate(gs)]
    dx_ref, gy1_ref, gy2_ref = _propagate_reference_backward(conv, b, ref_gs, F)
    for d in range(3):
        _gate(dxs[d], dx_ref[d], f'dx[{d}]')
        if gy1_ref[d] is not None:
            _gate(gys[d][0], gy1_ref[d], f'gY1[{d}]')
        if gy2_ref[d] is not None:
            _gate(gys[d][1], gy2_ref[d], f'gY2[{d}]')


def _macrocycle_complexes(n, seed, max_ring):
    """Molecules the ZINC generator does not make: a large ring of 9 .. 16 atoms with a few side chains and a fused hexagon,
    ring-lifted with max_ring (exp/scripts/cwn-zinc.sh:24 lifts ZINC with max_ring_size 18; SURVEY 8d's generator stops at 6)."""
    import numpy as np
    from cwn_amd.synthetic import ring_lift
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(rng.integers(9, 17))
        bonds = [(i, (i + 1) % k) for i in range(k)]
        m = k
        a = int(rng.integers(0, k))                       # a hexagon fused on the bond (a, a + 1)
        chain = [a] + list(range(m, m + 4)) + [(a + 1) % k]
        bonds += [(chain[i], chain[i + 1]) for i in range(5)]
        m += 4
        for _ in range(int(rng.integers(0, 4))):          # side chains
            bonds.append((int(rng.integers(0, m)), m))
            m += 1
        bonds = sorted({(min(u, v), max(u, v)) for u, v in bonds})
        vx = torch.from_numpy(rng.integers(0, 28, size=(m, 1))).float()
        ex = torch.from_numpy(rng.integers(0, 4, size=(len(bonds), 1))).float()
        out.append(ring_lift(m, bonds, vx, ex, max_k=max_ring))
    return out


@pytest.mark.parametrize('F', [128, 64])
def test_blocked_layer_on_rings_up_to_eighteen(F):
    """exp/scripts/cwn-zinc.sh:24 lifts with max_ring_size 18: 2-cells with up to 16 boundary edges here (a ring of k edges
    gives k (k - 1) upper-adjacency entries among them: 240 for one cell), through the blocked launch, the two-kernel path
    and the float64 oracle -- untimed, parity only."""
    from cwn_amd.complex import ComplexBatch
    cxs = _macrocycle_complexes(24, 3, 18)
    assert max(int(c.cochains[2].boundary_index[1].bincount().max()) for c in cxs) >= 12      # large rings really are cells
    b = ComplexBatch.from_complex_list(cxs, max_dim=2).to(DEV)
    g = torch.Generator().manual_seed(5)
    for d in range(3):
        b.cochains[d].x = torch.randn(b.cochains[d].num_cells, F, generator=g).to(DEV)
    conv = _conv(F, seed=8, eps=0.2)
    ref = _oracle_scope(conv, b)
    blocked, streamed = _run(conv, b, blocked=True), _run(conv, b, blocked=False)
    for d in range(3):
        _gate(blocked[2 * d], ref[d][0], f'max_ring 18, F = {F}: blocked out_up[{d}]')
        _gate(blocked[2 * d + 1], ref[d][1], f'max_ring 18, F = {F}: blocked out_b[{d}]')
        if F == 128:
            assert torch.equal(blocked[2 * d], streamed[2 * d]) and torch.equal(blocked[2 * d + 1], streamed[2 * d + 1])
        else:
            _gate(streamed[2 * d], ref[d][0], f'max_ring 18, F = {F}: two-kernel out_up[{d}]')


def test_blocked_training_step_builds_no_plans_of_the_upper_adjacencies():
    """Round 6: a training step whose forward runs as the blocked launch and whose backward is the owner-form launch reads no CSR
    plan of an upper adjacency -- the streams that describe the step to autograd get their plans UNBUILT (csr.deferred_builds)
    and nobody asks for them: one batched build per step (the boundary adjacencies, Complex.prepare(upper=False)), where there
    were three.  A reader that does want such a plan builds it on demand: the same batch through the streaming path afterwards."""
    from cwn_amd import csr, layers
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.synthetic import zinc_like_complexes
    from cwn_amd.train import TrainStep
    torch.manual_seed(0)
    model = EmbedSparseCIN(28, 4, 1, 2, 64, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(DEV)
    batches = [ComplexBatch.from_complex_list(zinc_like_complexes(24, i, 6), max_dim=2).to(DEV) for i in range(2)]
    for b in batches:
        b.y = torch.zeros(b.num_complexes, 1, device=DEV)
    ts = TrainStep(model, batches, task_type='regression', use_graph=True)
    for i in range(3):
        ts.step(i % 2)            # (the capture's warm-up settles _skip_upper_plans)
    assert ts._skip_upper_plans
    ts2 = TrainStep(model, batches, task_type='regression', use_graph=False)
    ts2._skip_upper_plans = True
    ts2.step(0)
    calls, orig = [], csr.build_many

    def traced(adjs, *a, **k):
        adjs = [x for x in adjs if k.get('force') or not x.built]
        if adjs:
            calls.append([(x.n_dst, x.n_entries) for x in adjs])
        return orig(adjs, *a, **k)
    csr.build_many = traced
    import cwn_amd.complex as cx_mod
    try:
        loss = ts2.step(1)
        torch.cuda.synchronize()
    finally:
        csr.build_many = orig
    assert len(calls) == 1, calls            # Complex.prepare(backward=True, upper=False): the boundary adjacencies (+ transposes)
    assert torch.isfinite(loss)
    # the same batch on the streaming path: the plans are built when they are read
    model.eval()
    x0 = [None if batches[1].cochains[d].x is None else batches[1].cochains[d].x.clone() for d in range(3)]
    keep, layers.BLOCKED_LAYER = layers.BLOCKED_LAYER, False
    try:
        with torch.no_grad():
            want = model(batches[1])
    finally:
        layers.BLOCKED_LAYER = keep
    batches[1].set_xs(x0)             # (the forward leaves its embeddings in the batch)
    with torch.no_grad():
        got = model(batches[1])
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
