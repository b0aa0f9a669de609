"""CPU checks of the host halves of the static-batch path (cwn_amd/static_batch.py, round 4): the per-complex CSRs a packed
dataset keeps, the numpy restatements of the device builders' per-complex fit tests, and the layout of the device-side
segment tables.  The launches themselves are in tests/test_gpu_static.py."""
import numpy as np
import pytest
import torch

from cwn_amd.complex import ComplexBatch
from cwn_amd.packed import PackedComplexes
from cwn_amd.synthetic import zinc_like_complexes


def _csr(index: np.ndarray, key_row: int, n_rows: int):
    """destination-sorted CSR of a [2, E] index keyed on row `key_row`, stable in entry order (= cwn_csr_build)."""
    key, val = index[key_row], index[1 - key_row]
    order = np.argsort(key, kind='stable')
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(key, minlength=n_rows))])
    return rowptr.astype(np.int64), val[order].astype(np.int64)


def test_per_complex_csr_concatenates_to_the_batch_csr():
    """The CSR of a batch's boundary adjacency = the concatenation of its complexes' CSRs: `col` + the source dimension's
    cell offset, the row pointers (kept without their leading zero) + the running entry count -- for the adjacency and its
    transpose, dims 1 and 2, on shuffled batches that include molecules without rings."""
    pool = zinc_like_complexes(60, seed=5, max_ring=6, n_lo=4, n_hi=30)
    p = PackedComplexes(pool, 'cpu', max_dim=2, with_csr=True)
    rng = np.random.default_rng(0)
    for _ in range(4):
        idx = rng.permutation(len(pool))[:23]
        ref = ComplexBatch.from_complex_list([pool[i] for i in idx], max_dim=2)
        for d in (1, 2):
            bi = ref.cochains[d].boundary_index
            if bi is None:
                continue
            bi = bi.numpy()
            n_here, n_below = ref.cochains[d].num_cells, ref.cochains[d - 1].num_cells
            for name, key_row, rows_of, src_rows_of, n_rows in (('b', 1, p.n_cells[d], p.n_down[d], n_here),
                                                                ('bt', 0, p.n_down[d], p.n_cells[d], n_below)):
                want_rp, want_col = _csr(bi, key_row, n_rows)
                rp_pk, col_pk = p.keys[d][name + '_rowptr'], p.keys[d][name + '_col']
                rp, col, ent, cell = [0], [], 0, 0
                for c in idx:
                    r = rp_pk.data[rp_pk.start[c]: rp_pk.start[c] + rp_pk.length[c]].numpy().astype(np.int64)
                    assert r.size == rows_of[c]
                    rp += (r + ent).tolist()
                    cl = col_pk.data[col_pk.start[c]: col_pk.start[c] + col_pk.length[c]].numpy().astype(np.int64)
                    col += (cl + cell).tolist()
                    ent += int(col_pk.length[c])
                    cell += int(src_rows_of[c])
                assert np.array_equal(np.asarray(rp), want_rp), (d, name)
                assert np.array_equal(np.asarray(col), want_col), (d, name)


def test_single_complex_fit_masks_agree_with_the_host_table_builders():
    """blockplan.single_fit_forward / single_fit_backward (what a static batch asks before it trusts a batch to the
    device-side table builders) against the host C++ builders run on each complex alone, molecules of 6 .. 48 atoms."""
    from cwn_amd.blockplan import BlockPlan, gemm_rows_cap, lds_bytes, LDS_BYTES, single_fit_backward, single_fit_forward
    pool = zinc_like_complexes(70, seed=9, max_ring=6, n_lo=6, n_hi=48)
    F = 128
    cells = [np.array([c.cochains[d].num_cells if d in c.cochains else 0 for c in pool]) for d in range(3)]
    up_len = [np.array([(c.cochains[d].upper_index.size(1) if (d in c.cochains and c.cochains[d].upper_index is not None) else 0)
                        for c in pool]) for d in range(3)]
    b_len = [np.array([(c.cochains[d].boundary_index.size(1) if (d in c.cochains and c.cochains[d].boundary_index is not None) else 0)
                       for c in pool]) for d in range(3)]
    has_up, has_b = [True, True, False], [False, True, True]
    cap = gemm_rows_cap(F)
    src_cap = min(cap, (LDS_BYTES - lds_bytes(F, cap, 0)) // (F * 4))
    fwd = single_fit_forward(cells, up_len, b_len, F, has_up, has_b, 0, cap, src_cap)
    bwd = single_fit_backward(cells, up_len, b_len, F, has_up, has_b)
    assert fwd.any() and not fwd.all() and bwd.any() and not bwd.all()         # the pool straddles the caps
    for i in range(len(pool)):
        ptr = lambda a: [0, int(a[i])]
        plan = BlockPlan([[int(cells[d][i])] for d in range(3)], [ptr(up_len[0]), ptr(up_len[1]), None],
                         [None, ptr(b_len[1]), ptr(b_len[2])])
        assert (plan.items(F, has_up, has_b) is not None) == bool(fwd[i]), i
        assert (plan.bwd_items(F, has_up, has_b) is not None) == bool(bwd[i]), i


def test_device_table_layout_is_the_host_collate_layout():
    """StaticBatch.host_tables (the checker of cwn_collate_tables on the GPU) lays the tables out as PackedComplexes._prepare
    does for a full batch: dst / src / off / seg, then the sizes."""
    from cwn_amd import static_batch as SB
    pool = zinc_like_complexes(40, seed=2, max_ring=6)
    p = PackedComplexes(pool, 'cpu', max_dim=2, with_csr=True)
    B, D, K = 16, 3, len(p._klist)
    idx = np.random.default_rng(1).permutation(40)[:B]
    _, _, tables, _ = p._prepare(idx)
    sb = SB.StaticBatch.__new__(SB.StaticBatch)         # the layout arithmetic only (the buffers need a GPU)
    sb.packed, sb.B, sb.D, sb.K = p, B, D, K
    sb.o_src = K * (B + 1)
    sb.o_off = sb.o_src + K * B
    sb.o_seg = sb.o_off + D * 5 * B
    sb.o_sizes = sb.o_seg + D * (B + 1)
    sb.n_tab = sb.o_sizes + 8 + K
    t = sb.host_tables(idx)
    assert np.array_equal(t[:tables.size], tables)
    assert t[sb.o_sizes + 3] == B and t[sb.o_sizes + 0] == sum(pool[i].cochains[0].num_cells for i in idx)
    # a short batch: absent complexes are zero-length segments, the prefix sums stay flat behind the last one
    t2 = sb.host_tables(idx[:5])
    assert t2[sb.o_sizes + 3] == 5
    seg0 = t2[sb.o_seg: sb.o_seg + B + 1]
    assert (seg0[5:] == seg0[5]).all() and seg0[5] == sum(pool[i].cochains[0].num_cells for i in idx[:5])


def test_static_drivers_refuse_layers_that_need_a_per_batch_csr_plan():
    """StaticForward / StaticTrainStep in mode 'blocked' serve SparseCINConv stacks and (round 6) CIN++ stacks as the reference's
    molecular models run them (lower stream off: the blocked layer launch writes their third output).  A CIN++ layer with the
    lower-adjacency stream on (streaming aggregation over a CSR plan of the lower adjacency, built per batch on the host's
    sizes) is refused at construction, not handed capacity-sized buffers."""
    from cwn_amd.layers import CINppConv
    from cwn_amd.models import EmbedCINpp, EmbedSparseCIN
    from cwn_amd.static_graph import refuse_unsupported_layers
    kw = dict(dropout_rate=0.0, max_dim=2, embed_edge=True, use_coboundaries=True)
    refuse_unsupported_layers(EmbedSparseCIN(28, 4, 1, 2, 16, **kw), 'StaticForward')
    model = EmbedCINpp(28, 4, 1, 2, 16, **kw)
    refuse_unsupported_layers(model, 'StaticForward')
    conv = next(m for m in model.modules() if isinstance(m, CINppConv))
    conv.mp_levels[1].use_down_msg = True
    with pytest.raises(NotImplementedError, match='CINppConv'):
        refuse_unsupported_layers(model, 'StaticForward')


def test_host_capacity_check_over_the_distinct_size_columns():
    """StaticBatch._check_capacity: a batch beyond a capacity raises ValueError naming what overflowed; columns that
    are copies of one another are checked once against the smallest of their capacities; a column whose B largest values fit
    is not gathered at all; `-1` (no complex) counts as nothing."""
    import numpy as np
    import pytest
    from cwn_amd.static_batch import StaticBatch

    class _P:
        pass
    sb = StaticBatch.__new__(StaticBatch)
    sb.D, sb.K, sb.B = 3, 4, 8
    rng = np.random.default_rng(0)
    cells = rng.integers(5, 30, size=(100, 3))
    meta = np.zeros((100, 9 + 12), dtype=np.int64)
    meta[:, 0:9:3] = cells
    meta[:, 9] = cells[:, 0] * 2          # upper_index of dim 0
    meta[:, 10] = cells[:, 0] * 2         # its shared-cell vector: a copy
    meta[:, 11] = cells[:, 1]             # b_rowptr of dim 1 = cells of dim 1: a copy of a cell column
    meta[:, 12] = 1                       # y
    pk = _P()
    pk._meta = meta
    pk._klist = [(0, 'upper_index', None), (0, 'shared_coboundaries', None), (1, 'b_rowptr', None), (-1, 'y', None)]
    sb.packed, sb.cap_cells, sb._caps, sb._cap_cols = pk, [150, 10 ** 6, 10 ** 6], [300, 310, 10 ** 6, 8], None
    heavy = np.argsort(-cells[:, 0])[:8]
    light = np.argsort(cells[:, 0])[:8]
    host = np.full((3, 8), -1, dtype=np.int64)
    host[0], host[1, :4], host[2] = light, light[:4], heavy
    with pytest.raises(ValueError, match='batch 2: .* cells of dimension 0 exceed the capacity 150'):
        sb._check_capacity(host)
    sb._check_capacity(host[:2])                                  # the light batches pass, the short one too
    checked = sorted(c for _, _, c in sb._cap_cols)
    assert checked == [0, 3], checked       # cells of dim 0, and ONE of the two entry columns (cap 300 < 310); y and the huge caps: skipped
    sb.cap_cells[0] = 10 ** 6                                     # now the entry column decides: 2 x cells > 300
    sb._cap_cols = None
    with pytest.raises(ValueError, match="elements of 'upper_index' \\(dimension 0\\) exceed the capacity 300"):
        sb._check_capacity(host)
