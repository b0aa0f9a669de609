"""GPU tests of the static-batch path (round 4): one captured graph per model serves batches it has never seen, the way the
reference's training loop meets them (data/data_loading.py:84-111 shuffles, exp/train_utils.py:35-75 steps through).
Everything a batch changes is computed on the device (cwn_amd/static_batch.py); these tests pin each device-side piece on
its host / per-batch counterpart and the captured forward / training step on the ordinary per-batch launches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _packed(n=220, seed=3, n_lo=9, n_hi=30, with_csr=True):
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.synthetic import zinc_like_complexes
    pool = zinc_like_complexes(n, seed=seed, max_ring=6, n_lo=n_lo, n_hi=n_hi)
    return pool, PackedComplexes(pool, DEV, max_dim=2, with_csr=with_csr)


def _batches(n, B, seed, sizes=None):
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n)
    out, lo = [], 0
    for b in (sizes or [B] * (n // B)):
        out.append(perm[lo:lo + b])
        lo += b
    return out


def test_device_tables_equal_the_host_tables_and_the_collate_its_per_batch_form():
    """cwn_collate_tables against the numpy restatement (every table, the sizes, bit for bit: full batches, a short last
    batch, one complex), then the arrays the static collate writes against PackedComplexes.collate of the same complexes
    (features, index rows with the second row at the capacity offset, shared cells, batch vectors, labels) and the
    collated CSR of the boundary adjacencies and their transposes against cwn_csr_build on the per-batch index."""
    from cwn_amd import csr
    from cwn_amd.static_batch import StaticBatch
    pool, p = _packed()
    B = 48
    sb = StaticBatch(p, B)
    for idx in _batches(len(pool), B, 1, sizes=[B, B, 17, 1, B]):
        sb.set_batch(idx)
        sb.fill()
        torch.cuda.synchronize()
        assert np.array_equal(sb.tables[0].cpu().numpy(), sb.host_tables(idx)), len(idx)
        ref = p.collate(idx)
        n = [ref.cochains[d].num_cells for d in range(3)]
        assert sb.sizes() == n + [len(idx)]
        for d in range(3):
            rc, sc = ref.cochains[d], sb.batch.cochains[d]
            if rc.x is not None:
                assert torch.equal(sc._x[:n[d]], rc.x)
            assert torch.equal(sc.batch[:n[d]], rc.batch)
            for key in ('upper_index', 'boundary_index'):
                a = getattr(rc, key)
                if a is not None:
                    e = a.size(1)
                    assert torch.equal(getattr(sc, key)[:, :e], a), (d, key)       # row 1 lives at offset `capacity`
            if rc.shared_coboundaries is not None:
                e = rc.shared_coboundaries.numel()
                assert torch.equal(sc.shared_coboundaries[:e], rc.shared_coboundaries)
            if d > 0 and rc.boundary_index is not None:
                adj = csr.cached_adjacency(rc.boundary_index, n[d], n[d - 1])
                t = adj.t_src
                e = rc.boundary_index.size(1)
                assert torch.equal(sb.slots[0].bufs[(d, 'b_rowptr')][:n[d] + 1], adj.rowptr)
                assert torch.equal(sb.slots[0].bufs[(d, 'b_col')][:e], adj.col)
                assert torch.equal(sb.slots[0].bufs[(d, 'bt_rowptr')][:n[d - 1] + 1], t.rowptr)
                assert torch.equal(sb.slots[0].bufs[(d, 'bt_col')][:e], t.col)
        assert torch.equal(sb.batch.y[:len(idx)], ref.y.view(-1))
    csr.check_errors(DEV)


def test_collate_of_many_small_complexes_with_a_few_large_ones_over_eight_slots():
    """The form of the collate launch that gives a wave FOUR segments (batches of many small complexes over several slots:
    csrc/cwn_collate.hip): tiny molecules (every 16 lanes copy their own segment), with a few large ones among them (a wave
    that meets a long segment takes its four in turn), a short last batch (segments past the batch inside a wave) -- every
    slot's arrays against PackedComplexes.collate of the same complexes."""
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.synthetic import zinc_like_complexes
    pool = zinc_like_complexes(1500, seed=5, max_ring=6, n_lo=3, n_hi=10) + zinc_like_complexes(40, seed=6, max_ring=6, n_lo=70, n_hi=90)
    p = PackedComplexes(pool, DEV, max_dim=2, with_csr=True)
    B, S = 256, 8
    sb = StaticBatch(p, B, slots=S)
    batches = _batches(len(pool), B, 11, sizes=[B] * 5 + [B - 3])
    assert bool(sb.fits(batches).all())
    sb.set_epoch(batches)
    sb.fill()
    torch.cuda.synchronize()
    for j, idx in enumerate(batches):
        slot = sb.slots[j]
        ref = p.collate(idx)
        n = [ref.cochains[d].num_cells for d in range(3)]
        for d in range(3):
            rc, sc = ref.cochains[d], slot.batch.cochains[d]
            if rc.x is not None:
                assert torch.equal(sc._x[:n[d]], rc.x), (j, d)
            assert torch.equal(sc.batch[:n[d]], rc.batch), (j, d)
            for key in ('upper_index', 'boundary_index'):
                a = getattr(rc, key)
                if a is not None:
                    assert torch.equal(getattr(sc, key)[:, :a.size(1)], a), (j, d, key)
            if rc.shared_coboundaries is not None:
                assert torch.equal(sc.shared_coboundaries[:rc.shared_coboundaries.numel()], rc.shared_coboundaries), (j, d)
        assert torch.equal(slot.batch.y[:len(idx)], ref.y.view(-1))
    from cwn_amd import csr
    csr.check_errors(DEV)


def test_epoch_cursor_takes_the_batches_in_order():
    """set_epoch uploads the complex numbers of a whole epoch once; every fill then takes the next batch by itself (the
    device-side cursor): the tables after fill j are those of batch j."""
    from cwn_amd.static_batch import StaticBatch
    pool, p = _packed(n=150)
    B = 32
    sb = StaticBatch(p, B)
    batches = _batches(len(pool), B, 7, sizes=[B, B, B, 20])
    sb.set_epoch(batches)
    for j, idx in enumerate(batches):
        sb.fill()
        torch.cuda.synchronize()
        assert np.array_equal(sb.tables[0].cpu().numpy(), sb.host_tables(idx)), j
    sb.rewind(1)
    sb.fill()
    assert np.array_equal(sb.tables[0].cpu().numpy(), sb.host_tables(batches[1]))


@pytest.mark.parametrize('F,group', [(128, 1), (128, 3), (64, 4)])
def test_device_item_tables_pass_the_host_check_and_cover_every_cell_once(F, group):
    """cwn_layer_items_build_dev / cwn_layer_bwd_items_build_dev: the forward table passes the host-side validator
    (cwn_layer_items_check: derived fields, ranges, caps), its records and the backward table's own every cell of the batch
    exactly once per set, and empty records sit behind a set's own."""
    from cwn_amd import _ffi, csr
    from cwn_amd.static_batch import StaticBatch
    pool, p = _packed(n_lo=9, n_hi=28 if F == 128 else 45)
    B = 40
    sb = StaticBatch(p, B, group=group)
    has_up, has_b = [True, True, False], [False, True, True]
    for idx in _batches(len(pool), B, 5, sizes=[B, 13, B]):
        sb.set_batch(idx)
        sb.fill()
        t = sb.plan.items(F, has_up, has_b)
        tb = sb.plan.bwd_items(F, has_up, has_b)
        torch.cuda.synchronize()
        csr.check_errors(DEV)
        items = t.items.cpu().numpy()
        rc = _ffi.lib().cwn_layer_items_check(items.ctypes.data, t.n_items, F, t.c_plan(False))
        assert rc == 0, rc
        n = sb.sizes()
        for s, (dims_of_set) in enumerate(([0], [1, 2])):
            rec = items[s * B:(s + 1) * B]
            live = rec[:, 8] > 0
            assert not live[np.argmin(live):].any() if not live.all() else True          # empties behind the set's own
            for k, d in enumerate(dims_of_set):
                r0, cnt = rec[live, 10 + 7 * k], rec[live, 11 + 7 * k]
                order = np.argsort(r0, kind='stable')
                assert (np.cumsum(cnt[order]) - cnt[order] == r0[order]).all() and cnt.sum() == n[d], (s, d)
        bw = tb.items.cpu().numpy()
        for s, d in enumerate((0, 1)):
            rec = bw[s * B:(s + 1) * B]
            live = rec[:, 3] > 0
            r0, cnt = rec[live, 2], rec[live, 3]
            order = np.argsort(r0, kind='stable')
            assert (rec[live, 1] == d).all() and (np.cumsum(cnt[order]) - cnt[order] == r0[order]).all() and cnt.sum() == n[d]
            if d == 1:                                                    # the rings ride with the edges (TOP)
                assert rec[live, 5].sum() == n[2]


def _model(hidden=128, layers=2, seed=0):
    from cwn_amd.models import EmbedSparseCIN
    torch.manual_seed(seed)
    return EmbedSparseCIN(28, 4, 1, layers, hidden, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu',
                          readout='sum', train_eps=False, final_hidden_multiplier=2, final_readout='sum', init_reduce='sum',
                          embed_edge=True, use_coboundaries=True, graph_norm='bn').to(DEV)


@pytest.mark.parametrize('hidden,variant', [(128, None), (64, None), (128, 1), (64, 1)])
def test_static_forward_replays_one_graph_for_unseen_batches_bit_identically(hidden, variant):
    """StaticForward: the whole eval forward (front, layers, update networks, head) captured ONCE; batches of 48, 48, 17, 1
    and 48 complexes it has never seen give predictions bit-identical to model(collate(batch)) -- the per-batch launches
    with host-built tables -- and inside the gate of ... nothing else: equality is the bar (every kernel's result per row /
    per complex is independent of how the batch was cut)."""
    from cwn_amd import csr
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticForward
    pool, p = _packed(n_hi=28)
    model = _model(hidden).eval()
    B = 48
    sb = StaticBatch(p, B, variant=variant)          # (1: the two-per-CU form of the blocked launch, the default beyond 128 complexes)
    sf = StaticForward(model, sb)
    graphs = set()
    with torch.no_grad():
        for idx in _batches(len(pool), B, 11, sizes=[B, B, 17, 1, B]):
            got = sf.run(idx).clone()
            graphs.add(id(sf.graph))
            want = model(p.collate(idx))
            assert torch.equal(got, want), float((got - want).abs().max())
    assert len(graphs) == 1
    csr.check_errors(DEV)
    assert sb.fits(_batches(len(pool), B, 11, sizes=[B, 17])).all()
    assert sb.variant == (variant or 0) and sb.group == 1


def test_static_batch_picks_the_two_per_cu_form_beyond_128_complexes_when_the_dataset_fits_it():
    """variant=None: B <= 128 -> the 16-wave form; B > 128 and every molecule within the two-per-CU workgroup -> that form (and
    the forward still bit-identical to per-batch launches); one 40-atom molecule in the dataset -> the 16-wave form."""
    from cwn_amd import csr
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticForward
    from cwn_amd.synthetic import zinc_like_complexes
    pool, p = _packed(n=400, n_hi=24)
    model = _model(128).eval()
    B = 160
    sb = StaticBatch(p, B)
    sf = StaticForward(model, sb)
    with torch.no_grad():
        for idx in _batches(len(pool), B, 3, sizes=[B, 77, B]):
            assert torch.equal(sf.run(idx).clone(), model(p.collate(idx)))
    assert sb.variant == 1                 # (settled by the first forward table that was cut)
    csr.check_errors(DEV)
    big = zinc_like_complexes(1, 5, 6, n_lo=40, n_hi=40)
    p2 = PackedComplexes(pool + big, DEV, max_dim=2, with_csr=True)
    with torch.no_grad():
        sb2 = StaticBatch(p2, B)
        StaticForward(model, sb2).run(np.arange(B))
        assert sb2.variant == 0
        sb3 = StaticBatch(p, 128)
        StaticForward(model, sb3).run(np.arange(100))
        assert sb3.variant == 0


def test_static_forward_over_an_epoch_needs_nothing_from_the_host_per_step():
    """set_epoch + replay, replay, ...: the predictions of step j are those of the epoch's j-th batch."""
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticForward
    pool, p = _packed(n=200, n_hi=28)
    model = _model(128).eval()
    B = 32
    sb = StaticBatch(p, B)
    sb.reserve_epoch(8)
    sf = StaticForward(model, sb)
    batches = _batches(len(pool), B, 2, sizes=[B] * 5 + [9])
    with torch.no_grad():
        for epoch in range(2):
            order = batches if epoch == 0 else batches[::-1]
            sb.set_epoch(order)
            for idx in order:
                got = sf.replay()[0][:len(idx)].clone()
                assert torch.equal(got, model(p.collate(idx)))


@pytest.mark.parametrize('variant', [None, 1])
def test_static_train_step_matches_the_per_batch_step(variant):
    """StaticTrainStep: zero_grad, forward, loss, backward and Adam of exp/train_utils.py:57-75 captured ONCE and replayed on
    three batches it has never seen (48, 48, 20 complexes) against TrainStep's eager per-batch step from the same state:
    loss, the flat gradient after every step and the parameters after the last (the weight-gradient / BatchNorm-backward
    sums are fp32 atomics in both: summation order is the only difference)."""
    from cwn_amd import csr
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    from cwn_amd.train import TrainStep
    pool, p = _packed(n_hi=28)
    B = 48
    m1, m2 = _model(128, 2, seed=4), _model(128, 2, seed=4)
    m2.load_state_dict(m1.state_dict())
    batches = _batches(len(pool), B, 13, sizes=[B, B, 20])
    sb = StaticBatch(p, B, variant=variant)
    sb.set_batch(batches[0])
    st = StaticTrainStep(m1, sb, lr=1e-3)
    ref = TrainStep(m2, [p.collate(idx) for idx in batches], lr=1e-3, use_graph=False)
    worst = 0.0
    for j, idx in enumerate(batches):
        l1 = st.step_on([idx])[0].clone()
        l2 = ref.step(j)
        torch.cuda.synchronize()
        assert abs(float(l1) - float(l2)) <= 1e-5 * max(1.0, abs(float(l2))), (j, float(l1), float(l2))
        g1, g2 = st.bucket.flat, ref.bucket.flat
        rel = float((g1 - g2).norm() / g2.norm())
        print(f'[static train] step {j}: loss {float(l1):.6f} vs {float(l2):.6f}, relative L2 distance of the gradient {rel:.2e}')
        assert rel < 2e-5, (j, rel)
        if j + 1 < len(batches):
            # every step starts from ONE state: Adam turns summation noise of noise-level gradients into +-lr steps
            # (test_gpu_parity.py: test_train_step_graph_replay_matches_eager), after which two trajectories part for good
            worst = max(worst, float((st.opt.flat_p - ref.opt.flat_p).abs().max()))
            ref.opt.flat_p.copy_(st.opt.flat_p)
            ref.opt.exp_avg.copy_(st.opt.exp_avg)
            ref.opt.exp_avg_sq.copy_(st.opt.exp_avg_sq)
            for (_, a), (_, b) in zip(m1.named_buffers(), m2.named_buffers()):
                b.copy_(a)
    csr.check_errors(DEV)
    assert len(st._graphs) == 1
    for (n_, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        if a.dtype.is_floating_point:
            worst = max(worst, float((a - b).abs().max()) / max(1.0, float(b.abs().max())))
    assert worst < 2 * 1e-3 * 1.1, worst             # (ONE Adam step apart at most: sign flips of noise-level gradients, test_gpu_parity)
    for (n_, a), (_, b) in zip(m1.named_buffers(), m2.named_buffers()):
        if a.dtype.is_floating_point:                # BatchNorm running statistics: the batch's own rows only
            assert torch.allclose(a, b, rtol=5e-3, atol=2e-3), n_
        else:
            assert torch.equal(a, b), n_


def test_a_complex_beyond_a_workgroup_is_refused_by_fits_and_flagged_by_the_device():
    """A 60-atom molecule does not fit one workgroup at width 128 (48 atoms + 48 bonds are the 96 staged rows): fits() says so for the batches that hold it (the caller
    routes them to PackedComplexes.collate), and pushing such a batch through anyway sets the sticky UNFIT bit."""
    from cwn_amd import csr
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticForward
    from cwn_amd.synthetic import zinc_like_complexes
    pool = zinc_like_complexes(60, seed=1, max_ring=6, n_lo=12, n_hi=26) + zinc_like_complexes(1, seed=2, max_ring=6, n_lo=60, n_hi=60)
    p = PackedComplexes(pool, DEV, max_dim=2, with_csr=True)
    sb = StaticBatch(p, 16, caps={'cells': [640, 740, 90]})
    sf = StaticForward(_model(128).eval(), sb)
    with torch.no_grad():
        sf.run(list(range(16)))
    good, bad = np.arange(16), np.concatenate([np.arange(15), [60]])
    assert sb.fits([good, bad]).tolist() == [True, False]
    csr.check_errors(DEV)
    with torch.no_grad():
        sf.run(bad)
    with pytest.raises(IndexError, match='beyond what one workgroup holds'):
        csr.check_errors(DEV)


def test_three_slots_put_three_steps_behind_one_replay():
    """StaticBatch(slots=3): the fill launches cut the tables, arrays and item tables of three batches at once and a replay
    runs three forwards; an epoch of seven batches takes three replays, the last with two empty slots.  Every prediction
    bit-identical to model(collate(batch)); the device tables of every slot equal to the host restatement."""
    from cwn_amd import csr
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticForward
    pool, p = _packed(n=260, n_hi=28)
    model = _model(128).eval()
    B, S = 32, 3
    sb = StaticBatch(p, B, slots=S)
    sb.reserve_epoch(9)
    sf = StaticForward(model, sb)
    batches = _batches(len(pool), B, 4, sizes=[B] * 6 + [11])
    with torch.no_grad():
        n_replays = sb.set_epoch(batches)
        assert n_replays == 3
        for r in range(n_replays):
            outs = [o.clone() for o in sf.replay()]
            torch.cuda.synchronize()
            for j in range(S):
                k = r * S + j
                if k < len(batches):
                    assert np.array_equal(sb.tables[j].cpu().numpy(), sb.host_tables(batches[k])), (r, j)
                    assert torch.equal(outs[j][:len(batches[k])], model(p.collate(batches[k]))), (r, j)
                else:
                    assert sb.slots[j].sizes() == [0, 0, 0, 0]
    csr.check_errors(DEV)


def test_static_train_two_slots_and_an_empty_slot_changes_nothing():
    """StaticTrainStep over StaticBatch(slots=2): three batches = two replays, the last slot of the second empty.  Against the
    per-batch eager steps from the same state: losses, and after the epoch parameters, BatchNorm statistics and the Adam
    step counter (an empty batch's step must leave model and optimizer alone)."""
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    from cwn_amd.train import TrainStep
    pool, p = _packed(n_hi=28)
    B = 40
    m1, m2 = _model(64, 2, seed=6), _model(64, 2, seed=6)
    m2.load_state_dict(m1.state_dict())
    batches = _batches(len(pool), B, 21, sizes=[B, B, 25])
    sb = StaticBatch(p, B, slots=2)
    sb.reserve_epoch(4)
    sb.set_epoch(batches)
    st = StaticTrainStep(m1, sb, lr=1e-3)
    ref = TrainStep(m2, [p.collate(idx) for idx in batches], lr=1e-3, use_graph=False)
    sb.set_epoch(batches)
    got = []
    for r in range(2):
        got += [float(l) for l in st.step()]
    want = [float(ref.step(j)) for j in range(3)]
    torch.cuda.synchronize()
    print('[static train, 2 slots] losses', got, 'vs', want)
    for a, b in zip(got[:3], want):
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (got, want)
    assert got[3] != got[3]                          # the empty batch: the mean of nothing
    assert int(st.opt.t) == 3 == int(ref.opt.t)
    for (n_, a), (_, b) in zip(m1.named_buffers(), m2.named_buffers()):
        if a.dtype.is_floating_point:            # (the two models part by +-lr per weight where Adam met summation noise)
            assert torch.allclose(a, b, rtol=5e-3, atol=2e-3), n_
        else:
            assert torch.equal(a, b), n_             # num_batches_tracked: three, not four
    worst = 0.0
    for (n_, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        if a.dtype.is_floating_point:
            worst = max(worst, float((a - b).abs().max()) / max(1.0, float(b.abs().max())))
    assert worst < 2 * 1e-3 * 3 * 1.1, worst


@pytest.mark.parametrize('variant', [None, 1])
def test_static_forward_and_train_step_on_the_molhiv_like_configuration(variant):
    """BASELINE configs[2] through the static path: OGBEmbedSparseCIN (nine atom-feature tables, three bond-feature tables,
    hidden 64, mean readout; exp/scripts/cwn-molhiv.sh:9-32) -- the eval forward of unseen batches bit-identical to the
    per-batch launches, and a captured training step (regression loss on synthetic targets) against TrainStep's eager
    per-batch step from the same state."""
    from cwn_amd import csr
    from cwn_amd.models import OGBEmbedSparseCIN
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticForward, StaticTrainStep
    from cwn_amd.synthetic import molhiv_like_complexes
    from cwn_amd.train import TrainStep
    pool = molhiv_like_complexes(150, 7, 6)
    g = torch.Generator().manual_seed(0)
    for c in pool:
        c.y = torch.randn(1, 1, generator=g)
    p = PackedComplexes(pool, DEV, max_dim=2, with_csr=True)

    def make(seed=1):
        torch.manual_seed(seed)
        return OGBEmbedSparseCIN(1, 2, 64, dropout_rate=0.0, max_dim=2, readout='mean', final_readout='sum', init_reduce='sum',
                                 embed_edge=True, use_coboundaries=True, graph_norm='bn').to(DEV)
    B = 40
    batches = _batches(len(pool), B, 5, sizes=[B, B, 23])
    model = make().eval()
    sb = StaticBatch(p, B, variant=variant)       # (1: what a batch of 512 gets by default -- the bench's molhiv legs)
    assert sb.fits(batches).all()
    sf = StaticForward(model, sb)
    with torch.no_grad():
        for idx in batches:
            got = sf.run(idx).clone()
            want = model(p.collate(idx))
            assert torch.equal(got, want), float((got - want).abs().max())
    csr.check_errors(DEV)
    # training
    m1, m2 = make(2), make(2)
    m2.load_state_dict(m1.state_dict())
    sb2 = StaticBatch(p, B, variant=variant)
    sb2.set_batch(batches[0])
    st = StaticTrainStep(m1, sb2, lr=1e-3)
    ref = TrainStep(m2, [p.collate(idx) for idx in batches], lr=1e-3, use_graph=False)
    for j, idx in enumerate(batches):
        l1 = st.step_on([idx])[0].clone()
        l2 = ref.step(j)
        torch.cuda.synchronize()
        assert abs(float(l1) - float(l2)) <= 1e-5 * max(1.0, abs(float(l2))), (j, float(l1), float(l2))
        rel = float((st.bucket.flat - ref.bucket.flat).norm() / ref.bucket.flat.norm())
        print(f'[static train, molhiv-like, variant {variant}] step {j}: loss {float(l1):.6f} vs {float(l2):.6f}, relative L2 distance of the gradient {rel:.2e}')
        assert rel < 2e-5, (j, rel)
        if j + 1 < len(batches):                     # (every step from ONE state: test_static_train_step_matches_the_per_batch_step)
            ref.opt.flat_p.copy_(st.opt.flat_p)
            ref.opt.exp_avg.copy_(st.opt.exp_avg)
            ref.opt.exp_avg_sq.copy_(st.opt.exp_avg_sq)
            for (_, a), (_, b_) in zip(m1.named_buffers(), m2.named_buffers()):
                b_.copy_(a)
    csr.check_errors(DEV)
    assert len(st._graphs) == 1


def test_the_readme_training_loop():
    """README.md's eight-line loop as written: a packed dataset, StaticBatch(slots=4), StaticTrainStep, two shuffled epochs of
    batches of 32 (the last one short, the last replay's spare slots empty): every real batch gives a finite loss, an empty slot
    NaN and no step; the step counter equals the number of real batches; the parameters move."""
    from cwn_amd import csr
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    pool, packed = _packed(n=150, n_hi=28)
    model = _model(64, 2, seed=3)
    before = [p.detach().clone() for p in model.parameters()]
    sb = StaticBatch(packed, batch_size=32, slots=4)
    step = StaticTrainStep(model, sb, lr=1e-3)
    rng = np.random.default_rng(0)
    real = 0
    for epoch in range(2):
        perm = rng.permutation(len(pool))
        order = [perm[i:i + 32] for i in range(0, len(perm), 32)]             # 5 batches: 32, 32, 32, 32, 22
        n_rep = sb.set_epoch(order)
        assert n_rep == 2                                                     # 4 + 1 (+ 3 empty slots)
        losses = []
        for _ in range(n_rep):
            losses += [float(l) for l in step.step()]
        assert len(losses) == 8
        assert all(np.isfinite(l) for l in losses[:5]) and all(np.isnan(l) for l in losses[5:]), losses
        real += 5
    torch.cuda.synchronize()
    csr.check_errors(DEV)
    assert int(step.opt.t) == real, (int(step.opt.t), real)
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
    assert sum(isinstance(k, tuple) for k in step._graphs) == 1          # ONE graph of four steps (+ the one-step graphs of its warm-up)


def test_static_train_step_in_the_data_parallel_form():
    """world > 1 (forced here on one GPU: the collectives are no-ops without a process group) runs the S steps of a fill as S
    replays of TrainStep's graph(forward + backward) -> all-reduce -> graph(Adam) form -- nothing is captured around a
    collective -- and must equal the one-graph form step for step."""
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    pool, p = _packed(n_hi=28)
    B = 40
    m1, m2 = _model(128, 2, seed=6), _model(128, 2, seed=6)
    m2.load_state_dict(m1.state_dict())
    batches = _batches(len(pool), B, 21, sizes=[B, B, B, 17])
    sa, sb_ = StaticBatch(p, B, slots=2), StaticBatch(p, B, slots=2)
    one = StaticTrainStep(m1, sa, lr=1e-3)
    two = StaticTrainStep(m2, sb_, lr=1e-3)
    two.world = 2
    for sb in (sa, sb_):
        assert sb.set_epoch(batches) == 2
    for rep in range(2):
        la, lb = one.step(), two.step()
        torch.cuda.synchronize()
        for a, b in zip(la, lb):
            # (step 0 from identical states; later steps one Adam sign flip of noise-level gradients apart per step at most)
            tol = 1e-5 if (rep == 0 and a is la[0]) else 1e-2
            assert abs(float(a) - float(b)) <= tol * max(1.0, abs(float(a))), (rep, float(a), float(b))
    assert two._graphs[0][1] is not None                # the Adam graph of the two-graph form
    worst = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(m1.parameters(), m2.parameters()))
    print(f'[static train, data-parallel form] parameters after four steps: max|delta| = {worst:.3e}')
    assert worst < 2 * 1e-3 * 4 * 1.1, worst           # (four Adam steps: sign flips of noise-level gradients at most)


def test_from_graphs_to_static_training_steps_without_per_complex_objects():
    """data/utils.py:501-544 + data/data_loading.py:84-111 + exp/train_utils.py:57-75 end to end on the device path: PyG-like
    graphs are ring-lifted by host threads straight into the packed dataset (with the per-complex CSRs), PackedLoader shuffles
    index lists, StaticTrainStep cuts the batches and steps -- against TrainStep on the collated batches of the same lists."""
    from cwn_amd import csr, lifting
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.packed import PackedLoader
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    from cwn_amd.train import TrainStep
    from tests.test_lifting import _pyg_like, random_molecule
    rng = np.random.default_rng(4)
    graphs = []
    for i in range(96):
        n, bonds = random_molecule(rng, 8, 26)
        g, _ = _pyg_like(rng, n, bonds, with_attr=True, long_x=False)
        g['x'] = torch.from_numpy(rng.integers(0, 28, size=(n, 1))).float()               # atom types
        g['y'] = torch.randn(1, 1, generator=torch.Generator().manual_seed(i))           # (edge_attr: bond types 0 .. 3)
        graphs.append(g)
    packed, dimension, feats = lifting.pack_graph_dataset_with_rings(graphs, max_ring_size=6, init_edges=True, init_rings=False,
                                                                     n_threads=2, device=DEV, with_csr=True)
    assert dimension == 2

    def make():
        torch.manual_seed(9)
        return EmbedSparseCIN(28, 4, 1, 2, 64, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum',
                              train_eps=False, final_hidden_multiplier=2, final_readout='sum', init_reduce='sum',
                              embed_edge=True, use_coboundaries=True, graph_norm='bn').to(DEV)
    loader = PackedLoader(packed, batch_size=24, shuffle=True, seed=5)
    loader.set_epoch(0)
    batches = loader.batches()
    assert len(batches) == 4
    sb = StaticBatch(packed, 24, slots=2)
    assert sb.fits(batches).all()
    m1, m2 = make(), make()
    st = StaticTrainStep(m1, sb, lr=1e-3)
    ref = TrainStep(m2, [packed.collate(idx) for idx in batches[:2]], lr=1e-3, use_graph=False)
    assert sb.set_epoch(batches) == 2
    losses = [float(l) for l in st.step()]
    want = [float(ref.step(0)), float(ref.step(1))]
    torch.cuda.synchronize()
    assert abs(losses[0] - want[0]) <= 1e-5 * max(1.0, abs(want[0])), (losses, want)
    assert abs(losses[1] - want[1]) <= 2e-3 * max(1.0, abs(want[1])), (losses, want)       # (one Adam step apart already)
    more = [float(l) for l in st.step()]
    assert all(np.isfinite(l) for l in more)
    csr.check_errors(DEV)


def test_static_train_step_on_a_multi_task_head_with_null_labels():
    """ogbg-mol* style: 12 binary tasks per molecule, ~30 % of the labels missing (NaN) -- exp/train_utils.py:62-73 masks them
    out of BCEWithLogits.  The captured static step (cwn_loss_cols_f32: *n_dev complexes x 12 columns are real) against the
    per-batch eager step on two unseen batches, one of them short."""
    from cwn_amd import csr
    from cwn_amd.models import OGBEmbedSparseCIN
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    from cwn_amd.synthetic import molhiv_like_complexes
    from cwn_amd.train import TrainStep
    T = 12
    pool = molhiv_like_complexes(90, 11, 6, n_hi=40)
    g = torch.Generator().manual_seed(2)
    for c in pool:
        y = (torch.rand(1, T, generator=g) > 0.5).float()
        y[torch.rand(1, T, generator=g) < 0.3] = float('nan')
        c.y = y
    p = PackedComplexes(pool, DEV, max_dim=2, with_csr=True)

    def make():
        torch.manual_seed(5)
        return OGBEmbedSparseCIN(T, 2, 64, dropout_rate=0.0, max_dim=2, readout='mean', final_readout='sum', init_reduce='sum',
                                 embed_edge=True, use_coboundaries=True, graph_norm='bn').to(DEV)
    B = 32
    batches = _batches(len(pool), B, 8, sizes=[B, 19])
    m1, m2 = make(), make()
    m2.load_state_dict(m1.state_dict())
    sb = StaticBatch(p, B)
    sb.set_batch(batches[0])
    st = StaticTrainStep(m1, sb, task_type='bin_classification', lr=1e-3)
    ref = TrainStep(m2, [p.collate(idx) for idx in batches], task_type='bin_classification', lr=1e-3, use_graph=False)
    for j, idx in enumerate(batches):
        l1 = st.step_on([idx])[0].clone()
        l2 = ref.step(j)
        torch.cuda.synchronize()
        assert np.isfinite(float(l2)) and abs(float(l1) - float(l2)) <= 1e-5 * max(1.0, abs(float(l2))), (j, float(l1), float(l2))
        rel = float((st.bucket.flat - ref.bucket.flat).norm() / ref.bucket.flat.norm())
        assert rel < 2e-5, (j, rel)
        ref.opt.flat_p.copy_(st.opt.flat_p)
        ref.opt.exp_avg.copy_(st.opt.exp_avg)
        ref.opt.exp_avg_sq.copy_(st.opt.exp_avg_sq)
        for (_, a), (_, b_) in zip(m1.named_buffers(), m2.named_buffers()):
            b_.copy_(a)
    csr.check_errors(DEV)


def test_the_example_script_runs():
    """examples/train_shuffled_epochs.py (the reference's loop end to end on the device path) at a small size."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr = subprocess.run([sys.executable, os.path.join(root, 'examples', 'train_shuffled_epochs.py'), '384', '2'],
                        capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert 'held-out MAE' in pr.stdout and 'epoch 1:' in pr.stdout, pr.stdout[-2000:]


def test_a_batch_beyond_the_capacity_is_refused_on_the_host_and_dropped_on_the_device():
    """the default capacities are a statistical bound and nothing compared a batch with them -- a size-sorted batch
    wrote past the buffers inside a replayed graph.  Now (a) set_batches / set_epoch raise for such a batch, and (b) a caller
    that writes the complex numbers itself gets the device-side guard: the slot runs as an EMPTY batch (every size word 0,
    nothing written past -- or into -- the arrays) and the sticky word reports it."""
    from cwn_amd import csr
    from cwn_amd.static_batch import StaticBatch
    pool, p = _packed(n=200, n_lo=9, n_hi=40)
    B = 32
    sb = StaticBatch(p, B, slots=2)
    heavy = np.argsort(-p._meta[:, 0])[:B]                  # the B largest molecules: beyond mean x B + 6 sigma sqrt(B)?
    tot = p._meta[heavy, 0].sum()
    if tot <= sb.cap_cells[0]:                              # (make sure the case is the one under test)
        sb = StaticBatch(p, B, slots=2, caps={'cells': [int(tot) - 5, sb.cap_cells[1], sb.cap_cells[2]]})
    light = np.argsort(p._meta[:, 0])[:B]
    assert not sb.fits([heavy])[0] and sb.fits([light])[0]
    with pytest.raises(ValueError, match='exceed the capacity'):
        sb.set_batches([light, heavy])
    with pytest.raises(ValueError, match='exceed the capacity'):
        sb.set_epoch([light, light, heavy])
    # (b) behind the host's back
    sb.set_batches([light, light])
    sb.fill()
    torch.cuda.synchronize()
    before = {k: v[1].clone() for k, v in sb.bufs.items()}
    sb.idx[B:2 * B].copy_(torch.from_numpy(heavy).to(DEV))
    sb.cursor.zero_()
    sb.fill()
    torch.cuda.synchronize()
    assert sb.sizes(0) == [int(p._meta[light, 3 * d].sum()) for d in range(3)] + [B]      # slot 0 is served as ever
    assert sb.sizes(1) == [0, 0, 0, 0]                                                   # slot 1: an empty batch
    assert not sb.tables[1].any()
    for k, v in sb.bufs.items():
        assert torch.equal(v[1], before[k]), k                                           # ... and nothing was written for it
    with pytest.raises(IndexError, match='capacity'):
        csr.check_errors(DEV)
    csr.check_errors(DEV)                                                                # (reported once)


def test_data_parallel_form_leaves_an_empty_tail_slot_alone():
    """in the data-parallel form (graph(forward + backward) -> all-reduce -> graph(Adam)) the Adam graph was recorded
    with `active` unset, so an empty tail slot took an Adam step on a zero gradient (momentum moved the parameters, t
    advanced).  Three batches over two slots, forced world 2 on one GPU: t == 3 and the second replay's empty slot changes
    no parameter."""
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    pool, p = _packed(n_hi=28)
    B = 40
    m = _model(64, 2, seed=6)
    batches = _batches(len(pool), B, 21, sizes=[B, B, 25])
    sb = StaticBatch(p, B, slots=2)
    st = StaticTrainStep(m, sb, lr=1e-3)
    st.world = 2
    sb.set_epoch(batches)
    st.step()
    torch.cuda.synchronize()
    assert int(st.opt.t) == 2
    # the second replay by hand: slot 0 real, slot 1 empty
    from cwn_amd.train import TrainStep
    TrainStep.step(st, 0)
    torch.cuda.synchronize()
    keep = [q.detach().clone() for q in m.parameters()]
    mom = st.opt.exp_avg.clone()
    assert int(st.opt.t) == 3
    loss = TrainStep.step(st, 1)
    torch.cuda.synchronize()
    assert float(loss) != float(loss)                        # the mean of nothing
    assert int(st.opt.t) == 3
    assert torch.equal(st.opt.exp_avg, mom)
    for a, b in zip(m.parameters(), keep):
        assert torch.equal(a.detach(), b)


def test_static_forward_short_replay_for_the_tail_of_an_epoch():
    """Round 6: an epoch that is not a multiple of S ends with a replay over a power-of-two number of slots instead of S - r empty
    ones; StaticForward.run_epoch returns every batch's predictions, bit for bit the per-batch launches', whatever the split."""
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticForward
    from cwn_amd.synthetic import zinc_like_complexes
    torch.manual_seed(1)
    pool = zinc_like_complexes(230, 3, 6)
    p = PackedComplexes(pool, DEV, max_dim=2, with_csr=True)
    model = EmbedSparseCIN(28, 4, 1, 2, 64, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(DEV).eval()
    B, S = 16, 8
    sb = StaticBatch(p, B, slots=S)
    sf = StaticForward(model, sb)
    perm = np.random.default_rng(2).permutation(len(pool))
    batches = [perm[k * B:(k + 1) * B] for k in range(14)] + [perm[224:230]]     # 15 batches: 8 + 4 + 2 + 1, the last one ragged
    assert [sf.slots_for(n) for n in (15, 8, 7, 5, 3, 2, 1)] == [8, 8, 8, 8, 4, 2, 1]
    with torch.no_grad():
        got = sf.run_epoch(batches)
        assert sorted(sf._graphs) == [1, 2, 4, 8] or sorted(sf._graphs) == [8]      # (7 left -> 8 slots: one replay with one empty slot)
        for k, idx in enumerate(batches):
            want = model(p.collate(idx))
            assert torch.equal(got[k], want), k
        # a second epoch in another order through the same graphs
        got2 = sf.run_epoch(batches[::-1])[::-1]
        for k in range(len(batches)):
            assert torch.equal(got2[k], got[k]), k
        got3 = sf.run_epoch(batches[:11])            # 8 + 4 (3 left -> 4 slots)
        assert len(got3) == 11 and all(torch.equal(got3[k], got[k]) for k in range(11))


def test_static_train_step_short_replay_for_the_tail_of_an_epoch():
    """Round 6: StaticTrainStep.run_epoch takes one optimisation step per batch -- S at a time and a shorter captured sequence
    for the tail -- and ends where the same epoch through full replays with empty slots ends (an empty slot's step changes
    nothing): the same parameters, Adam's counter = the number of batches."""
    import copy
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    from cwn_amd.synthetic import zinc_like_complexes
    torch.manual_seed(2)
    pool = zinc_like_complexes(176, 5, 6)
    p = PackedComplexes(pool, DEV, max_dim=2, with_csr=True)
    model = EmbedSparseCIN(28, 4, 1, 2, 64, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(DEV).train()
    B, S = 16, 4
    perm = np.random.default_rng(3).permutation(len(pool))
    batches = [perm[k * B:(k + 1) * B] for k in range(11)]              # 4 + 4 + (3 -> a replay of 4 slots) ... and 9: 4 + 4 + 1
    m1, m2, m3 = copy.deepcopy(model), copy.deepcopy(model), copy.deepcopy(model)
    t1 = StaticTrainStep(m1, StaticBatch(p, B, slots=S), task_type='regression', lr=1e-3)
    t2 = StaticTrainStep(m2, StaticBatch(p, B, slots=S), task_type='regression', lr=1e-3)
    t3 = StaticTrainStep(m3, StaticBatch(p, B, slots=S), task_type='regression', lr=1e-3)
    assert [t1.slots_for(n) for n in (9, 4, 3, 2, 1)] == [4, 4, 4, 2, 1]
    for ep in (batches[:9], batches[:10], batches):
        l1 = t1.run_epoch(ep)
        for t in (t2, t3):
            t.sb.set_epoch(ep)
            for _ in range(-(-len(ep) // S)):
                t.step()                                          # full replays: the tail's empty slots change nothing
        assert len(l1) == len(ep) and all(bool(torch.isfinite(x)) for x in l1)
    torch.cuda.synchronize()
    assert int(t1.opt.t) == int(t2.opt.t) == int(t3.opt.t) == 9 + 10 + 11
    # Thirty Adam steps over weight gradients summed by fp32 atomics: two runs of the SAME sequence (m2, m3) drift apart by
    # ~1e-2 (a parameter whose gradient is noise moves by lr per step in either direction) -- the tail replays (m1) must be
    # no further from them than they are from each other
    dist = lambda a, b: max(float((x - y).abs().max()) for (_, x), (_, y) in zip(a.named_parameters(), b.named_parameters()))
    floor = dist(m2, m3)
    print(f'[static tail] full vs full {floor:.3e}, tail vs full {dist(m1, m2):.3e}')
    assert dist(m1, m2) <= 2.0 * floor + 1e-4, (dist(m1, m2), floor)
