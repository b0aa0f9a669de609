"""CIN++ layers (mp/layers.py:216-260, 344-427) through the complex-blocked launch (round 6).

As the reference's molecular CIN++ models run the layer -- include_down_features=False (mp/molec_models.py:111), so down_index
is None and out_down = zeros + (1 + eps2) x (:253) -- the propagate scope is the SparseCIN one plus a third output per
dimension, which the launch writes from the registers that hold the row (cwn_layer_dim.out_down).  Checked here:
  * bit-identical to the streaming path (grouped GEMM + one aggregation launch over three streams) in every form of the launch
    (16 waves, two per CU, BIG records, mixed), sort / store / load modes;
  * against the float64 oracle (the SparseCIN scope of tests/test_gpu_blocked.py + the third output);
  * the whole EmbedCINpp forward equal to the streaming path's, and the reference-generated goldens (tests/test_gpu_parity.py::
    test_embed_cinpp_whole_stack_golden) through it at hidden 64;
  * a lower adjacency (feed_down_attr) or the co-boundary stream keeps the layer off the blocked launch, with a reason."""
import pytest
import torch

from tests.test_gpu_blocked import _batch, _gate, cpu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _conv(F, seed=0, eps=0.0, **kw):
    from cwn_amd.layers import CINppConv
    torch.manual_seed(seed)
    conv = CINppConv(F, F, F, None, None, None, None, None, None, max_dim=2, hidden=F, eps=eps, train_eps=True,
                     act_module=torch.nn.ReLU, layer_dim=F, use_coboundaries=True, **kw)
    with torch.no_grad():                # three different eps per level: a swapped pair would show
        for d, lvl in enumerate(conv.mp_levels):
            lvl.eps1.fill_(eps + 0.125 * d)
            lvl.eps2.fill_(eps + 0.5 + 0.0625 * d)
            lvl.eps3.fill_(eps - 0.25 + 0.03125 * d)
    return conv.to(DEV).eval()


def _run(conv, b, blocked):
    from cwn_amd import layers
    prev = layers.BLOCKED_LAYER
    layers.BLOCKED_LAYER = blocked
    try:
        with torch.no_grad():
            params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
            plans, outs = conv.propagate_all(*params)
        assert (plans[0] == 'blocked') == blocked, getattr(conv, 'blocked_reason', None)
    finally:
        layers.BLOCKED_LAYER = prev
    return outs


@pytest.mark.parametrize('kind,n,F,variant', [('zinc', 128, 128, '0'), ('zinc', 300, 128, '1'), ('zinc', 1, 128, '0'), ('zinctrees', 40, 128, '0'),
                                              ('zincwide', 64, 128, '0'),
                                              ('molhiv', 96, 64, '0'), ('molhiv', 400, 64, '1')])
def test_cinpp_blocked_is_bit_identical_to_the_streaming_path(kind, n, F, variant):
    from cwn_amd import csr, layers
    b = _batch(kind, n, F, seed=61)
    conv = _conv(F, seed=62, eps=0.25)
    prev = layers.LAYER_VARIANT
    try:
        layers.LAYER_VARIANT = variant
        layers._BLOCKED_CACHE.clear()
        b.block_plan().forget_csr()
        first = _run(conv, b, blocked=True)       # sorts + stores the per-item CSR
        second = _run(conv, b, blocked=True)      # loads it
        with torch.no_grad():
            table = conv._blocked_args(b.get_all_cochain_params(max_dim=2, include_down_features=False), 0)[2]
        assert table.variant == int(variant)
    finally:
        layers.LAYER_VARIANT = prev
        layers._BLOCKED_CACHE.clear()
    csr._cache.clear()
    plain = _run(conv, b, blocked=False)
    assert len(first) == len(plain) == 9
    for i, (f, s, p) in enumerate(zip(first, second, plain)):
        assert torch.equal(f, s), (i, 'store vs load')
        if F == 128 or i % 3 == 1:          # (width 64: the streaming path multiplies on fp32 MFMA; the third output has no product)
            assert torch.equal(f, p), (i, (f - p).abs().max().item())
        else:
            _gate(f, cpu(p).double(), f'{kind}-{n} F={F} stream {i} vs streaming path')
    # the third output against its definition, in float64
    for d in range(3):
        x = cpu(b.cochains[d].x).double()
        want = (1.0 + float(conv.mp_levels[d].eps2)) * x
        _gate(first[3 * d + 1], want, f'out_down[{d}]')
        assert torch.equal(first[3 * d + 1], (1.0 + conv.mp_levels[d].eps2) * b.cochains[d].x)
    print(f'[gate] CIN++ {kind}-{n} F={F} variant {variant}: nine outputs torch.equal across sort / load'
          + (' and to the streaming path' if F == 128 else '; third output torch.equal to (1 + eps2) x'))


def test_cinpp_blocked_with_big_records_and_mixed_launches():
    from cwn_amd import csr, layers
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    F = 128
    small, large = zinc_like_complexes(300, 71, 6), zinc_like_complexes(12, 72, 6, n_lo=28, n_hi=70)
    cxs = []
    for i, c in enumerate(small):
        cxs.append(c)
        if i % 23 == 5 and large:
            cxs.append(large.pop())
    cxs += large
    b = ComplexBatch.from_complex_list(cxs, max_dim=2).to(DEV)
    g = torch.Generator().manual_seed(73)
    for d in range(3):
        b.cochains[d].x = torch.randn(b.cochains[d].num_cells, F, generator=g).to(DEV)
    conv = _conv(F, seed=74, eps=0.1)
    keep = (layers.BIG_ITEMS, layers.LAYER_VARIANT)
    outs = {}
    try:
        for variant in ('0', 'mixed'):
            layers.BIG_ITEMS, layers.LAYER_VARIANT = 'always', variant
            layers._BLOCKED_CACHE.clear()
            with torch.no_grad():
                table = conv._blocked_args(b.get_all_cochain_params(max_dim=2, include_down_features=False), 0)[2]
            assert (table.variant == 'mixed') == (variant == 'mixed'), table.variant
            assert getattr(table, 'n_big', 0) or any(getattr(p, 'n_big', 0) for p in getattr(table, 'parts', []))
            outs[variant] = _run(conv, b, blocked=True)
    finally:
        layers.BIG_ITEMS, layers.LAYER_VARIANT = keep
        layers._BLOCKED_CACHE.clear()
    csr._cache.clear()
    plain = _run(conv, b, blocked=False)
    for v, got in outs.items():
        for i, (f, p) in enumerate(zip(got, plain)):
            assert torch.equal(f, p), (v, i, (f - p).abs().max().item())
    print('[gate] CIN++ with BIG records, 16-wave and mixed launches: nine outputs torch.equal to the streaming path')


@pytest.mark.parametrize('H', [64, 128])
def test_embed_cinpp_forward_blocked_equals_streaming(H):
    from cwn_amd import csr, layers
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedCINpp
    from cwn_amd.synthetic import zinc_like_complexes
    torch.manual_seed(5)
    model = EmbedCINpp(28, 4, 1, 3, H, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum',
                       train_eps=True, final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True,
                       use_coboundaries=True, graph_norm='bn').to(DEV).eval()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0.0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
        for conv in model.convs:
            for d, lvl in enumerate(conv.mp_levels):
                lvl.eps1.fill_(0.1 + 0.01 * d); lvl.eps2.fill_(-0.2); lvl.eps3.fill_(0.3)
    b = ComplexBatch.from_complex_list(zinc_like_complexes(64, 9, 6), max_dim=2).to(DEV)
    x0 = [None if b.cochains[d].x is None else b.cochains[d].x.clone() for d in range(3)]

    def fwd(blocked):
        prev = layers.BLOCKED_LAYER
        layers.BLOCKED_LAYER = blocked
        layers._BLOCKED_CACHE.clear()
        csr._cache.clear()
        try:
            for d in range(3):
                b.cochains[d]._x = x0[d]
            with torch.no_grad():
                y, res = model(b, include_partial=True)
            reason = [getattr(c, 'blocked_reason', 'x') for c in model.convs]
        finally:
            layers.BLOCKED_LAYER = prev
        return y, res, reason
    yb, rb, why_b = fwd(True)
    ys, rs, why_s = fwd(False)
    assert all(w is None for w in why_b), why_b
    assert all(w is not None for w in why_s)
    for k in rs:
        if H == 128:
            assert torch.equal(rb[k], rs[k]), (k, (rb[k] - rs[k]).abs().max().item())
        else:
            _gate(rb[k], cpu(rs[k]).double(), f'EmbedCINpp H={H} {k}: blocked vs streaming')
    _gate(yb, cpu(ys).double(), f'EmbedCINpp H={H} out: blocked vs streaming')
    print(f'[gate] EmbedCINpp forward (hidden {H}): blocked propagate ' + ('torch.equal to' if H == 128 else 'within 1e-5 of') + ' the streaming path')


def test_cinpp_layers_with_other_streams_stay_off_the_blocked_launch():
    b = _batch('zinc', 16, 64, seed=81)
    for kw, word in ((dict(coboundary_stream=True), 'co-boundary'),):
        conv = _conv(64, seed=82, **kw)
        with torch.no_grad():
            got = conv._blocked_args(b.get_all_cochain_params(max_dim=2, include_down_features=False), 0)
        assert isinstance(got, str) and word in got, got
    # a real lower adjacency (the synthetic batches above carry none: without one the layer's lower stream is its self term)
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    bd = ComplexBatch.from_complex_list(zinc_like_complexes(16, 81, 6, include_down_adj=True), max_dim=2).to(DEV)
    g = torch.Generator().manual_seed(84)
    for d in range(3):
        bd.cochains[d].x = torch.randn(bd.cochains[d].num_cells, 64, generator=g).to(DEV)
    conv = _conv(64, seed=83, feed_down_attr=True)
    with torch.no_grad():
        got = conv._blocked_args(bd.get_all_cochain_params(max_dim=2, include_down_features=True), 0)
        assert isinstance(got, str) and 'lower-adjacency' in got, got
        # ... and the same layer over params without it (include_down_features=False: down_index None) takes the launch
        assert not isinstance(conv._blocked_args(bd.get_all_cochain_params(max_dim=2, include_down_features=False), 0), str)


@pytest.mark.parametrize('F,rows', [(128, (3165, 3341, 304)), (64, (1, 65, 130)), (128, (31, 32, 33)), (64, (20000, 7, 0))])
def test_fused_update_mlp3_vs_float64_and_the_grouped_launches(F, rows):
    """cwn_update_mlp3_f32 (csrc/cwn_mlp3.hip): update_up_nn / update_down_nn / update_boundaries_nn + the 3F-wide combine_nn
    of every dimension in one launch (mp/layers.py:255-260) against the same torch modules evaluated in float64, and against
    the path it replaces (two grouped GEMM launches + torch.cat + combine_nn as torch modules)."""
    import copy
    from cwn_amd import layers, ops
    conv = _conv(F, seed=91)
    with torch.no_grad():
        for m in conv.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0.0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.2)
    g = torch.Generator().manual_seed(92)
    outs = [torch.randn(r, F, generator=g).to(DEV) for r in rows for _ in range(3)]
    plans = ['blocked'] * 3
    calls = []
    orig = ops.update_mlp3
    ops.update_mlp3 = lambda dims: (calls.append(len(dims)), orig(dims))[1]
    try:
        with torch.no_grad():
            fused = conv._dense_eval(plans, outs, 0)
            prev, layers.FUSED_UPDATE_MLP = layers.FUSED_UPDATE_MLP, False
            try:
                grouped = conv._dense_eval(plans, outs, 0)
            finally:
                layers.FUSED_UPDATE_MLP = prev
    finally:
        ops.update_mlp3 = orig
    assert calls == [3], calls
    ref = copy.deepcopy(conv).double().cpu().eval()
    for d in range(3):
        lvl = ref.mp_levels[d]
        xs = [cpu(outs[3 * d + k]).double() for k in range(3)]
        with torch.no_grad():
            want = lvl.combine_nn(torch.cat([lvl.update_up_nn(xs[0]), lvl.update_down_nn(xs[1]), lvl.update_boundaries_nn(xs[2])], dim=-1)) \
                if rows[d] else torch.zeros(0, F, dtype=torch.float64)
        _gate(fused[d], want, f'update_mlp3 F={F} dim {d} ({rows[d]} rows) vs float64')
        _gate(grouped[d], want, f'grouped launches F={F} dim {d} vs float64')
        assert fused[d].shape == (rows[d], F)


@pytest.mark.parametrize('kind,n,F', [('zinc', 64, 128), ('zinc', 300, 128), ('molhiv', 96, 64)])
def test_cinpp_training_step_through_the_blocked_launches(kind, n, F):
    """With autograd on, a CIN++ layer's propagate step runs as the blocked launch (CWN_LAYER_STORE_Y + the third output) and its
    backward as the owner-form launch + (1 + eps2) g_down added onto dx (ops._blocked_backward_impl): outputs and gradients
    of the streaming autograd node (grouped GEMM + one aggregation launch over three streams)."""
    from cwn_amd import layers, ops
    b = _batch(kind, n, F, seed=15)
    from cwn_amd.layers import CINppConv
    torch.manual_seed(16)
    conv = CINppConv(F, F, F, None, None, None, None, None, None, max_dim=2, hidden=F, eps=0.25, train_eps=False,
                     act_module=torch.nn.ReLU, layer_dim=F, use_coboundaries=True).to(DEV).train()
    with torch.no_grad():
        for d, lvl in enumerate(conv.mp_levels):
            lvl.eps1.fill_(0.25 + 0.125 * d); lvl.eps2.fill_(0.75 - 0.0625 * d); lvl.eps3.fill_(-0.5 + 0.03125 * d)
    g = torch.Generator().manual_seed(13)
    ws = [torch.randn(b.cochains[d].num_cells, F, generator=g).to(DEV) for d in range(3) for _ in range(3)]
    captured = {}
    orig = ops.gemm_aggregate

    def spy(specs, make_streams, precomputed=None):
        captured['pre'] = precomputed
        return orig(specs, make_streams, precomputed=precomputed)

    def run(flag):
        layers.BLOCKED_TRAIN_FORWARD = flag
        ops.gemm_aggregate = spy
        if flag:
            ops.pack_layer_weights_many([conv.mp_levels[d].msg_up_nn[1].weight for d in range(2)], transposed=True)
        try:
            conv.zero_grad(set_to_none=True)
            xin = [b.cochains[d].x.detach().clone().requires_grad_() for d in range(3)]
            b.set_xs(xin)
            plans, outs = conv.propagate_all(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
            assert len(outs) == 9
            sum((o * w).sum() for o, w in zip(outs, ws)).backward()
            return outs, xin, {k: v.grad.clone() for k, v in conv.named_parameters() if v.grad is not None}, captured.get('pre')
        finally:
            layers.BLOCKED_TRAIN_FORWARD = True
            ops.gemm_aggregate = orig

    before = list(ops.BLOCKED_BACKWARD_LAUNCHES)
    outs1, x1, g1, pre1 = run(True)
    assert ops.BLOCKED_BACKWARD_LAUNCHES[1] == before[1] + 1, 'the owner-form backward launch did not run'
    outs0, x0, g0, pre0 = run(False)
    assert pre1 is not None and pre0 is None, 'the training forward did not take the blocked kernel'
    for i, (a, c) in enumerate(zip(outs1, outs0)):
        if F == 128 or i % 3 == 1:
            assert torch.equal(a, c), i
        else:
            torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-5 * max(1.0, float(c.abs().max())))
    for a, c in zip(x1, x0):
        torch.testing.assert_close(a.grad, c.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(c.grad.abs().max())))
    assert g1.keys() == g0.keys() and g1
    for k in g0:
        torch.testing.assert_close(g1[k], g0[k], rtol=1e-4, atol=1e-4 * max(1.0, float(g0[k].abs().max())), msg=k)
    # dx against float64 for the part that is new here: the third output's piece
    for d in range(3):
        piece = (1.0 + float(conv.mp_levels[d].eps2)) * cpu(ws[3 * d + 1]).double()
        rest = cpu(x0[d].grad).double() - piece            # (what the other two outputs contribute, by the streaming path)
        _gate(x1[d].grad, rest + piece, f'CIN++ blocked backward dx[{d}]')
    print(f'[gate] CIN++ training step {kind}-{n} F={F}: blocked forward + owner-form backward = the streaming autograd node')


def test_static_blocked_batch_serves_embed_cinpp_forward_and_training():
    """a static batch in mode 'blocked' for EmbedCINpp -- one captured graph for every batch of an epoch."""
    import copy
    from cwn_amd.models import EmbedCINpp
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticForward, StaticTrainStep
    from cwn_amd.synthetic import zinc_like_complexes
    torch.manual_seed(21)
    H, B = 64, 12
    model = EmbedCINpp(28, 4, 1, 2, H, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum',
                       final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True,
                       use_coboundaries=True, graph_norm='bn').to(DEV).eval()
    pool = zinc_like_complexes(60, 22, 6)
    packed = PackedComplexes(pool, torch.device(DEV), max_dim=2, with_csr=True)
    sb = StaticBatch(packed, B, slots=2, mode='blocked')
    sf = StaticForward(model, sb)
    batches = [list(range(0, 12)), list(range(20, 29)), list(range(30, 42)), list(range(50, 55))]
    with torch.no_grad():
        for lo in (0, 2):
            outs = sf.run_many(batches[lo: lo + 2])
            for idx, got in zip(batches[lo: lo + 2], outs):
                want = model(packed.collate(idx))
                assert torch.equal(got, want), (idx[0], (got - want).abs().max().item())
    tmodel = copy.deepcopy(model).train()
    ts = StaticTrainStep(tmodel, sb, task_type='regression')
    sb.set_batches(batches[:2])
    before = [p.detach().clone() for p in tmodel.parameters()]
    losses = ts.step()
    assert all(bool(torch.isfinite(l).item()) for l in losses)
    assert any(not torch.equal(p, q) for p, q in zip(tmodel.parameters(), before))
    print('[gate] EmbedCINpp over a static batch in mode blocked: replayed forward torch.equal to per-batch launches; a captured training step runs')


def test_axpy_eps_many_vectors_in_one_launch_equals_addcmul():
    """cwn_axpy_eps_f32 (ABI 24): y_i += (1 + eps_i) x_i for several vectors per launch -- what adds the third output's piece
    (1 + eps2) g_down onto the dx of a CIN++ layer -- bit for bit the framework's `y.addcmul_(x, 1 + eps)` / `y.add_(x)`:
    ragged lengths (a tail shorter than a workgroup's span, an empty vector), eps NULL, more descriptors than one launch takes."""
    from cwn_amd import _ffi
    g = torch.Generator().manual_seed(5)
    sizes = [4, 1024 * 4 + 12, 3340 * 128, 0, 364 * 128, 100, 4096, 40, 8, 16, 20]          # 11 > CWN_AXPY_MAX_DESCS
    ys = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    xs = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    eps = [None if i % 3 == 2 else torch.randn(1, generator=g).to(DEV) for i in range(len(sizes))]
    want = [y.clone().add_(x) if e is None else y.clone().addcmul_(x, (1.0 + e)) for y, x, e in zip(ys, xs, eps)]
    descs = [_ffi.AxpyDesc(y=y.data_ptr(), x=x.data_ptr(), eps=_ffi.ptr(e), n=y.numel()) for y, x, e in zip(ys, xs, eps)]
    L = _ffi.lib()
    assert L.cwn_axpy_eps_f32((_ffi.AxpyDesc * len(descs))(*descs), len(descs), _ffi.stream_ptr(DEV)) == 1        # CWN_ERR_BAD_ARG
    for lo in range(0, len(descs), _ffi.AXPY_MAX_DESCS):
        chunk = descs[lo: lo + _ffi.AXPY_MAX_DESCS]
        _ffi.check(L.cwn_axpy_eps_f32((_ffi.AxpyDesc * len(chunk))(*chunk), len(chunk), _ffi.stream_ptr(DEV)), 'cwn_axpy_eps_f32')
    torch.cuda.synchronize()
    for i, (y, w) in enumerate(zip(ys, want)):
        assert torch.equal(y, w), (i, sizes[i])
    bad = _ffi.AxpyDesc(y=ys[2].data_ptr(), x=xs[2].data_ptr(), eps=None, n=6)              # not a multiple of 4
    assert L.cwn_axpy_eps_f32((_ffi.AxpyDesc * 1)(bad), 1, _ffi.stream_ptr(DEV)) == 1
