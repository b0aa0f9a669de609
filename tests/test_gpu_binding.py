"""The compiled binding of the eager path (csrc/cwn_torch_ext.cpp) against the ctypes binding: the same
launches through both, bit for bit; the same exceptions; prepared launches that notice when their parameters change.
mp/layers.py:184-199 is what a user layer calls per forward -- the path these launches serve."""
import pytest
import torch

from cwn_amd import _cext, csr, layers, ops

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _model_and_batch(B=24, hidden=128, L=2):
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.synthetic import zinc_like_complexes
    torch.manual_seed(0)
    model = EmbedSparseCIN(28, 4, 1, L, hidden, dropout_rate=0.0, max_dim=2, embed_edge=True, use_coboundaries=True).to(DEV).eval()
    with torch.no_grad():                      # running statistics that are not the identity
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0.0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
    b = ComplexBatch.from_complex_list(zinc_like_complexes(B, 5, 6), max_dim=2).to(DEV)
    x0 = [None if b.cochains[d].x is None else b.cochains[d].x.clone() for d in range(3)]

    def fwd(m=None):
        for d in range(3):
            b.cochains[d]._x = x0[d]
        return (m or model)(b)
    return model, b, fwd


def _only(d):
    assert len(d) == 1
    return next(iter(d.values()))


def _forget(model):
    for conv in model.convs:
        layers._BLOCKED_CACHE.pop(conv, None)
        layers._MLP_CACHE.pop(conv, None)


def test_the_module_is_built_and_speaks_this_abi():
    from cwn_amd import _ffi
    X = _cext.ext()
    assert X is not None, 'cwn_amd/_cwn_torch_ext.so is missing: python -m cwn_amd._build_ext (build() does it)'
    assert int(X.abi_version) == _ffi.ABI_VERSION and _cext.active() == 'compiled'


@pytest.mark.parametrize('hidden', [64, 128])
def test_eager_forward_is_bit_identical_through_both_bindings(hidden):
    model, b, fwd = _model_and_batch(hidden=hidden)
    outs = {}
    with torch.no_grad():
        for which in ('ctypes', 'compiled'):
            _forget(model)
            with _cext.binding(which):
                first = fwd().clone()                 # builds the prepared launches with this binding
                built = [next(iter(layers._BLOCKED_CACHE[c].values()))['launch'] for c in model.convs] + \
                        [layers._MLP_CACHE[c][0] for c in model.convs]
                second = fwd().clone()                # ... and runs them again from the caches
                again = [next(iter(layers._BLOCKED_CACHE[c].values()))['launch'] for c in model.convs] + \
                        [layers._MLP_CACHE[c][0] for c in model.convs]
            assert torch.equal(first, second)
            assert all(a is b_ for a, b_ in zip(built, again))       # (nothing in a forward invalidates its own prepared launches)
            ent = layers._BLOCKED_CACHE[model.convs[0]]
            launch = next(iter(ent.values()))['launch']
            assert (launch._c is not None) == (which == 'compiled')
            assert (layers._MLP_CACHE[model.convs[0]][0]._c is not None) == (which == 'compiled')
            outs[which] = first
    assert torch.equal(outs['ctypes'], outs['compiled'])
    print(f'[gate] eager EmbedSparseCIN forward (hidden {hidden}): ctypes and compiled bindings torch.equal, cached launches equal to fresh ones')


def test_training_forward_through_the_compiled_binding_stores_the_products():
    """LayerLaunch.run(ys=...) -- the training forward of the blocked kernel (CWN_LAYER_STORE_Y) -- through both bindings, in
    the deterministic mode: the output and every gradient of one forward + backward equal."""
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.synthetic import zinc_like_complexes
    res = {}
    ops.deterministic(True)
    try:
        for which in ('ctypes', 'compiled'):
            torch.manual_seed(1)
            model = EmbedSparseCIN(28, 4, 1, 2, 64, dropout_rate=0.0, max_dim=2, embed_edge=True, use_coboundaries=True).to(DEV).train()
            b = ComplexBatch.from_complex_list(zinc_like_complexes(16, 7, 6), max_dim=2).to(DEV)
            with _cext.binding(which):
                out = model(b)
                out.abs().sum().backward()
            res[which] = (out.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None])
    finally:
        ops.deterministic(False)
    assert torch.equal(res['ctypes'][0], res['compiled'][0])
    ga, gc = res['ctypes'][1], res['compiled'][1]
    assert len(ga) == len(gc) > 20 and all(torch.equal(a, c) for a, c in zip(ga, gc))


def test_torch_library_ops_run_the_same_launches():
    X = _cext.ext()
    model, b, fwd = _model_and_batch()
    with torch.no_grad():
        _forget(model)
        fwd()
        conv = model.convs[1]
        launch = next(iter(layers._BLOCKED_CACHE[conv].values()))['launch']
        xs = [torch.randn(r, launch.F, device=DEV) for r in launch.rows]
        want = launch.run(xs, 0)
        h = X.layer_register(launch._c)
        got = torch.ops.cwn.layer_fused(xs, h, 0)
        assert len(got) == len(want) and all(torch.equal(g, w) for g, w in zip(got, want))
        mlp = layers._MLP_CACHE[conv][0]
        hm = X.mlp_register(mlp._c)
        want2 = mlp.run(want[0::2], want[1::2])
        got2 = torch.ops.cwn.update_mlp(list(want[0::2]), list(want[1::2]), hm)
        assert all(torch.equal(g, w) for g, w in zip(got2, want2))
        X.release(h)
        X.release(hm)
        with pytest.raises(RuntimeError, match='unknown handle'):
            torch.ops.cwn.layer_fused(xs, h, 0)


def test_argument_errors_are_the_ctypes_bindings():
    model, b, fwd = _model_and_batch()
    with torch.no_grad():
        for which in ('ctypes', 'compiled'):
            _forget(model)
            with _cext.binding(which):
                fwd()
            launch = next(iter(layers._BLOCKED_CACHE[model.convs[0]].values()))['launch']
            xs = [torch.randn(r, launch.F, device=DEV) for r in launch.rows]
            with pytest.raises(ValueError, match='rows / width'):
                launch.run([xs[0][:-1]] + xs[1:], 0)
            with pytest.raises(TypeError, match='float32'):
                launch.run([xs[0].double()] + xs[1:], 0)
            # a non-contiguous feature matrix is taken (copied), as before
            wide = torch.randn(launch.rows[0], 2 * launch.F, device=DEV)
            got = launch.run([wide[:, ::2]] + xs[1:], 0)
            want = launch.run([wide[:, ::2].contiguous()] + xs[1:], 0)
            assert all(torch.equal(g, w) for g, w in zip(got, want))
    csr.check_errors(DEV)


@pytest.mark.parametrize('which', ['ctypes', 'compiled'])
def test_prepared_update_launch_notices_what_changes_under_it(which):
    """ops.MlpLaunch is derived once from a layer's parameters: an in-place write to ANY of them (here: one BatchNorm bias, one
    running variance), a replaced submodule, a switch to training mode -- each must be seen by the next call."""
    model, b, fwd = _model_and_batch(hidden=64)
    with torch.no_grad(), _cext.binding(which):
        _forget(model)
        base = fwd().clone()
        conv = model.convs[0]
        first = layers._MLP_CACHE[conv][0]
        assert torch.equal(fwd(), base) and layers._MLP_CACHE[conv][0] is first          # reused

        def fresh():
            _forget(model)
            return fwd().clone()
        bn = conv.mp_levels[1].update_up_nn[1]
        bn.bias.add_(0.5)                                                               # (1) in place, one tensor
        got = fwd().clone()
        assert layers._MLP_CACHE[conv][0] is not first and not torch.equal(got, base)
        assert torch.equal(got, fresh())
        conv.mp_levels[0].combine_nn[1].running_var.mul_(2.0)                          # (2) a buffer
        got = fwd().clone()
        assert torch.equal(got, fresh())
        lin = conv.mp_levels[2].update_boundaries_nn[0]                                # (3) a replaced submodule
        new = torch.nn.Linear(lin.in_features, lin.out_features).to(DEV)
        held = layers._MLP_CACHE[conv][0]
        conv.mp_levels[2].update_boundaries_nn[0] = new
        got = fwd().clone()
        assert layers._MLP_CACHE[conv][0] is not held
        assert torch.equal(got, fresh())
        lin2 = conv.mp_levels[0].msg_up_nn[1]                                          # (4) the message weight (LayerLaunch's side)
        lin2.weight.mul_(1.5)
        got = fwd().clone()
        assert torch.equal(got, fresh())
        model.train()                                                                  # (5) batch statistics: not this launch's
        held = layers._MLP_CACHE[conv][0]
        got = fwd().clone()
        assert not torch.equal(got, base)
        assert torch.equal(got, fresh())
        model.eval()
        assert torch.equal(fwd(), fresh())
        # ... and the running statistics that training-mode forward wrote (torch's native batch_norm: without moving their
        # version counters) are the ones the next eval forward folds: a deep copy has no cached fold
        import copy
        assert torch.equal(fwd(), fwd(copy.deepcopy(model)))


def test_a_forward_reads_the_error_word_once_at_its_end():
    """csr.deferred_checks: the front's range check no longer syncs in the middle of a forward; the IndexError still comes out of
    the same call."""
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    model, b, fwd = _model_and_batch()
    with torch.no_grad():
        fwd()
        n = [0]
        real = torch.Tensor.item

        def counting(self):
            n[0] += 1
            return real(self)
        torch.Tensor.item = counting
        try:
            fwd()
        finally:
            torch.Tensor.item = real
        assert n[0] <= 1, n[0]
        b2 = ComplexBatch.from_complex_list(zinc_like_complexes(8, 9, 6), max_dim=2).to(DEV)
        b2.cochains[0].x[3, 0] = 28.0
        with pytest.raises(IndexError):
            model(b2)
    csr._err_flag(DEV).zero_()


def test_prepared_front_and_head_launches_equal_the_long_way_and_notice_changes():
    """ops.FrontLaunch / ops.HeadLaunch: a second forward of a batch runs the prepared launches (no new entries), its output is
    the first forward's bit for bit; another batch, a changed embedding table or head weight, an include_partial call go the
    long way and come out right."""
    from cwn_amd import models
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    model, b, fwd = _model_and_batch()
    with torch.no_grad():
        first = fwd().clone()
        front, head = _only(layers._FRONT_CACHE[model.init_conv])[0], _only(models._HEAD_CACHE[model])[0]
        again = fwd().clone()
        assert torch.equal(first, again)
        assert _only(layers._FRONT_CACHE[model.init_conv])[0] is front and _only(models._HEAD_CACHE[model])[0] is head
        # reference for everything below: the same model with the prepared launches forgotten
        def fresh(batch_fwd):
            layers._FRONT_CACHE.pop(model.init_conv, None)
            models._HEAD_CACHE.pop(model, None)
            _forget(model)
            return batch_fwd().clone()
        model.v_embed_init.weight[3].add_(1.0)                      # an embedding row, in place
        got = fwd().clone()
        assert not torch.equal(got, first) and torch.equal(got, fresh(fwd))
        model.lin2.bias.add_(0.25)
        got = fwd().clone()
        assert torch.equal(got, fresh(fwd))
        model.lin1s[1].weight.mul_(0.5)
        got = fwd().clone()
        assert torch.equal(got, fresh(fwd))
        # another batch through the same model, then the first one again
        b2 = ComplexBatch.from_complex_list(zinc_like_complexes(17, 11, 6), max_dim=2).to(DEV)
        x2 = [None if b2.cochains[d].x is None else b2.cochains[d].x.clone() for d in range(3)]

        def fwd2():
            for d in range(3):
                b2.cochains[d]._x = x2[d]
            return model(b2)
        o2 = fwd2().clone()
        o1 = fwd().clone()
        assert torch.equal(o2, fresh(fwd2)) and torch.equal(o1, fresh(fwd))
        # side outputs: not the prepared launch's business
        for d in range(3):
            b2.cochains[d]._x = x2[d]
        out, res = model(b2, include_partial=True)
        assert torch.equal(out, o2) and 'pool_0' in res and 'layer0_0' in res
    csr.check_errors(DEV)


def test_the_front_checks_each_feature_version_once():
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    model, b, fwd = _model_and_batch()
    n = [0]
    real = torch.Tensor.item

    def counting(self):
        n[0] += 1
        return real(self)
    with torch.no_grad():
        fwd()
        fwd()
        torch.Tensor.item = counting
        try:
            fwd()
            fwd()
        finally:
            torch.Tensor.item = real
        assert n[0] == 0, n[0]                        # the batch's features were checked when they were first seen
    # a bad atom type written INTO a checked tensor is a new version: checked again, raised from that forward
    b3 = ComplexBatch.from_complex_list(zinc_like_complexes(24, 5, 6), max_dim=2).to(DEV)
    x3 = [None if b3.cochains[d].x is None else b3.cochains[d].x.clone() for d in range(3)]
    with torch.no_grad():
        def fwd3():
            for d in range(3):
                b3.cochains[d]._x = x3[d]
            return model(b3)
        fwd3()
        fwd3()
        x3[0][5, 0] = 31.0
        with pytest.raises(IndexError):
            fwd3()
        x3[0][5, 0] = 2.0
        good = fwd3().clone()
        assert torch.isfinite(good).all()
    csr._err_flag(DEV).zero_()


@pytest.mark.parametrize('which', ['ctypes', 'compiled'])
def test_forward_under_inference_mode_over_a_batch_built_inside_it(which):
    """the prepared-launch caches read `Tensor._version` of batch tensors; inference tensors keep none
    (torch raises).  `with torch.inference_mode(): model(batch.to(dev))` must run -- uncached -- and give the no_grad result."""
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    model, b, fwd = _model_and_batch(B=12, hidden=64)
    with torch.no_grad():
        ref = fwd().clone()
    _forget(model)
    with _cext.binding(which), torch.inference_mode():
        bi = ComplexBatch.from_complex_list(zinc_like_complexes(12, 5, 6), max_dim=2).to(DEV)
        assert bi.cochains[0].x.is_inference() and bi.cochains[0].upper_index.is_inference()
        x0 = [None if bi.cochains[d].x is None else bi.cochains[d].x.clone() for d in range(3)]
        out1 = model(bi).clone()
        for d in range(3):
            bi.cochains[d]._x = x0[d]
        out2 = model(bi).clone()
    assert torch.equal(out1, ref) and torch.equal(out2, ref)
    print(f'[gate] forward under torch.inference_mode() ({which} binding): torch.equal to the no_grad forward, twice')
