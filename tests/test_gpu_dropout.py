"""Dropout on the device path (round 5; VERDICT r4 "what's missing" #1): exp/scripts/cwn-molhiv.sh trains with --drop_rate 0.5,
mp/molec_models.py:298-300 drops out after every conv layer and :345-346 before lin2.  Here the multipliers are derived inside
the kernels from (seed, step, site, element) -- csrc/cwn_dropout.h -- in the epilogue of the launch that produces a value and
in the prologue of the launch that consumes its gradient; no mask tensor exists.  These tests pin
  * the generator and the keep rule on the numpy restatement (tests/_philox_ref.py, itself pinned on the published Philox
    known answers),
  * every fused application (cwn_norm_act_f32 / cwn_norm_bwd_reduce_f32, cwn_head_f32 / _bwd, cwn_dropout_f32) on the exported
    multipliers,
  * a whole captured training step of config 3 WITH dropout 0.5 on the float64 oracle applying the same multipliers where the
    reference applies F.dropout (tests/golden/dropout.npz pins those places on the reference itself),
  * fresh masks per replay, identity in eval mode, and the static-batch step."""
import numpy as np
import pytest
import torch

from oracle import cwn_oracle as O
from tests._philox_ref import multipliers
from tests._product import gate, to_double

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _state():
    from cwn_amd import ops
    st = ops.dropout_state(DEV).cpu().tolist()
    return int(st[0]), int(st[1])


def test_multipliers_equal_the_philox_restatement_on_both_kernel_paths():
    from cwn_amd import ops
    ops.dropout_seed(0x1234ABCD5678, DEV)
    seed, step = _state()
    assert (seed, step) == (0x1234ABCD5678, 0)
    for shape, p, site in (((300, 64), 0.5, 1), ((77, 128), 0.1, 2), ((16, 3), 0.25, 3), ((5, 7), 0.9, 4), ((1, 4), 0.5, 5)):
        got = ops.dropout_multipliers(shape, p, site, DEV).cpu().numpy()
        want = multipliers(shape, p, seed, 0, site)
        assert np.array_equal(got, want), (shape, p, site, float(np.abs(got - want).max()))
    # another step, another stream; the vector path (N % 4 == 0) and the scalar path number elements alike
    a = ops.dropout_multipliers((6, 8), 0.25, 9, DEV, step=41).cpu().numpy()
    b = ops.dropout_multipliers((16, 3), 0.25, 9, DEV, step=41).cpu().numpy()
    assert np.array_equal(a.reshape(-1), b.reshape(-1)) and np.array_equal(a, multipliers((6, 8), 0.25, seed, 41, 9))
    big = ops.dropout_multipliers((4096, 128), 0.5, 11, DEV).cpu().numpy()
    assert abs((big > 0).mean() - 0.5) < 2e-3 and set(np.unique(big)) == {0.0, 2.0}


def test_dropout_op_forward_and_backward_use_the_same_multipliers_and_eval_is_identity():
    from cwn_amd import ops
    ops.dropout_seed(7, DEV)
    ops.DROPOUT_TRACE = []
    try:
        x = torch.randn(130, 64, device=DEV, requires_grad=True)
        y = ops.dropout(x, 0.3, True)
        (site, p, _), = ops.DROPOUT_TRACE
    finally:
        ops.DROPOUT_TRACE = None
    m = torch.from_numpy(multipliers((130, 64), 0.3, 7, 0, site)).to(DEV)
    assert torch.equal(y.detach(), x.detach() * m)
    g = torch.randn_like(x)
    y.backward(g)
    assert torch.equal(x.grad, g * m)
    assert ops.dropout(x, 0.3, False) is x and ops.dropout(x, 0.0, True) is x
    # a strided input (a column slice) and a 3-D tensor
    z = torch.randn(40, 96, device=DEV)
    ops.DROPOUT_TRACE = []
    try:
        a = ops.dropout(z[:, 32:64], 0.5, True)
        b = ops.dropout(z.view(5, 8, 96), 0.5, True)
        (s1, _, _), (s2, _, _) = ops.DROPOUT_TRACE
    finally:
        ops.DROPOUT_TRACE = None
    assert torch.equal(a, z[:, 32:64] * torch.from_numpy(multipliers((40, 32), 0.5, 7, 0, s1)).to(DEV))
    assert torch.equal(b, z.view(5, 8, 96) * torch.from_numpy(multipliers((5, 8, 96), 0.5, 7, 0, s2)).to(DEV))


def _conv_and_batch(F=64, seed=3, B=24):
    from cwn_amd.layers import SparseCINConv
    from cwn_amd.synthetic import zinc_like_batch
    torch.manual_seed(seed)
    conv = SparseCINConv(F, F, F, None, None, None, None, max_dim=2, hidden=F, act_module=torch.nn.ReLU, layer_dim=F,
                         use_coboundaries=True).to(DEV).train()
    b = zinc_like_batch(B, seed=seed, max_ring=6, device=DEV)
    g = torch.Generator().manual_seed(seed)
    for d in range(3):
        b.cochains[d].x = torch.randn(b.cochains[d].num_cells, F, generator=g).to(DEV)
    return conv, b.prepare(backward=True)


@pytest.mark.parametrize('F,upstream', [(64, 1.0), (128, 1.0), (64, 2.0 ** -9), (128, 2.0 ** -9)])
def test_conv_output_dropout_fused_into_the_activation_and_the_reduce_launches(F, upstream):
    """SparseCINConv.forward(out_dropout=p) in training mode: outputs = (outputs without dropout) x multipliers, bit for bit
    (the activation launch multiplies what it would have stored); input and parameter gradients = those of the same layer
    followed by an explicit multiplication (autograd through the dropout-free fused path).
    `upstream = 2^-9` is the NORMALISED variant: the weight gradients of this layer are sums over ~1e3
    rows (|ref|_inf 50 .. 220) and pass the relative gate at 1e-5 .. 5e-5 absolute; with the upstream gradient scaled by a
    power of two every gradient is O(1) and the ABSOLUTE 1e-5 is asserted."""
    from cwn_amd import ops
    conv, b = _conv_and_batch(F)
    ops.dropout_seed(99, DEV)
    params = lambda: b.get_all_cochain_params(max_dim=2, include_down_features=False)
    xs0 = [b.cochains[d].x.clone() for d in range(3)]
    state0 = {k: v.clone() for k, v in conv.state_dict().items()}

    def run(p):
        conv.load_state_dict(state0)
        conv.zero_grad(set_to_none=True)
        xs = [x.clone().requires_grad_(True) for x in xs0]
        for d in range(3):
            b.cochains[d].x = xs[d]
        ops.DROPOUT_TRACE = []
        try:
            outs = conv(*params(), out_dropout=p)
            trace = list(ops.DROPOUT_TRACE)
        finally:
            ops.DROPOUT_TRACE = None
        return xs, outs, trace

    xs_a, outs_a, trace = run(0.5)
    assert [t[2] for t in trace] == [('conv', 0), ('conv', 1), ('conv', 2)], trace       # applied by dense_train's last launch
    ms = [torch.from_numpy(multipliers(tuple(o.shape), 0.5, 99, 0, t[0])).to(DEV) for o, t in zip(outs_a, trace)]
    w = [torch.randn_like(o) * upstream for o in outs_a]
    sum((o * wi).sum() for o, wi in zip(outs_a, w)).backward()
    ga = [x.grad.clone() for x in xs_a]
    pa = {n: q.grad.clone() for n, q in conv.named_parameters() if q.grad is not None}
    xs_b, outs_b, trace_b = run(0.0)
    assert trace_b == []
    for oa, ob, m in zip(outs_a, outs_b, ms):
        assert torch.equal(oa.detach(), ob.detach() * m)
        assert 0.35 < float((m > 0).float().mean()) < 0.65
    sum((o * m * wi).sum() for o, m, wi in zip(outs_b, ms, w)).backward()
    for d in range(3):
        gate(ga[d], xs_b[d].grad.double(), f'F={F}: dL/dx_{d} through the fused output dropout vs an explicit multiplication')
    refs = {n: q.grad.double() for n, q in conv.named_parameters() if q.grad is not None}
    for n, r in refs.items():
        # (the bias of a Linear in front of a BatchNorm has gradient ZERO in exact arithmetic -- the norm removes the column
        #  mean; both sides hold the rounding residue of a sum of ~1e3 terms of size |dW|: gated at the scale of that sum)
        tol = 1e-5 * max(1.0, float(refs[n[:-4] + 'weight'].abs().max())) if n.endswith('.bias') and n[:-4] + 'weight' in refs else 1e-5
        err = gate(pa[n], r, f'F={F}{"" if upstream == 1.0 else " (normalised)"}: dL/d{n} through the fused output dropout', tol=tol)
        if upstream != 1.0:
            assert float(r.abs().max()) <= 1.0 and err <= 1e-5, (n, float(r.abs().max()), err)
    # eval mode: the argument is ignored
    conv.eval()
    with torch.no_grad():
        ops.DROPOUT_TRACE = []
        try:
            for d in range(3):
                b.cochains[d].x = xs0[d]
            conv(*params(), out_dropout=0.5)
            assert ops.DROPOUT_TRACE == []
        finally:
            ops.DROPOUT_TRACE = None


@pytest.mark.parametrize('pos', ['lin2', 'lin1', 'final_readout'])
def test_head_dropout_positions_forward_and_backward(pos):
    """The fused head (cwn_head_f32 / _bwd) with the dropout of `apply_dropout_before` inside the launch against the unfused
    torch arithmetic of mp/molec_models.py:334-346 with the exported multipliers."""
    from cwn_amd import _ffi, ops
    from cwn_amd.synthetic import zinc_like_batch
    ops.dropout_seed(5, DEV)
    K, H2, O, p = 64, 128, 3, 0.5
    b = zinc_like_batch(20, seed=8, max_ring=6, device=DEV)
    plan = b.block_plan()
    C = b.num_complexes
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(b.cochains[d].num_cells, K, generator=g).to(DEV).requires_grad_(True) for d in range(3)]
    lin1 = [torch.nn.Linear(K, H2).to(DEV) for _ in range(3)]
    lin2 = torch.nn.Linear(H2, O).to(DEV)
    ptrs = [plan.cell_ptr_device(d, DEV) for d in range(3)]
    dpos = {'lin1': _ffi.HEAD_DROP_LIN1, 'final_readout': _ffi.HEAD_DROP_FINAL, 'lin2': _ffi.HEAD_DROP_LIN2}[pos]
    ops.DROPOUT_TRACE = []
    try:
        out, pooled = ops.head_train(xs, ptrs, C, [l.weight for l in lin1], [l.bias for l in lin1], lin2.weight, lin2.bias,
                                     mean_readout=True, mean_final=False, drop_p=p, drop_pos=dpos)
        (site, _, tag), = ops.DROPOUT_TRACE
    finally:
        ops.DROPOUT_TRACE = None
    assert tag == ('head', dpos)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    got = {'out': out.detach().clone(), 'dx': [x.grad.clone() for x in xs], 'dw1': [l.weight.grad.clone() for l in lin1],
           'db1': [l.bias.grad.clone() for l in lin1], 'dw2': lin2.weight.grad.clone(), 'db2': lin2.bias.grad.clone()}
    for t in xs + [q for l in lin1 + [lin2] for q in l.parameters()]:
        t.grad = None
    # the reference's arithmetic in float64
    mk = lambda shape: torch.from_numpy(multipliers(shape, p, 5, 0, site)).to(DEV).double()
    dd = lambda t: t.double()
    pooled_ref = []
    for d in range(3):
        seg = ptrs[d].cpu().tolist()
        pooled_ref.append(torch.stack([dd(xs[d])[seg[c]:seg[c + 1]].mean(0) if seg[c + 1] > seg[c] else torch.zeros(K, device=DEV, dtype=torch.double)
                                       for c in range(C)]))
    if pos == 'lin1':
        m = mk((3, C, K))
        pooled_ref = [pr * m[d] for d, pr in enumerate(pooled_ref)]
    hs = [torch.relu(pr @ dd(l.weight).t() + dd(l.bias)) for pr, l in zip(pooled_ref, lin1)]
    h = torch.stack(hs, 0)
    if pos == 'final_readout':
        h = h * mk((3, C, H2))
    h = h.sum(0)
    if pos == 'lin2':
        h = h * mk((C, H2))
    ref = h @ dd(lin2.weight).t() + dd(lin2.bias)
    (ref * w.double()).sum().backward()
    gate(got['out'], ref.detach(), f'head dropout before {pos}: prediction')
    for d in range(3):
        gate(got['dx'][d], xs[d].grad.double(), f'head dropout before {pos}: dL/dx_{d}')
        gate(got['dw1'][d], lin1[d].weight.grad.double(), f'head dropout before {pos}: dL/dW1_{d}')
        gate(got['db1'][d], lin1[d].bias.grad.double(), f'head dropout before {pos}: dL/db1_{d}')
    gate(got['dw2'], lin2.weight.grad.double(), f'head dropout before {pos}: dL/dW2')
    gate(got['db2'], lin2.bias.grad.double(), f'head dropout before {pos}: dL/db2')


def _oracle_cx(b):
    cpu = lambda t: None if t is None else t.detach().cpu()
    return {'dimension': b.dimension, 'y': None, 'num_complexes': b.num_complexes, 'cochains': [
        {k: cpu(b.cochains[d][k]) for k in ('x', 'upper_index', 'lower_index', 'shared_boundaries',
                                            'shared_coboundaries', 'boundary_index', 'y', 'batch')}
        for d in range(b.dimension + 1)]}


def _molhiv_model(hidden=64, layers=2, p=0.5, seed=0):
    from cwn_amd.models import OGBEmbedSparseCIN
    torch.manual_seed(seed)
    # exp/scripts/cwn-molhiv.sh:9-32: --drop_rate 0.5 --indrop_rate 0.0 --drop_position lin2 --readout mean --final_readout sum
    return OGBEmbedSparseCIN(1, layers, hidden, dropout_rate=p, indropout_rate=0.0, max_dim=2, readout='mean', final_readout='sum',
                             apply_dropout_before='lin2', init_reduce='sum', embed_edge=True, use_coboundaries=True,
                             graph_norm='bn').to(DEV).train()


def _oracle_step(state, ocx, y, sites, step, seed, sizes, C, H2, L=2, p=0.5, dtype=torch.float64):
    """loss and gradients of one training step of the oracle with the multipliers of `sites` (L x 3 conv outputs, then the head)."""
    leaves = {k: v.to(dtype).clone().requires_grad_(True) for k, v in state.items() if v.is_floating_point() and 'running' not in k}
    st = dict(to_double(state)) if dtype == torch.float64 else dict(state)
    st.update(leaves)
    drop = {}
    for l in range(L):
        for d in range(3):
            drop[('conv', l, d)] = torch.from_numpy(multipliers((sizes[d], st['lin2.weight'].size(1) // 2), p, seed, step, sites[3 * l + d])).to(dtype)
    drop[('head',)] = torch.from_numpy(multipliers((C, H2), p, seed, step, sites[3 * L])).to(dtype)
    out, _ = O.sparse_cin_model_forward(st, ocx, L, use_coboundaries=True, training=True, norm='bn', embed='ogb', readout='mean',
                                        dropout=drop, drop_position='lin2')
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, y.to(dtype).view(out.shape))
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in leaves.items()}


def test_config3_training_step_with_dropout_vs_float64_oracle():
    """BASELINE configs[2] as the reference trains it (exp/scripts/cwn-molhiv.sh: OGBEmbedSparseCIN 64 x 2, mean readout,
    dropout 0.5 after every conv layer and before lin2, BCE-with-logits) at the batch of 512: ONE step of the captured
    training graph -- every dropout inside a stage / head launch -- against float64 autograd over the oracle's forward with the
    multipliers the kernels derived (exported through the numpy Philox restatement) applied where the reference applies
    F.dropout.  Then: a second replay draws other masks, eval mode ignores dropout."""
    from cwn_amd import ops
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import molhiv_like_complexes
    from cwn_amd.train import TrainStep
    ops.dropout_seed(2024, DEV)
    model = _molhiv_model()
    b = ComplexBatch.from_complex_list(molhiv_like_complexes(512, 43, 6), max_dim=2)
    b.y = (torch.rand(512, 1, generator=torch.Generator().manual_seed(1)) < 0.3).float()
    b = b.to(DEV)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ocx = _oracle_cx(b)
    sizes = [b.cochains[d].num_cells for d in range(3)]
    ops.DROPOUT_TRACE = []
    try:
        ts = TrainStep(model, [b], task_type='bin_classification', lr=1e-4, use_graph=True)
        loss = ts.step(0)
        torch.cuda.synchronize()
        trace = list(ops.DROPOUT_TRACE)
    finally:
        ops.DROPOUT_TRACE = None
    per_step = 2 * 3 + 1
    assert len(trace) % per_step == 0 and len(trace) >= per_step, trace
    last = trace[-per_step:]
    assert [t[2][0] for t in last] == ['conv'] * 6 + ['head'], last      # no ('tensor', ..) entry: nothing ran as a launch of its own
    seed, step = _state()
    assert seed == 2024 and step >= 1
    sites = [t[0] for t in last]
    ref_loss, ref_g = _oracle_step(state, ocx, b.y.detach().cpu(), sites, step, seed, sizes, 512, 128)
    _, g32 = _oracle_step(state, ocx, b.y.detach().cpu(), sites, step, seed, sizes, 512, 128, dtype=torch.float32)
    gate(loss.detach().view(1), ref_loss.view(1), 'molhiv-512 with dropout 0.5: training loss vs float64 oracle with the same multipliers')
    worst, worst32, d2, d2_32, n2, n_par = 0.0, 0.0, 0.0, 0.0, 0.0, 0
    for name, q in model.named_parameters():
        r = ref_g[name]
        if r is None:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, name
            continue
        n_par += 1
        g, r32 = q.grad.detach().cpu().double(), g32[name].double()
        scale = max(1.0, float(r.abs().max()))
        worst, worst32 = max(worst, float((g - r).abs().max()) / scale), max(worst32, float((r32 - r).abs().max()) / scale)
        d2, d2_32, n2 = d2 + float(((g - r) ** 2).sum()), d2_32 + float(((r32 - r) ** 2).sum()), n2 + float((r ** 2).sum())
    rel, rel32 = (d2 / n2) ** 0.5, (d2_32 / n2) ** 0.5
    print(f'[gate] molhiv-512 with dropout 0.5: {n_par} parameter gradients vs float64 oracle autograd: worst max|delta| / max(1, |ref|_inf) = '
          f'{worst:.3e} (fp32 oracle: {worst32:.3e}); relative L2 {rel:.3e} (fp32 oracle: {rel32:.3e})')
    # The bar.  A WRONG multiplier anywhere in the backward moves the gradient by O(1) relative (half the entries of a stream
    # dropped or not); rounding moves it by what the reference's own fp32 arithmetic moves it -- and at p = 0.5 that depends on
    # the masks drawn: measured over mask realisations (tools/diag_dropout_grads.py) the whole-gradient relative L2 distance is
    # 1.3e-4 .. 4.0e-4 for the product and 0.9e-4 .. 1.7e-4 for the fp32 oracle, the worst entry 1.6e-5 .. 7.7e-5 vs 1.0e-5 ..
    # 1.6e-5 (single ReLU / BatchNorm-sensitive entries: the first-layer edge networks).  So: no further from float64 than
    # twice the fp32 oracle OR the absolute floors below, which sit three orders of magnitude under a mask error.  (The
    # dropout-free step keeps its tight bar: tests/test_gpu_train_full.py.)
    assert worst <= max(2.0 * worst32, 2e-4), (worst, worst32)
    assert rel <= max(2.0 * rel32, 1e-3), (rel, rel32)
    # a second replay: the step counter moved, the masks with it (the loss of the same batch differs beyond one Adam step's reach)
    l1 = float(loss)                         # (the graph's loss tensor: the next replay overwrites it)
    l2 = float(ts.step(0))
    torch.cuda.synchronize()
    assert _state()[1] == step + 1
    m_a, m_b = multipliers((sizes[0], 64), 0.5, seed, step, sites[0]), multipliers((sizes[0], 64), 0.5, seed, step + 1, sites[0])
    assert 0.4 < float((m_a != m_b).mean()) < 0.6
    assert abs(l2 - l1) > 1e-6
    # eval: identity, no site is drawn
    model.eval()
    ops.DROPOUT_TRACE = []
    try:
        with torch.no_grad():
            b2 = ComplexBatch.from_complex_list(molhiv_like_complexes(64, 44, 6), max_dim=2).to(DEV)
            p1 = model(b2).clone()
            b3 = ComplexBatch.from_complex_list(molhiv_like_complexes(64, 44, 6), max_dim=2).to(DEV)
            p2 = model(b3)
        assert ops.DROPOUT_TRACE == [] and torch.equal(p1, p2)
    finally:
        ops.DROPOUT_TRACE = None


def test_static_train_step_with_dropout_vs_float64_oracle():
    """The never-seen-batch path (StaticTrainStep: device-side collate, row counts in device memory, S steps per replay) with
    dropout 0.5: the loss of every slot's step against the float64 oracle on the collated batch with the multipliers of that
    step (element numbering does not depend on the capacity of the buffers), model state advanced by the product's own steps."""
    from cwn_amd import csr, ops
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    from cwn_amd.synthetic import molhiv_like_complexes
    ops.dropout_seed(31337, DEV)
    pool = molhiv_like_complexes(200, 5, 6)
    for i, c in enumerate(pool):
        c.y = torch.tensor([[float(i % 3 == 0)]])
    p = PackedComplexes(pool, DEV, max_dim=2, with_csr=True)
    B, S = 48, 2
    model = _molhiv_model(seed=2)
    sb = StaticBatch(p, B, slots=S)
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(pool))
    batches = [perm[0:B], perm[B:2 * B]]
    assert sb.fits(batches).all()
    ops.DROPOUT_TRACE = []
    try:
        st = StaticTrainStep(model, sb, task_type='bin_classification', lr=1e-4)
        sb.set_epoch(batches)
        state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        losses = [l.clone() for l in st.step()]
        torch.cuda.synchronize()
        trace = list(ops.DROPOUT_TRACE)
    finally:
        ops.DROPOUT_TRACE = None
    csr.check_errors(DEV)
    per_step = 7
    last = trace[-S * per_step:]
    assert [t[2][0] for t in last] == (['conv'] * 6 + ['head']) * S, last
    seed, step = _state()
    # slot 0's step against the oracle from the state before the replay (slot 1 starts from a state the product produced)
    ref = p.collate(batches[0])
    ocx = _oracle_cx(ref)
    sizes = [ref.cochains[d].num_cells for d in range(3)]
    ref_loss, _ = _oracle_step(state0, ocx, ref.y.detach().cpu(), [t[0] for t in last[:per_step]], step - (S - 1), seed, sizes, B, 128)
    gate(losses[0].view(1), ref_loss.view(1), 'static batch, slot 0, dropout 0.5: training loss vs float64 oracle')
    assert all(torch.isfinite(l) for l in losses)
