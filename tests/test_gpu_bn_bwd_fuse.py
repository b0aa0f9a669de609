"""Round 6: the reduce half of a conv layer's OUTPUT BatchNorm backward (d beta, d gamma: two column sums over dy) taken over by the
blocked backward launch of the NEXT layer (cwn_layer_bwd_dim.out_bn, ABI 23; ops.bn_out_register) -- exp/train_utils.py:57-75 is
the loop, mp/layers.py:322-325 the stage.  The same step with the take-over on and off: every gradient agrees (the sums are the
same numbers added in another order: fp32 atomics), the counter shows which path ran, and the cases that must NOT be taken over
(an output dropout, a dy that autograd has added something to) are not."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _grads(model, batch, fuse, use_graph=False):
    from cwn_amd import ops
    from cwn_amd.train import TrainStep
    keep = ops.BN_BWD_FUSE
    ops.BN_BWD_FUSE = fuse
    try:
        ts = TrainStep(model, [batch], task_type='regression', use_graph=use_graph, lr=0.0)
        n0 = ops.BN_BWD_FUSED[0]
        loss = ts.step(0)
        torch.cuda.synchronize()
        taken = ops.BN_BWD_FUSED[0] - n0
        g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    finally:
        ops.BN_BWD_FUSE = keep
    return float(loss), g, taken


@pytest.mark.parametrize('hidden,layers,n_lo,n_hi', [(128, 4, 18, 30), (64, 3, 18, 30), (128, 2, 4, 8), (64, 2, 4, 7)])
def test_reduce_of_the_output_batchnorm_taken_over_by_the_next_layers_backward(hidden, layers, n_lo, n_hi):
    """(molecules of 4 - 8 atoms: items whose staged region is smaller than the sixteen waves' partial sums -- the kernel's
    second way through LDS)"""
    import copy
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.synthetic import zinc_like_complexes
    torch.manual_seed(3)
    model = EmbedSparseCIN(28, 4, 1, layers, hidden, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(DEV).train()
    b = ComplexBatch.from_complex_list(zinc_like_complexes(96, 5, 6, n_lo=n_lo, n_hi=n_hi), max_dim=2).to(DEV)
    b.y = torch.randn(b.num_complexes, 1, device=DEV)
    m_off, m_on = copy.deepcopy(model), copy.deepcopy(model)
    l_off, g_off, t_off = _grads(m_off, b, fuse=False)
    l_on, g_on, t_on = _grads(m_on, b, fuse=True)
    assert t_off == 0
    # every conv layer but the last hands its three outputs to a blocked backward: 3 stages per layer taken over
    assert t_on == 3 * (layers - 1), t_on
    assert abs(l_on - l_off) <= 1e-6 * max(1.0, abs(l_off))
    assert g_on.keys() == g_off.keys() and len(g_on) > 20
    worst = 0.0
    for n in g_on:
        ref = g_off[n]
        err = float((g_on[n] - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        worst = max(worst, err)
        assert err <= 2e-5, (n, err)
    print(f'[gate] BN reduce taken over, hidden {hidden} x {layers}, molecules of {n_lo} - {n_hi} atoms: {t_on} stages, max gradient difference {worst:.2e} (relative to max(1, |g|))')


def test_take_over_survives_graph_capture_and_replay():
    """The slot sums live in the step's zeroed scratch: a captured step replayed again and again must see them zero every time
    (sums that survived a replay would grow d beta / d gamma of every taken-over stage with each step)."""
    import copy
    from cwn_amd import ops
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.synthetic import zinc_like_complexes
    from cwn_amd.train import TrainStep
    torch.manual_seed(4)
    model = EmbedSparseCIN(28, 4, 1, 3, 64, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(DEV).train()
    bs = [ComplexBatch.from_complex_list(zinc_like_complexes(48, 7 + i, 6), max_dim=2).to(DEV) for i in range(2)]
    for b in bs:
        b.y = torch.randn(b.num_complexes, 1, device=DEV)
    m_e, m_g = copy.deepcopy(model), copy.deepcopy(model)
    te = TrainStep(m_e, bs, task_type='regression', use_graph=False, lr=0.0)
    tg = TrainStep(m_g, bs, task_type='regression', use_graph=True, lr=0.0)
    n0 = ops.BN_BWD_FUSED[0]
    te.step(0)
    assert ops.BN_BWD_FUSED[0] - n0 == 6
    ge = {n: p.grad.detach().clone() for n, p in m_e.named_parameters() if p.grad is not None}
    for i in range(7):
        tg.step(i % 2)                 # (warm-up + capture + replays; lr = 0: the parameters do not move)
    tg.step(0)
    torch.cuda.synchronize()
    for n, p in m_g.named_parameters():
        if p.grad is None:
            continue
        err = float((p.grad - ge[n]).abs().max()) / max(1.0, float(ge[n].abs().max()))
        assert err <= 2e-5, (n, err)


def test_not_taken_over_with_an_output_dropout_or_a_second_consumer():
    from cwn_amd import ops
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import OGBEmbedSparseCIN, SparseCIN
    from cwn_amd.synthetic import molhiv_like_complexes, zinc_like_complexes
    from cwn_amd.train import TrainStep
    torch.manual_seed(5)
    # dropout after every conv layer (mp/molec_models.py:298-300): dy arrives w.r.t. the dropped activation
    m = OGBEmbedSparseCIN(1, 2, 64, dropout_rate=0.5, max_dim=2, readout='mean', final_readout='sum', init_reduce='sum',
                          embed_edge=True, use_coboundaries=True, graph_norm='bn').to(DEV).train()
    b = ComplexBatch.from_complex_list(molhiv_like_complexes(64, 3, 6), max_dim=2).to(DEV)
    b.y = torch.zeros(b.num_complexes, 1, device=DEV)
    n0 = ops.BN_BWD_FUSED[0]
    TrainStep(m, [b], task_type='bin_classification', use_graph=False).step(0)
    assert ops.BN_BWD_FUSED[0] == n0
    # jumping knowledge: every layer's output is ALSO read by the head -- autograd adds the head's piece to dx: another tensor
    m2 = SparseCIN(4, 1, 3, 64, dropout_rate=0.0, max_dim=2, jump_mode='cat', readout='sum', use_coboundaries=True, graph_norm='bn').to(DEV).train()
    b2 = ComplexBatch.from_complex_list(zinc_like_complexes(32, 9, 6), max_dim=2).to(DEV)
    for d in range(3):
        c = b2.cochains[d]
        c.x = torch.randn(c.num_cells, 4, device=DEV)
    b2.y = torch.zeros(b2.num_complexes, 1, device=DEV)
    outs = []
    for fuse in (False, True):
        import copy
        mm = copy.deepcopy(m2)
        keep, ops.BN_BWD_FUSE = ops.BN_BWD_FUSE, fuse
        try:
            ts = TrainStep(mm, [b2], task_type='regression', use_graph=False, lr=0.0)
            ts.step(0)
        finally:
            ops.BN_BWD_FUSE = keep
        outs.append({n: p.grad.detach().clone() for n, p in mm.named_parameters() if p.grad is not None})
    for n in outs[0]:
        err = float((outs[0][n] - outs[1][n]).abs().max()) / max(1.0, float(outs[0][n].abs().max()))
        assert err <= 2e-5, (n, err)


def test_a_second_backward_over_the_same_forward_reduces_by_itself():
    """retain_graph: the slots hold the FIRST backward's sums -- the take-over is one shot, the second backward launches its own
    reduce and gives the same gradients."""
    from cwn_amd import ops
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.synthetic import zinc_like_complexes
    torch.manual_seed(6)
    model = EmbedSparseCIN(28, 4, 1, 3, 64, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(DEV).train()
    b = ComplexBatch.from_complex_list(zinc_like_complexes(40, 11, 6), max_dim=2).to(DEV)
    ops.bn_registry_clear()
    loss = model(b).square().mean()
    n0 = ops.BN_BWD_FUSED[0]
    loss.backward(retain_graph=True)
    first = ops.BN_BWD_FUSED[0] - n0
    g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    loss.backward()
    second = ops.BN_BWD_FUSED[0] - n0 - first
    torch.cuda.synchronize()
    assert first == 6 and second == 0, (first, second)
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        err = float((p.grad - g1[n]).abs().max()) / max(1.0, float(g1[n].abs().max()))
        assert err <= 2e-5, (n, err)
