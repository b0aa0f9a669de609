"""The training step pinned AT THE SIZE THAT IS TIMED (VERDICT r3 "what's weak" #1): BASELINE configs[1] exactly -- EmbedSparseCIN,
hidden 128, 4 layers, batch 128, BatchNorm in training mode, L1 loss -- and configs[2] (molhiv-like, batch 512, hidden 64,
2 layers, mean readout), one `TrainStep.step` through the captured graph against torch autograd over the ORACLE's forward in
float64 on the CPU (exp/train_utils.py:57-75 is the loop, mp/test_layers.py:72-112 the reference's own gradient test):
the loss, every parameter gradient, and the parameters after the Adam step.  These are the kernels `secondary.train_step`
times: cwn_layer_fused_f32 (STORE_Y), cwn_dense_stage_f32 / _bwd_f32 on the packed bf16-split blocks,
cwn_layer_bwd_own_f32, the merged cwn_gemm_tn_f32 launches, cwn_head_f32 / _bwd_f32, cwn_adam_f32."""
import numpy as np
import pytest
import torch

from oracle import cwn_oracle as O
from tests._product import gate, to_double

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _oracle_cx(b):
    cpu = lambda t: None if t is None else t.detach().cpu()
    return {'dimension': b.dimension, 'y': None, 'num_complexes': b.num_complexes, 'cochains': [
        {k: cpu(b.cochains[d][k]) for k in ('x', 'upper_index', 'lower_index', 'shared_boundaries',
                                            'shared_coboundaries', 'boundary_index', 'y', 'batch')}
        for d in range(b.dimension + 1)]}


@pytest.mark.parametrize('cfg', ['zinc128', 'molhiv512', 'cinpp64'])
def test_training_step_at_the_timed_size_vs_float64_oracle(cfg):
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedCINpp, EmbedSparseCIN, OGBEmbedSparseCIN
    from cwn_amd.synthetic import molhiv_like_complexes, zinc_like_complexes
    from cwn_amd.train import TrainStep
    torch.manual_seed(0)
    if cfg == 'zinc128':          # bench.py's model, exp/scripts/cwn-zinc.sh:14-30
        L = 4
        model = EmbedSparseCIN(28, 4, 1, L, 128, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum',
                               train_eps=False, final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True,
                               use_coboundaries=True, graph_norm='bn')
        b = ComplexBatch.from_complex_list(zinc_like_complexes(128, 41, 6), max_dim=2)
        okw = dict(embed='zinc')
    elif cfg == 'cinpp64':        # round 4: the CIN++ stack (mp/molec_models.py:167-199), three update networks per dimension on the
        L = 2                     # stage kernels (a plan without combine stages), the 3F-wide combine on torch; trained eps
        model = EmbedCINpp(28, 4, 1, L, 64, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum',
                           train_eps=True, final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True,
                           use_coboundaries=True, graph_norm='bn')
        b = ComplexBatch.from_complex_list(zinc_like_complexes(64, 47, 6), max_dim=2)
        okw = dict(embed='zinc', conv='cinpp')
    else:                         # exp/scripts/cwn-molhiv.sh:9-32 at BASELINE's batch of 512
        L = 2
        model = OGBEmbedSparseCIN(1, L, 64, dropout_rate=0.0, max_dim=2, readout='mean', final_readout='sum', init_reduce='sum',
                                  embed_edge=True, use_coboundaries=True, graph_norm='bn')
        b = ComplexBatch.from_complex_list(molhiv_like_complexes(512, 43, 6), max_dim=2)
        b.y = torch.randn(512, 1, generator=torch.Generator().manual_seed(1))
        okw = dict(embed='ogb', readout='mean')
    model = model.to(DEV).train()
    b = b.to(DEV)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # ---- the oracle in float64: forward in training mode, L1 loss, autograd
    ocx = _oracle_cx(b)
    leaves = {k: v.double().requires_grad_(True) for k, v in state.items() if v.is_floating_point() and 'running' not in k}
    ostate = dict(to_double(state))
    ostate.update(leaves)
    ref_out, _ = O.sparse_cin_model_forward(ostate, ocx, L, use_coboundaries=True, training=True, norm='bn', **okw)
    y = b.y.detach().cpu().double().view(ref_out.shape)
    ref_loss = (ref_out - y).abs().mean()
    ref_loss.backward()
    # ... and in float32: what the REFERENCE's own arithmetic (torch fp32 on the CPU) makes of the same step.  At this size a
    # step holds ~1e8 ReLU pre-activations and four layers of BatchNorm backward (differences of large sums): fp32 gradients
    # of any implementation sit 1e-4 .. 3e-3 from the float64 ones (measured below, printed), so the bar for the product is
    # the reference's own distance, not 1e-5.
    leaves32 = {k: v.float().clone().requires_grad_(True) for k, v in state.items() if v.is_floating_point() and 'running' not in k}
    st32 = dict(state)
    st32.update(leaves32)
    out32, _ = O.sparse_cin_model_forward(st32, ocx, L, use_coboundaries=True, training=True, norm='bn', **okw)
    (out32 - b.y.detach().cpu().float().view(out32.shape)).abs().mean().backward()
    # ---- the product: ONE step of the captured training graph
    lr = 1e-3
    ts = TrainStep(model, [b], task_type='regression', lr=lr, use_graph=True)
    p0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    loss = ts.step(0)
    torch.cuda.synchronize()
    gate(loss.detach().view(1), ref_loss.detach().view(1), f'{cfg}: training loss (BatchNorm batch statistics) vs float64 oracle')
    worst, above, n_par = 0.0, [], 0
    worst32, above32 = 0.0, 0
    per_tensor = []
    d2, d2_32, n2 = 0.0, 0.0, 0.0
    for name, p in model.named_parameters():
        r = leaves[name].grad
        if r is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        n_par += 1
        g, r, r32 = p.grad.detach().cpu().double(), r.double(), leaves32[name].grad.double()
        scale = max(1.0, float(r.abs().max()))
        err, err32 = float((g - r).abs().max()), float((r32 - r).abs().max())
        worst, worst32 = max(worst, err / scale), max(worst32, err32 / scale)
        above32 += err32 > 1e-5 * scale
        if err > 1e-5 * scale:
            above.append((name, err, scale))
            q = lambda t: float(torch.quantile(t.abs().flatten()[:1 << 24], 0.95)) / scale if t.numel() > 1 else float(t.abs().max()) / scale
            per_tensor.append((err / scale, err32 / scale, name, q(g - r), q(r32 - r)))
        d2, d2_32, n2 = d2 + float(((g - r) ** 2).sum()), d2_32 + float(((r32 - r) ** 2).sum()), n2 + float((r ** 2).sum())
    rel, rel32 = (d2 / n2) ** 0.5, (d2_32 / n2) ** 0.5
    print(f'[gate] {cfg}: {n_par} parameter gradients of one training step vs float64 oracle autograd: worst max|delta| / max(1, |ref|_inf) '
          f'= {worst:.3e} (the fp32 oracle itself: {worst32:.3e}); above 1e-5: {len(above)} (fp32 oracle: {above32}); relative L2 distance '
          f'of the whole gradient {rel:.3e} (fp32 oracle: {rel32:.3e})')
    # per tensor: every gradient the product has above 1e-5, with the fp32 oracle's own distance on the same
    # tensor (maximum and 95th percentile).  FINDING (zinc128): on 12 of the 127 tensors the product is 10 - 2000 x further than
    # the fp32 oracle -- all of them the networks of ONE branch of one layer (convs.3.mp_levels.1.update_boundaries_nn.*,
    # convs.1.mp_levels.2.*): the weight of the second Linear is off in ONE row (max 3.5e-4, rms 1.3e-5, top singular share
    # 0.59), beta.grad of the BatchNorm behind it by 7.8e-5 and gamma.grad by 4e-8, and the first Linear of the branch everywhere
    # by ~3e-5 -- the signature of ONE ReLU whose pre-activation lies within rounding of zero taking the other branch than in
    # float64 (its gradient reaches beta but, xhat gamma + beta ~ 0, not gamma; through the BatchNorm backward of the stage in
    # front it moves s1 / s2 and with them every row).  Identical with every GEMM on the exact fp32 MFMA kernels and with the
    # two-kernel backward (tools/diag_grad_outliers.py, CWN_GEMM_SPLIT=0 CWN_TN_SPLIT=0 CWN_STAGE_KERNEL=0): not the split, not a
    # kernel -- torch's fp32 has its own such units elsewhere (its worst tensor is 2.7e-3).  No per-tensor bar can tell a
    # flipped unit from an error of that size, so the per-tensor list is REPORTED and the bars below stay on the whole gradient;
    # a per-tensor cap of 5e-3 catches what is not rounding.
    per_tensor.sort(reverse=True)
    lone = [t for t in per_tensor if t[1] <= 1e-5]
    print(f'[gate] {cfg}: {len(per_tensor)} tensors above 1e-5, {len(lone)} of them where the fp32 oracle is not; the worst ten '
          f'(product max / fp32 oracle max / product p95 / fp32 oracle p95 / name): ' +
          '; '.join(f'{e:.2e} / {e32:.2e} / {q:.2e} / {q32:.2e} / {n_}' for e, e32, n_, q, q32 in per_tensor[:10]))
    far = [(e, e32, n_) for e, e32, n_, q, q32 in per_tensor if e > 2e-5 and e > 4.0 * e32]
    print(f'[gate] {cfg}: {len(far)} tensors where the product is > 2e-5 and > 4 x the fp32 oracle (flipped ReLU units, see the test): ' +
          '; '.join(f'{e:.2e} / {e32:.2e} / {n_}' for e, e32, n_ in far[:12]))
    assert all(e <= 5e-3 for e, _, _, _, _ in per_tensor), per_tensor[:3]
    # the bar: the north star's 1e-5 . max(1, |ref|_inf) where the reference's own fp32 arithmetic meets it, else no further
    # from the float64 gradient than twice what that arithmetic is
    assert worst <= 2.0 * max(worst32, 1e-5), (worst, worst32)
    assert rel <= 2.0 * max(rel32, 1e-6), (rel, rel32)
    # ---- the Adam step (torch.optim.Adam, first step: m = (1 - b1) g, v = (1 - b2) g^2) from the product's own gradient, in float64
    b1, b2, eps = 0.9, 0.999, 1e-8
    worst_p = 0.0
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().cpu().double()
        m, v = (1 - b1) * g, (1 - b2) * g * g
        want = p0[name].cpu().double() - (lr / (1 - b1)) * (m / (v.sqrt() / np.sqrt(1 - b2) + eps))
        worst_p = max(worst_p, float((p.detach().cpu().double() - want).abs().max()))
    print(f'[gate] {cfg}: parameters after the Adam step vs float64 Adam on the same gradient: max|delta| = {worst_p:.3e} (lr {lr})')
    assert worst_p <= 2e-3 * lr, worst_p
