"""Diagnostic: per-parameter gradient distance (product vs float64 oracle vs fp32 oracle) of the molhiv-512 step, with and
without dropout, BCE and L1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_dropout import _molhiv_model, _oracle_cx, _oracle_step, _state, DEV
from tests._product import to_double
from oracle import cwn_oracle as O
from cwn_amd import ops
from cwn_amd.complex import ComplexBatch
from cwn_amd.synthetic import molhiv_like_complexes
from cwn_amd.train import TrainStep

LOSS = os.environ.get('LOSS', 'bce')
for p in (0.5, 0.0, 0.5, 0.0):
    ops.dropout_seed(2024, DEV)
    model = _molhiv_model(p=p)
    b = ComplexBatch.from_complex_list(molhiv_like_complexes(512, 43, 6), max_dim=2)
    b.y = (torch.rand(512, 1, generator=torch.Generator().manual_seed(1)) < 0.3).float()
    b = b.to(DEV)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ocx = _oracle_cx(b)
    sizes = [b.cochains[d].num_cells for d in range(3)]
    ops.DROPOUT_TRACE = []
    ts = TrainStep(model, [b], task_type='bin_classification' if LOSS == 'bce' else 'regression', lr=1e-4, use_graph=os.environ.get('GRAPH', '1') == '1')
    loss = ts.step(0)
    torch.cuda.synchronize()
    trace = list(ops.DROPOUT_TRACE)
    ops.DROPOUT_TRACE = None
    seed, step = _state()
    if p > 0:
        sites = [t[0] for t in trace[-7:]]
        ref_loss, ref_g = _oracle_step(state, ocx, b.y.cpu(), sites, step, seed, sizes, 512, 128)
        _, g32 = _oracle_step(state, ocx, b.y.cpu(), sites, step, seed, sizes, 512, 128, dtype=torch.float32)
    else:
        def plain(dtype):
            leaves = {k: v.to(dtype).clone().requires_grad_(True) for k, v in state.items() if v.is_floating_point() and 'running' not in k}
            st = dict(to_double(state)) if dtype == torch.float64 else dict(state)
            st.update(leaves)
            out, _ = O.sparse_cin_model_forward(st, ocx, 2, use_coboundaries=True, training=True, norm='bn', embed='ogb', readout='mean')
            yy = b.y.cpu().to(dtype).view(out.shape)
            l = torch.nn.functional.binary_cross_entropy_with_logits(out, yy) if LOSS == 'bce' else (out - yy).abs().mean()
            l.backward()
            return l.detach(), {k: v.grad for k, v in leaves.items()}
        ref_loss, ref_g = plain(torch.float64)
        _, g32 = plain(torch.float32)
    rows = []
    for name, q in model.named_parameters():
        r = ref_g[name]
        if r is None:
            continue
        g = q.grad.detach().cpu().double()
        scale = max(1.0, float(r.abs().max()))
        rows.append((float((g - r).abs().max()) / scale, float((g32[name].double() - r).abs().max()) / scale, name, float(r.abs().max()),
                     float((g - r).norm() / max(1e-30, r.norm()))))
    d2 = sum(float(((q.grad.detach().cpu().double() - ref_g[n]) ** 2).sum()) for n, q in model.named_parameters() if ref_g[n] is not None)
    d32 = sum(float(((g32[n].double() - ref_g[n]) ** 2).sum()) for n, q in model.named_parameters() if ref_g[n] is not None)
    n2 = sum(float((ref_g[n] ** 2).sum()) for n, q in model.named_parameters() if ref_g[n] is not None)
    print(f'=== p = {p} loss {LOSS}: whole-gradient relative L2: product {(d2 / n2) ** 0.5:.3e}  fp32 oracle {(d32 / n2) ** 0.5:.3e}')
    rows.sort(reverse=True)
    print(f'--- p = {p}: loss {float(loss):.8f} vs {float(ref_loss):.8f}')
    for e, e32, n, m, rl in rows[:3]:
        print(f'  {e:.3e}  fp32 {e32:.3e}  |ref| {m:.3e}  relL2 {rl:.3e}  {n}')
