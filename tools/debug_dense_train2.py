import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import layers, dense_train as DT
from cwn_amd.synthetic import zinc_like_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
H = 128
kw = dict(passed_msg_up_nn=None, passed_msg_boundaries_nn=None, passed_update_up_nn=None,
          passed_update_boundaries_nn=None, train_eps=True, max_dim=2, hidden=H,
          act_module=torch.nn.ReLU, layer_dim=H, graph_norm=torch.nn.BatchNorm1d, use_coboundaries=True)
a = layers.SparseCINConv(H, H, H, **kw).to(dev).train()
c = layers.SparseCINConv(H, H, H, **kw).to(dev).train()
c.load_state_dict(a.state_dict())
b = zinc_like_batch(24, seed=5, device=dev)
g = torch.Generator().manual_seed(1)
xs = [torch.randn(b.cochains[d].num_cells, H, generator=g).to(dev) for d in range(3)]
ws = [torch.randn(b.cochains[d].num_cells, H, generator=g).to(dev) for d in range(3)]
grads_at_outs = {}

def run(conv, fused_on, tag):
    layers.FUSED_DENSE_TRAINING = fused_on
    xin = [x.clone().requires_grad_() for x in xs]
    b.set_xs(xin)
    params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
    plans, outs = conv.propagate_all(*params)
    for k, o in enumerate(outs):
        o.register_hook(lambda gr, k=k: grads_at_outs.__setitem__((tag, k), gr.clone()))
    dense = conv._dense_train(plans, outs, 0) if fused_on else None
    if dense is None:
        dense = [conv.mp_levels[d].finish(outs[2 * d], outs[2 * d + 1]) for d in range(3)]
    sum((o * w).sum() for o, w in zip(dense, ws)).backward()
    return dense, xin

of, xf = run(a, True, 'f')
op, xp = run(c, False, 'p')
for d in range(3):
    print('H', d, float((of[d] - op[d]).abs().max()))
for k in range(6):
    print('d outs', k, float((grads_at_outs[('f', k)] - grads_at_outs[('p', k)]).abs().max()))
for d in range(3):
    print('dx', d, float((xf[d].grad - xp[d].grad).abs().max()), float(xp[d].grad.abs().max()))
of2, xf2 = run(a, True, 'f2')
for d in range(3):
    print('dx fused again', d, float((xf[d].grad - xf2[d].grad).abs().max()))
op2, xp2 = run(c, False, 'p2')
for d in range(3):
    print('dx plain again', d, float((xp[d].grad - xp2[d].grad).abs().max()))
