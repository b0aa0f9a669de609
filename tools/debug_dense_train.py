import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import layers, dense_train as DT
dev = torch.device('cuda:0')
torch.manual_seed(0)
H = 128
kw = dict(passed_msg_up_nn=None, passed_msg_boundaries_nn=None, passed_update_up_nn=None,
          passed_update_boundaries_nn=None, train_eps=True, max_dim=2, hidden=H,
          act_module=torch.nn.ReLU, layer_dim=H, graph_norm=torch.nn.BatchNorm1d, use_coboundaries=True)
a = layers.SparseCINConv(H, H, H, **kw).to(dev).train()
b = layers.SparseCINConv(H, H, H, **kw).to(dev).train()
b.load_state_dict(a.state_dict())
Ms = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [500, 700, 90]
outs = [torch.randn(M, H, device=dev) for M in Ms for _ in range(2)]
ws = [torch.randn(M, H, device=dev) for M in Ms]
ups, bds, cbs = [], [], []
for lvl in a.mp_levels:
    st = [layers._mlp_stages(n) for n in (lvl.update_up_nn, lvl.update_boundaries_nn, lvl.combine_nn)]
    ch = [[DT.Stage(l, n) for l, n in s] for s in st]
    ups.append(ch[0]); bds.append(ch[1]); cbs.append(ch[2][0])
oa = [o.clone().requires_grad_() for o in outs]
Hs = DT.dense_train(DT._Plan(ups, bds, cbs), oa)
sum((h * w).sum() for h, w in zip(Hs, ws)).backward()
ob = [o.clone().requires_grad_() for o in outs]
Hr = [b.mp_levels[d].finish(ob[2 * d], ob[2 * d + 1]) for d in range(3)]
sum((h * w).sum() for h, w in zip(Hr, ws)).backward()
for d in range(3):
    print('H', d, float((Hs[d] - Hr[d]).abs().max()))
for k in range(6):
    print('dX', k, float((oa[k].grad - ob[k].grad).abs().max()), float(ob[k].grad.abs().max()))
pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
for n, p in pb.items():
    if p.grad is not None:
        ga = pa[n].grad
        print(n, None if ga is None else float((ga - p.grad).abs().max()), float(p.grad.abs().max()))
