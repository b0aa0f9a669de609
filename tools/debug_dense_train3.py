import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import layers, dense_train as DT, ops, _ffi
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
b.set_xs(xs)
params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
with torch.no_grad():
    plans, outs = a.propagate_all(*params)
outs = [o.detach().clone() for o in outs]
print('out stats', [(float(o.mean()), float(o.std()), float(o.abs().max())) for o in outs[:2]])
# fused
ups, bds, cbs = [], [], []
for lvl in a.mp_levels:
    st = [layers._mlp_stages(n) for n in (lvl.update_up_nn, lvl.update_boundaries_nn, lvl.combine_nn)]
    ch = [[DT.Stage(l, n) for l, n in s] for s in st]
    ups.append(ch[0]); bds.append(ch[1]); cbs.append(ch[2][0])
oa = [o.clone().requires_grad_() for o in outs]
Hs = DT.dense_train(DT._Plan(ups, bds, cbs), oa)
sum((h * w).sum() for h, w in zip(Hs, ws)).backward()
# torch modules with intermediates for dim 0
ob = [o.clone().requires_grad_() for o in outs]
inter = {}
def fwd(lvl, up, bd, d):
    t = up
    for k, m in enumerate(lvl.update_up_nn):
        t = m(t)
        if d == 0:
            t.retain_grad(); inter[('up', k)] = t
    u = bd
    for k, m in enumerate(lvl.update_boundaries_nn):
        u = m(u)
        if d == 0:
            u.retain_grad(); inter[('bd', k)] = u
    v = torch.cat([t, u], -1)
    for k, m in enumerate(lvl.combine_nn):
        v = m(v)
        if d == 0:
            v.retain_grad(); inter[('cb', k)] = v
    return v
Hr = [fwd(c.mp_levels[d], ob[2 * d], ob[2 * d + 1], d) for d in range(3)]
sum((h * w).sum() for h, w in zip(Hr, ws)).backward()
for k in range(6):
    print('dX', k, float((oa[k].grad - ob[k].grad).abs().max()), float(ob[k].grad.abs().max()))
for key, t in inter.items():
    print(key, 'val absmax', float(t.abs().max()), 'grad absmax', float(t.grad.abs().max()),
          'frac zero grad', float((t.grad == 0).float().mean()))
pa, pb = dict(a.named_parameters()), dict(c.named_parameters())
for n, p in pb.items():
    if p.grad is not None and n.startswith('mp_levels.0'):
        print(n, float((pa[n].grad - p.grad).abs().max()), float(p.grad.abs().max()))

print('---- kernel pieces on the dim-0 combine data')
from cwn_amd.dense_train import _norm_desc
z = inter[('cb', 0)].detach().contiguous()
bn = c.mp_levels[0].combine_nn[1]
M, N = z.shape
mean = z.double().mean(0); var = z.double().var(0, unbiased=False)
rstd = 1.0 / torch.sqrt(var + bn.eps)
aff = torch.stack([(bn.weight.double() * rstd), (bn.bias.double() - mean * bn.weight.double() * rstd), mean, rstd]).float().contiguous()
dH = ws[0].contiguous()
s12 = torch.zeros(2, N, device=dev)
dZ = torch.empty(M, N, device=dev)
_ffi.norm_bwd_reduce([_norm_desc(z, dy=dH, aff=aff, s12=s12)], dev)
_ffi.norm_bwd_apply([_norm_desc(z, dy=dH, out=dZ, aff=aff, s12=s12)], dev)
print('s1 err', float((s12[0] - bn.bias.grad).abs().max()), 's2 err', float((s12[1] - bn.weight.grad).abs().max()))
print('dZ err', float((dZ - inter[('cb', 0)].grad).abs().max()))
print('fused dbeta err per col (top5)', torch.topk((pa['mp_levels.0.combine_nn.1.bias'].grad - bn.bias.grad).abs(), 5))
