"""The blocked backward launch alone on the ZINC batch, graph-replayed, under its debug knobs: the atomic form
(cwn_layer_bwd_f32; CWN_LBWD_DBG: 1 no entry scatter, 2 no MFMA / product atomics, 4 no self terms / boundary transposes,
8 no gY store) or, with a third argument `own`, the owner form (cwn_layer_bwd_own_f32; CWN_LBWD_DBG: 1 no entry walk,
2 no matrix cores, 4 no gY store).
usage: ubench_layer_bwd.py [batch] [F] [own]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import ops
from cwn_amd.complex import ComplexBatch
from cwn_amd.layers import SparseCINConv
from cwn_amd.synthetic import zinc_like_complexes
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
OWN = len(sys.argv) > 3 and sys.argv[3] == 'own'
torch.manual_seed(0)
conv = SparseCINConv(F, F, F, None, None, None, None, max_dim=2, hidden=F, act_module=torch.nn.ReLU, layer_dim=F,
                     use_coboundaries=True).to(dev).train()
b = ComplexBatch.from_complex_list(zinc_like_complexes(B, 1, 6), max_dim=2).to(dev)
for d in range(3):
    b.cochains[d].x = torch.randn(b.cochains[d].num_cells, F, device=dev)
params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
dims, plan, table, key = conv._blocked_args(params, 0, training=True)
rows = [int(D.x.size(0)) for D in dims]
ys_of = [[None, None] for _ in range(3)]
for d in range(2):
    ys_of[d][0] = torch.empty(rows[d], F, device=dev)
    ys_of[d + 1][1] = torch.empty(rows[d + 1], F, device=dev)
outs = ops.LayerLaunch(dims, table).run([D.x for D in dims], 0, ys=[tuple(p) for p in ys_of])
gs = [torch.randn_like(o) for o in outs]
ws = [conv.mp_levels[d].msg_up_nn[1].weight for d in range(2)]
ops.pack_layer_weights_many(ws, transposed=True)
wt_of = [ops.packed_layer_weight_t(ws[0]), ops.packed_layer_weight_t(ws[1]), None]
bwd_table = plan.bwd_items(F, [True, True, False], [D.b_index is not None for D in dims]) if OWN else None
assert not OWN or bwd_table is not None
go = lambda: ops.layer_backward(dims, table, [tuple(p) for p in ys_of], [(gs[2 * d], gs[2 * d + 1]) for d in range(3)], wt_of, bwd_table=bwd_table)
go(); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    go()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        go()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    g.replay()
e1.record(); torch.cuda.synchronize()
print(f'{"own" if OWN else "atomic"} dbg={os.environ.get("CWN_LBWD_DBG", "0")}: {1e3 * e0.elapsed_time(e1) / 60:.2f} us per ({"alloc" if OWN else "fill"} + launch), '
      f'{bwd_table.n_items if OWN else table.n_items} items' + (f', {bwd_table.lds_bytes} B of LDS' if OWN else ''))
