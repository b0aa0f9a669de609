"""Phase timing of the owner form of the blocked backward launch (csrc/cwn_layer_bwd_own.hip) from the instrumented build:
    make -C cwn_amd/csrc bwdtiming && CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_bwdtiming.so python tools/time_layer_bwd_phases.py
Wave 0 (first product) and the first wave of the second product stamp s_memtime at the end of each phase; this prints, per
set, the mean / max phase durations (shader clock cycles; ~2.4 GHz) and the record of the slowest workgroups.
usage: time_layer_bwd_phases.py [batch] [F]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cwn_amd import _ffi, ops                               # noqa: E402
from cwn_amd.complex import ComplexBatch                    # noqa: E402
from cwn_amd.layers import SparseCINConv                    # noqa: E402
from cwn_amd.synthetic import zinc_like_complexes           # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device('cuda:0')
L = _ffi.lib()
assert hasattr(L, 'cwn_layer_bwd_own_debug_stamps'), 'needs the timing build (CWN_HIP_LIB=.../libcwn_hip_bwdtiming.so)'
L.cwn_layer_bwd_own_debug_stamps.argtypes = [C.c_void_p]
L.cwn_layer_bwd_own_debug_stamps.restype = None
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
bt = plan.bwd_items(F, [True, True, False], [D.b_index is not None for D in dims])
stamps = torch.zeros(bt.n_items, 32, dtype=torch.int64, device=dev)
L.cwn_layer_bwd_own_debug_stamps(stamps.data_ptr())
go = lambda: ops.layer_backward(dims, table, [tuple(p) for p in ys_of], [(gs[2 * d], gs[2 * d + 1]) for d in range(3)], wt_of, bwd_table=bt)
for _ in range(5):
    go()
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.int64)
tab = bt.host
names = ['record', 'requests', 'rows in', 'own walk', 'top walk', 'barrier 2', 'gy + planes', 'barrier 3', 'product(s)', 'barrier X',
         'adds / third', 'barrier Y', 'dx out']
t0 = st[:, 0].min()
print(f'{bt.n_items} items, LDS {bt.lds_bytes} B; launch span {(st[:, [13, 29]].max() - t0) } cycles; start spread {(st[:, 0].max() - t0) } cycles')
for s_ in sorted(set((tab[:, 0] >> 8).tolist())):
    sel = (tab[:, 0] >> 8) == s_
    for w, off in (('first-product wave', 0), ('second-product wave', 16)):
        d = np.diff(st[sel][:, off:off + 14], axis=1)
        tot = (st[sel][:, off + 13] - st[sel][:, off])
        print(f'set {s_} ({sel.sum()} items), {w}: total mean {tot.mean():.0f} max {tot.max():.0f} cycles')
        print('   ' + '  '.join(f'{n} {d[:, k].mean():.0f}/{d[:, k].max():.0f}' for k, n in enumerate(names)))
order = np.argsort(-(st[:, 13] - st[:, 0]))[:4]
for i in order:
    print('slow item', i, 'record', tab[i, :15].tolist(), 'cycles per phase (wave 0):', np.diff(st[i, :14]).tolist(),
          '(second):', np.diff(st[i, 16:30]).tolist())
