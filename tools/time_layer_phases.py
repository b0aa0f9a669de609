"""Phase timing of the complex-blocked layer kernel (csrc/cwn_layer.hip) from the instrumented build:
    make -C cwn_amd/csrc timing && CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_timing.so python tools/time_layer_phases.py
Every workgroup stamps s_memtime at the end of each phase; this prints, per item kind, the mean / max
of the phase durations and of the whole workgroup, and the spread of workgroup start times."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cwn_amd import _ffi, layers, ops                      # noqa: E402
from cwn_amd.complex import ComplexBatch                    # noqa: E402
from cwn_amd.layers import SparseCINConv                    # noqa: E402
from cwn_amd.synthetic import zinc_like_complexes           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device('cuda:0')
L = _ffi.lib()
assert hasattr(L, 'cwn_layer_debug_stamps'), 'needs the timing build (CWN_HIP_LIB=.../libcwn_hip_timing.so)'
L.cwn_layer_debug_stamps.argtypes = [C.c_void_p]
L.cwn_layer_debug_stamps.restype = None
b = ComplexBatch.from_complex_list(zinc_like_complexes(n, 0, 6), max_dim=2).to(dev)
for d in range(3):
    b.cochains[d].x = torch.randn(b.cochains[d].num_cells, F, device=dev)
torch.manual_seed(0)
conv = SparseCINConv(F, F, F, None, None, None, None, max_dim=2, hidden=F, act_module=torch.nn.ReLU,
                     layer_dim=F, use_coboundaries=True).to(dev).eval()
params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
with torch.no_grad():
    args = conv._blocked_args(params, 0)
assert not isinstance(args, str), args
dims, plan, table, key = args
items, mr, ms = table.items, table.max_rows, table.max_src
MODE = int(os.environ.get('MODE', '0'))   # 0 sort, 1 sort + store, 2 load (after one storing launch)
REP = 1
stamps = torch.zeros(items.size(0), 96, dtype=torch.int64, device=dev)      # (= CWN_STAMP_REC)
L.cwn_layer_debug_stamps(stamps.data_ptr())
with torch.no_grad():
    if MODE == 2:
        ops.layer_fused(dims, table, 1)
    for _ in range(5):
        ops.layer_fused(dims, table, MODE)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.layer_fused(dims, table, MODE)
    e1.record()
    torch.cuda.synchronize()
print(f'items {items.size(0)}  rows_cap {mr}  src_cap {ms}  lds {L.cwn_layer_fused_lds_bytes(F, mr, ms)}  '
      f'eager launch avg {e0.elapsed_time(e1) / 20 * 1e3:.2f} us')
st = stamps.cpu().numpy().astype(np.int64)
it = items.cpu().numpy()
names = ['item+issue loads', 'entries->LDS', 'rank', 'rowptr+split+stage', 'boundary+self', 'MFMA', 'Y write', 'upper reduce']
# (the shader-clock counters of different XCDs are not synchronised: no chip-wide span from the stamps)
n1 = it.shape[0] // REP
rnd = np.arange(it.shape[0]) // n1
for kind, mask in [(f'g={g} items, round {r}', (it[:, 0] & 1 == 1) & (it[:, 1] == g) & (rnd == r)) for r in range(REP) for g in (0, 1)]:
    s = st[mask]
    if not len(s):
        continue
    d = np.diff(s[:, :9], axis=1)
    print(f'{kind}: {len(s)} workgroups, whole {np.mean(s[:, 8] - s[:, 0]):.0f} mean / {np.max(s[:, 8] - s[:, 0])} max ticks')
    for k, nm in enumerate(names):
        print(f'   {nm:22s} mean {d[:, k].mean():8.0f}  max {d[:, k].max():8d}')
    sub = s[:, [0, 9, 10, 11, 12, 1]]
    if (sub[:, 1:5] > 0).all():
        dd = np.diff(sub, axis=1)
        for k, nm in enumerate(['item record arrives', 'set record (scalar)', 'check + entries/eps/bias issued', 'W issued', 'rows issued']):
            print(f'      - {nm:32s} mean {dd[:, k].mean():8.0f}  max {dd[:, k].max():8d}')
# per-wave points (16-wave build): when each wave starts, has its row requests out, has its rows staged
w = st[:, 16:64].reshape(-1, 3, 16)
if (w[:, 0, :] > 0).all():
    rel = w - st[:, 0][:, None, None]
    for k, nm in enumerate(['wave start', 'row requests out', 'rows split + staged']):
        r = rel[:, k, :]
        print(f'{nm:22s} per wave, mean over workgroups: ' + ' '.join(f'{v:5.0f}' for v in r.mean(axis=0)) + f'   | last wave mean {r.max(axis=1).mean():.0f}')
