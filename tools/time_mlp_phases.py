"""Stage timing of the fused update / combine kernel (csrc/cwn_mlp.hip) from the instrumented build:
    make -C cwn_amd/csrc mlptiming && CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_mlptiming.so python tools/time_mlp_phases.py
Workgroup 0's thread 0 of every workgroup stamps s_memtime between the steps of the chain."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cwn_amd import _ffi, ops
from cwn_amd.layers import SparseCINConv
dev = torch.device('cuda:0')
L = _ffi.lib()
L.cwn_mlp_debug_stamps.argtypes = [C.c_void_p]
L.cwn_mlp_debug_stamps.restype = None
torch.manual_seed(0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
conv = SparseCINConv(F, F, F, None, None, None, None, max_dim=2, hidden=F, act_module=torch.nn.ReLU, layer_dim=F,
                     use_coboundaries=True).to(dev).eval()
rows = tuple(int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (3165, 3341, 304)
TM = int(os.environ.get('TM', 4096 // F))
outs = []
for n in rows:
    outs += [torch.randn(n, F, device=dev), torch.randn(n, F, device=dev)]
nblk = sum((n + TM - 1) // TM for n in rows)
stamps = torch.zeros(nblk, 16, dtype=torch.int64, device=dev)
L.cwn_mlp_debug_stamps(stamps.data_ptr())
with torch.no_grad():
    for _ in range(5):
        conv._dense_eval(['blocked'] * 3, outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        conv._dense_eval(['blocked'] * 3, outs)
    e1.record()
    torch.cuda.synchronize()
print(f'{nblk} workgroups, eager avg {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call')
st = stamps.cpu().numpy()
d = np.diff(st[:, :9], axis=1)
names = ['rows + W0 + split (both branches)', 'multiply 1u', 'multiply 1b + finish 1u + barrier', 'multiply 2u + finish 1b + barrier',
         'multiply 2b + finish 2u + barrier', 'multiply c(up) + finish 2b + barrier', 'multiply c(b)', 'finish + store']
print(f'whole: mean {np.mean(st[:, 8] - st[:, 0]):.0f} max {np.max(st[:, 8] - st[:, 0])}')
for k, nm in enumerate(names):
    print(f'  {nm:40s} mean {d[:, k].mean():8.0f}  max {d[:, k].max():8d}')

# when do the workgroups start and end (ticks from the first start): rounds of the launch
t0 = st[:, 0].min()
start, end = np.sort(st[:, 0] - t0), np.sort(st[:, 8] - t0)
q = [0, 10, 25, 50, 75, 90, 100]
print('start percentiles', [int(np.percentile(start, p)) for p in q])
print('end   percentiles', [int(np.percentile(end, p)) for p in q])
