"""Split the weight-gradient GEMM's time on the ZINC-128 training shape (six descriptors, one
launch): CWN_TN_DBG=1 no output, 2 no MFMA, 4 no bias sum (combine by adding)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import _ffi
dev = torch.device('cuda:0')
Ms = [3165, 3165, 3341, 3341, 304, 304]
K2 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dZ = [torch.randn(m, 128, device=dev) for m in Ms]
X = [torch.randn(m, 128, device=dev) for m in Ms]
X2 = [torch.randn(m, K2, device=dev) for m in Ms] if K2 else [None] * 6
dW = [torch.zeros(128, 128 + K2, device=dev) for _ in Ms]
db = [torch.zeros(128, device=dev) for _ in Ms]
def go():
    _ffi.gemm_tn([_ffi.GemmTnDesc(dZ=a.data_ptr(), X=b.data_ptr(), X2=_ffi.ptr(c), in_scale=None, in_shift=None,
                                  in_scale2=None, in_shift2=None, dW=w.data_ptr(), db=v.data_ptr(), M=a.size(0),
                                  lddz=128, ldx=128, ldx2=K2, lddw=128 + K2, N=128, K=128, K2=K2, in_relu=0)
                  for a, b, c, w, v in zip(dZ, X, X2, dW, db)], dev)
go(); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): go()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50): go()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): g.replay()
e1.record(); torch.cuda.synchronize()
print(f'dbg={os.environ.get("CWN_TN_DBG", "0")} K2={K2}: {1e3 * e0.elapsed_time(e1) / 150:.2f} us/launch')
