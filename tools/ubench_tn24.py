"""The merged weight-gradient launch of a ZINC-128 training step (24 descriptors, a third of them with a K-concatenated
second operand) under CWN_TN_DBG: 1 no output, 2 no MFMA, 4 no bias sum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import _ffi
dev = torch.device('cuda:0')
Ms = [3165, 3341, 304] * 8
K2s = [0, 0, 0, 0, 0, 0, 128, 128, 128] * 3
K2s = K2s[:24]
pro = len(sys.argv) > 1 and sys.argv[1] == 'pro'
dZ = [torch.randn(m, 128, device=dev) for m in Ms]
X = [torch.randn(m, 128, device=dev) for m in Ms]
X2 = [torch.randn(m, k2, device=dev) if k2 else None for m, k2 in zip(Ms, K2s)]
dW = [torch.zeros(128, 128 + k2, device=dev) for k2 in K2s]
db = [torch.zeros(128, device=dev) for _ in Ms]
sc, sh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev)
def go():
    _ffi.gemm_tn([_ffi.GemmTnDesc(dZ=a.data_ptr(), X=b.data_ptr(), X2=_ffi.ptr(c), in_scale=sc.data_ptr() if pro else None,
                                  in_shift=sh.data_ptr() if pro else None,
                                  in_scale2=None, in_shift2=None, dW=w.data_ptr(), db=v.data_ptr(), M=a.size(0),
                                  lddz=128, ldx=128, ldx2=k2, lddw=128 + k2, N=128, K=128, K2=k2, in_relu=1 if pro else 0)
                  for a, b, c, w, v, k2 in zip(dZ, X, X2, dW, db, K2s)], dev)
go(); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): go()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): go()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): g.replay()
e1.record(); torch.cuda.synchronize()
flops = sum(2 * m * 128 * (128 + k2) for m, k2 in zip(Ms, K2s))
us = 1e3 * e0.elapsed_time(e1) / 60
print(f'dbg={os.environ.get("CWN_TN_DBG", "0")} pro={pro}: {us:.2f} us/launch  {flops / us * 1e-6:.1f} TF fp32-equivalent')
