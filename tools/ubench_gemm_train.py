"""The grouped GEMM launches of a ZINC-128 TRAINING step (cwn_amd/dense_train.py) under the kernel's debug knobs:
stage 1 (6 GEMMs, batch statistics in the epilogue), stage 2 (+ BatchNorm / ReLU prologue), combine (K = 256),
dX (transposed weight).  hipGraph replay of back-to-back launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)
Ms6 = [3165, 3165, 3341, 3341, 304, 304]
Ms3 = [3165, 3341, 304]


def bench(make, reps=30):
    gs = make()
    ops.run_gemm(gs, dev); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            ops.run_gemm(gs, dev)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (3 * reps)


W = [torch.randn(128, 128, device=dev) / 11 for _ in range(6)]
Wc = [torch.randn(128, 256, device=dev) / 16 for _ in range(3)]
b = torch.randn(128, device=dev)
sc, sh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev)
X6 = [torch.randn(m, 128, device=dev) for m in Ms6]
X3 = [torch.randn(m, 128, device=dev) for m in Ms3]
X3b = [torch.randn(m, 128, device=dev) for m in Ms3]
st6 = [torch.empty(2, ops.stat_rows(m), 128, dtype=torch.float64, device=dev) for m in Ms6]
st3 = [torch.empty(2, ops.stat_rows(m), 128, dtype=torch.float64, device=dev) for m in Ms3]
cases = {
    'stage 1 (stats)': lambda d: [ops.Gemm(X=x, W=w, bias=b, col_stats=s, debug=d) for x, w, s in zip(X6, W, st6)],
    'stage 1 no stats': lambda d: [ops.Gemm(X=x, W=w, bias=b, debug=d, exact=True) for x, w in zip(X6, W)],
    'stage 1 split kernel': lambda d: [ops.Gemm(X=x, W=w, bias=b) for x, w in zip(X6, W)],
    'stage 2 (prologue + stats)': lambda d: [ops.Gemm(X=x, W=w, bias=b, in_scale=sc, in_shift=sh, in_relu=1, col_stats=s, debug=d)
                                             for x, w, s in zip(X6, W, st6)],
    'combine (K 256, prologue, stats)': lambda d: [ops.Gemm(X=x, X2=x2, W=w, bias=b, in_scale=sc, in_shift=sh, in_scale2=sc,
                                                            in_shift2=sh, in_relu=3, col_stats=s, debug=d)
                                                   for x, x2, w, s in zip(X3, X3b, Wc, st3)],
    'dX (w_trans)': lambda d: [ops.Gemm(X=x, W=w, w_trans=True, debug=d) for x, w in zip(X6, W)],
    'dX combine (w_trans, N 256)': lambda d: [ops.Gemm(X=x, W=w, w_trans=True, debug=d) for x, w in zip(X3, Wc)],
}
for name, mk in cases.items():
    flop = sum(g.X.size(0) * g.W.numel() * 2 for g in mk(0))
    row = []
    for dbg in ((0, 1, 2, 4, 7) if 'split' not in name else (0,)):
        us = bench(lambda: mk(dbg))
        row.append(f'{("full", "noMFMA", "noW", "", "noStore", "", "", "loads")[dbg]} {us:6.2f}')
    print(f'{name:34s} ' + '  '.join(row) + f'   ({flop / 1e9:.2f} GFLOP)')
