"""Focused check of the bf16-split GEMM path of cwn_gemm_f32 (a few seconds on the GPU box):
accuracy against float64 next to the exact fp32-MFMA kernel on the shapes that matter (grouped,
ragged, persistent multi-tile workgroups), then the time of both kernels on the ZINC-128 layer shape
and at 64x the rows.   python tools/check_gemm_split.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import _ffi, ops

dev = torch.device('cuda:0')
torch.manual_seed(0)
W2 = (torch.randn(128, 256, device=dev) / 16)
b = torch.randn(128, device=dev)


def gemms(Ms, Xs=None):
    Xs = Xs or [torch.randn(m, 128, device=dev) for m in Ms]
    return Xs, [ops.Gemm(X=x, W=W2[:, :128] if i % 2 == 0 else W2[:, 128:], bias=b if i % 2 == 0 else None)
                for i, x in enumerate(Xs)]


def graph_us(fn, reps):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (3 * reps)


ok = True
for Ms in ([1], [63, 64, 65], [3165, 3341, 3341, 304], [100_000, 7], [40_000, 40_000, 129]):
    Xs, _ = gemms(Ms)
    res = {}
    for name, en in (('split', True), ('exact', False)):
        _ffi.gemm_set_split(en)
        _, gs = gemms(Ms, Xs)
        if en:
            assert ops.gemm_uses_split(gs, dev)
        res[name] = [y.clone() for y in ops.run_gemm(gs, dev)]
    _ffi.gemm_set_split(True)
    worst = {}
    for i, x in enumerate(Xs):
        Wd = (W2[:, :128] if i % 2 == 0 else W2[:, 128:]).double()
        ref = x.double() @ Wd.t() + (b.double() if i % 2 == 0 else 0)
        bound = x.double().abs() @ Wd.abs().t() + 1.0
        for name in res:
            e = float(((res[name][i].double() - ref).abs() / bound).max())
            worst[name] = max(worst.get(name, 0.0), e)
    good = worst['split'] < 2e-6
    ok = ok and good
    print(f'M={Ms}: max err / (|x|.|w| + 1): split {worst["split"]:.2e}, exact {worst["exact"]:.2e}  {"ok" if good else "FAIL"}')
W = W2[:, :128].contiguous()
Y = ops.run_gemm([ops.Gemm(X=torch.eye(128, device=dev), W=W)], dev)[0]
print('identity exact:', bool(torch.equal(Y, W.t())))
ok = ok and bool(torch.equal(Y, W.t()))

for name, Ms, reps in (('zinc128', [3165, 3341, 3341, 304], 50), ('x64', [202560, 213824, 213824, 19456], 5)):
    Xs, gs = gemms(Ms)
    outs = [torch.empty(m, 128, device=dev) for m in Ms]
    for g, o in zip(gs, outs):
        g.out = o
    t = {}
    for label, en in (('split', True), ('exact', False)):
        _ffi.gemm_set_split(en)
        t[label] = graph_us(lambda: ops.run_gemm(gs, dev), reps)
    _ffi.gemm_set_split(True)
    print(f'{name:8s} split {t["split"]:8.2f} us   exact {t["exact"]:8.2f} us')
print('ALL OK' if ok else 'FAILED')
