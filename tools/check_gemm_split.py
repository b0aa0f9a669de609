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
        ops.set_gemm_exact(not en)
        _, gs = gemms(Ms, Xs)
        if en:
            assert ops.gemm_uses_split(gs, dev)
        res[name] = [y.clone() for y in ops.run_gemm(gs, dev)]
    ops.set_gemm_exact(False)
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

# The candidate kernel (csrc/cwn_gemm_split_v2.hip, `make -C cwn_amd/csrc v2`, CWN_HIP_LIB=.../libcwn_hip_v2.so)
# also serves the BatchNorm prologue, the band statistics and w_trans: each against the exact kernel.
if os.environ.get('CWN_HIP_LIB', '').endswith('_v2.so'):
    for M in (1, 95, 3341, 70_000):
        X = torch.randn(M, 128, device=dev)
        sc, sh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev)
        cases = {
            'prologue+relu': lambda: ops.Gemm(X=X, W=W, bias=b, in_scale=sc, in_shift=sh, in_relu=1),
            'statistics': lambda: ops.Gemm(X=X, W=W, bias=b,
                                           col_stats=torch.zeros(2, ops.stat_rows(M), 128, dtype=torch.float64, device=dev)),
            'prologue+statistics': lambda: ops.Gemm(X=X, W=W, in_scale=sc, in_shift=sh, in_relu=1,
                                                    col_stats=torch.zeros(2, ops.stat_rows(M), 128, dtype=torch.float64, device=dev)),
            'w_trans': lambda: ops.Gemm(X=X, W=W, w_trans=True),
        }
        for cname, make in cases.items():
            got = {}
            for label, en in (('split', True), ('exact', False)):
                ops.set_gemm_exact(not en)
                g = make()
                if en:
                    assert ops.gemm_uses_split([g], dev), cname
                y = ops.run_gemm([g], dev)[0]
                got[label] = (y.clone(), None if g.col_stats is None else g.col_stats.clone())
            ops.set_gemm_exact(False)
            ys, ye = got['split'][0].double(), got['exact'][0].double()
            e = float((ys - ye).abs().max() / (ye.abs().max() + 1e-30))
            good = e < 2e-6
            if got['split'][1] is not None:
                ss, se = got['split'][1], got['exact'][1]
                es = float(((ss - se).abs() / (se.abs() + 1.0)).max())
                good = good and es < 1e-5
                cname += f' (stats rel {es:.1e})'
            ok = ok and good
            print(f'v2 M={M} {cname}: max |split - exact| / max|exact| = {e:.2e}  {"ok" if good else "FAIL"}')

for name, Ms, reps in (('zinc128', [3165, 3341, 3341, 304], 50), ('x64', [202560, 213824, 213824, 19456], 5)):
    Xs, gs = gemms(Ms)
    outs = [torch.empty(m, 128, device=dev) for m in Ms]
    for g, o in zip(gs, outs):
        g.out = o
    t = {}
    for label, en in (('split', True), ('exact', False)):
        ops.set_gemm_exact(not en)
        t[label] = graph_us(lambda: ops.run_gemm(gs, dev), reps)
    ops.set_gemm_exact(False)
    print(f'{name:8s} split {t["split"]:8.2f} us   exact {t["exact"]:8.2f} us')
if os.environ.get('CWN_HIP_LIB', '').endswith('_v2.so'):
    print('(v2: re-run with CWN_SPLIT_TM32=1 for 32-row tiles on launches that do not fill the chip; '
          f'this run: CWN_SPLIT_TM32={os.environ.get("CWN_SPLIT_TM32", "0")})')
print('ALL OK' if ok else 'FAILED')
