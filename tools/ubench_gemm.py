"""Bisect the grouped-GEMM kernel's time on the ZINC-128 shape (4 GEMMs, one launch) and on a
large shape: hipGraph replay of back-to-back launches, with the kernel's debug knobs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)


def bench(Ms, debug, reps=50):
    W = torch.randn(128, 256, device=dev) / 16
    b = torch.randn(128, device=dev)
    Xs = [torch.randn(m, 128, device=dev) for m in Ms]
    gs = [ops.Gemm(X=x, W=W[:, :128] if i % 2 == 0 else W[:, 128:], bias=b if i % 2 == 0 else None, debug=debug)
          for i, x in enumerate(Xs)]
    ops.run_gemm(gs, dev); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
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


for name, Ms in (('zinc128', [3165, 3341, 3341, 304]), ('x64', [202560, 213824, 213824, 19456])):
    flop = sum(Ms) * 128 * 128 * 2
    for dbg, label in ((0, 'full'), (1, 'no MFMA'), (2, 'no W staging'), (4, 'no store'), (3, 'no MFMA, no W'), (7, 'loads + LDS only')):
        us = bench(Ms, dbg, reps=50 if name == 'zinc128' else 5)
        print(f'{name:8s} {label:18s} {us:9.2f} us/launch   {flop / us / 1e6:7.1f} TF')
