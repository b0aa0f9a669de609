"""Prototype check (run ON the GPU box from the repo root): accuracy and time of the fp32-accurate
bf16x3-split GEMM (tools/proto/gemm_bf16x3.hip) against the production fp32-MFMA kernel and float64.
    make -C tools/proto   (or the hipcc line in the .hip header) ; python tools/proto/run_gemm_bf16x3.py"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from cwn_amd import ops

dev = torch.device('cuda:0')
L = C.CDLL(os.path.join(HERE, 'libproto_gemm.so'))
L.proto_gemm_bf16x3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]


def proto(X, W, flags=0, max_blocks=512, out=None):
    Y = torch.empty(X.size(0), 128, device=dev) if out is None else out
    rc = L.proto_gemm_bf16x3(X.data_ptr(), W.data_ptr(), Y.data_ptr(), X.size(0), flags, max_blocks,
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return Y


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


torch.manual_seed(0)
# 1. layout: X = I (asymmetric W) must return W^T EXACTLY (1.0 splits into (1, 0, 0))
W = (torch.randn(128, 128, device=dev) / 16).contiguous()
Y = proto(torch.eye(128, device=dev), W)
print('identity check exact:', bool(torch.equal(Y, W.t().contiguous())))

# 2. accuracy on random data, ragged M
for M in (1, 63, 64, 65, 10151):
    X = torch.randn(M, 128, device=dev)
    ref = X.double() @ W.double().t()
    bound = X.double().abs() @ W.double().abs().t()          # |x|.|w|: the natural error scale
    res = {'bf16x3': proto(X, W), 'bf16 (hi*hi only)': proto(X, W, flags=2),
           'fp32 MFMA (production)': ops.run_gemm([ops.Gemm(X=X, W=W)], dev)[0]}
    print(f'M={M}: ' + ', '.join(f'{k}: max err/|x||w| {float(((v.double() - ref).abs() / bound).max()):.2e}, '
                                 f'max rel-to-max {float((v.double() - ref).abs().max() / ref.abs().max()):.2e}'
                                 for k, v in res.items()))

# 3. time
for name, M, reps in (('zinc128 (10151 rows)', 10151, 50), ('x64 (649664 rows)', 649664, 5)):
    X = torch.randn(M, 128, device=dev)
    out = torch.empty(M, 128, device=dev)
    flop = 2.0 * M * 128 * 128
    g = ops.Gemm(X=X, W=W)
    t = {'production fp32 MFMA': graph_us(lambda: ops.run_gemm([g], dev), reps)}
    for mb in (256, 512, 1 << 30):
        t[f'bf16x3 (<= {mb if mb < 1 << 30 else "one/tile"} blocks)'] = graph_us(lambda: proto(X, W, 0, mb, out), reps)
    t['bf16x3 without the MFMAs (512)'] = graph_us(lambda: proto(X, W, 1, 512, out), reps)
    t['bf16 hi*hi only (512)'] = graph_us(lambda: proto(X, W, 2, 512, out), reps)
    for k, us in t.items():
        print(f'{name:22s} {k:36s} {us:9.2f} us   {flop / us / 1e6:7.1f} TF-equivalent')
