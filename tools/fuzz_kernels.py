"""Randomised cross-check of the C-ABI kernels against plain torch (float64 / sequential references):
aggregate (all message ops x reductions, random degree distributions incl. hubs, random widths),
grouped GEMM (random M/N/K, K-concat, prologue/epilogue options, transposed weights), weight-gradient
GEMM, plan builds.  usage: fuzz_kernels.py [iterations] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cwn_amd import _ffi, ops
from cwn_amd.csr import Adjacency

dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
g = torch.Generator().manual_seed(seed)


def rand_index(n_dst, n_src, n_aux):
    kind = rng.integers(0, 4)
    if kind == 0:
        deg = rng.integers(0, 5, n_dst)
    elif kind == 1:
        deg = rng.integers(0, 70, n_dst)
    elif kind == 2:
        deg = rng.integers(0, 4, n_dst)
        for _ in range(rng.integers(1, 4)):
            deg[rng.integers(0, n_dst)] = rng.integers(65, 2000)
    else:
        deg = np.zeros(n_dst, dtype=np.int64)
        deg[rng.integers(0, n_dst, max(1, n_dst // 10))] = rng.integers(1, 30)
    dst = torch.repeat_interleave(torch.arange(n_dst), torch.from_numpy(np.asarray(deg, dtype=np.int64)))
    dst = dst[torch.randperm(dst.numel(), generator=g)]
    E = dst.numel()
    src = torch.randint(0, n_src, (E,), generator=g)
    aux = torch.randint(0, n_aux, (E,), generator=g)
    return torch.stack([src, dst]), aux


def check_aggregate():
    ops.ALLOW_SMALL_OPERANDS = bool(rng.integers(0, 2))      # both addressing variants of the kernel
    n_dst, n_src, n_aux = int(rng.integers(1, 900)), int(rng.integers(1, 700)), int(rng.integers(1, 300))
    F = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 31, 32, 64, 100, 128, 130, 256]))
    idx, aux = rand_index(n_dst, n_src, n_aux)
    if idx.size(1) == 0:
        return
    x = torch.randn(n_src, F, generator=g)
    ua = torch.randn(n_aux, F if rng.integers(0, 3) else 1, generator=g)
    sx = torch.randn(n_dst, F, generator=g) if rng.integers(0, 2) else None
    op = rng.choice(['id', 'plus', 'times', 'relu'])
    red = 'add' if op == 'relu' else str(rng.choice(['add', 'mean', 'max']))
    adj = Adjacency.from_index(idx.to(dev), n_dst, n_src, aux.to(dev), n_aux)
    kw = dict(reduce=red, self_x=None if sx is None else sx.to(dev))
    a, b = x[idx[0]].double(), ua[aux].double()
    if op == 'id':
        msg = a
    elif op == 'plus':
        msg, kw['msg_op'], kw['B'] = a + b, ops.MSG_A_PLUS_B, ua.to(dev)
    elif op == 'times':
        msg, kw['msg_op'], kw['B'] = a * b, ops.MSG_A_TIMES_B, ua.to(dev)
    else:
        msg, kw['msg_op'], kw['B'] = torch.relu(a + b), ops.MSG_RELU_A_PLUS_B, ua.to(dev)
    if op != 'id' and ua.size(1) == 1 and F != 1:
        msg = msg  # broadcast already applied by the double arithmetic above
    got = ops.aggregate(adj, n_dst, x.to(dev), **kw).cpu().double()
    ref = torch.zeros(n_dst, F, dtype=torch.float64)
    if red in ('add', 'mean'):
        ref.index_add_(0, idx[1], msg.expand(-1, F) if msg.size(1) != F else msg)
        if red == 'mean':
            ref /= torch.bincount(idx[1], minlength=n_dst).clamp(min=1).unsqueeze(1)
    else:
        m = msg.expand(-1, F) if msg.size(1) != F else msg
        ref = torch.full((n_dst, F), -float('inf'), dtype=torch.float64)
        ref = ref.scatter_reduce(0, idx[1].unsqueeze(1).expand(-1, F), m, reduce='amax', include_self=True)
        ref[torch.isinf(ref)] = 0.0
    if sx is not None:
        ref += sx.double()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= 2e-5 * scale, ('aggregate', op, red, F, n_dst, idx.size(1), err, scale)


def check_gemm():
    M, N = int(rng.integers(0, 3000)), int(rng.choice([1, 3, 16, 40, 64, 65, 128, 130, 256]))
    K = int(rng.choice([1, 4, 7, 24, 64, 100, 128, 200, 256]))
    K2 = int(rng.choice([0, 0, 4, 64, 128])) if K % 4 == 0 else 0
    if K + K2 > 256:
        K2 = 0
    X, X2 = torch.randn(M, K, generator=g), (torch.randn(M, K2, generator=g) if K2 else None)
    wt = bool(rng.integers(0, 2)) and K2 == 0
    W = torch.randn(N, K + K2, generator=g) / (K + K2) ** 0.5
    bias = torch.randn(N, generator=g) if rng.integers(0, 2) else None
    pro = (not wt) and bool(rng.integers(0, 2))
    isc, ish = (torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)) if pro else (None, None)
    relu = bool(rng.integers(0, 2))
    d = lambda t: None if t is None else t.to(dev)
    gm = ops.Gemm(X=d(X), X2=d(X2), W=d(W.t().contiguous() if wt else W), bias=d(bias), relu=relu, w_trans=wt,
                  in_scale=d(isc), in_shift=d(ish), in_relu=1 if pro else 0)
    Y, = ops.run_gemm([gm], dev)
    A = X.double()
    if pro:
        A = torch.relu(A * isc.double() + ish.double())
    if X2 is not None:
        A = torch.cat([A, X2.double()], 1)
    ref = A @ W.double().t()
    if bias is not None:
        ref = ref + bias.double()
    if relu:
        ref = ref.clamp(min=0)
    if M:
        err = float((Y.cpu().double() - ref).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref.abs().max())), ('gemm', M, N, K, K2, wt, pro, err)


def check_tn():
    M, N = int(rng.integers(1, 4000)), int(rng.choice([1, 5, 64, 100, 128, 256]))
    K = int(rng.choice([1, 4, 24, 64, 128, 200]))
    K2 = int(rng.choice([0, 64, 128])) if K % 4 == 0 else 0
    dZ, X = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
    X2 = torch.randn(M, K2, generator=g) if K2 else None
    dZd, Xd, X2d = dZ.to(dev), X.to(dev), None if X2 is None else X2.to(dev)
    dW, db = torch.zeros(N, K + K2, device=dev), torch.zeros(N, device=dev)
    _ffi.DETERMINISTIC_TN = bool(rng.integers(0, 2))
    try:
        _ffi.gemm_tn([_ffi.GemmTnDesc(dZ=dZd.data_ptr(), X=Xd.data_ptr(), X2=_ffi.ptr(X2d), in_scale=None,
                                      in_shift=None, in_scale2=None, in_shift2=None, dW=dW.data_ptr(), db=db.data_ptr(),
                                      M=M, lddz=N, ldx=K, ldx2=K2, lddw=K + K2, N=N, K=K, K2=K2, in_relu=0)], dev)
    finally:
        _ffi.DETERMINISTIC_TN = False
    A = X.double() if X2 is None else torch.cat([X.double(), X2.double()], 1)
    ref = dZ.double().t() @ A
    err = float((dW.cpu().double() - ref).abs().max())
    assert err <= 3e-5 * max(1.0, float(ref.abs().max())), ('tn', M, N, K, K2, err)
    errb = float((db.cpu().double() - dZ.double().sum(0)).abs().max())
    assert errb <= 3e-5 * max(1.0, float(dZ.double().sum(0).abs().max())), ('tn bias', M, N, errb)


def check_aggregate_backward():
    ops.ALLOW_SMALL_OPERANDS = bool(rng.integers(0, 2))
    n_dst, n_src, n_aux = int(rng.integers(1, 500)), int(rng.integers(1, 400)), int(rng.integers(1, 200))
    F = int(rng.choice([1, 3, 4, 16, 64, 128]))
    idx, aux = rand_index(n_dst, n_src, n_aux)
    if idx.size(1) == 0:
        return
    op = rng.choice(['id', 'plus', 'relu'])
    red = 'add' if op == 'relu' else str(rng.choice(['add', 'mean']))
    same = n_dst == n_src or bool(rng.integers(0, 2))     # x doubles as the self term when shapes allow
    x = torch.randn(n_src, F, generator=g, dtype=torch.float64)
    ua = torch.randn(n_aux, F, generator=g, dtype=torch.float64)
    sx = torch.randn(n_dst, F, generator=g, dtype=torch.float64)
    w = torch.randn(n_dst, F, generator=g, dtype=torch.float64)
    eps = torch.tensor([0.25], dtype=torch.float64)
    xr, ur, sr = x.clone().requires_grad_(), ua.clone().requires_grad_(), sx.clone().requires_grad_()
    a, b = xr[idx[0]], ur[aux]
    msg = a if op == 'id' else (a + b if op == 'plus' else torch.relu(a + b))
    ref = torch.zeros(n_dst, F, dtype=torch.float64).index_add_(0, idx[1], msg)
    if red == 'mean':
        ref = ref / torch.bincount(idx[1], minlength=n_dst).clamp(min=1).unsqueeze(1)
    ref = ref + (1 + eps) * sr
    (ref * w).sum().backward()
    adj = Adjacency.from_index(idx.to(dev), n_dst, n_src, aux.to(dev), n_aux)
    xg, ug, sg = (t.float().to(dev).requires_grad_() for t in (x, ua, sx))
    kw = dict(reduce=red, self_x=sg, eps=eps.float().to(dev))
    if op != 'id':
        kw.update(msg_op=ops.MSG_A_PLUS_B if op == 'plus' else ops.MSG_RELU_A_PLUS_B, B=ug)
    got = ops.aggregate(adj, n_dst, xg, **kw)
    (got * w.float().to(dev)).sum().backward()
    for name, mine, theirs in (('x', xg.grad, xr.grad), ('self', sg.grad, sr.grad)) + \
            ((('attr', ug.grad, ur.grad),) if op != 'id' else ()):
        scale = max(1.0, float(theirs.abs().max()))
        err = float((mine.cpu().double() - theirs).abs().max())
        # ReLU kinks: a pre-activation within 1e-6 of zero may flip between fp32 and fp64
        tol = 2e-5 * scale if op != 'relu' else 5e-2 * scale
        assert err <= tol, ('aggregate backward', name, op, red, F, n_dst, idx.size(1), err, scale)


for it in range(iters):
    check_aggregate()
    check_aggregate_backward()
    check_gemm()
    check_tn()
    if (it + 1) % 50 == 0:
        print(f'{it + 1} rounds ok', flush=True)
torch.cuda.synchronize()
print(f'fuzz ok: {iters} rounds, seed {seed}')
