"""Randomised checks of the kernels round 5 added or changed, against independent restatements:

  * cwn_csr_build with a device-side entry count (cwn_csr_desc.e_dev) vs the exact build of the live entries;
  * cwn_embedding_bwd_f32 (the ballot form over several tables) vs index_add in float64;
  * cwn_dropout_f32 vs the numpy Philox restatement (tests/_philox_ref.py), odd shapes and strides;
  * cwn_head_f32 / cwn_head_bwd_f32 with jumping-knowledge blocks, split pooling, empty complexes and absent dimensions vs float64;
  * cwn_loss_cols_f32(CWN_LOSS_CE) vs torch.nn.functional.cross_entropy in float64;
  * the prepared launches of the eager forward (both bindings) vs a deep copy of the model that has none, while parameters,
    submodules and the mode change under them.

usage: python tools/fuzz_round5.py [rounds] [seed]      (prints one line per failure, exits non-zero if any)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cwn_amd import _ffi, csr, ops
from cwn_amd.train import fused_loss
from tests._philox_ref import multipliers

dev = torch.device('cuda', 0)
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = []


def check(ok, what):
    if not ok:
        fails.append(what)
        print('FAIL', what, flush=True)


def fuzz_csr():
    n = int(rng.integers(1, 5000))
    cap_n = n + int(rng.integers(0, 500))
    E = int(rng.integers(0, 40000))
    cap_E = E + int(rng.integers(1, 3000))
    hub = rng.random() < 0.3
    key = rng.integers(0, n, size=cap_E)
    if hub:
        key[: E // 3] = int(rng.integers(0, n))
    val = rng.integers(0, n, size=cap_E)
    aux = rng.integers(0, max(1, n // 2), size=cap_E)
    kt, vt, at = (torch.from_numpy(a).to(dev) for a in (key, val, aux))
    adj = csr.Adjacency(kt, vt, cap_n, cap_n, at, max(1, cap_n // 2))
    e_dev = torch.tensor([E], dtype=torch.int64, device=dev)
    adj.e_dev_ptr = e_dev.data_ptr()        # (round 6, ADVICE r5: the live entry count is an explicit tag of the adjacency, as a
    csr.build_many([adj], validate=False)   #  static batch sets it -- no longer looked up through the capacity)
    ref = csr.Adjacency(kt[:E].contiguous(), vt[:E].contiguous(), n, n, at[:E].contiguous(), max(1, n // 2))
    csr.build_many([ref], validate=False)
    torch.cuda.synchronize()
    ok = (torch.equal(adj.rowptr[:n + 1], ref.rowptr) and bool((adj.rowptr[n:] == E).all()) and torch.equal(adj.col[:E], ref.col)
          and torch.equal(adj.perm[:E], ref.perm) and torch.equal(adj.aux[:E], ref.aux)
          and sorted(adj.long_row_list().tolist()) == sorted(ref.long_row_list().tolist()))
    check(ok, f'csr e_dev n={n} E={E} cap={cap_n}/{cap_E} hub={hub}')


def fuzz_embedding():
    H = int(rng.choice([64, 128, 256]))
    cols = int(rng.integers(1, 10))
    dims = [int(rng.integers(1, 120)) for _ in range(cols)]
    n = int(rng.integers(1, 3000))
    tables = [torch.nn.Parameter(torch.randn(d, H, device=dev)) for d in dims]
    idx = torch.stack([torch.from_numpy(rng.integers(0, d, size=n)) for d in dims], 1).to(dev)
    out = ops.embedding_sum(tables, idx)
    g = torch.randn(n, H, device=dev)
    out.backward(g)
    worst = 0.0
    for c, t in enumerate(tables):
        ref = torch.zeros(dims[c], H, dtype=torch.float64, device=dev).index_add_(0, idx[:, c], g.double())
        worst = max(worst, float((t.grad.double() - ref).abs().max()) / max(1.0, float(ref.abs().max())))
    ref_out = sum(t.detach()[idx[:, c]] for c, t in enumerate(tables))
    check(worst <= 1e-5 and torch.equal(out.detach(), ref_out), f'embedding bwd H={H} cols={cols} n={n} dims={dims} worst={worst:.2e}')


def fuzz_dropout():
    M, N = int(rng.integers(1, 400)), int(rng.integers(1, 300))
    p = float(rng.choice([0.1, 0.25, 0.5, 0.9]))
    seed = int(rng.integers(0, 2 ** 62))
    ops.dropout_seed(seed, dev)
    pad = int(rng.integers(0, 5))
    base = torch.randn(M, N + pad, device=dev)
    x = base[:, :N] if pad else base
    ops.DROPOUT_TRACE = []
    y = ops.dropout(x, p, True)
    site = ops.DROPOUT_TRACE[0][0]
    ops.DROPOUT_TRACE = None
    m = torch.from_numpy(multipliers((M, N), p, seed, 0, site)).to(dev)
    check(torch.equal(y, x * m), f'dropout M={M} N={N} p={p} pad={pad}')


def fuzz_head():
    n_dims = int(rng.integers(1, 4))
    n_parts = int(rng.choice([1, 1, 2, 4]))
    Kp = int(rng.choice([4, 16, 64]))
    K = Kp * n_parts
    H2 = int(rng.choice([8, 64, 128]))
    O = int(rng.integers(1, 4))
    C = int(rng.integers(1, 12))
    big = rng.random() < 0.4
    counts = [[int(rng.integers(0, 900 if big else 40)) for _ in range(C)] for _ in range(n_dims)]
    for d in range(n_dims):
        counts[d][int(rng.integers(0, C))] = 0                       # a complex without cells of this dimension
    ptrs = [torch.tensor(np.concatenate([[0], np.cumsum(cs)]), dtype=torch.int64, device=dev) for cs in counts]
    blocks = [[torch.randn(max(1, sum(counts[d])), Kp, device=dev)[:sum(counts[d])].requires_grad_(True) for _ in range(n_parts)]
              for d in range(n_dims)]
    absent = n_dims > 1 and rng.random() < 0.2
    bias = rng.random() < 0.5
    lin1 = [torch.nn.Linear(K, H2, bias=bias).to(dev) for _ in range(n_dims)]
    lin2 = torch.nn.Linear(H2, O).to(dev)
    mean_r, mean_f = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    xs = [None if (absent and d == n_dims - 1) else (blocks[d] if n_parts > 1 else blocks[d][0]) for d in range(n_dims)]
    if any(x is not None and (x[0] if isinstance(x, list) else x).size(0) == 0 for x in xs):
        return
    prev = ops.HEAD_POOL_SPLIT
    ops.HEAD_POOL_SPLIT = str(int(rng.choice([1, 2, 5, 16])))
    try:
        if absent:            # (the training form needs every dimension: inference only)
            with torch.no_grad():
                out = ops.head(xs, ptrs, C, [l.weight for l in lin1], [l.bias for l in lin1], lin2.weight, lin2.bias, mean_readout=mean_r,
                               mean_final=mean_f)
        else:
            out, _ = ops.head_train(xs, ptrs, C, [l.weight for l in lin1], [l.bias for l in lin1], lin2.weight, lin2.bias,
                                    mean_readout=mean_r, mean_final=mean_f)
    finally:
        split, ops.HEAD_POOL_SPLIT = ops.HEAD_POOL_SPLIT, prev
    # float64
    hs = []
    for d in range(n_dims):
        if xs[d] is None:
            pooled = torch.zeros(C, K, dtype=torch.float64, device=dev)
        else:
            cat = torch.cat([b.double() for b in blocks[d]], dim=-1)
            rows = []
            for c in range(C):
                seg = cat[int(ptrs[d][c]):int(ptrs[d][c + 1])]
                rows.append((seg.mean(0) if mean_r else seg.sum(0)) if seg.size(0) else torch.zeros(K, dtype=torch.float64, device=dev))
            pooled = torch.stack(rows)
        h = pooled @ lin1[d].weight.double().t()
        if bias:
            h = h + lin1[d].bias.double()
        hs.append(torch.relu(h))
    s = torch.stack(hs, 0)
    s = s.mean(0) if mean_f else s.sum(0)
    ref = s @ lin2.weight.double().t() + lin2.bias.double()
    err = float((out.detach().double() - ref.detach()).abs().max()) / max(1.0, float(ref.abs().max()))
    tag = f'head dims={n_dims} parts={n_parts} Kp={Kp} H2={H2} C={C} big={big} absent={absent} bias={bias} mean={mean_r}/{mean_f} split={split}'
    check(err <= 1e-5, tag + f' fwd err={err:.2e}')
    if not absent:
        w = torch.randn_like(out)
        (out * w).sum().backward()
        got = [[b.grad.clone() for b in blk] for blk in blocks]
        gw = [l.weight.grad.clone() for l in lin1]
        for blk in blocks:
            for b in blk:
                b.grad = None
        for l in lin1 + [lin2]:
            l.zero_grad()
        (ref * w.double()).sum().backward()
        worst = 0.0
        for d in range(n_dims):
            for q in range(n_parts):
                r = blocks[d][q].grad.double()
                worst = max(worst, float((got[d][q].double() - r).abs().max()) / max(1.0, float(r.abs().max())))
            r = lin1[d].weight.grad.double()
            worst = max(worst, float((gw[d].double() - r).abs().max()) / max(1.0, float(r.abs().max())))
        check(worst <= 1e-5, tag + f' bwd err={worst:.2e}')


def fuzz_ce():
    rows, cols = int(rng.integers(1, 600)), int(rng.integers(2, 12))
    pred = (4 * torch.randn(rows, cols, device=dev)).requires_grad_(True)
    y = torch.from_numpy(rng.integers(0, cols, size=rows)).to(dev)
    if rows > 2:
        y[int(rng.integers(0, rows))] = -100
    loss = fused_loss('classification', pred, y)
    if not bool((y >= 0).any()):
        return
    loss.backward()
    p64 = pred.detach().double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(p64, y)
    ref.backward()
    ok = abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))) and float((pred.grad.double() - p64.grad).abs().max()) <= 1e-6
    check(ok, f'cross-entropy rows={rows} cols={cols}')


_prep = {}


def fuzz_prepared():
    """The prepared launches of the eager forward (ops.LayerLaunch / MlpLaunch / FrontLaunch / HeadLaunch and their caches in
    layers / models): a model is run over a few batches in random order while, between calls, a random parameter or buffer is
    written in place, a submodule replaced, the mode switched or the binding changed -- every output must equal that of a deep
    copy of the model made at that moment (a copy has no prepared launches: it goes the long way)."""
    import copy
    from cwn_amd import _cext
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedSparseCIN
    from cwn_amd.synthetic import zinc_like_complexes
    if not _prep or rng.random() < 0.05:
        hidden = int(rng.choice([64, 128]))
        torch.manual_seed(int(rng.integers(0, 1 << 30)))
        model = EmbedSparseCIN(28, 4, int(rng.integers(1, 3)), int(rng.integers(1, 4)), hidden, dropout_rate=0.0, max_dim=2,
                               jump_mode=rng.choice([None, 'cat']), readout=str(rng.choice(['sum', 'mean'])), embed_edge=True,
                               use_coboundaries=True).to(dev).eval()
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.normal_(0.0, 0.3)
                    m.running_var.uniform_(0.5, 1.5)
        bs = [ComplexBatch.from_complex_list(zinc_like_complexes(int(rng.integers(1, 40)), int(rng.integers(0, 1 << 20)), 6, n_lo=6, n_hi=26),
                                             max_dim=2).to(dev) for _ in range(3)]
        x0 = [[None if b.cochains[d].x is None else b.cochains[d].x.clone() for d in range(3)] for b in bs]
        _prep.clear()
        _prep.update(model=model, bs=bs, x0=x0, cfg=f'hidden {hidden} layers {len(model.convs)} jump {model.jump_mode} readout {model.readout} '
                                                    f'edge table {model.e_embed_init is not None}')
    model, bs, x0 = _prep['model'], _prep['bs'], _prep['x0']
    what = int(rng.integers(0, 8))
    with torch.no_grad():
        params = [p for p in model.parameters()] + [b for b in model.buffers() if b.dtype.is_floating_point]
        if what == 0:
            t = params[int(rng.integers(0, len(params)))]
            t.mul_(1.0 + 0.1 * float(rng.random()))
        elif what == 1:
            conv = model.convs[int(rng.integers(0, len(model.convs)))]
            lvl = conv.mp_levels[int(rng.integers(0, 3))]
            net = [lvl.update_up_nn, lvl.update_boundaries_nn, lvl.combine_nn][int(rng.integers(0, 3))]
            old = net[0]
            net[0] = torch.nn.Linear(old.in_features, old.out_features).to(dev)
            _prep['last'] = f'replaced conv {model.convs.index(conv) if hasattr(model.convs, "index") else "?"} dim {list(conv.mp_levels).index(lvl)} net {[id(n) for n in (lvl.update_up_nn, lvl.update_boundaries_nn, lvl.combine_nn)].index(id(net))}'
        elif what == 2:
            model.train(bool(rng.random() < 0.3))
    k = int(rng.integers(0, len(bs)))

    def run(m, which):
        for d in range(3):
            bs[k].cochains[d]._x = x0[k][d]
        with torch.no_grad(), _cext.binding(which):
            return m(bs[k]).clone()
    which = 'compiled' if (rng.random() < 0.7 and _cext.ext() is not None) else 'ctypes'
    # (the copy first: building ANY module moves ops.STRUCT_EPOCH, so the model's first run below rebuilds its launches and
    #  the second runs them from the caches -- both must equal the copy's)
    ref = run(copy.deepcopy(model), 'ctypes')
    got1 = run(model, which)
    got2 = run(model, which)
    if not (torch.equal(got1, ref) and torch.equal(got2, ref)) and not _prep.get('diag'):
        _prep['diag'] = True                       # the first failure: which stage of the forward differs, and whose update launch

        def partial(m):
            for d in range(3):
                bs[k].cochains[d]._x = x0[k][d]
            with torch.no_grad():
                return {kk: v.clone() for kk, v in m(bs[k], include_partial=True)[1].items()}
        cp = copy.deepcopy(model)
        a, c = partial(model), partial(cp)
        print('  first difference by stage (model vs a fresh copy):',
              {kk: float((a[kk] - c[kk]).abs().max()) for kk in a if not torch.equal(a[kk], c[kk])}, flush=True)
        sd, sc = model.state_dict(), cp.state_dict()
        print('  state_dict differences:', [kk for kk in sd if not torch.equal(sd[kk], sc[kk])], flush=True)
    check(torch.equal(got1, ref) and torch.equal(got2, ref),
          f'prepared launches: change {what}, batch {k}, binding {which}, training {model.training}, max|delta| '
          f'{float((got1 - ref).abs().max()):.2e} / {float((got2 - ref).abs().max()):.2e}; {_prep["cfg"]}; {_prep.get("last")}')
    model.eval()


for r in range(ROUNDS):
    for fn in (fuzz_csr, fuzz_embedding, fuzz_dropout, fuzz_head, fuzz_ce, fuzz_prepared):
        try:
            fn()
        except Exception as e:                       # an exception is a failure of the case, not of the run
            if not fails:
                import traceback
                traceback.print_exc()
            check(False, f'{fn.__name__} raised {type(e).__name__}: {e}')
torch.cuda.synchronize()
try:
    csr.check_errors(dev)
except IndexError as e:
    check(False, f'device error word: {e}')
print(f'{ROUNDS} rounds x 6 checks: {len(fails)} failures')
sys.exit(1 if fails else 0)
