"""Randomised cross-check of the training-stage kernels (csrc/cwn_stage.hip: cwn_dense_stage_f32, cwn_dense_stage_bwd_f32)
against float64: random row counts (ends inside bands and workgroups, single rows), widths 64 / 128, every combination of
bias / prologue / statistics / norm, one to eight products per launch.
usage: fuzz_stage.py [seconds] [seed]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cwn_amd import _ffi, ops                                   # noqa: E402

dev = torch.device('cuda:0')
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
coin = lambda: bool(ri(0, 1))
D = lambda t: t.double()
t0, n_fwd, n_bwd, worst = time.time(), 0, 0, 0.0
while time.time() - t0 < budget:
    F = 64 if coin() else 128
    n = ri(1, 8)
    wide = coin()
    Ms = [ri(1, 70) if coin() else ri(1, 6000) for _ in range(n)]
    lins = [torch.nn.Linear(2 * F if wide else F, F).to(dev) for _ in Ms]
    ops.pack_stage_weights_many([l.weight for l in lins])
    # ---- forward ----
    gemms = []
    for M, lin in zip(Ms, lins):
        X, X2 = rn(M, F), (rn(M, F) if wide else None)
        pro = coin()
        gm = ops.Gemm(X=X, X2=X2, W=lin.weight, bias=lin.bias.detach() if coin() else None,
                      in_scale=(rn(F).abs() + 0.5) if pro else None, in_shift=rn(F) if pro else None,
                      in_scale2=(rn(F).abs() + 0.5) if (pro and wide) else None, in_shift2=rn(F) if (pro and wide) else None,
                      in_relu=(3 if wide else 1) if coin() else 0,
                      col_stats=torch.zeros(2, ops.stat_rows(M), F, dtype=torch.float64, device=dev) if coin() else None)
        gemms.append(gm)
    got = ops.run_stage(gemms, dev)
    assert got is not None
    for gm, y in zip(gemms, got):
        x = D(gm.X)
        if gm.in_scale is not None:
            x = x * D(gm.in_scale) + D(gm.in_shift)
        if gm.in_relu & 1:
            x = x.relu()
        if gm.X2 is not None:
            x2 = D(gm.X2)
            if gm.in_scale2 is not None:
                x2 = x2 * D(gm.in_scale2) + D(gm.in_shift2)
            if gm.in_relu & 2:
                x2 = x2.relu()
            x = torch.cat([x, x2], 1)
        z = x @ D(gm.W.detach()).t() + (D(gm.bias) if gm.bias is not None else 0.0)
        err = float((D(y) - z).abs().max()) / max(1.0, float(z.abs().max()))
        worst = max(worst, err)
        assert err <= 1e-5, ('forward', F, Ms, err)
        if gm.col_stats is not None:
            M = z.size(0)
            zp = torch.cat([z, z.new_zeros((-M) % 32, F)]).view(-1, 32, F)
            want = torch.stack([zp.sum(1), (zp * zp).sum(1)])
            assert (gm.col_stats - want).abs().max() <= 32e-6 * max(1.0, float(z.abs().max()) ** 2), ('stats', F, Ms)
    n_fwd += 1
    # ---- backward ----
    entries, refs, keep = [], [], []
    for M, lin in zip(Ms, lins):
        dy, z = rn(M, F), rn(M, F)
        with_norm = coin()
        aff = torch.stack([rn(F), rn(F), rn(F) * 0.1, rn(F).abs() + 0.5])
        s12 = torch.stack([rn(F), rn(F)]) * M ** 0.5
        dz = torch.empty(M, F, device=dev)
        relu = coin()
        b = _ffi.GemmBnb(z=z.data_ptr(), dz=dz.data_ptr(), ldz=F, lddz=F, relu=int(relu))
        if with_norm:
            b.scale, b.shift, b.mean, b.rstd = (aff[r].data_ptr() for r in range(4))
            b.s1, b.s2 = s12[0].data_ptr(), s12[1].data_ptr()
        out = torch.full((M, 2 * F if wide else F), float('nan'), device=dev)
        entries.append((dy, b, lin.weight, out[:, :F], out[:, F:] if wide else None))
        keep += [dy, z, aff, s12, dz, out]
        if with_norm:
            sc, sh, mu, rs = (D(aff[r]) for r in range(4))
            mask = ((z * aff[0] + aff[1]) > 0) if relu else torch.ones_like(z, dtype=torch.bool)
            dyh = D(dy) * mask
            want_dz = sc * dyh - sc * D(s12[0]) / M - sc * rs * D(s12[1]) / M * (D(z) - mu)
        else:
            want_dz = D(dy) * ((z > 0) if relu else 1.0)
        refs.append((dz, want_dz, out, want_dz @ D(lin.weight.detach())))
    assert ops.run_stage_bwd(entries, dev)
    for dz, want_dz, out, want_dx in refs:
        e1 = float((D(dz) - want_dz).abs().max()) / max(1.0, float(want_dz.abs().max()))
        e2 = float((D(out) - want_dx).abs().max()) / max(1.0, float(want_dx.abs().max()))
        worst = max(worst, e1, e2)
        assert e1 <= 1e-5 and e2 <= 1e-5, ('backward', F, Ms, e1, e2)
    n_bwd += 1
print(f'{n_fwd} forward and {n_bwd} backward launches of random shape: worst relative deviation {worst:.2e} (gate 1e-5)')
