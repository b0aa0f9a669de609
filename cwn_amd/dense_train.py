"""Training-mode dense networks of a SparseCINConv layer (update_up_nn, update_boundaries_nn,
combine_nn of every dimension; mp/layers.py:193-199, 303-325) as grouped launches of the C-ABI
kernels, forward AND backward, behind one torch.autograd.Function.

Forward, per layer (all dimensions and both branches in each launch):

    for every update stage:   cwn_gemm_f32      Z = prologue(previous Z) W^T + b, batch statistics
                                                of Z accumulated in the epilogue (fp64)
                              cwn_bn_finalize   statistics -> per-column affine (+ running stats)
    combine:                  cwn_gemm_f32      Z3 = [A_up | A_bd] Wc^T + bc  (torch.cat = K-concat)
                              cwn_bn_finalize
                              cwn_norm_act      H = ReLU(Z3 * scale + shift)

`prologue` is the BatchNorm apply + ReLU of the producing stage, done while the tile is staged, so
no normalised activation between two Linear layers is ever written to memory.

Backward, per stage from the last to the first:

    cwn_norm_bwd_reduce   s1 = sum dyh, s2 = sum dyh * xhat          (= d beta, d gamma)
    cwn_norm_bwd_apply    dZ = scale * (dyh - s1/M - xhat * s2/M)
    cwn_gemm_tn_f32       dW += dZ^T prologue(X),  db += sum dZ      (fp32 atomics, MFMA)
    cwn_gemm_f32(w_trans) dX = dZ W                                    (the layer's own weight)

Semantics are those of torch.nn.Linear / BatchNorm1d(train) / ReLU; tests/test_gpu_parity.py checks
outputs, every gradient and the running statistics against the torch modules.

CINppConv (mp/layers.py:216-260: three update networks per dimension, four with the co-boundary stream, and a 3F / 4F-wide
combine) uses the same Function with a plan of N branches: the combine stage takes the third / fourth branch as extra
K-blocks of the same launch (cwn_dense_stage_ex_f32), its backward is two entries of the backward-stage launch and two K-pairs
of the weight-gradient launch per dimension.  Where that form does not apply (widths other than 64 / 128, CWN_LIVE_BN=0) the
plan comes WITHOUT a combine stage: the last stage of every branch is activated by cwn_norm_act and handed back, torch.cat +
combine_nn run as torch modules.
"""
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor
from torch.nn import BatchNorm1d, Identity, Linear

from . import _ffi, ops

# True: BatchNorm backward as ONE launch per stage (cwn_norm_bwd_f32: column-owning workgroups, sums bit-reproducible and added
# straight into gamma.grad / beta.grad).  OFF: measured on the ZINC-128 step the launch takes 28 us against 6.3 + 5.9 us
# for reduce + apply -- a workgroup that owns 4 columns reads and writes 16 B of every 128-B line, and the partial-line
# traffic of 32 workgroups per matrix costs far more than the launch it saves (step 1.17 -> 1.34 ms).  Kept for runs that
# want reproducible BatchNorm gradients (with _ffi.DETERMINISTIC_TN for the weights).
FUSED_NORM_BACKWARD = False
# The apply half of the BatchNorm backward as the PROLOGUE of the input-gradient GEMM that consumes it (cwn_gemm_bnb): no
# apply launch, dz written once on the way.  False: cwn_norm_bwd_apply_f32 + a plain transposed-weight GEMM (A/B, tests).
FUSED_NORM_APPLY = os.environ.get('CWN_FUSED_NORM_APPLY') != '0'
# A BatchNorm without a launch of its own (include/cwn_hip.h: cwn_bn_live; round 4): the stage launch adds its workgroups'
# column sums into 8 fp64 slot rows, every workgroup of the NEXT launch (the following stage, or the activation of the
# layer's last stage) derives the affine from them in its prologue and its first workgroup writes what the backward reads and
# the running statistics -- the 12 cwn_bn_finalize_f32 launches of a ZINC training step (5.9 us each) are gone.  Needs every
# stage on cwn_dense_stage_f32 (width 64 / 128, packed blocks); CWN_LIVE_BN=0 restores the finalize launches (whose per-band
# partials summed in band order are the bit-reproducible form).
LIVE_BN = os.environ.get('CWN_LIVE_BN', '1') != '0'
# ... and the REDUCE half of its backward (cwn_bn_bwd_live): the backward-stage launch that produces a stage's dy adds the
# column sums of dyh, dyh * xhat into 8 fp32 slot rows from its epilogue (the tile still in registers, one coalesced atomic
# instruction per 64 columns and workgroup); the launch that consumes them sums the slots in its prologue.  8 of the 12
# cwn_norm_bwd_reduce_f32 launches of a ZINC step go (the combine stages' dy comes from autograd: theirs stay).  Round 3
# tried this with one atomic per column and lane group straight into s1 / s2 -- 110 k four-lane atomic instructions onto 256
# addresses per launch, 9.9 -> 20.5 us; staged through LDS it is four 64-lane instructions per workgroup onto 8 x 256.
LIVE_BN_BWD = os.environ.get('CWN_LIVE_BN_BWD', '1') != '0'
# (Measured and dropped: the REDUCE half of the next stage -- column sums of dyh, dyh * xhat -- taken in the epilogue of the
# backward-stage launch that produces its dy, the tile still in registers, so that 8 of the 12 reduce launches of a ZINC step
# go away.  A workgroup of cwn_dense_stage_bwd_f32 owns 32 rows: 428 workgroups x 256 column sums = 110 k fp32 atomics per
# launch against the ~7 k of cwn_norm_bwd_reduce_f32's 128-row bands -- the launch went from 9.9 to 20.5 us, the step
# from 0.811 to 0.884 ms.)


class CombineNeedsStageKernel(RuntimeError):
    """A plan with a combine stage over three / four branches met conditions under which only the two-branch form exists (the
    caller -- CINppConv -- then runs the branches here and the combine network as torch modules).  Raised before any launch."""


@dataclass
class Stage:
    """One Linear -> norm -> ReLU group.  `norm` is a BatchNorm1d in training mode or Identity."""
    lin: Linear
    norm: torch.nn.Module

    @property
    def is_bn(self) -> bool:
        return isinstance(self.norm, BatchNorm1d)


def supported(stages: Sequence[Stage], max_k: Optional[int] = None) -> bool:
    for st in stages:
        if not isinstance(st.lin, Linear) or st.lin.in_features > (ops.GEMM_MAX_K if max_k is None else max_k):
            return False
        n = st.norm
        if isinstance(n, Identity):
            continue
        if not isinstance(n, BatchNorm1d):
            return False
        if (not n.training or not n.affine or not n.track_running_stats or n.momentum is None
                or n.num_features != st.lin.out_features):
            return False
    return True


def _norm_desc(z: Tensor, *, dy: Optional[Tensor] = None, out: Optional[Tensor] = None, aff=None,
               s12=None, relu: bool = True) -> _ffi.NormDesc:
    """aff = (scale, shift, mean, rstd) rows of one [4, N] tensor, or None for an identity norm."""
    M, N = z.shape
    ld = lambda t: t.stride(0) if t.size(0) > 1 else t.size(1)
    return _ffi.NormDesc(
        dy=_ffi.ptr(dy), z=z.data_ptr(),
        scale=None if aff is None else aff[0].data_ptr(), shift=None if aff is None else aff[1].data_ptr(),
        mean=None if aff is None else aff[2].data_ptr(), rstd=None if aff is None else aff[3].data_ptr(),
        s1=None if s12 is None else s12[0].data_ptr(), s2=None if s12 is None else s12[1].data_ptr(),
        out=_ffi.ptr(out), M=M, lddy=0 if dy is None else ld(dy), ldz=ld(z),
        ldout=0 if out is None else ld(out), N=N, relu=int(relu))


class _Plan:
    """Static description of one layer's dense networks: per dimension, `depth` update stages for
    each of the branches (SparseCINConv: two, upper and boundary) and one combine stage over the K-concatenation of the
    two.  `chains` (round 4, CINppConv): two to four branches per dimension; with `cb` the combine stage takes the third /
    fourth branch as extra K-blocks (cwn_dense_stage_ex_f32; live BatchNorm mode only -- CombineNeedsStageKernel otherwise),
    without it the Function returns every branch's activated output and the caller concatenates and combines them."""

    def __init__(self, up: Optional[List[List[Stage]]], bd: Optional[List[List[Stage]]], cb: Optional[List[Stage]],
                 chains: Optional[List[List[List[Stage]]]] = None, out_drop: float = 0.0):
        # out_drop: dropout probability of the layer's OUTPUT (OGBEmbedSparseCIN drops out after every conv layer,
        # mp/molec_models.py:298-300): applied by the activation launch of the combine stage, re-derived by the reduce launch of
        # its backward (ops: "dropout without a mask tensor"); plans with a combine stage only
        self.out_drop = float(out_drop) if cb is not None else 0.0
        self.chains = chains if chains is not None else [[u, b] for u, b in zip(up, bd)]     # [dim][branch] -> stages
        self.cb = cb
        assert cb is None or all(2 <= len(c) <= 4 for c in self.chains)
        self.nd = len(self.chains)
        self.nb = len(self.chains[0])
        self.depth = len(self.chains[0][0])

    def stages(self):
        """Every stage in the flattening order of the Function's tensor arguments."""
        for i in range(self.nd):
            for chain in self.chains[i]:
                for st in chain:
                    yield st
            if self.cb is not None:
                yield self.cb[i]


def _stage_tensors(st: Stage) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor], Optional[Tensor]]:
    return (st.lin.weight, st.lin.bias, st.norm.weight if st.is_bn else None,
            st.norm.bias if st.is_bn else None)


class _DenseTrain(torch.autograd.Function):
    """tensors = [out_up_0, out_bd_0, ..., out_up_{nd-1}, out_bd_{nd-1}] + 4 per stage (W, b, gamma,
    beta) in _Plan.stages() order.  Returns H_0 .. H_{nd-1} (a plan without combine stages: the activated output of
    every branch, [dim][branch] flattened)."""

    @staticmethod
    def forward(ctx, plan: _Plan, *tensors):
        nd, depth, nb = plan.nd, plan.depth, plan.nb
        dev = tensors[0].device
        A0 = [[ops._rowmajor(tensors[nb * i + br], 'stream output') for br in range(nb)] for i in range(nd)]
        stages = list(plan.stages())
        par = tensors[nb * nd:]
        P = {id(st): par[4 * k: 4 * k + 4] for k, st in enumerate(stages)}
        bns = [st for st in stages if st.is_bn]
        # one fp64 buffer for all batch statistics (per-band partials written by the GEMM
        # epilogue: nothing to zero), one fp32 buffer for all affines
        rows_of = {}
        for i in range(nd):
            for st in [t for chain in plan.chains[i] for t in chain] + ([plan.cb[i]] if plan.cb is not None else []):
                rows_of[id(st)] = ops.stat_rows(A0[i][0].size(0))
        widths = [st.lin.out_features for st in bns]
        F0 = int(A0[0][0].size(1))
        live = bool(LIVE_BN and ops.STAGE_KERNEL and bns and len(bns) == len(stages) and F0 in (64, 128) and nb * nd <= _ffi.MAX_DESCS
                    and all(a.size(1) == F0 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 for pair in A0 for a in pair)
                    and all(tuple(P[id(st)][0].shape) == (F0, F0) and ops.packed_stage_block(P[id(st)][0], 0) is not None
                            for i in range(nd) for chain in plan.chains[i] for st in chain)
                    and all(tuple(P[id(st)][0].shape) == (F0, nb * F0)
                            and all(ops.packed_stage_block(P[id(st)][0], c * F0) is not None for c in range(nb)) for st in (plan.cb or []))
                    and all(t is None or (t.numel() == F0 and t.data_ptr() % 16 == 0 and t.is_contiguous())
                            for st in stages for t in P[id(st)][1:]))
        if plan.cb is not None and nb > 2 and not live:
            raise CombineNeedsStageKernel('a combine stage over more than two branches runs on cwn_dense_stage_ex_f32 with live '
                                          'BatchNorm records only')
        affs = torch.empty(4 * sum(widths), dtype=torch.float32, device=dev)
        stat_of, aff_of, sum_of, slot_of, o, so = {}, {}, {}, {}, 0, 0
        if live:
            # slot sums (fp64) and the backward's s1 / s2 (fp32) of every BatchNorm: ONE zeroed region (the step arena's, when a
            # step driver brackets the step: no fill of their own)
            nw = sum(widths)
            zero = ops.zeros_scratch(8 * _ffi.BN_SLOTS * 2 * nw + 4 * 2 * nw, dev)
            slots = zero[:8 * _ffi.BN_SLOTS * 2 * nw].view(torch.float64)
            sums = zero[8 * _ffi.BN_SLOTS * 2 * nw: 8 * _ffi.BN_SLOTS * 2 * nw + 4 * 2 * nw].view(torch.float32)
            stats = None
        else:
            stats = torch.empty(2 * sum(rows_of[id(st)] * w for st, w in zip(bns, widths)),
                                dtype=torch.float64, device=dev)
            # the backward pass' column sums (s1, s2 per BatchNorm stage): allocated here and CLEARED by cwn_bn_finalize_f32,
            # so that the backward launches no fill for them
            sums = torch.empty(2 * sum(widths), dtype=torch.float32, device=dev)
        for st, w in zip(bns, widths):
            r = rows_of[id(st)]
            if live:
                slot_of[id(st)] = slots[_ffi.BN_SLOTS * 2 * o: _ffi.BN_SLOTS * 2 * (o + w)].view(_ffi.BN_SLOTS, 2, w)
            else:
                stat_of[id(st)] = stats[so: so + 2 * r * w].view(2, r, w)
                so += 2 * r * w
            aff_of[id(st)] = affs[4 * o: 4 * o + 4 * w].view(4, w)
            sum_of[id(st)] = sums[2 * o: 2 * o + 2 * w].view(2, w)
            o += w

        def live_record(st: Optional[Stage]):
            """The cwn_bn_live record with which a consumer of `st`'s output derives its BatchNorm."""
            if not live or st is None:
                return None
            n = st.norm
            _, _, gamma, beta = P[id(st)]
            return _ffi.BnLive(slots=slot_of[id(st)].data_ptr(), gamma=_ffi.ptr(gamma), beta=_ffi.ptr(beta),
                               running_mean=n.running_mean.data_ptr(), running_var=n.running_var.data_ptr(),
                               num_batches_tracked=n.num_batches_tracked.data_ptr(), aff=aff_of[id(st)].data_ptr(),
                               eps=float(n.eps), momentum=float(n.momentum))

        def finalize(group: Sequence[Tuple[Stage, int]]):
            if live:
                return
            descs = []
            for st, M in group:
                if not st.is_bn:
                    continue
                n, s, a = st.norm, stat_of[id(st)], aff_of[id(st)]
                W, b, gamma, beta = P[id(st)]
                descs.append(_ffi.BnDesc(
                    col_sum=s[0].data_ptr(), col_sumsq=s[1].data_ptr(), gamma=_ffi.ptr(gamma),
                    beta=_ffi.ptr(beta), running_mean=n.running_mean.data_ptr(),
                    running_var=n.running_var.data_ptr(), scale=a[0].data_ptr(), shift=a[1].data_ptr(),
                    mean=a[2].data_ptr(), rstd=a[3].data_ptr(), M=M, N=s.size(2), eps=float(n.eps),
                    momentum=float(n.momentum), num_batches_tracked=n.num_batches_tracked.data_ptr(),
                    bwd_sums=sum_of[id(st)].data_ptr()))
            if descs:
                _ffi.bn_finalize(descs, dev)

        def prologue(st: Optional[Stage]):
            """(scale, shift) of the producing stage for the consumer's prologue."""
            if st is None or not st.is_bn or live:
                return None, None
            a = aff_of[id(st)]
            return a[0], a[1]

        def run(gemms):
            res = ops.run_stage(gemms, dev)          # cwn_dense_stage_f32 when the blocks are packed (ops.STAGE_KERNEL)
            if res is None:
                if live:
                    raise RuntimeError('dense_train: a stage left cwn_dense_stage_f32 in live-BatchNorm mode (set CWN_LIVE_BN=0)')
                res = ops.run_gemm(gemms, dev)
            return res

        Z = [[[None] * depth for _ in range(nb)] for _ in range(nd)]   # Z[dim][branch][stage]
        for s in range(depth):
            gemms, group = [], []
            for i in range(nd):
                for br, chain in enumerate(plan.chains[i]):
                    st = chain[s]
                    W, b, _, _ = P[id(st)]
                    X = A0[i][br] if s == 0 else Z[i][br][s - 1]
                    sc, sh = prologue(chain[s - 1] if s > 0 else None)
                    gemms.append(ops.Gemm(X=X, W=W, bias=b, in_scale=sc, in_shift=sh,
                                          in_relu=1 if s > 0 else 0,
                                          col_stats=stat_of.get(id(st)), stat_slots=slot_of.get(id(st)),
                                          in_bn=live_record(chain[s - 1] if s > 0 else None)))
                    group.append((st, X.size(0)))
            res = run(gemms)
            k = 0
            for i in range(nd):
                for br in range(nb):
                    Z[i][br][s] = res[k]
                    k += 1
            finalize(group)
        if plan.cb is not None:
            gemms, group = [], []
            for i in range(nd):
                st = plan.cb[i]
                W, b, _, _ = P[id(st)]
                last_up, last_bd = plan.chains[i][0][-1], plan.chains[i][1][-1]
                sc, sh = prologue(last_up)
                sc2, sh2 = prologue(last_bd)
                # (a third / fourth branch -- CIN++: live mode only, checked above -- rides as an extra K-block each)
                more = tuple((Z[i][br][-1], True, live_record(plan.chains[i][br][-1])) for br in range(2, nb))
                gemms.append(ops.Gemm(X=Z[i][0][-1], X2=Z[i][1][-1], W=W, bias=b, in_scale=sc, in_shift=sh,
                                      in_scale2=sc2, in_shift2=sh2, in_relu=3, col_stats=stat_of.get(id(st)),
                                      stat_slots=slot_of.get(id(st)), in_bn=live_record(last_up),
                                      in_bn2=live_record(last_bd), more=more))
                group.append((st, Z[i][0][-1].size(0)))
            Z3 = run(gemms)
            finalize(group)
            last = [(z, plan.cb[i]) for i, z in enumerate(Z3)]
        else:               # no combine stage here: every branch's last stage is activated and handed back
            Z3 = []
            last = [(Z[i][br][-1], plan.chains[i][br][-1]) for i in range(nd) for br in range(nb)]
        H = [torch.empty_like(z) for z, _ in last]
        acts, drop_sites = [], []
        for (z, st), h in zip(last, H):
            site = None
            if z.numel():
                d = _norm_desc(z, out=h, aff=None if live else aff_of.get(id(st)))
                if live:
                    d.bn = live_record(st)
                if plan.out_drop > 0.0:
                    d.drop = ops.dropout_record(dev, plan.out_drop, tag=('conv', len(drop_sites)))
                    site = int(d.drop.site)
                acts.append(d)
            drop_sites.append(site)
        ctx.drop_sites = drop_sites
        if acts:
            _ffi.norm_act(acts, dev)
        # (round 6) the outputs of the combine stage are the NEXT conv layer's inputs: registered so that the blocked backward of
        # that layer takes over the reduce half of this stage's BatchNorm backward (ops.bn_out_register; the slot sums live in
        # the step's zeroed scratch).  Not with an output dropout (dy then arrives w.r.t. the dropped activation).
        ctx.bn_ext = None
        if (plan.cb is not None and live and LIVE_BN_BWD and ops.BN_BWD_FUSE and not FUSED_NORM_BACKWARD and plan.out_drop == 0.0
                and ops.STAGE_KERNEL and ops.BLOCKED_BACKWARD == 2):
            ext = []
            for i, (z, h) in enumerate(zip(Z3, H)):
                st = plan.cb[i]
                Fw = int(z.size(1)) if z.dim() == 2 else 0
                if (not st.is_bn or Fw not in (64, 128) or not z.numel() or z.stride(1) != 1 or z.stride(0) % 4
                        or z.data_ptr() % 16):
                    ext.append(None)
                    continue
                slots = ops.zeros_scratch(4 * _ffi.BN_SLOTS * 2 * Fw, dev).view(torch.float32)[:_ffi.BN_SLOTS * 2 * Fw]
                slots = slots.view(_ffi.BN_SLOTS, 2, Fw)
                ext.append((ops.bn_out_register(h, z, aff_of[id(st)], slots), slots))
            ctx.bn_ext = ext
        if bns:
            ops.state_changed()      # bn_finalize wrote the running statistics (and the batch counters) through raw pointers
        ctx.plan = plan
        ctx.aff_of = {k: v for k, v in aff_of.items()}
        ctx.sum_of, ctx.sums, ctx.sums_clean = sum_of, sums, True
        flatZ = [Z[i][br][s] for i in range(nd) for br in range(nb) for s in range(depth)]
        ctx.save_for_backward(*[a for pair in A0 for a in pair], *flatZ, *Z3, affs,
                              *[t for t in par])
        ctx.n_par = len(par)
        return tuple(H)

    @staticmethod
    def backward(ctx, *dH):
        plan: _Plan = ctx.plan
        nd, depth, nb = plan.nd, plan.depth, plan.nb
        saved = ctx.saved_tensors
        A0 = [[saved[nb * i + br] for br in range(nb)] for i in range(nd)]
        o = nb * nd
        Z = [[[None] * depth for _ in range(nb)] for _ in range(nd)]
        for i in range(nd):
            for br in range(nb):
                for s in range(depth):
                    Z[i][br][s] = saved[o]
                    o += 1
        n3 = nd if plan.cb is not None else 0
        Z3 = list(saved[o: o + n3])
        o += n3 + 1                                   # + the flat affine buffer (kept alive)
        par = saved[o: o + ctx.n_par]
        stages = list(plan.stages())
        P = {id(st): par[4 * k: 4 * k + 4] for k, st in enumerate(stages)}
        aff_of = ctx.aff_of
        dev = A0[0][0].device
        # gradient targets.  Weights / biases: the parameter's own .grad when it is allocated
        # (ops._grad_target: a FlatGradBucket or zero_grad(set_to_none=False)) -- the TN kernel adds
        # into it and autograd gets None -- else a slice of one zeroed scratch buffer.  The
        # BatchNorm sums s1 / s2 always go to scratch (the apply kernel needs THIS pass's sums): the buffer the forward
        # allocated and cwn_bn_finalize_f32 cleared; the apply launch adds them to gamma.grad / beta.grad where those
        # exist (round 2: a fill per layer before, a multi-tensor add after).
        # With the one-launch BatchNorm backward (opt-in; every matrix within its row cap, 16-byte aligned) the sums go
        # straight INTO gamma.grad / beta.grad.
        fused_norm = FUSED_NORM_BACKWARD and all(
            z.size(0) <= _ffi.NORM_BWD_FUSED_MAX_ROWS and z.size(1) % 4 == 0 and z.stride(0) % 4 == 0 and z.data_ptr() % 16 == 0
            for z in [zz for i in range(nd) for br in range(nb) for zz in Z[i][br]] + Z3)
        sizes, targets, norm_targets = [], [], {}
        for st in stages:
            W, b, gamma, beta = P[id(st)]
            tw, tb = ops._grad_target(st.lin.weight), ops._grad_target(st.lin.bias)
            targets.append((tw, tb))
            direct = None
            if st.is_bn:
                tg, tbeta = ops._grad_target(st.norm.weight), ops._grad_target(st.norm.bias)
                if tg is not None and tbeta is not None and (not fused_norm or (tg.data_ptr() % 16 == 0 and tbeta.data_ptr() % 16 == 0)):
                    direct = (tbeta, tg)                      # (s1 = d beta, s2 = d gamma)
            norm_targets[id(st)] = direct
            sizes.append((0 if tw is not None else W.numel(), 0 if (b is None or tb is not None) else b.numel()))
        n_flat = sum(a + b for a, b in sizes)
        flat = torch.zeros(n_flat, dtype=torch.float32, device=dev) if n_flat else None
        if not ctx.sums_clean:            # a second backward over the same forward (retain_graph): the scratch holds the first's sums
            ctx.sums.zero_()
        ctx.sums_clean = False
        G, q = {}, 0
        for st, (nw, nbias), (tw, tb) in zip(stages, sizes, targets):
            W, b = P[id(st)][0], P[id(st)][1]
            dW = flat[q: q + nw].view_as(W) if tw is None else tw
            q += nw
            db = (flat[q: q + nbias] if nbias else None) if tb is None else tb
            q += nbias
            G[id(st)] = (dW, db, ctx.sum_of.get(id(st)))

        # slot sums of the reduce halves the backward-stage launches take over (LIVE_BN_BWD): one zeroed region
        upd = [st for i in range(nd) for chain in plan.chains[i] for st in chain if st.is_bn]
        F0 = int(Z[0][0][-1].size(1))
        live_bwd = bool(LIVE_BN_BWD and not fused_norm and ops.STAGE_KERNEL and upd and F0 in (64, 128)
                        and all(st.lin.out_features == F0 for st in upd))
        bslot_of, filled = {}, set()
        if live_bwd:
            zb = ops.zeros_scratch(4 * _ffi.BN_SLOTS * 2 * F0 * len(upd), dev).view(torch.float32)
            for k, st in enumerate(upd):
                bslot_of[id(st)] = zb[k * _ffi.BN_SLOTS * 2 * F0: (k + 1) * _ffi.BN_SLOTS * 2 * F0].view(_ffi.BN_SLOTS, 2, F0)

        def out_record(st: Optional[Stage], z: Tensor):
            """The cwn_bn_bwd_live record with which a launch whose dx is `st`'s dy takes over the reduce of its backward."""
            if not live_bwd or st is None or not st.is_bn or not z.numel() or z.stride(1) != 1 or z.stride(0) % 4 or z.data_ptr() % 16:
                return None
            return _ffi.BnBwdLive(z=z.data_ptr(), aff=aff_of[id(st)].data_ptr(), slots=bslot_of[id(st)].data_ptr(),
                                  ldz=z.stride(0))

        drop_of = {}          # id(combine stage) -> (cwn_dropout record, the multiplied gradient the reduce launch writes)
        ld = lambda t: t.stride(0) if t.size(0) > 1 else t.size(1)

        def bnb_ok(dy, z):
            return (FUSED_NORM_APPLY and not fused_norm and z.size(1) in (64, 128) and z.numel() > 0
                    and all(t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.stride(1) == 1 for t in (dy, z)))

        def norm_backward(items, lazy=False):
            """items: (stage, dy, z) -> dz list; BatchNorm stages reduce first.  The sums are handed on to gamma.grad /
            beta.grad by the apply launch itself where those buffers exist (`norm_targets`).  `lazy`: only the reduce is
            launched and every entry comes back as (dz, bnb) -- dz still EMPTY, bnb the cwn_gemm_bnb extension with which
            the consuming transposed-weight GEMM forms (and writes) it in its prologue; bnb None = dz is complete."""
            red, app, outs = [], [], []
            direct, scratch = [], []
            lazy = lazy and all(bnb_ok(dy, z) for _, dy, z in items if z.numel())
            for st, dy, z in items:
                dz = torch.empty(z.shape, dtype=torch.float32, device=dev)
                if not z.numel():
                    outs.append((dz, None) if lazy else dz)
                    continue
                # (the layer's output dropout: `dy` arrives w.r.t. the dropped activation; the reduce launch multiplies it on
                #  the way in and writes the product, which everything behind it reads)
                dy_in, drop = dy, drop_of.get(id(st))
                if drop is not None:
                    dy = drop[1]
                aff = aff_of.get(id(st))
                s12 = G[id(st)][2]
                tgt = norm_targets[id(st)]
                if fused_norm and dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and (s12 is None or s12.data_ptr() % 16 == 0):
                    (direct if tgt is not None else scratch).append(
                        _norm_desc(z, dy=dy, out=dz, aff=aff, s12=tgt if tgt is not None else s12))
                    outs.append(dz)
                    continue
                have_slots = id(st) in filled
                if have_slots and not lazy:          # (the sums are needed as arrays: rare -- the lazy form is the stage launch's)
                    s12.copy_(bslot_of[id(st)].sum(0))
                    have_slots = False
                elif st.is_bn and not have_slots:
                    rd = _norm_desc(z, dy=dy_in, aff=aff, s12=s12)
                    if drop is not None:
                        rd.drop, rd.dy_out, rd.lddy_out = drop[0], dy.data_ptr(), ld(dy)
                    red.append(rd)
                if lazy:
                    b = _ffi.GemmBnb(z=z.data_ptr(), dz=dz.data_ptr(), ldz=z.stride(0), lddz=dz.stride(0), relu=1)
                    b.s_slots = bslot_of[id(st)] if have_slots else None
                    if aff is not None:
                        b.scale, b.shift, b.mean, b.rstd = (aff[r].data_ptr() for r in range(4))
                        b.s1, b.s2 = s12[0].data_ptr(), s12[1].data_ptr()
                        if tgt is not None:
                            b.acc1, b.acc2 = tgt[0].data_ptr(), tgt[1].data_ptr()
                    outs.append((dz, b))
                    continue
                d = _norm_desc(z, dy=dy, out=dz, aff=aff, s12=s12)
                if st.is_bn and tgt is not None:
                    d.acc1, d.acc2 = tgt[0].data_ptr(), tgt[1].data_ptr()
                app.append(d)
                outs.append(dz)
            if direct:
                _ffi.norm_bwd(direct, dev, accumulate=True)
            if scratch:
                _ffi.norm_bwd(scratch, dev, accumulate=False)
            if red:
                _ffi.norm_bwd_reduce(red, dev)
            if app:
                _ffi.norm_bwd_apply(app, dev)
            return outs

        def second_view(b):
            """The extension for a second GEMM over the same input: same dz, nothing written twice."""
            c = _ffi.GemmBnb.from_buffer_copy(b)
            c.dz, c.acc1, c.acc2 = None, None, None
            return c

        def prologue(st: Optional[Stage]):
            if st is None or not st.is_bn:
                return None, None
            a = aff_of[id(st)]
            return a[0], a[1]

        ld = lambda t: t.stride(0) if t.size(0) > 1 else t.size(1)

        # weight gradients whose targets are all the parameters' own .grad buffers may wait for the end of the backward
        can_defer = ops.ACCUMULATE_INTO_GRAD and all(tw is not None and (st.lin.bias is None or tb is not None)
                                                     for st, (tw, tb) in zip(stages, targets))
        def mark_filled(recs, to):
            for (_, o1, o2), (t1, t2) in zip(recs, to):
                if o1 is not None:
                    filled.add(id(t1))
                if o2 is not None:
                    filled.add(id(t2))

        # ---- combine stage -------------------------------------------------------------------
        if plan.cb is None:          # (the caller's combine network runs on torch: dH is the gradient of every branch's activated output)
            flat_dH = [g if g is not None else torch.zeros_like(Z[k // nb][k % nb][-1]) for k, g in enumerate(dH)]
            dy = [[ops._rowmajor(flat_dH[nb * i + br], 'grad') for br in range(nb)] for i in range(nd)]
        else:
            dH = [g if g is not None else torch.zeros_like(z) for g, z in zip(dH, Z3)]
            dH = [ops._rowmajor(g, 'grad') for g in dH]
            dH_in = list(dH)
            for i, site in enumerate(ctx.drop_sites):
                if site is None or not dH[i].numel():
                    continue
                rec = ops.dropout_record(dev, plan.out_drop, site)
                if plan.cb[i].is_bn and not fused_norm:           # the reduce launch of this stage takes it
                    dH[i] = torch.empty_like(dH[i])
                    drop_of[id(plan.cb[i])] = (rec, dH[i])
                else:                                             # no reduce launch (an identity norm): a launch of its own
                    dH[i] = dH_in[i] = ops.dropout_apply(dH[i], rec)
            # (round 6) a dy that IS the dx the next layer's blocked backward returned comes with its column sums in the slots this
            # stage's forward registered (ops.bn_out_register): no reduce launch for it
            if live_bwd and getattr(ctx, 'bn_ext', None):
                for i, e in enumerate(ctx.bn_ext):
                    if e is not None and dH_in[i].numel() and ops.bn_sums_ready(dH_in[i], e[0]):
                        bslot_of[id(plan.cb[i])] = e[1]
                        filled.add(id(plan.cb[i]))
                        ops.BN_BWD_FUSED[0] += 1
                        ctx.bn_ext[i] = None            # (one shot: a second backward over this forward reduces by itself)
            pend3 = norm_backward([(plan.cb[i], dH_in[i], Z3[i]) for i in range(nd)], lazy=True)
            lazy3 = bool(pend3) and isinstance(pend3[0], tuple)
            dZ3 = [p[0] for p in pend3] if lazy3 else pend3
            tn, nn = [], []
            dA = []
            stage_bwd = []          # the same products for cwn_dense_stage_bwd_f32 (ops.run_stage_bwd), when every one has the lazy form
            live_recs, live_to = [], []   # per entry: the slot-sum forms (ops.run_stage_bwd) and the stages whose reduce they take over

            # branch pairs of a dimension: one K-pair of the weight gradient and one backward-stage entry each (nb = 2: the pair)
            pairs = [tuple(range(lo, min(lo + 2, nb))) for lo in range(0, nb, 2)]
            for i in range(nd):
                st = plan.cb[i]
                W = P[id(st)][0]
                dW, db, _ = G[id(st)]
                lasts = [plan.chains[i][br][-1] for br in range(nb)]
                hu = lasts[0].lin.out_features
                same_width = all(l.lin.out_features == hu for l in lasts) and W.size(1) == nb * hu
                if dZ3[i].numel():
                    for pr in pairs:
                        Xa, Xb = Z[i][pr[0]][-1], (Z[i][pr[1]][-1] if len(pr) > 1 else None)
                        sc, sh = prologue(lasts[pr[0]])
                        sc2, sh2 = prologue(lasts[pr[1]]) if len(pr) > 1 else (None, None)
                        c0 = sum(l.lin.out_features for l in lasts[:pr[0]])
                        tn.append(_ffi.GemmTnDesc(
                            dZ=dZ3[i].data_ptr(), X=Xa.data_ptr(), X2=_ffi.ptr(Xb), in_scale=_ffi.ptr(sc),
                            in_shift=_ffi.ptr(sh), in_scale2=_ffi.ptr(sc2), in_shift2=_ffi.ptr(sh2),
                            dW=dW.data_ptr() + 4 * c0, db=_ffi.ptr(db) if pr[0] == 0 else None, M=dZ3[i].size(0), lddz=ld(dZ3[i]),
                            ldx=ld(Xa), ldx2=0 if Xb is None else ld(Xb), lddw=dW.stride(0), N=W.size(0), K=Xa.size(1),
                            K2=0 if Xb is None else Xb.size(1), in_relu=3 if Xb is not None else 1))
                if lazy3 and pend3[i][1] is not None:
                    # the pieces of dA as products over the same dz (each 128 or 64 columns wide: the kernel's shapes)
                    out = torch.empty(dZ3[i].size(0), W.size(1), dtype=torch.float32, device=dev)
                    b = pend3[i][1]
                    c0 = 0
                    for br in range(nb):
                        h = lasts[br].lin.out_features
                        nn.append(ops.Gemm(X=dH[i], W=W[:, c0:c0 + h], w_trans=True, out=out[:, c0:c0 + h],
                                           bnb=b if br == 0 else second_view(b)))
                        c0 += h
                    dA.append(out)
                    for pr in pairs:
                        c0 = pr[0] * hu
                        ent = (dH[i], b if pr[0] == 0 else second_view(b), W, out[:, c0:c0 + hu],
                               out[:, c0 + hu:c0 + 2 * hu] if len(pr) > 1 else None)
                        stage_bwd.append((ent if nb == 2 else ent + (c0,)) if same_width else None)
                        live_recs.append((getattr(b, 's_slots', None) if pr[0] == 0 else None,
                                          out_record(lasts[pr[0]], Z[i][pr[0]][-1]),
                                          out_record(lasts[pr[1]], Z[i][pr[1]][-1]) if len(pr) > 1 else None))
                        live_to.append((lasts[pr[0]], lasts[pr[1]] if len(pr) > 1 else None))
                else:
                    nn.append(ops.Gemm(X=dZ3[i], W=W, w_trans=True))
                    dA.append(None)
                    stage_bwd.append(None)
                    live_recs.append((None, None, None))
                    live_to.append((None, None))
            keep_all = [dZ3, Z, A0, aff_of, dH]
            # [M, sum of the branch widths] per dimension (before the weight gradients: with the lazy form the launch WRITES dZ3)
            if all(e is not None for e in stage_bwd) and ops.run_stage_bwd(stage_bwd, dev, live_recs if live_bwd else None):
                if live_bwd:
                    mark_filled(live_recs, live_to)
            else:
                res = ops.run_gemm(nn, dev)
                k = 0
                for i in range(nd):
                    if dA[i] is None:
                        dA[i] = res[k]
                        k += 1
                    else:
                        k += nb
            if tn:
                _ffi.gemm_tn(tn, dev, keep=keep_all, deferrable=can_defer)
            dy = []
            for i in range(nd):
                cols, c0 = [], 0
                for br in range(nb):
                    h = plan.chains[i][br][-1].lin.out_features
                    cols.append(dA[i][:, c0:c0 + h])
                    c0 += h
                dy.append(cols)
        # ---- update stages, last to first ------------------------------------------------------
        for s in range(depth - 1, -1, -1):
            items = []
            for i in range(nd):
                for br, chain in enumerate(plan.chains[i]):
                    items.append((chain[s], dy[i][br], Z[i][br][s]))
            # (the lazy form hands dz to the input-gradient GEMM's prologue, which exists for 16-byte rows only: a first layer
            #  over 1-wide features -- REDDIT-BINARY's constant vertex feature, mp/models.py:112-260 -- takes the apply launch)
            wide = all(P[id(chain[s])][0].size(1) % 4 == 0 and P[id(chain[s])][0].stride(0) % 4 == 0
                       for i in range(nd) for chain in plan.chains[i])
            pend = norm_backward(items, lazy=wide)
            lazy_s = bool(pend) and isinstance(pend[0], tuple)
            dZ = [p[0] for p in pend] if lazy_s else pend
            tn, nn, k = [], [], 0
            stage_bwd, live_recs, live_to = [], [], []
            for i in range(nd):
                for br, chain in enumerate(plan.chains[i]):
                    st = chain[s]
                    W = P[id(st)][0]
                    dW, db, _ = G[id(st)]
                    X = A0[i][br] if s == 0 else Z[i][br][s - 1]
                    sc, sh = prologue(chain[s - 1] if s > 0 else None)
                    dz = dZ[k]
                    if dz.numel():
                        tn.append(_ffi.GemmTnDesc(
                            dZ=dz.data_ptr(), X=X.data_ptr(), X2=None, in_scale=_ffi.ptr(sc),
                            in_shift=_ffi.ptr(sh), in_scale2=None, in_shift2=None, dW=dW.data_ptr(),
                            db=_ffi.ptr(db), M=dz.size(0), lddz=ld(dz), ldx=ld(X), ldx2=0,
                            lddw=dW.stride(0), N=W.size(0), K=X.size(1), K2=0, in_relu=1 if s > 0 else 0))
                    if lazy_s and pend[k][1] is not None:
                        nn.append(ops.Gemm(X=dy[i][br], W=W, w_trans=True, bnb=pend[k][1]))
                        stage_bwd.append((dy[i][br], pend[k][1], W, torch.empty(dz.size(0), W.size(1), dtype=torch.float32, device=dev),
                                          None))
                        live_recs.append((getattr(pend[k][1], 's_slots', None),
                                          out_record(chain[s - 1], Z[i][br][s - 1]) if s > 0 else None, None))
                        live_to.append((chain[s - 1] if s > 0 else None, None))
                    else:
                        nn.append(ops.Gemm(X=dz, W=W, w_trans=True))
                        stage_bwd.append(None)
                        live_recs.append((None, None, None))
                        live_to.append((None, None))
                    k += 1
            # (before the weight gradients: with the lazy form the launch writes dZ)
            if all(e is not None for e in stage_bwd) and ops.run_stage_bwd(stage_bwd, dev, live_recs if live_bwd else None):
                res = [e[3] for e in stage_bwd]
                if live_bwd:
                    mark_filled(live_recs, live_to)
            else:
                for st_, _, _ in items:              # (cwn_gemm_bnb reads s1 / s2 as arrays)
                    if id(st_) in filled:
                        G[id(st_)][2].copy_(bslot_of[id(st_)].sum(0))
                res = ops.run_gemm(nn, dev)
            if tn:
                _ffi.gemm_tn(tn, dev, keep=[dZ, Z, A0, aff_of, dy], deferrable=can_defer)
            k = 0
            for i in range(nd):
                for br in range(nb):
                    dy[i][br] = res[k]
                    k += 1
        grads: List[Optional[Tensor]] = [None]
        for i in range(nd):
            grads += list(dy[i])
        for st, (tw, tb) in zip(stages, targets):
            dW, db, s12 = G[id(st)]
            W, b, gamma, beta = P[id(st)]
            gg = gb = None
            if st.is_bn and norm_targets[id(st)] is None:     # (else the launches above have added the sums to the .grad buffers)
                gg = s12[1] if gamma is not None else None
                gb = s12[0] if beta is not None else None
            grads += [dW if tw is None else None,
                      (db if tb is None else None) if b is not None else None, gg, gb]
        return tuple(grads)


def dense_train(plan: _Plan, outs: Sequence[Tensor]) -> List[Tensor]:
    """outs = [out_up_0, out_bd_0, out_up_1, ...] (what SparseCINConv.propagate_all returns)."""
    flat = []
    for st in plan.stages():
        flat += list(_stage_tensors(st))
    return list(_DenseTrain.apply(plan, *outs, *flat))
