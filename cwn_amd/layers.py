"""Cellular message-passing layers on the MI355X engine.

Host-side mirror of the reference's mp/layers.py: same class names, constructor arguments,
`forward` signatures and parameter names (state_dicts are interchangeable -- the golden tests load
the reference's state_dict), so mp/models.py / mp/molec_models.py-style model code runs on it
unchanged.  What changes is HOW a layer runs:

  * SparseCINConv.forward (mp/layers.py:333-342) issues ONE aggregation launch for all cochain
    dimensions and both adjacencies (upper + boundary), with the GIN self terms
    `+(1+eps)*x` (:191-192) and the zero rows of absent adjacencies folded into it;
  * the coboundary message  ReLU(Linear_{2F->F}(cat(x_j, up_attr)))  (:290-293) is evaluated as
    ReLU(Y1[j] + Y2[c]) with Y1 = X_d W[:, :F]^T + b and Y2 = X_{d+1} W[:, F:]^T: the per-message
    `[E, 2F]` matrix and GEMM disappear (GEMM rows drop from E to N_d + N_{d+1}); the result
    differs from the reference only by fp32 summation order (< 1e-5, tested);
  * `up_attr` arrives lazily (IndexedRows), so the K3 gather of data/complex.py:579-580 is never
    materialised;
  * anything not recognised (custom message nets, other activations) takes the generic
    gather-kernel -> hook -> segmented-reduce-kernel path of CochainMessagePassing and is still
    correct.
"""
import os
import weakref
from abc import ABC, abstractmethod
from typing import Any, Callable, List, Optional

import torch
from torch import Tensor
from torch.nn import BatchNorm1d as BN, Linear, ReLU, Sequential

from . import ops
from .cell_mp import CochainMessagePassing, CochainMessagePassingParams, IndexedRows, dense
from .csr import Adjacency, cached_adjacency


def reset(nn):
    """torch_geometric.nn.inits.reset semantics (used at mp/layers.py:88-92, 201-208)."""
    def _reset(item):
        if hasattr(item, 'reset_parameters'):
            item.reset_parameters()
    if nn is not None:
        if hasattr(nn, 'children') and len(list(nn.children())) > 0:
            for item in nn.children():
                _reset(item)
        else:
            _reset(nn)


class Catter(torch.nn.Module):
    """mp/layers.py:263-268."""

    def forward(self, x):
        return torch.cat(x, dim=-1)


class FirstOf(torch.nn.Module):
    """The `lambda xs: xs[0]` of mp/layers.py:295 as a named module, so the engine can recognise
    the identity message and fuse it."""

    def forward(self, xs):
        return xs[0]


class Passthrough(torch.nn.Module):
    """The `lambda x: x` of mp/layers.py:299."""

    def forward(self, x):
        return x


def _attr_operand(attr):
    """(B matrix, ib_mode) for an attribute that is lazy (gather through the shared-cell index) or
    dense (one row per entry)."""
    if isinstance(attr, IndexedRows):
        return attr.src, 'aux'
    return attr, 'perm'


def _is_cat_linear_relu(nn) -> bool:
    return (isinstance(nn, Sequential) and len(nn) == 3 and isinstance(nn[0], Catter)
            and isinstance(nn[1], Linear) and isinstance(nn[2], ReLU))


FUSED_DENSE_TRAINING = True   # set False to run the update / combine networks as torch modules
FUSED_CINPP_COMBINE = os.environ.get('CWN_FUSED_CINPP_COMBINE') != '0'    # False: CINppConv's combine network as torch modules behind the fused branches
FUSED_CIN_TRAINING = os.environ.get('CWN_FUSED_CIN_TRAINING') != '0'      # False: CINCochainConv's training forward on the generic path
FUSED_UPDATE_MLP = os.environ.get('CWN_FUSED_UPDATE_MLP') != '0'   # False: the update / combine networks as three grouped GEMM launches
BLOCKED_TRAIN_FORWARD = os.environ.get('CWN_BLOCKED_TRAIN_FORWARD') != '0'    # the training forward through the blocked kernel too
BLOCKED_LAYER = os.environ.get('CWN_BLOCKED_LAYER') != '0'   # False: propagate scope as grouped GEMM + CSR aggregation
CSR_REUSE = True              # blocked layer kernel: sort a batch's adjacencies once, later layers load the result
# One workgroup per item and one item per CU at a time: the blocked kernel wins while the items fit the chip
# a few times over (measured on ZINC-like batches, tools/range_of_use.sh, M cells/s blocked vs CSR path: 256
# complexes 646 vs 428, 512: 761 vs 588, 1024: 788 vs 759, 2048: 836 vs 900, 8192: 863 vs 981); beyond
# ~2600 items (two per complex) the two-kernel path's streaming wins.
BLOCKED_MAX_ITEMS = int(os.environ.get('CWN_BLOCKED_MAX_ITEMS', '2600'))
# Which form of the blocked kernel a launch takes (include/cwn_hip.h, cwn_layer_plan.variant): 'auto' = the 16-wave
# one-per-CU form while the items fit the chip once (TWO_PER_CU_MIN_ITEMS), the 8-wave two-per-CU form beyond that
# when every complex fits its smaller caps; '0' / '1' force one (A/B measurements, tests).
LAYER_VARIANT = os.environ.get('CWN_LAYER_VARIANT', 'auto')
# A complex beyond a workgroup's LDS (a molecule of more than ~44 atoms at width 128, ~115 at 64) is streamed by its own
# workgroup inside the blocked launch (BIG records) instead of sending the whole batch to the two-kernel path -- while
# such complexes are the exception (at most BIG_MAX_SHARE of the items: a batch of hub complexes IS the streaming case).
BIG_ITEMS = {'0': False, 'always': 'always'}.get(os.environ.get('CWN_BIG_ITEMS', '1'), True)   # 'always': skip the cost model below (tests, A/B)
BIG_MAX_SHARE = 0.25
TWO_PER_CU_MIN_ITEMS = int(os.environ.get('CWN_TWO_PER_CU_MIN_ITEMS', '256'))
def _streaming_pays(table, F: int) -> bool:
    """BIG records against sending the whole batch to the two-kernel path.  The launch lasts as long as its longest
    workgroup: a streamed complex takes ~0.07 us per staged row at width 128 (~0.035 at 64; a 200-atom molecule = 415
    rows ~ 30 us), the rest of the batch ~8.7 us per round of 256 items, and the two-kernel path ~1.9 x the blocked time
    of the same batch without its giants.  Calibrated on tools/big_items_bench.py (share of the all-small rate, streamed
    / two-kernel path): molhiv-512 with 5 % giants of 60 - 200 atoms 0.87 / 0.43, ZINC-512 0.77 / 0.67, ZINC-128 0.38 /
    0.54 -- the one case that keeps the two-kernel path: a launch of 9 us cannot hide a 30-us workgroup."""
    recs = table.big_records
    rows = max(int(r[11]) + int(r[5]) for r in recs)
    t_big = (0.07 if F == 128 else 0.035) * rows
    rounds = max(1, -(-(table.n_items - table.n_big) // TWO_PER_CU_MIN_ITEMS))
    return t_big < 1.9 * 8.7 * rounds


def _mixed_wins(tm, table0, lower: int, F: int = 128) -> bool:
    """Two launches (two-per-CU form for the complexes that fit + 16-wave form for the rest) against one launch of the
    16-wave form for all: rounds as in _two_per_cu_wins, plus one launch boundary.  Measured on batches with the size
    spread of the real ZINC subset (9 - 38 atoms, tools/ab_mixed.sh; M cells/s mixed / 16-wave form with big items /
    two-kernel path): batch 512: 798 / 849 / 577 (the one-round second launch costs more than it saves), batch 2048:
    1054 / 954 / 851; batch 128 has one round either way: 558 / 558 / 325."""
    slots = TWO_PER_CU_MIN_ITEMS
    rounds = lambda n, per: -(-n // per) if n <= 2 * per else n / per      # (a partial LAST round of many costs its share)
    t1, t0 = tm.parts
    if t0.n_big and BIG_ITEMS != 'always' and not (BIG_ITEMS and _streaming_pays(t0, F)):
        return False
    mixed = 1.3 * rounds(t1.n_items, 2 * slots) + rounds(t0.n_items, slots) + 0.2
    alone = rounds(table0.n_items if table0 is not None else lower, slots) if table0 is not None else 2.0 * rounds(lower, slots)
    return mixed < alone


def _two_per_cu_wins(n0: int, n1: int) -> bool:
    """A launch of n0 items in the 16-wave form (256 at a time) against n1 items in the two-per-CU form (512 at a
    time, each ~1.3 x as long: 8 waves, a neighbour on the CU): whole rounds while a launch is a few rounds (the last
    round costs as much as a full one), the ratio beyond.  Calibrated on tools/ab_variant.sh (M cells/s, 16-wave /
    two-per-CU): ZINC 160 complexes 555 / 639, 192: 649 / 744, 256: 756 / 905, 384: 844 / 1009; molhiv 256: 1351 /
    1575, 512: 2143 / 1974 (445 items in two rounds against 677 in two longer ones: the one case the 16-wave form
    keeps), 1024: 2354 / 2550, 2048: 2603 / 2905."""
    slots = TWO_PER_CU_MIN_ITEMS                       # = CUs
    r0 = -(-n0 // slots) if n0 <= 4 * slots else n0 / slots
    r1 = -(-n1 // (2 * slots)) if n1 <= 8 * slots else n1 / (2 * slots)
    return 1.3 * r1 < r0


class ByModule:
    """A mapping keyed by a module's IDENTITY that forgets a module when it dies -- weakref.WeakKeyDictionary's job at a plain
    dict's price: `get` there builds a weak reference per call (~0.7 us; five look-ups per layer of an eager forward).  An
    entry holds a weak reference with a callback of its own, so a recycled id() never answers for another module."""

    def __init__(self):
        self._d = {}

    def get(self, module, default=None):
        hit = self._d.get(id(module))
        return hit[1] if hit is not None and hit[0]() is module else default

    def __contains__(self, module) -> bool:
        hit = self._d.get(id(module))
        return hit is not None and hit[0]() is module

    def __getitem__(self, module):
        hit = self._d.get(id(module))
        if hit is None or hit[0]() is not module:
            raise KeyError(module)
        return hit[1]

    def __setitem__(self, module, value) -> None:
        key = id(module)
        self._d[key] = (weakref.ref(module, lambda _r, k=key, d=self._d: d.pop(k, None) if (d.get(k) or (None,))[0] is _r else None), value)

    def __delitem__(self, module) -> None:
        if module not in self:
            raise KeyError(module)
        del self._d[id(module)]

    def pop(self, module, *default):
        if module in self:
            return self._d.pop(id(module))[1]
        if default:
            return default[0]
        raise KeyError(module)

    def setdefault(self, module, default):
        if module not in self:
            self[module] = default
        return self._d[id(module)][1]

    def __len__(self) -> int:
        return len(self._d)

    def clear(self) -> None:
        self._d.clear()

    def values(self):
        return [v for _r, v in self._d.values()]


# prepared launches per (layer module, batch): kept OUTSIDE the modules (ctypes records do not deepcopy / pickle)
_BLOCKED_CACHE = ByModule()
_MLP_CACHE = ByModule()            # layer module -> (ops.MlpLaunch, (start, dims, outputs), BatchNorm modules)
FUSED_OGB_FRONT = os.environ.get('CWN_FUSED_OGB_FRONT', '1') != '0'      # (A/B: '0' = the separate launches of the OGB encoders' front)
_FRONT_CACHE = ByModule()          # embedding front module -> {id(boundary_index_1): (ops.FrontLaunch, (rings?, reduce))}


def _ffi_dyn() -> bool:
    from . import _ffi
    return bool(_ffi.DYN_ROWS)


def _fold_norm(norm, width: int):
    """Eval-mode normalisation as a per-column affine (scale, shift) for the GEMM epilogue:
    BatchNorm1d with running statistics -> (w / sqrt(rv + eps), b - rm * scale); Identity ->
    (None, None).  Cached on the module and refreshed when any of its tensors changes.
    Returns None when the module cannot be folded (LayerNorm, BatchNorm without running stats)."""
    if isinstance(norm, torch.nn.Identity):
        return (None, None)
    if (not isinstance(norm, BN) or norm.training or norm.running_mean is None
            or norm.num_features != width):
        return None     # batch statistics (training) are not a fixed affine
    # (num_batches_tracked: torch's native batch_norm writes the running statistics of a TRAINING-mode forward without moving
    # their version counters; the module's own `num_batches_tracked.add_(1)` does move one -- found by tools/fuzz_round5.py in
    # round 5: an eval forward after a training-mode forward through the torch modules folded the OLD statistics)
    tensors = [t for t in (norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.num_batches_tracked) if t is not None]
    key = (ops.STATE_EPOCH,) + tuple((t.data_ptr(), ops._ffi.tver(t)) for t in tensors)
    hit = getattr(norm, '_cwn_fold', None)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        scale = torch.rsqrt(norm.running_var + norm.eps)
        if norm.weight is not None:
            scale = scale * norm.weight
        shift = -norm.running_mean * scale
        if norm.bias is not None:
            shift = shift + norm.bias
        fold = (scale.contiguous(), shift.contiguous())
    norm._cwn_fold = (key, fold)
    return fold


def _mlp_stages(nn):
    """[(Linear, norm), ...] of a Sequential of (Linear, norm, ReLU) groups, or None."""
    if not isinstance(nn, Sequential) or len(nn) % 3 != 0 or len(nn) == 0:
        return None
    stages = []
    for i in range(0, len(nn), 3):
        lin, norm, act = nn[i], nn[i + 1], nn[i + 2]
        if not (isinstance(lin, Linear) and isinstance(act, ReLU)):
            return None
        stages.append((lin, norm))
    return stages


# ------------------------------------------------------------------------------------------------
# test / toy layers
# ------------------------------------------------------------------------------------------------
class DummyCochainMessagePassing(CochainMessagePassing):
    """Parameter-free layer used by the reference's tests (mp/layers.py:14-40): messages are
    x_j + attr.  Fused here as CWN_MSG_A_PLUS_B."""

    def __init__(self, up_msg_size, down_msg_size, boundary_msg_size=None, use_boundary_msg=False,
                 use_down_msg=True):
        super().__init__(up_msg_size, down_msg_size, boundary_msg_size=boundary_msg_size,
                         use_boundary_msg=use_boundary_msg, use_down_msg=use_down_msg)

    def message_up(self, up_x_j: Tensor, up_attr: Tensor) -> Tensor:
        return up_x_j + up_attr

    def message_down(self, down_x_j: Tensor, down_attr: Tensor) -> Tensor:
        return down_x_j + down_attr

    def message_and_aggregate_up(self, up_adj_t: Adjacency, x, up_attr) -> Tensor:
        B, mode = _attr_operand(up_attr)
        return ops.aggregate(up_adj_t, up_adj_t.n_dst, x, msg_op=ops.MSG_A_PLUS_B, B=B, ib_mode=mode,
                             reduce=self.aggr_up)

    def message_and_aggregate_down(self, down_adj_t: Adjacency, x, down_attr) -> Tensor:
        B, mode = _attr_operand(down_attr)
        return ops.aggregate(down_adj_t, down_adj_t.n_dst, x, msg_op=ops.MSG_A_PLUS_B, B=B,
                             ib_mode=mode, reduce=self.aggr_down)

    def forward(self, cochain: CochainMessagePassingParams):
        up_out, down_out, boundary_out = self.propagate(
            cochain.up_index, cochain.down_index, cochain.boundary_index, x=cochain.x,
            up_attr=cochain.kwargs['up_attr'], down_attr=cochain.kwargs['down_attr'],
            boundary_attr=cochain.kwargs['boundary_attr'])
        return cochain.x + up_out + down_out + boundary_out


class DummyCellularMessagePassing(torch.nn.Module):
    """mp/layers.py:43-59."""

    def __init__(self, input_dim=1, max_dim: int = 2, use_boundary_msg=False, use_down_msg=True):
        super().__init__()
        self.max_dim = max_dim
        self.mp_levels = torch.nn.ModuleList(
            DummyCochainMessagePassing(input_dim, input_dim, boundary_msg_size=input_dim,
                                       use_boundary_msg=use_boundary_msg, use_down_msg=use_down_msg)
            for _ in range(max_dim + 1))

    def forward(self, *cochain_params: CochainMessagePassingParams):
        assert len(cochain_params) <= self.max_dim + 1
        return [self.mp_levels[d].forward(cochain_params[d]) for d in range(len(cochain_params))]


# ------------------------------------------------------------------------------------------------
# CIN (upper + lower adjacencies, per-message networks)
# ------------------------------------------------------------------------------------------------
class CINCochainConv(CochainMessagePassing):
    """mp/layers.py:62-103.  The message networks are arbitrary callables on cat(x_j, attr): in
    general they run on the generic path (gather kernel -> network -> segmented-reduce kernel);
    the Linear -> ReLU -> BatchNorm form the reference's models build (mp/models.py:40-47) has a
    fused inference path (_fused_inference)."""

    def __init__(self, up_msg_size: int, down_msg_size: int, msg_up_nn: Callable,
                 msg_down_nn: Callable, update_nn: Callable, eps: float = 0., train_eps: bool = False):
        super().__init__(up_msg_size, down_msg_size, use_boundary_msg=False)
        self.msg_up_nn = msg_up_nn
        self.msg_down_nn = msg_down_nn
        self.update_nn = update_nn
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer('eps', torch.Tensor([eps]))
        self.reset_parameters()

    def forward(self, cochain: CochainMessagePassingParams):
        fused = self._fused_inference(cochain)
        if fused is None:
            fused = self._fused_training(cochain)
        if fused is not None:
            return self.update_nn(fused)
        out_up, out_down, _ = self.propagate(cochain.up_index, cochain.down_index, None, x=cochain.x,
                                             up_attr=cochain.kwargs['up_attr'],
                                             down_attr=cochain.kwargs['down_attr'])
        out_up = out_up + (1 + self.eps) * cochain.x
        out_down = out_down + (1 + self.eps) * cochain.x
        return self.update_nn(out_up + out_down)

    # ---- fused inference -------------------------------------------------------------------------
    @staticmethod
    def _message_form(nn):
        """(Linear, (scale, shift) | None) when `nn` is Linear -> ReLU [-> BatchNorm1d(eval) | Identity]
        (the conv_up / conv_down of mp/models.py:40-47), else None."""
        if not isinstance(nn, Sequential) or len(nn) not in (2, 3):
            return None
        if not (isinstance(nn[0], Linear) and isinstance(nn[1], ReLU)):
            return None
        if len(nn) == 2:
            return nn[0], (None, None)
        fold = _fold_norm(nn[2], nn[0].out_features)
        return None if fold is None else (nn[0], fold)

    def _fused_plan(self, cochain: CochainMessagePassingParams):
        """The dense products and aggregation streams of _fused_inference, unlaunched (CINConv
        groups the plans of all dimensions into one GEMM launch + one aggregation launch), or None
        when the fused path does not apply."""
        x = cochain.x
        if torch.is_grad_enabled() or x is None or not x.is_cuda:
            return None
        if (self.aggr_up or 'add') != 'add' or (self.aggr_down or 'add') != 'add':
            return None
        n, F = x.size(0), x.size(1)
        jobs = []
        for index, name, attr, nn in ((cochain.up_index, 'up', cochain.kwargs.get('up_attr'), self.msg_up_nn),
                                      (cochain.down_index, 'down', cochain.kwargs.get('down_attr'), self.msg_down_nn)):
            if index is None or (name == 'down' and not self.use_down_msg):
                continue
            form = self._message_form(nn)
            if form is None or attr is None:
                return None
            lin = form[0]
            attr_src, mode = _attr_operand(attr)
            if lin.in_features != F + attr_src.size(1) or lin.out_features != F \
                    or max(F, attr_src.size(1)) > ops.GEMM_MAX_K:
                return None
            jobs.append((index, name, attr_src, mode, form))
        kw = dict(x=x, up_attr=cochain.kwargs.get('up_attr'), down_attr=cochain.kwargs.get('down_attr'))
        gemms, adjs = [], []
        for index, name, attr_src, mode, (lin, _) in jobs:
            gemms += [ops.Gemm(X=x, W=lin.weight, w_col0=0, bias=lin.bias),
                      ops.Gemm(X=attr_src, W=lin.weight, w_col0=F)]
            size = self.__check_input_separately__(index, None)
            adjs.append(self._adjacency(index, name, size, kw))
        return dict(x=x, jobs=jobs, gemms=gemms, adjs=adjs)

    def _fused_streams(self, plan, ys: List[Tensor]) -> List[ops.Stream]:
        x = plan['x']
        return [ops.Stream(adj=adj, n_dst=x.size(0), width=x.size(1), A=ys[2 * k], B=ys[2 * k + 1],
                           msg_op=ops.MSG_RELU_A_PLUS_B, ib_mode=job[3])
                for k, (job, adj) in enumerate(zip(plan['jobs'], plan['adjs']))]

    def _fused_finish(self, plan, outs: List[Tensor]) -> Tensor:
        """out_up + out_down (with both self terms) from the aggregated ReLU sums."""
        total = (2 * (1 + self.eps)) * plan['x']
        for out, adj, job in zip(outs, plan['adjs'], plan['jobs']):
            scale, shift = job[4][1]
            if scale is None:
                total = total + out
            else:
                deg = (adj.rowptr[1:] - adj.rowptr[:-1]).to(torch.float32).unsqueeze(1)
                total = total + out * scale + deg * shift
        return total

    def _fused_inference(self, cochain: CochainMessagePassingParams) -> Optional[Tensor]:
        """out_up + out_down of forward() without materialising a message, when nothing needs a
        gradient and both message networks are Linear -> ReLU -> BatchNorm(eval).  The per-ENTRY
        network splits exactly:  BN(relu(W [x_j | a_e] + b)) = s * relu(Y1[j] + Y2[c]) + t  with
        Y1 = X W[:, :F]^T + b, Y2 = X_attr W[:, F:]^T computed once per CELL, so
            sum_e msg_e = s * (sum_e relu(Y1[j_e] + Y2[c_e])) + deg_i * t
        -- one grouped GEMM launch and one aggregation launch for both adjacencies.  Returns None
        when it does not apply (training, other network shapes, CPU tensors)."""
        plan = self._fused_plan(cochain)
        if plan is None:
            return None
        ys = ops.run_gemm(plan['gemms'], plan['x'].device) if plan['gemms'] else []
        streams = self._fused_streams(plan, ys)
        outs = ops.aggregate_many(streams) if streams else []
        return self._fused_finish(plan, outs)

    # ---- fused training (round 4) ----------------------------------------------------------------
    @staticmethod
    def _message_form_train(nn):
        """(Linear, norm | None) when `nn` is Linear -> ReLU [-> BatchNorm1d | Identity] in a state the fused training path
        reproduces, else None."""
        if not isinstance(nn, Sequential) or len(nn) not in (2, 3):
            return None
        if not (isinstance(nn[0], Linear) and isinstance(nn[1], ReLU)):
            return None
        norm = nn[2] if len(nn) == 3 else None
        if norm is None or isinstance(norm, torch.nn.Identity):
            return nn[0], None
        if not isinstance(norm, BN) or norm.num_features != nn[0].out_features:
            return None
        if norm.training and norm.track_running_stats and norm.momentum is None:
            return None                  # (cumulative moving average: not restated)
        if not norm.training and norm.running_mean is None:
            return None
        return nn[0], norm

    def _fused_training(self, cochain: CochainMessagePassingParams) -> Optional[Tensor]:
        """out_up + out_down of forward() WITH autograd and without a message ever materialised, for the
        message networks the reference's models build (mp/models.py:40-47: Linear -> ReLU -> BatchNorm).  The split of
        _fused_inference holds in training too -- relu(W [x_j | a_e] + b) = relu(Y1[j] + Y2[c]), Y1 / Y2 once per CELL on
        the MFMA kernel (ops.gemm_many: differentiable) -- and BatchNorm in TRAINING mode normalises over the ENTRIES of
        the adjacency, whose batch statistics are column sums of two aggregations:
            S_i = sum_{e -> i} r_e (CWN_MSG_RELU_A_PLUS_B),   Q_i = sum_{e -> i} r_e^2 (CWN_MSG_RELU_A_PLUS_B_SQ)
            mean = sum_i S_i / E,  var = sum_i Q_i / E - mean^2 (float64),   sum_e BN(r_e) = scale S_i + deg_i shift
        with the running statistics updated as torch does (momentum, unbiased variance).  Gradients: autograd over these
        few [n, F] tensor ops + the aggregations' own transposed launches (the squared form's is CWN_MSG_A_TIMES_2RELU).
        Per adjacency: 1 grouped GEMM + 1 aggregation launch (both sums) forward, against gather -> per-entry Linear ->
        ReLU -> BatchNorm -> scatter over E x 2F / E x F matrices.  None when it does not apply."""
        x = cochain.x
        if (not FUSED_CIN_TRAINING or not torch.is_grad_enabled() or x is None or not x.is_cuda or x.dtype != torch.float32
                or x.dim() != 2):
            return None
        if (self.aggr_up or 'add') != 'add' or (self.aggr_down or 'add') != 'add':
            return None
        n, F = x.size(0), x.size(1)
        jobs = []
        for index, name, attr, nn in ((cochain.up_index, 'up', cochain.kwargs.get('up_attr'), self.msg_up_nn),
                                      (cochain.down_index, 'down', cochain.kwargs.get('down_attr'), self.msg_down_nn)):
            if index is None or (name == 'down' and not self.use_down_msg):
                continue
            form = self._message_form_train(nn)
            if form is None or attr is None:
                return None
            lin, norm = form
            attr_src, mode = _attr_operand(attr)
            if lin.in_features != F + attr_src.size(1) or lin.out_features != F or max(F, attr_src.size(1)) > ops.GEMM_MAX_K:
                return None
            E = int(index.size(1))
            if isinstance(norm, BN) and norm.training and E < 2:
                return None              # (torch refuses BatchNorm over fewer than two values per channel: let it)
            jobs.append((index, name, attr_src, mode, lin, norm, E))
        kw = dict(x=x, up_attr=cochain.kwargs.get('up_attr'), down_attr=cochain.kwargs.get('down_attr'))
        total = (2 * (1 + self.eps)) * x
        if not jobs:
            return total
        gemms = []
        for index, name, attr_src, mode, lin, norm, E in jobs:
            gemms += [ops.Gemm(X=x, W=lin.weight, w_col0=0, bias=lin.bias), ops.Gemm(X=attr_src, W=lin.weight, w_col0=F)]
        ys = ops.gemm_many(gemms)
        streams, adjs = [], []
        for k, (index, name, attr_src, mode, lin, norm, E) in enumerate(jobs):
            size = self.__check_input_separately__(index, None)
            adj = self._adjacency(index, name, size, kw)
            adjs.append(adj)
            streams.append(ops.Stream(adj=adj, n_dst=n, width=F, A=ys[2 * k], B=ys[2 * k + 1], msg_op=ops.MSG_RELU_A_PLUS_B,
                                      ib_mode=mode))
            if isinstance(norm, BN) and norm.training:
                streams.append(ops.Stream(adj=adj, n_dst=n, width=F, A=ys[2 * k], B=ys[2 * k + 1],
                                          msg_op=ops.MSG_RELU_A_PLUS_B_SQ, ib_mode=mode))
        outs = ops.aggregate_many(streams)
        o = 0
        for (index, name, attr_src, mode, lin, norm, E), adj in zip(jobs, adjs):
            S = outs[o]
            o += 1
            if norm is None:
                total = total + S
                continue
            deg = (adj.rowptr[1:] - adj.rowptr[:-1]).to(torch.float32).unsqueeze(1)
            w = norm.weight.double() if norm.weight is not None else None
            if norm.training:
                Q = outs[o]
                o += 1
                mean = S.sum(0, dtype=torch.float64) / E
                var = (Q.sum(0, dtype=torch.float64) / E - mean * mean).clamp_min(0.0)      # biased, as BatchNorm normalises
                if norm.track_running_stats and norm.running_mean is not None:
                    with torch.no_grad():
                        m = float(norm.momentum)
                        norm.running_mean.mul_(1 - m).add_(mean.to(norm.running_mean.dtype), alpha=m)
                        norm.running_var.mul_(1 - m).add_((var * (E / (E - 1))).to(norm.running_var.dtype), alpha=m)
                        norm.num_batches_tracked += 1
            else:
                mean, var = norm.running_mean.double(), norm.running_var.double()
            scale = torch.rsqrt(var + norm.eps)
            if w is not None:
                scale = scale * w
            shift = -mean * scale
            if norm.bias is not None:
                shift = shift + norm.bias.double()
            total = total + S * scale.float() + deg * shift.float()
        return total

    def reset_parameters(self):
        reset(self.msg_up_nn)
        reset(self.msg_down_nn)
        reset(self.update_nn)
        self.eps.data.fill_(self.initial_eps)

    def message_up(self, up_x_j: Tensor, up_attr: Tensor) -> Tensor:
        if up_attr is not None:
            return self.msg_up_nn(torch.cat([up_x_j, up_attr], dim=-1))
        return self.msg_up_nn(up_x_j)

    def message_down(self, down_x_j: Tensor, down_attr: Tensor) -> Tensor:
        return self.msg_down_nn(torch.cat([down_x_j, down_attr], dim=-1))


class CINConv(torch.nn.Module):
    """mp/layers.py:106-124."""

    def __init__(self, up_msg_size: int, down_msg_size: int, msg_up_nn: Callable,
                 msg_down_nn: Callable, update_nn: Callable, eps: float = 0.,
                 train_eps: bool = False, max_dim: int = 2):
        super().__init__()
        self.max_dim = max_dim
        self.mp_levels = torch.nn.ModuleList(
            CINCochainConv(up_msg_size, down_msg_size, msg_up_nn, msg_down_nn, update_nn, eps, train_eps)
            for _ in range(max_dim + 1))

    def forward(self, *cochain_params: CochainMessagePassingParams):
        assert len(cochain_params) <= self.max_dim + 1
        n = len(cochain_params)
        # inference: the fused plans of ALL dimensions in one GEMM launch (per <= 8) + one aggregation
        plans = [self.mp_levels[d]._fused_plan(cochain_params[d]) for d in range(n)]
        if all(p is not None for p in plans):
            gemms = [gm for p in plans for gm in p['gemms']]
            ys = ops.run_gemm(gemms, cochain_params[0].x.device) if gemms else []
            streams, k = [], 0
            for d, p in enumerate(plans):
                m = len(p['gemms'])
                streams.append(self.mp_levels[d]._fused_streams(p, ys[k:k + m]))
                k += m
            flat = [st for sts in streams for st in sts]
            outs = ops.aggregate_many(flat) if flat else []
            res, k = [], 0
            for d, (p, sts) in enumerate(zip(plans, streams)):
                lvl = self.mp_levels[d]
                res.append(lvl.update_nn(lvl._fused_finish(p, outs[k:k + len(sts)])))
                k += len(sts)
            return res
        return [self.mp_levels[d].forward(cochain_params[d]) for d in range(n)]


class EdgeCINConv(torch.nn.Module):
    """mp/layers.py:127-151: CIN up to 1-cells."""

    def __init__(self, up_msg_size: int, down_msg_size: int, v_msg_up_nn: Callable,
                 e_msg_down_nn: Callable, e_msg_up_nn: Callable, v_update_nn: Callable,
                 e_update_nn: Callable, eps: float = 0., train_eps=False):
        super().__init__()
        self.max_dim = 1
        self.mp_levels = torch.nn.ModuleList([
            CINCochainConv(up_msg_size, down_msg_size, v_msg_up_nn, lambda *args: None, v_update_nn,
                           eps, train_eps),
            CINCochainConv(up_msg_size, down_msg_size, e_msg_up_nn, e_msg_down_nn, e_update_nn, eps,
                           train_eps)])

    def forward(self, *cochain_params: CochainMessagePassingParams):
        assert len(cochain_params) <= self.max_dim + 1
        return [self.mp_levels[d].forward(cochain_params[d]) for d in range(len(cochain_params))]


# ------------------------------------------------------------------------------------------------
# SparseCIN (upper + boundary adjacencies): the flagship layer of the hot path
# ------------------------------------------------------------------------------------------------
class SparseCINCochainConv(CochainMessagePassing):
    """mp/layers.py:154-214."""

    def __init__(self, dim: int, up_msg_size: int, down_msg_size: int,
                 boundary_msg_size: Optional[int], msg_up_nn: Callable, msg_boundaries_nn: Callable,
                 update_up_nn: Callable, update_boundaries_nn: Callable, combine_nn: Callable,
                 eps: float = 0., train_eps: bool = False):
        super().__init__(up_msg_size, down_msg_size, boundary_msg_size=boundary_msg_size,
                         use_down_msg=False)
        self.dim = dim
        self.msg_up_nn = msg_up_nn
        self.msg_boundaries_nn = msg_boundaries_nn
        self.update_up_nn = update_up_nn
        self.update_boundaries_nn = update_boundaries_nn
        self.combine_nn = combine_nn
        self.initial_eps = eps
        if train_eps:
            self.eps1 = torch.nn.Parameter(torch.Tensor([eps]))
            self.eps2 = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer('eps1', torch.Tensor([eps]))
            self.register_buffer('eps2', torch.Tensor([eps]))
        self.reset_parameters()

    def reset_parameters(self):
        reset(self.msg_up_nn)
        reset(self.msg_boundaries_nn)
        reset(self.update_up_nn)
        reset(self.update_boundaries_nn)
        reset(self.combine_nn)
        self.eps1.data.fill_(self.initial_eps)
        self.eps2.data.fill_(self.initial_eps)

    # ---- hooks (generic path, and what the reference's tests call) -----------------------------
    def message_up(self, up_x_j: Tensor, up_attr: Tensor) -> Tensor:
        return self.msg_up_nn((up_x_j, up_attr))

    def message_boundary(self, boundary_x_j: Tensor) -> Tensor:
        return self.msg_boundaries_nn(boundary_x_j)

    # ---- fused forms -----------------------------------------------------------------------------
    def _up_kind(self) -> str:
        if isinstance(self.msg_up_nn, FirstOf):
            return 'first'
        if _is_cat_linear_relu(self.msg_up_nn):
            return 'cat_linear_relu'
        return 'custom'

    def gemm_specs(self, cochain: CochainMessagePassingParams) -> List[ops.Gemm]:
        """The dense products the fused coboundary message needs: Y1 = X_d W[:, :F]^T + b and
        Y2 = X_{d+1} W[:, F:]^T ([] when the message network is another form or there is no upper
        adjacency).  SparseCINConv groups the specs of all dimensions into one MFMA launch."""
        x, up_attr = cochain.x, cochain.kwargs.get('up_attr')
        if (cochain.up_index is None or up_attr is None or self._up_kind() != 'cat_linear_relu'
                or (self.aggr_up or 'add') != 'add'):
            return []
        lin = self.msg_up_nn[1]
        F = x.size(1)
        attr_src, _ = _attr_operand(up_attr)
        if lin.in_features != F + attr_src.size(1) or max(F, attr_src.size(1)) > ops.GEMM_MAX_K:
            return []
        # (column ranges of the one weight: autograd sees the Parameter itself, not two slices)
        return [ops.Gemm(X=x, W=lin.weight, w_col0=0, bias=lin.bias),
                ops.Gemm(X=attr_src, W=lin.weight, w_col0=F)]

    def _up_stream(self, adj: Adjacency, x: Tensor, up_attr, self_x=None, eps=None,
                   ys: Optional[List[Tensor]] = None) -> Optional[ops.Stream]:
        """The upper-adjacency aggregation as one fused stream, or None when the message network
        is not a recognised form.  `ys` = precomputed [Y1, Y2] of gemm_specs()."""
        kind = self._up_kind()
        if kind == 'first':
            return ops.Stream(adj=adj, n_dst=adj.n_dst, width=int(x.size(1)), A=x, self_x=self_x,
                              eps=eps, reduce=self.aggr_up or 'add')
        if kind == 'cat_linear_relu' and up_attr is not None and (self.aggr_up or 'add') == 'add':
            if ys is None:
                specs = SparseCINCochainConv.gemm_specs(self, CochainMessagePassingParams(x, adj, up_attr=up_attr))
                if not specs:
                    return None
                ys = ops.gemm_many(specs)
            _, mode = _attr_operand(up_attr)
            return ops.Stream(adj=adj, n_dst=adj.n_dst, width=int(ys[0].size(1)), A=ys[0], B=ys[1],
                              msg_op=ops.MSG_RELU_A_PLUS_B, ib_mode=mode, self_x=self_x, eps=eps)
        return None

    def _boundary_fusable(self) -> bool:
        return isinstance(self.msg_boundaries_nn, Passthrough)

    def _boundary_eps(self):
        return self.eps2          # mp/layers.py:192 (CIN++ numbers its three eps differently, :252-254)

    def message_and_aggregate_up(self, up_adj_t: Adjacency, x, up_attr) -> Tensor:
        st = self._up_stream(up_adj_t, x, up_attr)
        if st is not None:
            return ops.aggregate_many([st])[0]
        # unrecognised message network: gather -> network -> segmented reduce
        x_j = ops.gather_rows(x, up_adj_t.val, lambda a=up_adj_t: a.t_src)
        msg = self.message_up(x_j, dense(up_attr))
        return ops.aggregate(up_adj_t, up_adj_t.n_dst, msg, ia_mode='perm', reduce=self.aggr_up or 'add')

    def message_and_aggregate_boundary(self, boundary_adj_t: Adjacency, boundary_attr) -> Tensor:
        if self._boundary_fusable():
            return ops.aggregate(boundary_adj_t, boundary_adj_t.n_dst, boundary_attr,
                                 reduce=self.aggr_boundary or 'add')
        x_j = ops.gather_rows(boundary_attr, boundary_adj_t.val, lambda a=boundary_adj_t: a.t_src)
        return ops.aggregate(boundary_adj_t, boundary_adj_t.n_dst, self.message_boundary(x_j),
                             ia_mode='perm', reduce=self.aggr_boundary or 'add')

    # ---- forward, split so that SparseCINConv can batch all dimensions into one launch ------------
    def streams(self, cochain: CochainMessagePassingParams,
                ys: Optional[List[Tensor]] = None) -> Optional[List[ops.Stream]]:
        """[upper stream, boundary stream] with the self terms folded in, or None when a message
        network is not fusable (the caller then uses `forward_unfused`)."""
        x = cochain.x
        n, dev = x.size(0), x.device
        up_attr, b_attr = cochain.kwargs.get('up_attr'), cochain.kwargs.get('boundary_attr')
        kw = dict(x=x, up_attr=up_attr, boundary_attr=b_attr)
        if cochain.up_index is not None:
            size = self.__check_input_separately__(cochain.up_index, None)
            up = self._up_stream(self._adjacency(cochain.up_index, 'up', size, kw), x, up_attr,
                                 self_x=x, eps=self.eps1, ys=ys)
            if up is None:
                return None
        else:
            up = ops.Stream(adj=None, n_dst=n, width=int(x.size(1)), self_x=x, eps=self.eps1)
        if self.use_boundary_msg and b_attr is not None:
            if not self._boundary_fusable() or cochain.boundary_index is None:
                return None
            size = self.__check_input_separately__(cochain.boundary_index, None)
            adj = self._adjacency(cochain.boundary_index, 'boundary', size, kw)
            bnd = ops.Stream(adj=adj, n_dst=n, width=int(b_attr.size(1)), A=b_attr, self_x=x,
                             eps=self._boundary_eps(), reduce=self.aggr_boundary or 'add')
        else:
            bnd = ops.Stream(adj=None, n_dst=n, width=int(x.size(1)), self_x=x, eps=self._boundary_eps())
        if up.width != x.size(1) or bnd.width != x.size(1):
            return None   # self term needs message width == feature width
        return [up, bnd]

    def finish(self, out_up: Tensor, out_boundaries: Tensor) -> Tensor:
        """mp/layers.py:193-199."""
        out_up = self.update_up_nn(out_up)
        out_boundaries = self.update_boundaries_nn(out_boundaries)
        return self.combine_nn(torch.cat([out_up, out_boundaries], dim=-1))

    def forward_unfused(self, cochain: CochainMessagePassingParams) -> Tensor:
        """The reference's own sequence (mp/layers.py:184-199) through propagate()."""
        out_up, _, out_boundaries = self.propagate(cochain.up_index, cochain.down_index,
                                                   cochain.boundary_index, x=cochain.x,
                                                   up_attr=cochain.kwargs['up_attr'],
                                                   boundary_attr=cochain.kwargs['boundary_attr'])
        out_up = out_up + (1 + self.eps1) * cochain.x
        out_boundaries = out_boundaries + (1 + self.eps2) * cochain.x
        return self.finish(out_up, out_boundaries)

    def forward(self, cochain: CochainMessagePassingParams):
        sts = self.streams(cochain)
        if sts is None:
            return self.forward_unfused(cochain)
        out_up, out_boundaries = ops.aggregate_many(sts)
        return self.finish(out_up, out_boundaries)


def _update_mlp(layer_dim, hidden, graph_norm, act_module):
    return Sequential(Linear(layer_dim, hidden), graph_norm(hidden), act_module(),
                      Linear(hidden, hidden), graph_norm(hidden), act_module())


class SparseCINConv(torch.nn.Module):
    """mp/layers.py:271-342.  Cellular GIN over upper neighbours and boundaries."""

    def __init__(self, up_msg_size: int, down_msg_size: int, boundary_msg_size: Optional[int],
                 passed_msg_up_nn: Optional[Callable], passed_msg_boundaries_nn: Optional[Callable],
                 passed_update_up_nn: Optional[Callable],
                 passed_update_boundaries_nn: Optional[Callable], eps: float = 0.,
                 train_eps: bool = False, max_dim: int = 2, graph_norm=BN, use_coboundaries=False,
                 **kwargs):
        super().__init__()
        self.max_dim = max_dim
        self.mp_levels = torch.nn.ModuleList()
        for dim in range(max_dim + 1):
            msg_up_nn = passed_msg_up_nn
            if msg_up_nn is None:
                if use_coboundaries:
                    msg_up_nn = Sequential(Catter(),
                                           Linear(kwargs['layer_dim'] * 2, kwargs['layer_dim']),
                                           kwargs['act_module']())
                else:
                    msg_up_nn = FirstOf()
            msg_boundaries_nn = passed_msg_boundaries_nn
            if msg_boundaries_nn is None:
                msg_boundaries_nn = Passthrough()
            update_up_nn = passed_update_up_nn
            if update_up_nn is None:
                update_up_nn = _update_mlp(kwargs['layer_dim'], kwargs['hidden'], graph_norm,
                                           kwargs['act_module'])
            update_boundaries_nn = passed_update_boundaries_nn
            if update_boundaries_nn is None:
                update_boundaries_nn = _update_mlp(kwargs['layer_dim'], kwargs['hidden'], graph_norm,
                                                   kwargs['act_module'])
            combine_nn = Sequential(Linear(kwargs['hidden'] * 2, kwargs['hidden']),
                                    graph_norm(kwargs['hidden']), kwargs['act_module']())
            self.mp_levels.append(SparseCINCochainConv(
                dim, up_msg_size, down_msg_size, boundary_msg_size=boundary_msg_size,
                msg_up_nn=msg_up_nn, msg_boundaries_nn=msg_boundaries_nn, update_up_nn=update_up_nn,
                update_boundaries_nn=update_boundaries_nn, combine_nn=combine_nn, eps=eps,
                train_eps=train_eps))

    def propagate_all(self, *cochain_params: CochainMessagePassingParams, start_to_process=0):
        """Everything `propagate` does for all dimensions of this layer (the 3 propagate calls of
        mp/layers.py:337-341, plus the self terms) in TWO launches: one grouped MFMA GEMM for the
        coboundary-message products, one fused aggregation.  Returns (plans, outs): plans[dim] is
        None for dimensions that are not fusable / not processed, outs holds (up, boundary) pairs
        of the fusable ones in order."""
        n = len(cochain_params)
        fused = self._propagate_blocked(cochain_params, start_to_process)
        if fused is not None:
            return ['blocked'] * n, fused
        specs, owner = [], []
        for dim in range(start_to_process, n):
            sp = self.mp_levels[dim].gemm_specs(cochain_params[dim])
            specs += sp
            owner += [dim] * len(sp)
        plans = [None] * n
        pre = self._propagate_blocked_train(cochain_params, start_to_process, specs, owner) if specs else None

        def make_streams(ys):
            # the forward has run as the blocked launch: the streams only DESCRIBE the step for the autograd node, and its
            # backward is the owner-form launch over its own item table -- the CSR plans of the adjacencies are built when
            # somebody reads them (csr.deferred_builds; ops.gemm_aggregate builds them at once when the backward will)
            from .csr import deferred_builds
            import contextlib
            with (deferred_builds() if pre is not None else contextlib.nullcontext()):
                for dim in range(start_to_process, n):
                    mine = [y for y, o in zip(ys, owner) if o == dim]
                    plans[dim] = self.mp_levels[dim].streams(cochain_params[dim], mine or None)
            return [st for p in plans if p is not None for st in p]

        if specs:       # (training: ONE autograd node around the products and the aggregation, ops._GemmAggregate)
            _, outs = ops.gemm_aggregate(specs, make_streams, precomputed=pre)
        else:
            fused = make_streams([])
            outs = ops.aggregate_many(fused) if fused else []
        return plans, outs

    def _propagate_blocked(self, cochain_params, start_to_process) -> Optional[List[Tensor]]:
        """The whole propagate scope of this layer in ONE launch (csrc/cwn_layer.hip): Y1 / Y2 on
        the matrix cores into LDS, the batch's COO entries sorted per complex in LDS, both
        aggregations and the self terms out of LDS; no CSR plan, nothing but the two output
        streams written.  Applies to the form the reference's molecular models build (coboundary
        message ReLU(Linear(cat(x_j, up_attr))), identity boundary message, 'add' everywhere,
        mp/layers.py:286-299) on a batch that carries its per-complex tables, without autograd;
        None otherwise (the caller then runs the grouped GEMM + CSR aggregation;
        `self.blocked_reason` says why)."""
        from . import _ffi
        n = len(cochain_params)
        plan = getattr(cochain_params[0], 'block_plan', None)
        ent = None
        if plan is not None and BLOCKED_LAYER and not ops.GEMM_EXACT and start_to_process == 0:
            # fast path: everything about (this layer, this batch) that does not change between calls was
            # checked and laid out once (`_blocked_args`); per call only identities and the features are looked at
            ckey = id(plan)
            ent = _BLOCKED_CACHE.get(self, {}).get(ckey)
            if ent is not None and not self._blocked_still_valid(ent, cochain_params):
                ent = None
        if ent is None:
            args = self._blocked_args(cochain_params, start_to_process)
            if isinstance(args, str):
                self.blocked_reason = args
                return None
            dims, plan, table, key = args
            ckey = id(plan)
            lins = [self.mp_levels[d].msg_up_nn[1] for d, D in enumerate(dims) if D.msg_w_packed is not None]
            ent = dict(plan=plan, table=table, key=key, F=int(dims[0].x.size(1)), epochs=(ops.STATE_EPOCH, ops.STRUCT_EPOCH),
                       idx=[(c.up_index, c.boundary_index, getattr(c.kwargs.get('up_attr'), 'index', None))
                            for c in cochain_params],
                       # the packed weights follow the message weights' versions; bias / eps are read in place (their addresses)
                       marks=ops._marks([t for lin in lins for t in (lin.weight, lin.bias)]
                                        + self._blocked_eps(n)),   # (no ModuleList slice: a new container moves STRUCT_EPOCH)
                       launch=ops.LayerLaunch(dims, table))
            cache = _BLOCKED_CACHE.setdefault(self, {})
            if len(cache) > 64:
                cache.clear()
            cache[ckey] = ent
        self.__dict__['blocked_reason'] = None        # (plain attributes: Module.__setattr__ costs ~1 us each, per layer per call)
        table, key = ent['table'], ent['key']
        # the layers of one forward share their index tensors (mp/molec_models.py:110-116): the first
        # launch on them stores every item's sorted adjacency, the following ones load it back
        mode = _ffi.LAYER_CSR_LOAD if (CSR_REUSE and table.csr_key == key) else (_ffi.LAYER_CSR_STORE if CSR_REUSE else 0)
        try:
            outs = ent['launch'].run([c.x for c in cochain_params], mode)
        except (ValueError, TypeError):
            # features of another shape / type than the prepared launch takes (its own check, in C++ when the compiled
            # binding runs it): the long way decides what serves them
            _BLOCKED_CACHE.get(self, {}).pop(id(ent['plan']), None)
            args = self._blocked_args(cochain_params, start_to_process)
            if isinstance(args, str):
                self.blocked_reason = args
                return None
            raise
        if mode == _ffi.LAYER_CSR_STORE:
            table.csr_key = key
        plan = ent['plan']
        if not plan.validated and not torch.cuda.is_current_stream_capturing():
            from . import csr
            csr.check_errors(cochain_params[0].x.device)     # once per batch: the table belongs to these index tensors
            plan.validated = True
        return outs

    def _propagate_blocked_train(self, cochain_params, start_to_process, specs, owner):
        """The TRAINING forward of the propagate step through the blocked layer kernel (CWN_LAYER_STORE_Y): one launch
        (+ one small launch per message weight to pack it) instead of the grouped GEMM + the aggregation, with Y1 / Y2
        written out for the backward pass, which stays on the CSR path (ops._GemmAggregate).  Returns (ys, outs) in the
        order of `specs` / of the streams, or None when the blocked form does not apply to (this layer, this batch)."""
        if not (BLOCKED_TRAIN_FORWARD and torch.is_grad_enabled() and ops.FUSED_PROPAGATE_NODE):
            return None
        n = len(cochain_params)
        if len(specs) != 2 * sum(1 for d in range(n) if owner.count(d) == 2) or any(owner.count(d) not in (0, 2) for d in range(n)):
            return None
        args = self._blocked_args(cochain_params, start_to_process, training=True)
        if isinstance(args, str):
            self.blocked_reason = args
            return None
        dims, plan, table, key = args
        if getattr(table, 'variant', 0) == 'mixed' or getattr(table, 'n_big', 0):
            return None                  # (two launches into the same outputs / streamed complexes: inference only)
        dev, F = dims[0].x.device, int(dims[0].x.size(1))
        ys_of = [[None, None] for _ in range(n)]
        ys, ydims = [], []
        for d in range(n):
            if owner.count(d) == 2:
                if d + 1 >= n:
                    return None
                y1 = torch.empty(dims[d].x.size(0), F, dtype=torch.float32, device=dev)
                y2 = torch.empty(dims[d + 1].x.size(0), F, dtype=torch.float32, device=dev)
                ys_of[d][0], ys_of[d + 1][1] = y1, y2
                ys += [y1, y2]
                ydims += [(d, 'y1'), (d + 1, 'y2')]
        from . import _ffi
        launch = ops.LayerLaunch(dims, table)
        # (the layers of one forward share their index tensors: the first launch stores every item's sorted adjacency,
        # the following ones load it back -- as in the inference path)
        mode = _ffi.LAYER_CSR_LOAD if (CSR_REUSE and table.csr_key == key) else (_ffi.LAYER_CSR_STORE if CSR_REUSE else 0)
        outs = launch.run([c.x.detach() for c in cochain_params], mode, ys=[tuple(p) for p in ys_of])
        if mode == _ffi.LAYER_CSR_STORE:
            table.csr_key = key
        if not plan.validated and not torch.cuda.is_current_stream_capturing():
            from . import csr
            csr.check_errors(dev)
            plan.validated = True
        self.blocked_reason = None
        bwd_table = None
        if ops.BLOCKED_BACKWARD == 2:               # the owner form of the backward launch cuts a table of its own
            bwd_table = plan.bwd_items(F, [D.up_index is not None and D.up_index.size(1) > 0 for D in dims],
                                       [D.b_index is not None for D in dims])
        return ys, outs, (dims, table, ydims, bwd_table)    # (the last: what the blocked BACKWARD launch needs, ops._GemmAggregate)

    def _blocked_still_valid(self, ent, cochain_params) -> bool:
        """The per-call part of `_blocked_args`: autograd state, feature tensors, lazy attributes, and the
        packed weights (re-packed when an optimizer step has bumped the weight's version)."""
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                        or any(c.x.requires_grad for c in cochain_params)):
            return False
        n = len(cochain_params)
        if n != len(ent['idx']) or cochain_params[0].block_plan is not ent['plan']:
            return False
        if ent['epochs'] != (ops.STATE_EPOCH, ops.STRUCT_EPOCH):
            return False              # a training kernel wrote the parameters (ops.state_changed) / a module tree changed
        for d, c in enumerate(cochain_params):
            up, bi, sh = ent['idx'][d]
            kw = c.kwargs
            attr = kw.get('up_attr')
            if c.up_index is not up or c.boundary_index is not bi or getattr(attr, 'index', None) is not sh:
                return False          # other index tensors than the prepared launch points at
            if up is not None and (not isinstance(attr, IndexedRows) or d + 1 >= n or attr.src is not cochain_params[d + 1].x):
                return False
            b_attr = kw.get('boundary_attr')
            if b_attr is not None and (d == 0 or b_attr is not cochain_params[d - 1].x):
                return False
        # (the features' shape / type / device: the launch's own check -- LayerLaunch.run raises, _propagate_blocked falls back)
        return ops._marks_current(ent['marks'])   # weights changed: rebuild (re-pack) through _blocked_args

    def _blocked_args(self, cochain_params, start_to_process, training=False):
        if not BLOCKED_LAYER:
            return 'layers.BLOCKED_LAYER is off'
        if ops.GEMM_EXACT:
            return 'exact fp32 GEMMs requested (the blocked kernel has only the split form)'
        if start_to_process != 0:
            return 'start_to_process != 0'
        n = len(cochain_params)
        plan = getattr(cochain_params[0], 'block_plan', None)
        if plan is None:
            return 'the batch carries no per-complex tables (ptr / __slices__)'
        if n > 3 or n != plan.n_dims:
            return 'dimension count'
        if not training and torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                                         or any(c.x.requires_grad for c in cochain_params)):
            return 'autograd is recording (inference path only)'
        F = int(cochain_params[0].x.size(1))
        if F not in (64, 128):
            return f'feature width {F} (64 or 128)'
        dims, has_up = [], []
        for d, c in enumerate(cochain_params):
            lvl = self.mp_levels[d]
            x = c.x
            if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.size(1) != F:
                return f'dim {d}: features must be fp32 [n, {F}] on the GPU'
            if (lvl.aggr_up or 'add') != 'add' or (lvl.aggr_boundary or 'add') != 'add':
                return f'dim {d}: reduce is not add'
            if not lvl._boundary_fusable() or lvl.up_msg_size != F:
                return f'dim {d}: boundary message network / message width'
            D = ops.LayerDim(x=x, eps1=lvl.eps1, eps2=lvl._boundary_eps())
            why = self._blocked_dim_extra(D, lvl, c, d)       # (CIN++: the third output; a stream the launch does not have)
            if why is not None:
                return why
            up = c.up_index is not None and c.up_index.size(1) > 0
            if c.up_index is not None:
                attr = c.kwargs.get('up_attr')
                lin = lvl.msg_up_nn[1] if lvl._up_kind() == 'cat_linear_relu' else None
                if lin is None or lin.in_features != 2 * F or lin.out_features != F:
                    return f'dim {d}: message network is not ReLU(Linear(cat))'
                if not isinstance(attr, IndexedRows) or d + 1 >= n or attr.src is not cochain_params[d + 1].x:
                    return f'dim {d}: up_attr is not the lazy gather of the next dimension\'s features'
                D.up_index, D.up_shared = c.up_index, attr.index
                D.msg_w_packed, D.msg_bias = ops.pack_layer_weight(lin.weight, fresh=training), lin.bias
            b_index, b_attr = c.boundary_index, c.kwargs.get('boundary_attr')
            if lvl.use_boundary_msg and b_attr is not None:
                if b_index is None or d == 0 or b_attr is not cochain_params[d - 1].x:
                    return f'dim {d}: boundary_attr is not the previous dimension\'s features'
                D.b_index = b_index
            dims.append(D)
            has_up.append(bool(up))
        has_b = [D.b_index is not None for D in dims]
        # Tables are keyed on the streams THIS layer runs (a layer without the boundary stream must not get
        # records that carry boundary entries) and on the form of the kernel.  Beyond one item per CU the two-per-CU
        # form is taken when every complex fits its smaller caps: it beats the 16-wave form AND the streaming CSR path
        # at every size measured (tools/ab_variant.sh, M cells/s, two-per-CU / 16-wave / CSR: 512 complexes 1090 / 874 /
        # 582, 2048: 1273 / 994 / 837, 8192: 1207 / 960 / 905; at 128 complexes = 256 items 633 / 711 / 322).
        lower = plan.at_least(F, has_up, has_b)
        table = None
        if LAYER_VARIANT == '1':
            table = plan.items(F, has_up, has_b, variant=1)
        elif LAYER_VARIANT == 'auto' and lower > 4 * TWO_PER_CU_MIN_ITEMS:
            table = plan.items(F, has_up, has_b, variant=1)          # many rounds either way: two per CU (if every complex fits)
        if table is None and LAYER_VARIANT != '1' and lower > BLOCKED_MAX_ITEMS:
            # too many items for the 16-wave form alone: the two-per-CU form for whatever fits it + the rest (mixed), or
            # the streaming path
            tm = plan.items_mixed(F, has_up, has_b) if LAYER_VARIANT in ('auto', 'mixed') else None
            if tm is None or tm.parts[1].n_items > BLOCKED_MAX_ITEMS or (tm.parts[1].n_big and BIG_ITEMS != 'always' and not (
                    BIG_ITEMS and _streaming_pays(tm.parts[1], F))):
                return f'more than {BLOCKED_MAX_ITEMS} items: beyond the range where one workgroup per item beats the streaming CSR path'
            table = tm
        if table is None and LAYER_VARIANT != '1':
            table = plan.items(F, has_up, has_b)
            if table is not None and LAYER_VARIANT == 'auto' and table.n_items > TWO_PER_CU_MIN_ITEMS:
                t1 = plan.items(F, has_up, has_b, variant=1)
                if t1 is not None and _two_per_cu_wins(table.n_items, t1.n_items):
                    table = t1
        if LAYER_VARIANT in ('auto', 'mixed') and lower > TWO_PER_CU_MIN_ITEMS and (table is None or table.variant == 0) \
                and lower <= BLOCKED_MAX_ITEMS:
            # more items than CUs, but some complex is too large for the two-per-CU form (its 80 KiB hold ~30 atoms at
            # width 128): that form for the complexes that fit, the 16-wave form (BIG records where needed) for the
            # rest -- two launches into the same outputs (blockplan.MixedTable)
            tm = plan.items_mixed(F, has_up, has_b)
            if tm is not None and (LAYER_VARIANT == 'mixed' or _mixed_wins(tm, table, lower, F)):
                table = tm
        if table is None and LAYER_VARIANT != '1' and BIG_ITEMS and lower <= BLOCKED_MAX_ITEMS:
            # some complex does not fit a workgroup's LDS: ITS workgroup streams it (BIG records, include/cwn_hip.h), the
            # rest of the batch stays blocked -- unless most of the batch is like that (REDDIT-like hub complexes)
            t = plan.items(F, has_up, has_b, variant=0, allow_big=True)
            if t is not None and (BIG_ITEMS == 'always' or (t.n_big <= max(2, BIG_MAX_SHARE * t.n_items) and _streaming_pays(t, F))):
                table = t
        if table is None:
            return 'a complex does not fit one workgroup (row / entry caps)'
        if table.variant == 0 and table.n_items > BLOCKED_MAX_ITEMS:     # (a MixedTable's variant is 'mixed')
            return f'{table.n_items} items: beyond the range where one workgroup per item beats the streaming CSR path'
        key = tuple((id(t), ops._ffi.tver(t)) for D in dims for t in (D.up_index, D.up_shared, D.b_index) if t is not None)
        return dims, plan, table, key

    def _blocked_dim_extra(self, D, lvl, cochain, d: int) -> Optional[str]:
        """Hook of `_blocked_args`: what a subclass adds to the launch's descriptor of dimension d, or why it cannot."""
        return None

    def _blocked_eps(self, n: int) -> List[Tensor]:
        """The eps tensors a prepared launch reads in place (its marks follow their versions)."""
        return [t for d in range(n) for t in (self.mp_levels[d].eps1, self.mp_levels[d].eps2)]

    def _dense_eval(self, plans, outs, start: int = 0) -> Optional[List[Tensor]]:
        """The update / combine networks of ALL dimensions (mp/layers.py:193-199) as three grouped
        MFMA launches -- [Linear+norm+ReLU] x2 for both branches, then combine with its torch.cat
        folded into the K-concatenation -- with eval-mode BatchNorm folded into the epilogue.
        Inference only (no autograd, running statistics); returns None when it does not apply and
        the caller runs the torch modules instead."""
        if torch.is_grad_enabled():
            return None
        ent = _MLP_CACHE.get(self)
        if ent is not None:
            # the prepared launch of this layer's networks (ops.MlpLaunch): per call only the rows are looked at
            launch, shape, norms = ent
            if (shape == (start, len(plans), len(outs)) and FUSED_UPDATE_MLP and not ops.GEMM_EXACT
                    and not any(m.training for m in norms) and None not in plans[start:] and launch.current()):
                res = launch.run(outs[0::2], outs[1::2])
                if res is not None:
                    return res
            else:
                del _MLP_CACHE[self]
        active = list(range(start, len(plans)))
        if not active or any(plans[d] is None for d in active) or len(outs) != 2 * len(active):
            return None
        levels = [self.mp_levels[d] for d in active]
        chains = []
        for lvl in levels:
            up, bd, cb = (_mlp_stages(lvl.update_up_nn), _mlp_stages(lvl.update_boundaries_nn),
                          _mlp_stages(lvl.combine_nn))
            if up is None or bd is None or cb is None or len(up) != len(bd) or len(cb) != 1:
                return None
            if any(lin.in_features > ops.GEMM_MAX_K for lin, _ in up + bd + cb):
                return None
            folds = []
            for stages in (up, bd, cb):
                fs = [_fold_norm(norm, lin.out_features) for lin, norm in stages]
                if any(f is None for f in fs):
                    return None
                folds.append(fs)
            chains.append((up, bd, cb, folds))
        dev = outs[0].device
        hs_up = [outs[2 * i] for i in range(len(levels))]
        hs_bd = [outs[2 * i + 1] for i in range(len(levels))]
        depth = len(chains[0][0])
        if any(len(c[0]) != depth for c in chains):
            return None
        if FUSED_UPDATE_MLP and depth == 2 and not ops.GEMM_EXACT:
            # all five Linear layers of every dimension in ONE launch (csrc/cwn_mlp.hip): the activations
            # between them stay in LDS
            mdims = [ops.MlpDim(x_up=hs_up[i], x_b=hs_bd[i],
                                linears=[up[0][0], up[1][0], bd[0][0], bd[1][0], cb[0][0]],
                                folds=[folds[0][0], folds[0][1], folds[1][0], folds[1][1], folds[2][0]])
                     for i, (up, bd, cb, folds) in enumerate(chains)]
            if ops.update_mlp_applies(mdims):
                sources, norms = [], []
                for up, bd, cb, _ in chains:
                    for lin, norm in up + bd + cb:
                        sources += [lin.weight, lin.bias]
                        if isinstance(norm, BN):
                            norms.append(norm)
                            sources += [norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.num_batches_tracked]
                launch = ops.MlpLaunch(mdims, sources)
                _MLP_CACHE[self] = (launch, (start, len(plans), len(outs)), norms)
                res = launch.run(hs_up, hs_bd)
                if res is not None:
                    return res
                return ops.update_mlp(mdims)
        for st in range(depth):
            gemms = []
            for i, (up, bd, cb, folds) in enumerate(chains):
                for hs, stages, fs in ((hs_up, up, folds[0]), (hs_bd, bd, folds[1])):
                    lin = stages[st][0]
                    gemms.append(ops.Gemm(X=hs[i], W=lin.weight, bias=lin.bias, relu=True,
                                          out_scale=fs[st][0], out_shift=fs[st][1],
                                          w_packed=ops.pack_gemm_weight(lin.weight)))
            res = ops.run_gemm(gemms, dev)
            hs_up, hs_bd = res[0::2], res[1::2]
        gemms = []
        for i, (up, bd, cb, folds) in enumerate(chains):
            lin = cb[0][0]
            gemms.append(ops.Gemm(X=hs_up[i], X2=hs_bd[i], W=lin.weight, bias=lin.bias, relu=True,
                                  out_scale=folds[2][0][0], out_shift=folds[2][0][1]))
        return ops.run_gemm(gemms, dev)

    def _dense_train(self, plans, outs, start: int = 0) -> Optional[List[Tensor]]:
        """The same networks with autograd and training-mode BatchNorm (batch statistics, running
        statistics updated) as grouped launches forward and backward: cwn_amd/dense_train.py.
        Returns None when it does not apply (LayerNorm, custom networks, fewer than two cells in
        a dimension) and the caller runs the torch modules instead."""
        from . import dense_train as DT
        if not torch.is_grad_enabled() or not FUSED_DENSE_TRAINING:
            return None
        active = list(range(start, len(plans)))
        if not active or any(plans[d] is None for d in active) or len(outs) != 2 * len(active):
            return None
        if any(o.size(0) < 2 for o in outs):
            return None      # BatchNorm1d(train) needs more than one row; let torch raise as the reference does
        ups, bds, cbs = [], [], []
        for d in active:
            lvl = self.mp_levels[d]
            up, bd, cb = (_mlp_stages(lvl.update_up_nn), _mlp_stages(lvl.update_boundaries_nn),
                          _mlp_stages(lvl.combine_nn))
            if up is None or bd is None or cb is None or len(up) != len(bd) or len(cb) != 1:
                return None
            chains = [[DT.Stage(lin, norm) for lin, norm in st] for st in (up, bd, cb)]
            if not all(DT.supported(c) for c in chains):
                return None
            if any(isinstance(s.norm, BN) and not s.norm.training for c in chains for s in c):
                return None
            ups.append(chains[0])
            bds.append(chains[1])
            cbs.append(chains[2][0])
        if len({len(u) for u in ups}) != 1:
            return None
        p = self._out_drop
        self._out_dropped = p > 0.0              # (the combine stage's activation launch applies it: dense_train._Plan.out_drop)
        return DT.dense_train(DT._Plan(ups, bds, cbs, out_drop=p), outs)

    # the caller's dropout of this layer's OUTPUT (OGBEmbedSparseCIN: after every conv, mp/molec_models.py:298-300), handed in
    # so that the fused training path applies it inside its last launch: forward(..., out_dropout=p).  _dense_train sets
    # _out_dropped when it has; every other path gets ops.dropout (one launch per dimension, no mask tensor).
    _out_drop = 0.0
    _out_dropped = False

    def _finish_dropout(self, out: List[Tensor], start: int) -> List[Tensor]:
        p, done = self._out_drop, self._out_dropped
        self.__dict__['_out_drop'], self.__dict__['_out_dropped'] = 0.0, False
        if p <= 0.0 or done:
            return out
        return [x if dim < start or x is None else ops.dropout(x, p, True) for dim, x in enumerate(out)]

    def forward(self, *cochain_params: CochainMessagePassingParams, start_to_process=0, out_dropout: float = 0.0):
        assert len(cochain_params) <= self.max_dim + 1
        n = len(cochain_params)
        self.__dict__['_out_drop'], self.__dict__['_out_dropped'] = (float(out_dropout) if self.training else 0.0), False
        plans, outs = self.propagate_all(*cochain_params, start_to_process=start_to_process)
        dense = self._dense_eval(plans, outs, start_to_process)
        if dense is None:
            dense = self._dense_train(plans, outs, start_to_process)
        if dense is not None:
            it = iter(dense)
            return self._finish_dropout([cochain_params[dim].x if dim < start_to_process else next(it) for dim in range(n)],
                                        start_to_process)
        # update / combine networks per dimension
        out, k = [], 0
        for dim in range(n):
            if dim < start_to_process:
                out.append(cochain_params[dim].x)
            elif plans[dim] is None:
                out.append(self.mp_levels[dim].forward_unfused(cochain_params[dim]))
            else:
                out.append(self.mp_levels[dim].finish(outs[k], outs[k + 1]))
                k += 2
        return self._finish_dropout(out, start_to_process)


# ------------------------------------------------------------------------------------------------
# CIN++ (upper + lower + boundary)
# ------------------------------------------------------------------------------------------------
class CINppCochainConv(SparseCINCochainConv):
    """mp/layers.py:216-260.  Quirk kept by default: forward does not pass `down_attr` (:244-247), and
    the molecular CIN++ models ask for include_down_features=False, so the lower stream is zeros
    there.  Two explicit switches do it properly (no reference oracle
    for either: parity unpinned, property-tested):
      feed_down_attr=True       `down_attr` reaches message_down -- the lower-adjacency stream CIN++
                                describes, msg_down_nn((x_j, shared-boundary features));
      coboundary_stream=True    a FOURTH stream: the sum over the cofaces of a cell (the co-boundary
                                aggregation mp/cell_mp.py:44 leaves as a TODO), with its own eps and
                                update network, concatenated before combine_nn (which then takes
                                4 * hidden inputs).  Needs params from Complex.get_cochain_params
                                (they carry `coboundary_index` / `coboundary_attr`)."""

    def __init__(self, dim: int, up_msg_size: int, down_msg_size: int, boundary_msg_size: int,
                 msg_up_nn: Callable[..., Any], msg_boundaries_nn: Callable[..., Any],
                 msg_down_nn: Callable[..., Any], update_up_nn: Callable[..., Any],
                 update_boundaries_nn: Callable[..., Any], update_down_nn: Callable[..., Any],
                 combine_nn: Callable[..., Any], eps: float = 0, train_eps: bool = False,
                 feed_down_attr: bool = False, update_coboundaries_nn: Optional[Callable[..., Any]] = None):
        super().__init__(dim, up_msg_size, down_msg_size, boundary_msg_size, msg_up_nn,
                         msg_boundaries_nn, update_up_nn, update_boundaries_nn, combine_nn, eps,
                         train_eps)
        # the reference inherits use_down_msg=False from SparseCINCochainConv (mp/layers.py:167-168): the
        # lower stream is OFF unless the caller asks for the proper form
        self.use_down_msg = bool(feed_down_attr)
        self.msg_down_nn = msg_down_nn
        self.update_down_nn = update_down_nn
        self.feed_down_attr = feed_down_attr
        self.update_coboundaries_nn = update_coboundaries_nn       # not None = the fourth stream is on
        if train_eps:
            self.eps3 = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer('eps3', torch.Tensor([eps]))
        reset(self.msg_down_nn)
        reset(self.update_down_nn)
        self.eps3.data.fill_(self.initial_eps)
        if update_coboundaries_nn is not None:
            if train_eps:
                self.eps4 = torch.nn.Parameter(torch.Tensor([eps]))
            else:
                self.register_buffer('eps4', torch.Tensor([eps]))
            reset(self.update_coboundaries_nn)

    def message_down(self, down_x_j: Tensor, down_attr: Tensor) -> Tensor:
        return self.msg_down_nn((down_x_j, down_attr))

    # ---- fused forms (round 4): every stream of the level in the layer's ONE aggregation launch --
    def _boundary_eps(self):
        return self.eps3          # mp/layers.py:254

    def _down_kind(self) -> str:
        if isinstance(self.msg_down_nn, FirstOf):
            return 'first'
        return 'cat_linear_relu' if _is_cat_linear_relu(self.msg_down_nn) else 'custom'

    def _down_active(self, cochain) -> bool:
        return bool(self.use_down_msg) and cochain.down_index is not None

    def _down_specs(self, cochain) -> List[ops.Gemm]:
        """The lower-adjacency message ReLU(Linear(cat(x_j, down_attr))) as Y1[j] + Y2[shared boundary] (the form of the
        coboundary message, SparseCINCochainConv.gemm_specs); [] when the lower stream is off (the reference's quirk, the
        default) or its network is another form."""
        x, attr = cochain.x, cochain.kwargs.get('down_attr')
        if (not self._down_active(cochain) or attr is None or self._down_kind() != 'cat_linear_relu'
                or (self.aggr_down or 'add') != 'add'):
            return []
        lin, F = self.msg_down_nn[1], x.size(1)
        src, _ = _attr_operand(attr)
        if lin.in_features != F + src.size(1) or max(F, src.size(1)) > ops.GEMM_MAX_K:
            return []
        return [ops.Gemm(X=x, W=lin.weight, w_col0=0, bias=lin.bias), ops.Gemm(X=src, W=lin.weight, w_col0=F)]

    def gemm_specs(self, cochain):
        return SparseCINCochainConv.gemm_specs(self, cochain) + self._down_specs(cochain)

    def streams(self, cochain, ys=None):
        """[upper, lower, boundary (, co-boundary)] with the self terms folded in -- (1 + eps1 / eps2 / eps3 / eps4) x,
        mp/layers.py:252-254 -- or None when a message network is not a recognised form (the caller then runs
        `forward_unfused`).  The lower stream of the default layer is its self term alone (the quirk: zeros + (1+eps2) x)."""
        n_up = len(SparseCINCochainConv.gemm_specs(self, cochain)) if ys else 0
        base = SparseCINCochainConv.streams(self, cochain, (list(ys[:n_up]) or None) if ys else None)
        if base is None:
            return None
        up, bnd = base
        x = cochain.x
        n, F = x.size(0), int(x.size(1))
        if self._down_active(cochain):
            attr = cochain.kwargs.get('down_attr') if self.feed_down_attr else None
            size = self.__check_input_separately__(cochain.down_index, None)
            adj = self._adjacency(cochain.down_index, 'down', size, dict(x=x, down_attr=attr))
            kind = self._down_kind()
            if kind == 'first':
                down = ops.Stream(adj=adj, n_dst=adj.n_dst, width=F, A=x, self_x=x, eps=self.eps2,
                                  reduce=self.aggr_down or 'add')
            elif kind == 'cat_linear_relu' and attr is not None and (self.aggr_down or 'add') == 'add':
                yd = list(ys[n_up:]) if ys else []
                if not yd:
                    specs = self._down_specs(cochain)
                    if not specs:
                        return None
                    yd = ops.gemm_many(specs)
                _, mode = _attr_operand(attr)
                down = ops.Stream(adj=adj, n_dst=adj.n_dst, width=int(yd[0].size(1)), A=yd[0], B=yd[1],
                                  msg_op=ops.MSG_RELU_A_PLUS_B, ib_mode=mode, self_x=x, eps=self.eps2)
            else:
                return None
        else:
            down = ops.Stream(adj=None, n_dst=n, width=F, self_x=x, eps=self.eps2)
        out = [up, down, bnd]
        if self.update_coboundaries_nn is not None:
            cob_index, cob_attr = getattr(cochain, 'coboundary_index', None), getattr(cochain, 'coboundary_attr', None)
            if cob_index is not None and cob_attr is not None:
                if cob_attr.dim() != 2 or cob_index.dim() != 2 or cob_index.size(0) != 2:
                    return None          # (propagate_coboundary raises the errors)
                t = cached_adjacency(cob_index, int(cob_attr.size(0)), int(n)).t_src     # keyed on this dimension's cell
                out.append(ops.Stream(adj=t, n_dst=n, width=int(cob_attr.size(1)), A=cob_attr, self_x=x, eps=self.eps4))
            else:
                out.append(ops.Stream(adj=None, n_dst=n, width=F, self_x=x, eps=self.eps4))
        if any(st.width != F for st in out):
            return None
        return out

    def finish(self, out_up: Tensor, out_down: Tensor, out_boundaries: Tensor, out_cob: Optional[Tensor] = None) -> Tensor:
        """mp/layers.py:255-260 (+ the fourth stream)."""
        parts = [self.update_up_nn(out_up), self.update_down_nn(out_down), self.update_boundaries_nn(out_boundaries)]
        if self.update_coboundaries_nn is not None:
            parts.append(self.update_coboundaries_nn(out_cob))
        return self.combine_nn(torch.cat(parts, dim=-1))

    def forward(self, cochain: CochainMessagePassingParams):
        sts = self.streams(cochain)
        if sts is None:
            return self.forward_unfused(cochain)
        return self.finish(*ops.aggregate_many(sts))

    def forward_unfused(self, cochain: CochainMessagePassingParams):
        """The reference's own sequence (mp/layers.py:243-260) through propagate() and the hooks."""
        kw = dict(x=cochain.x, up_attr=cochain.kwargs['up_attr'], boundary_attr=cochain.kwargs['boundary_attr'])
        down_index = cochain.down_index
        if self.feed_down_attr:
            kw['down_attr'] = cochain.kwargs.get('down_attr')
        out_up, out_down, out_boundaries = self.propagate(cochain.up_index, down_index, cochain.boundary_index, **kw)
        out_up = out_up + (1 + self.eps1) * cochain.x
        out_down = out_down + (1 + self.eps2) * cochain.x
        out_boundaries = out_boundaries + (1 + self.eps3) * cochain.x
        parts = [self.update_up_nn(out_up), self.update_down_nn(out_down), self.update_boundaries_nn(out_boundaries)]
        if self.update_coboundaries_nn is not None:
            cob_index = getattr(cochain, 'coboundary_index', None)
            cob_attr = getattr(cochain, 'coboundary_attr', None)
            if cob_index is not None and cob_attr is not None:
                out_cob = self.propagate_coboundary(cob_index, cob_attr, cochain.x.size(0))
            else:
                out_cob = ops.zeros_rows(cochain.x.size(0), cochain.x.size(1), cochain.x.device)
            parts.append(self.update_coboundaries_nn(out_cob + (1 + self.eps4) * cochain.x))
        return self.combine_nn(torch.cat(parts, dim=-1))


class CINppConv(SparseCINConv):
    """mp/layers.py:344-427; `feed_down_attr` / `coboundary_stream`: see CINppCochainConv."""

    def __init__(self, up_msg_size: int, down_msg_size: int, boundary_msg_size: Optional[int],
                 passed_msg_up_nn: Optional[Callable], passed_msg_down_nn: Optional[Callable],
                 passed_msg_boundaries_nn: Optional[Callable],
                 passed_update_up_nn: Optional[Callable], passed_update_down_nn: Optional[Callable],
                 passed_update_boundaries_nn: Optional[Callable], eps: float = 0.,
                 train_eps: bool = False, max_dim: int = 2, graph_norm=BN, use_coboundaries=False,
                 feed_down_attr: bool = False, coboundary_stream: bool = False, **kwargs):
        super().__init__(up_msg_size, down_msg_size, boundary_msg_size, passed_msg_up_nn,
                         passed_msg_boundaries_nn, passed_update_up_nn, passed_update_boundaries_nn,
                         eps, train_eps, max_dim, graph_norm, use_coboundaries, **kwargs)
        self.mp_levels = torch.nn.ModuleList()
        ld, hid, act = kwargs['layer_dim'], kwargs['hidden'], kwargs['act_module']
        n_streams = 4 if coboundary_stream else 3
        for dim in range(max_dim + 1):
            def msg_net(passed):
                if passed is not None:
                    return passed
                return Sequential(Catter(), Linear(ld * 2, ld), act()) if use_coboundaries else FirstOf()
            self.mp_levels.append(CINppCochainConv(
                dim, up_msg_size, down_msg_size, boundary_msg_size=boundary_msg_size,
                msg_up_nn=msg_net(passed_msg_up_nn), msg_down_nn=msg_net(passed_msg_down_nn),
                msg_boundaries_nn=passed_msg_boundaries_nn or Passthrough(),
                update_up_nn=passed_update_up_nn or _update_mlp(ld, hid, graph_norm, act),
                update_down_nn=passed_update_down_nn or _update_mlp(ld, hid, graph_norm, act),
                update_boundaries_nn=passed_update_boundaries_nn or _update_mlp(ld, hid, graph_norm, act),
                combine_nn=Sequential(Linear(hid * n_streams, hid), graph_norm(hid), act()),
                eps=eps, train_eps=train_eps, feed_down_attr=feed_down_attr,
                update_coboundaries_nn=_update_mlp(ld, hid, graph_norm, act) if coboundary_stream else None))

    # Round 6: the layer as the reference's molecular models run it -- lower stream OFF (the quirk: include_down_features=False,
    # mp/molec_models.py:111 -> down_index None -> out_down = zeros + (1 + eps2) x, mp/layers.py:253) -- takes the blocked launch
    # of SparseCINConv: upper + boundary streams as there (boundary self term with eps3, :254), and the launch writes the
    # third output (1 + eps2) x from the registers that hold the row (cwn_layer_dim.out_down, ABI 22).  Outputs per dimension
    # in the order of torch.cat in :260: [out_up, out_down, out_b].  With `feed_down_attr` (a real lower adjacency) or the
    # co-boundary stream the layer keeps the grouped message GEMM + ONE aggregation launch for its three (four) streams.
    def _blocked_dim_extra(self, D, lvl, cochain, d: int) -> Optional[str]:
        if lvl._down_active(cochain):
            return f'dim {d}: CIN++ with a lower-adjacency stream (the blocked launch has upper + boundary + the self-only third)'
        if lvl.update_coboundaries_nn is not None:
            return f'dim {d}: CIN++ with the co-boundary stream (four streams)'
        D.eps3, D.want_down = lvl.eps2, True
        return None

    def _blocked_eps(self, n: int) -> List[Tensor]:
        return [t for d in range(n) for t in (self.mp_levels[d].eps1, self.mp_levels[d].eps2, self.mp_levels[d].eps3)]

    def _blocked_still_valid(self, ent, cochain_params) -> bool:
        if any(self.mp_levels[d]._down_active(c) for d, c in enumerate(cochain_params)):
            return False         # a lower adjacency has appeared: `_blocked_args` says why the launch does not serve it
        return super()._blocked_still_valid(ent, cochain_params)

    # (training: SparseCINConv._propagate_blocked_train serves this layer too -- the launch stores Y1 / Y2 and writes the third
    #  output; the blocked backward launch takes the gradients of out_up / out_b and ops._blocked_backward_impl adds
    #  (1 + eps2) g_down onto dx)

    @staticmethod
    def _n_streams(plan) -> int:
        return 3 if isinstance(plan, str) else len(plan)       # ('blocked': the launch's three outputs)

    def _update_chains(self, plans, outs, start: int):
        """(active dimensions, streams per dimension, [dim][stream] -> [(Linear, norm), ...]) or None."""
        active = list(range(start, len(plans)))
        if not active or any(plans[d] is None for d in active):
            return None
        nb = self._n_streams(plans[active[0]])
        if any(self._n_streams(plans[d]) != nb for d in active) or len(outs) != nb * len(active):
            return None
        chains = []
        for d in active:
            lvl = self.mp_levels[d]
            nets = [lvl.update_up_nn, lvl.update_down_nn, lvl.update_boundaries_nn]
            if nb == 4:
                nets.append(lvl.update_coboundaries_nn)
            sts = [_mlp_stages(net) for net in nets]
            if any(st is None for st in sts):
                return None
            chains.append(sts)
        if len({len(st) for sts in chains for st in sts}) != 1:
            return None
        return active, nb, chains

    def _dense_eval(self, plans, outs, start: int = 0) -> Optional[List[Tensor]]:
        """Inference: the update networks of every stream and dimension as grouped MFMA launches, one per stage, with
        eval-mode BatchNorm and the ReLU folded into the epilogue (SparseCINConv._dense_eval's form); torch.cat + combine_nn
        as torch modules."""
        from . import _ffi
        if torch.is_grad_enabled():
            return None
        got = self._update_chains(plans, outs, start)
        if got is None:
            return None
        active, nb, chains = got
        flat = [st for sts in chains for st in sts]                      # [dim][stream] flattened, as `outs`
        folds = [[_fold_norm(norm, lin.out_features) for lin, norm in st] for st in flat]
        if any(f is None for fs in folds for f in fs) or any(lin.in_features > ops.GEMM_MAX_K for st in flat for lin, _ in st):
            return None
        if nb == 3 and FUSED_UPDATE_MLP and not ops.GEMM_EXACT and all(len(st) == 2 for st in flat):
            # three update networks + the 3F-wide combine of every dimension in ONE launch (csrc/cwn_mlp3.hip, round 6): the
            # activations between the seven Linear layers stay in LDS, the combine is accumulated branch by branch
            mdims = []
            for k, d in enumerate(active):
                cb = _mlp_stages(self.mp_levels[d].combine_nn)
                cfold = None if cb is None or len(cb) != 1 else _fold_norm(cb[0][1], cb[0][0].out_features)
                if cfold is None:
                    mdims = None
                    break
                sts, fs = flat[nb * k: nb * (k + 1)], folds[nb * k: nb * (k + 1)]
                mdims.append(ops.Mlp3Dim(xs=list(outs[nb * k: nb * (k + 1)]),
                                         linears=[st[j][0] for st in sts for j in range(2)] + [cb[0][0]],
                                         folds=[f[j] for f in fs for j in range(2)] + [cfold]))
            if mdims and ops.update_mlp3_applies(mdims):
                return ops.update_mlp3(mdims)
        hs, dev = list(outs), outs[0].device
        for s in range(len(flat[0])):
            gemms = [ops.Gemm(X=h, W=st[s][0].weight, bias=st[s][0].bias, relu=True, out_scale=fs[s][0], out_shift=fs[s][1],
                              w_packed=ops.pack_gemm_weight(st[s][0].weight)) for h, st, fs in zip(hs, flat, folds)]
            hs = []
            for lo in range(0, len(gemms), _ffi.MAX_DESCS):
                hs += ops.run_gemm(gemms[lo: lo + _ffi.MAX_DESCS], dev)
        return [self.mp_levels[d].combine_nn(torch.cat(hs[nb * k: nb * (k + 1)], dim=-1)) for k, d in enumerate(active)]

    def _dense_train(self, plans, outs, start: int = 0) -> Optional[List[Tensor]]:
        """Training mode: the update networks of EVERY stream and dimension (three or four chains of Linear ->
        BatchNorm(train) -> ReLU stages per dimension) through dense_train's stage launches, forward and backward -- a plan
        without combine stages, at most _ffi.MAX_DESCS chains per autograd node -- then torch.cat + combine_nn as torch
        modules (mp/layers.py:255-260).  None when it does not apply (LayerNorm, custom networks, fewer than two cells)."""
        from . import _ffi, dense_train as DT
        if not torch.is_grad_enabled() or not FUSED_DENSE_TRAINING or any(o.size(0) < 2 for o in outs):
            return None
        got = self._update_chains(plans, outs, start)
        if got is None:
            return None
        active, nb, raw = got
        chains = []
        for sts in raw:
            cs = [[DT.Stage(lin, norm) for lin, norm in st] for st in sts]
            if not all(DT.supported(c) for c in cs) or any(isinstance(s.norm, BN) and not s.norm.training for c in cs for s in c):
                return None
            chains.append(cs)
        per = max(1, _ffi.MAX_DESCS // nb)                    # dimensions per autograd node
        # the combine stage in the same node (its third / fourth branch as extra K-blocks of cwn_dense_stage_ex_f32) ...
        cbs = []
        for d in active:
            cb = _mlp_stages(self.mp_levels[d].combine_nn)
            cbs.append(DT.Stage(*cb[0]) if cb is not None and len(cb) == 1 else None)
        full = (FUSED_CINPP_COMBINE and ops.STAGE_KERNEL and all(c is not None and DT.supported([c], max_k=4 * 128) and c.is_bn and c.norm.training
                                                                for c in cbs))
        if ops.STAGE_KERNEL:
            # This layer packs its own blocks on every training forward (one launch: update and combine networks), next to
            # whatever a model packed (fresh=False).  ALWAYS: a block is keyed on its weight's storage address, and a look-up
            # that trusted an earlier entry could be handed the block of a dead layer whose parameters lived at the same
            # address (seen in the tests: two layers built one after the other).
            ws = [st.lin.weight for cs in chains for c in cs for st in c if st.lin.weight.is_cuda]
            if full:
                ws += [c.lin.weight for c in cbs if c.lin.weight.is_cuda]
            if ws:
                ops.pack_stage_weights_many(ws, fresh=False)
        if full:
            res: List[Tensor] = []
            try:
                for lo in range(0, len(active), per):
                    grp = chains[lo: lo + per]
                    res += DT.dense_train(DT._Plan(None, None, cbs[lo: lo + per], chains=grp, out_drop=self._out_drop),
                                          outs[nb * lo: nb * (lo + len(grp))])
                self._out_dropped = self._out_drop > 0.0
                return res
            except DT.CombineNeedsStageKernel:
                if res:         # (a later group refused after an earlier one has run: running it again would count its batch twice)
                    raise
                # (raised before any launch of the first group: the whole layer takes the other form)
        # ... or on torch, behind the branches
        hs: List[Tensor] = []
        for lo in range(0, len(active), per):
            grp = chains[lo: lo + per]
            hs += DT.dense_train(DT._Plan(None, None, None, chains=grp), outs[nb * lo: nb * (lo + len(grp))])
        return [self.mp_levels[d].combine_nn(torch.cat(hs[nb * k: nb * (k + 1)], dim=-1)) for k, d in enumerate(active)]

    def forward(self, *cochain_params: CochainMessagePassingParams, start_to_process=0, out_dropout: float = 0.0):
        """mp/layers.py:418-427."""
        assert len(cochain_params) <= self.max_dim + 1
        self.__dict__['_out_drop'], self.__dict__['_out_dropped'] = (float(out_dropout) if self.training else 0.0), False
        plans, outs = self.propagate_all(*cochain_params, start_to_process=start_to_process)
        dense = self._dense_eval(plans, outs, start_to_process)
        if dense is None:
            dense = self._dense_train(plans, outs, start_to_process)
        if dense is not None:
            it = iter(dense)
            return self._finish_dropout([c.x if dim < start_to_process else next(it) for dim, c in enumerate(cochain_params)],
                                        start_to_process)
        out, k = [], 0
        for dim, cochain in enumerate(cochain_params):
            if dim < start_to_process:
                out.append(cochain.x)
            elif plans[dim] is None:
                out.append(self.mp_levels[dim].forward_unfused(cochain))
            else:
                m = self._n_streams(plans[dim])
                out.append(self.mp_levels[dim].finish(*outs[k:k + m]))
                k += m
        return self._finish_dropout(out, start_to_process)


# ------------------------------------------------------------------------------------------------
# OrientedConv, InitReduceConv, embedding front-ends
# ------------------------------------------------------------------------------------------------
class OrientedConv(CochainMessagePassing):
    """mp/layers.py:430-470: messages x_j * orientation (a +-1 scalar per adjacency entry).
    Fused as CWN_MSG_A_TIMES_B with a width-1 per-entry attribute."""

    def __init__(self, dim: int, up_msg_size: int, down_msg_size: int,
                 update_up_nn: Optional[Callable], update_down_nn: Optional[Callable],
                 update_nn: Optional[Callable], act_fn, orient=True):
        super().__init__(up_msg_size, down_msg_size, use_boundary_msg=False)
        self.dim = dim
        self.update_up_nn = update_up_nn
        self.update_down_nn = update_down_nn
        self.update_nn = update_nn
        self.act_fn = act_fn
        self.orient = orient

    def _stream(self, adj: Adjacency, x: Tensor, attr: Tensor, aggr) -> ops.Stream:
        if not self.orient:
            return ops.Stream(adj=adj, n_dst=adj.n_dst, width=int(x.size(1)), A=x, reduce=aggr or 'add')
        return ops.Stream(adj=adj, n_dst=adj.n_dst, width=int(x.size(1)), A=x, B=dense(attr).to(torch.float32),
                          msg_op=ops.MSG_A_TIMES_B, ib_mode='perm', reduce=aggr or 'add')

    def propagate_both(self, cochain) -> Optional[List[Tensor]]:
        """(out_up, out_down) of mp/layers.py:441-446 in ONE aggregation launch (one autograd node in training) instead of
        one per adjacency; None when a subclass has replaced the hooks (forward then runs propagate())."""
        x = cochain.x
        if (not isinstance(x, Tensor) or not x.is_cuda or type(self).message_up is not OrientedConv.message_up
                or type(self).message_down is not OrientedConv.message_down
                or type(self).message_and_aggregate_up is not OrientedConv.message_and_aggregate_up
                or type(self).message_and_aggregate_down is not OrientedConv.message_and_aggregate_down
                or self._overrides['aggregate_up'] or self._overrides['aggregate_down'] or self._overrides['update']):
            return None
        up_attr, down_attr = cochain.upper_orient.view(-1, 1), cochain.lower_orient.view(-1, 1)
        kw = dict(x=x, up_attr=up_attr, down_attr=down_attr)
        up_size = self.__check_input_separately__(cochain.upper_index, None)
        down_size = self.__check_input_separately__(cochain.lower_index, None)
        self.__check_input_together__(cochain.upper_index, cochain.lower_index, up_size, down_size)
        sts = [self._stream(self._adjacency(cochain.upper_index, 'up', up_size, kw), x, up_attr, self.aggr_up),
               self._stream(self._adjacency(cochain.lower_index, 'down', down_size, kw), x, down_attr, self.aggr_down)]
        return ops.aggregate_many(sts)

    def forward(self, cochain):
        assert len(cochain.upper_orient) == cochain.upper_index.size(1)
        assert len(cochain.lower_orient) == cochain.lower_index.size(1)
        both = self.propagate_both(cochain)
        if both is not None:
            out_up, out_down = both
        else:
            out_up, out_down, _ = self.propagate(
                cochain.upper_index, cochain.lower_index, None, x=cochain.x,
                up_attr=cochain.upper_orient.view(-1, 1), down_attr=cochain.lower_orient.view(-1, 1))
        out_up = self.update_up_nn(out_up)
        out_down = self.update_down_nn(out_down)
        x = self.update_nn(cochain.x)
        return self.act_fn(x + out_up + out_down)

    def reset_parameters(self):
        reset(self.update_up_nn)
        reset(self.update_down_nn)
        reset(self.update_nn)

    def message_up(self, up_x_j: Tensor, up_attr: Tensor) -> Tensor:
        return up_x_j * up_attr if self.orient else up_x_j

    def message_down(self, down_x_j: Tensor, down_attr: Tensor) -> Tensor:
        return down_x_j * down_attr if self.orient else down_x_j

    def _fused(self, adj: Adjacency, x, attr, aggr) -> Tensor:
        if not self.orient:
            return ops.aggregate(adj, adj.n_dst, x, reduce=aggr or 'add')
        attr = dense(attr).to(torch.float32)
        return ops.aggregate(adj, adj.n_dst, x, msg_op=ops.MSG_A_TIMES_B, B=attr, ib_mode='perm',
                             reduce=aggr or 'add')

    def message_and_aggregate_up(self, up_adj_t: Adjacency, x, up_attr) -> Tensor:
        return self._fused(up_adj_t, x, up_attr, self.aggr_up)

    def message_and_aggregate_down(self, down_adj_t: Adjacency, x, down_attr) -> Tensor:
        return self._fused(down_adj_t, x, down_attr, self.aggr_down)


class InitReduceConv(torch.nn.Module):
    """mp/layers.py:473-487: initial features of d-cells = reduce of their boundary cells'.

    The reference sizes the output by `boundary_index[1].max() + 1` (a device sync on the GPU);
    pass `num_cells` to avoid it -- EmbedVEWithReduce does."""

    def __init__(self, reduce='add'):
        super().__init__()
        self.reduce = reduce

    def forward(self, boundary_x, boundary_index, num_cells: Optional[int] = None):
        from .csr import cached_adjacency
        if num_cells is None:
            num_cells = int(boundary_index[1, :].max()) + 1
        red = 'add' if self.reduce in ('add', 'sum') else self.reduce
        adj = cached_adjacency(boundary_index, num_cells, boundary_x.size(0))
        if red == 'min':
            return -ops.aggregate(adj, num_cells, -boundary_x, reduce='max')
        return ops.aggregate(adj, num_cells, boundary_x, reduce=red)


def _embed(layer, idx: Tensor) -> Tensor:
    """A torch.nn.Embedding, or a module holding a list of them whose outputs are summed over the
    index columns (the OGB Atom/BondEncoder form), as one fused gather-sum (ops.embedding_sum);
    anything else is called as it is."""
    def plain(e):
        return (isinstance(e, torch.nn.Embedding) and e.padding_idx is None and e.max_norm is None
                and not e.sparse and not e.scale_grad_by_freq)
    if idx.is_cuda and idx.numel() > 0:
        if plain(layer) and idx.dim() == 1:
            return ops.embedding_sum([layer.weight], idx)
        tables = None
        for name in ('atom_embedding_list', 'bond_embedding_list'):
            if hasattr(layer, name):
                tables = list(getattr(layer, name))
        if tables is not None and idx.dim() == 2 and idx.size(1) == len(tables) and all(plain(e) for e in tables):
            return ops.embedding_sum([e.weight for e in tables], idx)
    return layer(idx)


class AbstractEmbedVEWithReduce(torch.nn.Module, ABC):
    """mp/layers.py:490-547."""

    def __init__(self, v_embed_layer: Callable, e_embed_layer: Optional[Callable],
                 init_reduce: InitReduceConv):
        super().__init__()
        self.v_embed_layer = v_embed_layer
        self.e_embed_layer = e_embed_layer
        self.init_reduce = init_reduce

    @abstractmethod
    def _prepare_v_inputs(self, v_params):
        pass

    @abstractmethod
    def _prepare_e_inputs(self, e_params):
        pass

    def _check_v_inputs(self, v_params):      # the shape asserts of _prepare_*_inputs without the dtype conversion
        assert v_params.x is not None and v_params.x.dim() == 2

    def _check_e_inputs(self, e_params):
        assert self.e_embed_layer is not None and e_params.x.dim() == 2

    def forward(self, *cochain_params: CochainMessagePassingParams):
        assert 1 <= len(cochain_params) <= 3
        v_params = cochain_params[0]
        e_params = cochain_params[1] if len(cochain_params) >= 2 else None
        c_params = cochain_params[2] if len(cochain_params) == 3 else None
        fused = self._forward_fused(v_params, e_params, c_params)
        if fused is not None:
            return fused
        vx = _embed(self.v_embed_layer, self._prepare_v_inputs(v_params))
        out = [vx]
        if e_params is None:
            assert c_params is None
            return out
        n_e = getattr(e_params, 'num_cells', None) or (e_params.x.size(0) if e_params.x is not None else None)
        reduced_ex = self.init_reduce(vx, e_params.boundary_index, n_e)
        ex = reduced_ex
        if e_params.x is not None:
            ex = _embed(self.e_embed_layer, self._prepare_e_inputs(e_params))
            assert ex.size(1) == vx.size(1)
        out.append(ex)
        if c_params is not None:
            n_c = getattr(c_params, 'num_cells', None) or (c_params.x.size(0) if c_params.x is not None else None)
            # halved as in the reference (:538-540)
            out.append(self.init_reduce(reduced_ex, c_params.boundary_index, n_c) / 2.)
        return out

    def _forward_fused(self, v_params, e_params, c_params) -> Optional[List[Tensor]]:
        """The whole front in ONE launch (ops.embed_front, csrc/cwn_ends.hip): both embeddings, the reduction of
        the vertex embeddings onto the edges and of that onto the rings, halved -- the 8 launches below it were 29 us
        of a 167 us forward at the ZINC batch of 128.  Plain embedding tables and 'sum' reduction; with autograd
        (training) the same launch behind ops._EmbedFrontTrain.  None otherwise (the caller runs the separate launches)."""
        if not ops.FUSED_ENDS or e_params is None or v_params.x is None or not v_params.x.is_cuda:
            return None
        infer = not torch.is_grad_enabled()
        bi1, bi2 = e_params.boundary_index, (c_params.boundary_index if c_params is not None else None)
        ent = _FRONT_CACHE.get(self, {}).get(id(bi1)) if infer else None
        if (ent is not None and ent[1] == (c_params is not None, self.init_reduce.reduce) and ent[0].current(bi1, bi2)
                and (ent[0].te is not None) == (e_params.x is not None)):
            xs = ent[0].run(v_params.x, e_params.x)
            if xs is not None:
                return xs if c_params is not None else xs[:2]
        if self.init_reduce.reduce not in ('add', 'sum'):
            return None
        vt = _embedding_tables(self.v_embed_layer)
        et = _embedding_tables(self.e_embed_layer) if e_params.x is not None else None
        if vt is None or (e_params.x is not None and et is None):
            return None
        train = torch.is_grad_enabled() and any(w.requires_grad for w in vt + (et or []))
        if (len(vt) > 1 or (et is not None and len(et) > 1)) and not FUSED_OGB_FRONT:
            # OGB-style encoders (a table per integer feature column: 9 + 3 at molhiv).  ONE launch is slower there (44 - 60
            # us at the molhiv batch of 512: every ring row walks twelve vertices of nine columns), so cwn_embed_front_f32
            # makes it two -- the embeddings of both cell types, then the reductions from the x0 rows (round 6: 15 us against
            # the 40 us of the seven separate launches; in training the same forward behind ops._EmbedFrontTrain, whose
            # backward is the launches the separate path's autograd ran).
            return None
        if train and not ops.FUSED_FRONT_TRAINING:
            return None
        H = int(vt[0].size(1))
        if H % 4 != 0 or any(w.size(1) != H or not w.is_cuda or w.dtype != torch.float32 for w in vt + (et or [])):
            return None
        self._check_v_inputs(v_params)
        if e_params.x is not None:
            self._check_e_inputs(e_params)
        from .csr import cached_adjacency
        n0 = int(v_params.x.size(0))
        n1 = getattr(e_params, 'num_cells', None) or (e_params.x.size(0) if e_params.x is not None else None)
        if n1 is None or n0 == 0:
            return None
        n1 = int(n1)
        adj1 = cached_adjacency(e_params.boundary_index, n1, n0) if e_params.boundary_index is not None and n1 > 0 else None
        n2, adj2 = 0, None
        if c_params is not None:
            n2 = getattr(c_params, 'num_cells', None) or (c_params.x.size(0) if c_params.x is not None else None)
            if n2 is None:
                return None
            n2 = int(n2)
            if c_params.boundary_index is not None and n2 > 0 and n1 > 0:
                adj2 = cached_adjacency(c_params.boundary_index, n2, n1)
        if e_params.x is None and adj1 is None:
            return None                   # edges without features and without boundaries: let the plain path raise
        if train:
            if H * sum(int(w.size(0)) for w in vt) * 4 > 60 * 1024 or (et and H * sum(int(w.size(0)) for w in et) * 4 > 60 * 1024):
                return None               # tables beyond the LDS-table backward kernel: the generic path
            xs = ops.embed_front_train(vt, v_params.x, et, e_params.x if et is not None else None, n1, adj1, n2, adj2, halve=True)
        else:
            ex = e_params.x if et is not None else None
            if infer and not _ffi_dyn() and all(f is None or (f.dim() == 2 and f.is_contiguous() and f.dtype in (torch.float32, torch.long))
                                                for f in (v_params.x, ex)):
                # the prepared form of this launch for the next call on this batch (ops.FrontLaunch)
                launch = ops.FrontLaunch(vt, et, n0, n1, adj1, n2, adj2, True, bi1, bi2, v_params.x, ex)
                cache = _FRONT_CACHE.setdefault(self, {})            # (per batch, like _BLOCKED_CACHE: loops over a few batches)
                if len(cache) >= 16:
                    cache.clear()
                cache[id(bi1)] = (launch, (c_params is not None, self.init_reduce.reduce))
                xs = launch.run(v_params.x, ex)
                if xs is not None:
                    return xs if c_params is not None else xs[:2]
            xs = ops.embed_front(vt, v_params.x, et, ex, n1, adj1, n2, adj2, halve=True)
        return xs if c_params is not None else xs[:2]

    def reset_parameters(self):
        reset(self.v_embed_layer)
        reset(self.e_embed_layer)


def _embedding_tables(layer) -> Optional[List[Tensor]]:
    """The weight(s) of a plain torch.nn.Embedding or of an OGB-style encoder (a list of them, summed over the
    feature columns); None for anything else."""
    def plain(e):
        return (isinstance(e, torch.nn.Embedding) and e.padding_idx is None and e.max_norm is None
                and not e.sparse and not e.scale_grad_by_freq)
    if layer is None:
        return None
    if plain(layer):
        return [layer.weight]
    for name in ('atom_embedding_list', 'bond_embedding_list'):
        if hasattr(layer, name):
            tables = list(getattr(layer, name))
            return [e.weight for e in tables] if tables and all(plain(e) for e in tables) else None
    return None


class EmbedVEWithReduce(AbstractEmbedVEWithReduce):
    """mp/layers.py:550-570."""

    def _check_v_inputs(self, v_params):
        assert v_params.x is not None
        assert v_params.x.dim() == 2
        assert v_params.x.size(1) == 1

    def _check_e_inputs(self, e_params):
        assert self.e_embed_layer is not None
        assert e_params.x.dim() == 2
        assert e_params.x.size(1) == 1

    def _prepare_v_inputs(self, v_params):
        self._check_v_inputs(v_params)
        return v_params.x.squeeze(1).to(dtype=torch.long)

    def _prepare_e_inputs(self, e_params):
        self._check_e_inputs(e_params)
        return e_params.x.squeeze(1).to(dtype=torch.long)


class OGBEmbedVEWithReduce(AbstractEmbedVEWithReduce):
    """mp/layers.py:573-593 (the OGB Atom/Bond encoders themselves are third-party; any module
    mapping integer feature columns to embeddings fits)."""

    def _check_v_inputs(self, v_params):
        assert v_params.x is not None
        assert v_params.x.dim() == 2

    def _check_e_inputs(self, e_params):
        assert self.e_embed_layer is not None
        assert e_params.x.dim() == 2

    def _prepare_v_inputs(self, v_params):
        self._check_v_inputs(v_params)
        return v_params.x.to(dtype=torch.long)

    def _prepare_e_inputs(self, e_params):
        self._check_e_inputs(e_params)
        return e_params.x.to(dtype=torch.long)
