"""One optimisation step of a cell-complex model on a GPU-resident batch, as the reference's
training loop runs it (exp/train_utils.py:57-75: zero_grad, forward, loss, backward, step), laid
out for MI355X:

  * all gradients live in ONE flat fp32 buffer (dist.FlatGradBucket): zeroing them is one fill,
    the weight-gradient kernels accumulate straight into it (ops.ACCUMULATE_INTO_GRAD), and the
    data-parallel collective of a step is a single RCCL all-reduce of that buffer;
  * the optimiser is Adam over ONE flat parameter buffer (`FlatAdam`, cwn_adam_f32: one launch
    for the whole model; torch's fused multi-tensor Adam needs 8 x 22 us for the 265 tensors);
  * the whole step -- plan reuse, forward, backward, optimiser -- is captured once per distinct batch
    in a hipGraph and replayed (world size 1);
  * under data parallelism the backward is cut at the outputs of the message-passing layers into S pieces
    (dist.StagedBackward), one hipGraph each, and the gradients of a piece are all-reduced (asynchronously, on
    RCCL's stream) while the next piece replays: only the last chunk's collective -- the first layer and the
    embeddings -- is exposed before the optimiser graph.

Loss functions follow exp/train_utils.py:10-13 (L1 for 'regression', MSE, BCE-with-logits, CE).
"""
import os
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import ops
import torch.distributed as dist

from . import _ffi
from .dist import FlatGradBucket, StagedBackward

# 'thread_local': a capture is only invalidated by calls of the capturing thread, not by another
# thread of the process touching the runtime meanwhile (the RCCL watchdog of torch.distributed)
CAPTURE_MODE = 'thread_local'

_LOSSES: Dict[str, Callable] = {
    'regression': torch.nn.L1Loss(),                 # reg_criterion
    'mse_regression': torch.nn.MSELoss(),            # msereg_criterion
    'bin_classification': torch.nn.BCEWithLogitsLoss(),
    'classification': torch.nn.CrossEntropyLoss(),
}


class _FusedMeanLoss(torch.autograd.Function):
    """criterion(pred, y) for the elementwise-mean criteria, value AND gradient from ONE launch (cwn_loss_f32); the
    framework's own form is ~9 launches of a few hundred elements (45 us of a 1.5 ms step)."""

    @staticmethod
    def forward(ctx, pred, y, kind):
        p = pred.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        n_dev = _ffi.dyn(p.size(0)) if p.dim() >= 1 else None          # a static batch: the complexes that exist (device int64)
        cols = p.numel() // p.size(0) if (p.dim() >= 1 and p.size(0) > 0) else 1         # (multi-task heads: [complexes, tasks])
        # (cross-entropy: `y` holds one int64 class per row -- CWN_LOSS_CE reads it as such)
        _ffi.check(_ffi.lib().cwn_loss_cols_f32(kind, p.data_ptr(), y.contiguous().data_ptr(), p.numel(), cols, loss.data_ptr(),
                                                grad.data_ptr(), n_dev, _ffi.stream_ptr(p.device)), 'cwn_loss_cols_f32')
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        one = _ONES.get(g.device)
        if one is not None and g.data_ptr() == one.data_ptr():      # the seed of TrainStep's backward: exactly 1
            return grad, None, None
        return grad * g, None, None


_ONES = {}


def _one(device) -> torch.Tensor:
    """A cached scalar 1.0: the seed of loss.backward() without the framework's fill."""
    t = _ONES.get(device)
    if t is None:
        t = _ONES[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


_FUSED_KIND = {'regression': 0, 'mse_regression': 1, 'bin_classification': 2, 'classification': 3}      # = CWN_LOSS_*
FUSED_LOSS = True


def fused_loss(task_type: str, pred: torch.Tensor, y: torch.Tensor) -> Optional[torch.Tensor]:
    """The task's criterion through cwn_loss_f32, or None when it does not apply (CPU tensors, other dtypes / shapes,
    CrossEntropy)."""
    kind = _FUSED_KIND.get(task_type)
    if not FUSED_LOSS or kind is None or not pred.is_cuda or pred.dtype != torch.float32 or pred.numel() == 0 or not y.is_cuda \
            or y.device != pred.device:
        return None
    if task_type == 'classification':
        # torch.nn.CrossEntropyLoss() (exp/train_utils.py:21-22): logits [complexes, classes], one int64 class per complex
        if pred.dim() != 2 or y.dtype != torch.long or y.dim() != 1 or y.numel() != pred.size(0):
            return None
    elif y.dtype != torch.float32 or pred.shape != y.shape:
        return None
    return _FusedMeanLoss.apply(pred, y, kind)


class FlatAdam:
    """torch.optim.Adam semantics (no amsgrad) on flat buffers: the parameters of `bucket` are
    re-homed into one contiguous fp32 buffer (their `.data` become views of it, so modules keep
    working and state_dicts are unchanged), the moments are two more flat buffers, and `step()` is
    one kernel launch reading the bucket's flat gradient.  Graph-capturable (the step counter
    lives on the device)."""

    def __init__(self, bucket: FlatGradBucket, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        self.bucket, self.lr, self.betas, self.eps, self.weight_decay = bucket, lr, betas, eps, weight_decay
        g = bucket.flat
        self.flat_p = torch.empty_like(g)
        self.flat_p.zero_()                     # the pad elements between parameters (bucket.offsets) stay zero
        with torch.no_grad():
            for p, off in zip(bucket.params, bucket.offsets):
                view = self.flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
        self.exp_avg = torch.zeros_like(g)
        self.exp_avg_sq = torch.zeros_like(g)
        self.t = torch.zeros(1, dtype=torch.int32, device=g.device)
        # a device int64 (or None): the complexes of the batch this step belongs to -- a step on an EMPTY batch of a static
        # epoch changes nothing (static_graph.StaticTrainStep sets it per slot)
        self.active: Optional[torch.Tensor] = None
        self.counted = False        # the next step()'s count has been taken by the step's opening launch (TrainStep._begin)
        # what TrainStep snapshots around its warm-up
        self.param_groups = [{'params': list(bucket.params)}]
        self.state = {}

    def state_tensors(self) -> List[torch.Tensor]:
        return [self.exp_avg, self.exp_avg_sq, self.t]

    def zero_grad(self, set_to_none: bool = False):
        self.bucket.zero_()

    @torch.no_grad()
    def step(self):
        if self.counted:                 # (the step's opening launch has advanced the counter: cwn_step_begin)
            self.counted = False
        elif self.active is None:
            self.t.add_(1)
        else:
            self.t.add_((self.active > 0).to(torch.int32).view(1))
        g = self.bucket.flat
        _ffi.check(_ffi.lib().cwn_adam_f32(
            self.flat_p.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            g.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
            self.t.data_ptr(), _ffi.ptr(self.active), _ffi.stream_ptr(g.device)), 'cwn_adam_f32')
        ops.weights_changed()        # the parameters were written through a raw pointer: tensor versions did not move


class TrainStep:
    """step(i) runs one optimisation step on batches[i] and returns the (detached) loss.

    `batches` are ComplexBatch objects already on the device; their input features are restored
    before every step because the models overwrite them layer by layer (set_xs, as the reference
    does).  With `use_graph` the first call of step(i) for each i captures, later calls replay."""

    def __init__(self, model: torch.nn.Module, batches: Sequence, task_type: str = 'regression',
                 lr: float = 1e-3, use_graph: bool = True, optimizer: Optional[torch.optim.Optimizer] = None,
                 rebuild_plans: bool = True, stages: Optional[int] = None, share: Optional['TrainStep'] = None):
        """`stages`: number of pieces the backward is cut into so that the gradient all-reduce overlaps with it
        (None: one per message-passing layer, at most 4, when the process group has more than one rank, else 1;
        env CWN_TRAIN_STAGES overrides).  Needs `model.convs`; a network whose layers cannot be cut (see
        dist.StagedBackward.stages) falls back to 1."""
        if task_type not in _LOSSES:
            raise NotImplementedError('Training on task type {} not yet supported.'.format(task_type))
        self.model, self.batches = model.train(), list(batches)
        self.loss_fn = _LOSSES[task_type]
        self.task_type = task_type
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if live else 1
        from . import dist as _cd
        if _cd.force_dp():                 # the data-parallel form on one rank (cwn_amd/dist.py: FORCE_DP)
            self.world = max(self.world, 2)
        if self.world > 1 and live:
            # a rank with fewer batches would leave the others waiting in a collective
            n = torch.tensor([len(self.batches), -len(self.batches)], dtype=torch.int64,
                             device=next(model.parameters()).device)
            dist.all_reduce(n, op=dist.ReduceOp.MAX)
            if int(n[0]) != -int(n[1]):
                raise ValueError('TrainStep: every rank must hold the same number of batches '
                                 f'(this rank {len(self.batches)}, max {int(n[0])}, min {-int(n[1])})')
        for b in self.batches:
            b.prepare(backward=True)
        self.inputs = [[None if c.x is None else c.x.clone() for c in self._cochains(b)]
                       for b in self.batches]
        if os.environ.get('CWN_TRAIN_STAGES'):
            stages = int(os.environ['CWN_TRAIN_STAGES'])
        convs = list(getattr(model, 'convs', []))
        if stages is None:
            stages = min(len(convs), 4) if self.world > 1 else 1
        stages = max(1, min(int(stages), len(convs)))
        self.staged, stage_of = None, None
        if stages > 1 and self.batches:
            # cuts behind layers k_1 > k_2 > ...: evenly spread over layers 0 .. L-2 (the head rides with the last layer)
            L = len(convs)
            ks = sorted({round(v * (L - 2) / max(1, stages - 2)) for v in range(stages - 1)} if stages > 2 else {0})
            self.staged = StagedBackward([convs[k] for k in ks])
            stage_of = self._probe_stages(convs, ks)
            if stage_of is None:
                import warnings
                warnings.warn('TrainStep: the backward cannot be cut behind the message-passing layers (jumping '
                              'knowledge / a skip connection around layers): the gradient all-reduce runs as ONE '
                              'collective after the backward, without overlap')
                self.staged.remove()
                self.staged = None
        self.n_stages = self.staged.n_stages if self.staged is not None else 1
        # `share`: a second step driver over the SAME model (static_graph.RoutedTrainStep: one per static batch) takes the
        # first one's gradient bucket and optimizer -- the parameters have one flat gradient and one Adam state
        if share is not None:
            if share.model is not model or share.n_stages != self.n_stages:
                raise ValueError('TrainStep(share=...): the same model and the same number of backward pieces')
            self.bucket = share.bucket
            optimizer = optimizer or share.opt
        else:
            self.bucket = FlatGradBucket(model.parameters(), stage_of, self.n_stages)
        if self.world > 1 and live:
            # the ranks reduce the bucket chunk by chunk: its layout must be the same everywhere
            lay = torch.tensor([hi for _, hi in self.bucket.chunks], dtype=torch.int64, device=self.bucket.flat.device)
            lay = torch.cat([lay, -lay])
            dist.all_reduce(lay, op=dist.ReduceOp.MAX)
            if not torch.equal(lay[:self.n_stages], -lay[self.n_stages:]):
                raise RuntimeError('TrainStep: the gradient bucket is laid out differently on different ranks')
        self.stage_params = [[p for p in self.bucket.params if stage_of[id(p)] == self.n_stages - 1 - j]
                             for j in range(self.n_stages)] if stage_of is not None else None
        self.opt = optimizer or FlatAdam(self.bucket, lr=lr)
        if os.environ.get('CWN_TRAIN_TWO_GRAPH') == '1':     # debugging: the data-parallel form on one rank
            self.world = max(self.world, 2)
        self.use_graph = use_graph
        self._graphs: Dict[int, tuple] = {}
        self._warm = False
        # a new batch needs its adjacency plans (forward + transposed) built: part of the step
        # unless the caller trains on a fixed set of batches and says so
        self.rebuild_plans = rebuild_plans
        # set by the warm-up of the first capture: every layer of every batch took the complex-blocked launches forward AND
        # backward, so a step's plan build leaves the upper adjacencies out (one csr launch per step instead of two: the ten
        # plans of a ZINC batch -- 4 + 6 transposes -- are two batched calls, the four boundary ones are one)
        self._skip_upper_plans = False

    def _count_at_begin(self) -> bool:
        """May the step's opening launch advance FlatAdam's counter (it knows whether the step is real)?"""
        return True

    def _begin(self):
        """The bracket of one step (ops.step_arena): gradients zeroed, arena zeroed, FlatAdam's counter advanced."""
        mine = isinstance(self.opt, FlatAdam) and self._count_at_begin()
        if mine:
            self.opt.counted = True
        return ops.step_arena(self.bucket.flat.device, flat=self.bucket.flat, counter=self.opt.t if mine else None,
                              active=self.opt.active if mine else None)

    def _abort_step(self) -> None:
        """A step that raised between its opening launch and the optimizer: the count _begin took is given back."""
        if isinstance(self.opt, FlatAdam) and self.opt.counted:
            self.opt.counted = False
            with torch.no_grad():
                if self.opt.active is None:
                    self.opt.t.sub_(1)
                else:
                    self.opt.t.sub_((self.opt.active > 0).to(torch.int32).view(1))

    # ---- pieces ------------------------------------------------------------------------------
    @staticmethod
    def _cochains(b):
        return [b.cochains[d] for d in range(b.dimension + 1)]

    def _restore(self, i: int):
        b = self.batches[i]
        for c, x in zip(self._cochains(b), self.inputs[i]):
            c._x = x
        return b

    def _state_tensors(self) -> List[torch.Tensor]:
        """Parameters, buffers, then the optimiser's state tensors (in a fixed order)."""
        ts = list(self.model.parameters()) + list(self.model.buffers())
        if hasattr(self.opt, 'state_tensors'):
            return ts + self.opt.state_tensors()
        for group in self.opt.param_groups:
            for p in group['params']:
                st = self.opt.state.get(p, {})
                ts += [st[k] for k in sorted(st) if torch.is_tensor(st[k])]
        return ts

    def _probe_stages(self, convs, ks) -> Optional[dict]:
        """{id(parameter): stage}, the same on every rank, or None when the network cannot be cut behind the
        layers `ks`.  The parameters of a layer take the stage of the layer; every other parameter (embeddings,
        head) the stage one forward per batch (buffers put back afterwards) finds it at in the autograd graph --
        the maximum over the ranks, whose batches may reach different parameters (a shard without 2-cells) --
        and stage 0, reduced last, when no batch of any rank reaches it."""
        params = [p for p in self.model.parameters() if p.requires_grad]
        keep = [t.detach().clone() for t in self.model.buffers()]
        # the probe forwards run in training mode: the dropout masks they draw must not move the caller's
        # random stream -- the generator states are put back like the buffers
        rng_cpu = torch.get_rng_state()
        rng_dev = torch.cuda.get_rng_state(params[0].device) if params and params[0].is_cuda else None
        reached, ok = {}, True
        for i in range(len(self.batches)):
            self.staged.begin()
            loss = self._loss(self._restore(i))
            so = self.staged.stages(loss, params, default=None)
            del loss
            self.staged._cuts = []
            self._restore(i)
            if so is None:
                ok = False
                break
            ok = ok and all(reached.setdefault(k, v) == v for k, v in so.items())
        with torch.no_grad():
            for t, old in zip(self.model.buffers(), keep):
                t.copy_(old)
        torch.set_rng_state(rng_cpu)
        if rng_dev is not None:
            torch.cuda.set_rng_state(rng_dev, params[0].device)
        for k, conv in enumerate(convs):
            st = sum(1 for c in ks if c < k)
            ok = ok and all(reached.setdefault(id(p), st) == st for p in conv.parameters() if p.requires_grad)
        mine = torch.tensor([reached.get(id(p), -1) for p in params] + [0 if ok else 1], dtype=torch.int64,
                            device=params[0].device)
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            both = mine.clone()
            dist.all_reduce(both, op=dist.ReduceOp.MAX)
            clash = ((mine >= 0) & (mine != both)).any().to(torch.int64).view(1)
            dist.all_reduce(clash, op=dist.ReduceOp.MAX)
            both[-1] = torch.maximum(both[-1], clash[0])
            mine = both
        mine = mine.tolist()
        if mine[-1] != 0:
            return None
        return {id(p): max(0, st) for p, st in zip(params, mine)}

    def _loss(self, b) -> torch.Tensor:
        pred = self.model(b)
        y = b.y.view(-1,) if self.task_type == 'classification' else b.y.view(pred.shape).to(pred.dtype)
        loss = fused_loss(self.task_type, pred, y)
        if loss is not None:
            return loss
        if self.task_type != 'classification' and not (pred.is_cuda and torch.cuda.is_current_stream_capturing()):
            # null labels (exp/train_utils.py:64-68): `mask = ~torch.isnan(targets)`; the fused kernel does the same in place
            # (boolean indexing is a host sync: not inside a capture, where only the fused form masks)
            mask = ~torch.isnan(y)
            return self.loss_fn(pred[mask], y[mask])
        return self.loss_fn(pred, y)

    def _forward_backward(self, i: int, pieces: Optional[Sequence[int]] = None):
        """zero the gradients, forward, backward.  With a staged backward `pieces` selects what runs now:
        [0] = zero + forward + the first piece (returns the loss, which the later pieces need), [j] = piece j."""
        if self.staged is None:
            b = self._restore(i)
            if self.rebuild_plans:
                b.forget_plans().prepare(backward=True, upper=not self._skip_upper_plans)
            # zero_grad + what the step's kernels need zero on entry + the optimizer's step counter: one launch
            try:
                with self._begin():
                    loss = self._loss(b)
                    with ops.accumulate_into_grad():        # gradients are views into self.bucket: kernels add in place
                        loss.backward(gradient=_one(loss.device))      # (no fill for the seed; _FusedMeanLoss hands its gradient on as is)
            except BaseException:
                self._abort_step()
                raise
            finally:
                self._restore(i)                      # drop the references to the autograd graph
            return loss.detach()
        S = self.n_stages
        try:
            for j in (range(S) if pieces is None else pieces):
                if j == 0:
                    b = self._restore(i)
                    if self.rebuild_plans:
                        b.forget_plans().prepare(backward=True, upper=not self._skip_upper_plans)
                    self._arena = self._begin()
                    self._arena.__enter__()
                    self.staged.begin()
                    self._live_loss = self._loss(b)
                with ops.accumulate_into_grad():
                    self.staged.piece(j, self._live_loss, self.stage_params[j])
                if j == S - 1:
                    loss, self._live_loss = self._live_loss.detach(), None
                    arena, self._arena = self._arena, None
                    arena.__exit__(None, None, None)
                    self._restore(i)
                    return loss
        except BaseException as e:
            # a forward / backward that raised (out of memory, a refused layer): close the step's bracket and hand the
            # optimizer's count back, or the NEXT step would skip its increment against a stale count
            arena, self._arena = getattr(self, '_arena', None), None
            if arena is not None:
                arena.__exit__(type(e), e, e.__traceback__)
            self._live_loss = None
            self._abort_step()
            self._restore(i)
            raise
        return self._live_loss.detach()

    def _n_local(self, i: int):
        """The samples this rank's loss of step i averages over: its weight in the gradient all-reduce (a static batch hands a
        device tensor)."""
        return self.batches[i].num_complexes

    def _before_optimizer(self, i: int) -> None:
        """Between the (reduced) gradient and the optimizer's step (static_graph.StaticTrainStep: the global sample count)."""

    def _eager(self, i: int) -> torch.Tensor:
        n_local = self._n_local(i)
        if self.staged is None:
            loss = self._forward_backward(i)
            if self.world > 1:
                self.bucket.all_reduce_mean(n_local=n_local)
        else:
            for j in range(self.n_stages):
                loss = self._forward_backward(i, [j])
                self.bucket.reduce_chunk(j, n_local)      # overlaps with piece j + 1
            self.bucket.finish()
        self._optimizer_step(i)
        return loss

    def _optimizer_step(self, i: int) -> None:
        """What follows the (reduced) gradient: eagerly, and as the data-parallel form's last captured graph."""
        self._before_optimizer(i)
        self.opt.step()

    def _capture(self, i: int):
        if not self._warm:
            # side-stream warm-up of every batch, once (allocator pools, lazy optimiser state,
            # plan caches) -- with the model / optimiser state put back afterwards, so that a
            # graphed run takes exactly the steps an eager run takes
            keep = [t.detach().clone() for t in self._state_tensors()]
            n_before = len(keep)
            # ... including the device-side dropout stream {seed, step}: every warm-up step's opening launch advances `step`
            # (ops.step_arena), and the masks of a graphed run must be the ones an eager run draws on the same seed
            from . import ops as _ops
            dev_ = next(self.model.parameters()).device
            ds_before = _ops._drop_states.get(dev_)
            ds_keep = None if ds_before is None else ds_before.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            n_blocked = _ops.BLOCKED_BACKWARD_LAUNCHES[1]
            with torch.cuda.stream(s):
                for k in range(len(self.batches)):
                    self._eager(k)
            torch.cuda.current_stream().wait_stream(s)
            n_convs = len(list(getattr(self.model, 'convs', [])))
            if (self.rebuild_plans and n_convs and os.environ.get('CWN_TRAIN_UPPER_PLANS') != '1'
                    and _ops.BLOCKED_BACKWARD_LAUNCHES[1] - n_blocked == n_convs * len(self.batches)):
                self._skip_upper_plans = True
            with torch.no_grad():
                now = self._state_tensors()
                for t, old in zip(now[:n_before], keep):
                    t.copy_(old)
                for t in now[n_before:]:          # optimiser state created by the warm-up
                    t.zero_()
                ds_now = _ops._drop_states.get(dev_)
                if ds_now is not None:
                    if ds_keep is not None:
                        ds_now.copy_(ds_keep)
                    else:                         # created by the warm-up: its seed stays, its step counter rewinds
                        ds_now[1] = 0
            torch.cuda.synchronize()
            self._warm = True
        g1 = torch.cuda.CUDAGraph()
        if self.world == 1 and self.staged is None:
            with torch.cuda.graph(g1, capture_error_mode=CAPTURE_MODE):
                loss = self._eager(i)
            return ([g1], None, loss)
        # data parallel: graph(s) of forward + backward, the collective(s) issued eagerly between them, graph(Adam)
        pieces = [g1]
        with torch.cuda.graph(g1, capture_error_mode=CAPTURE_MODE):
            loss = self._forward_backward(i, None if self.staged is None else [0])
        for j in range(1, self.n_stages):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=g1.pool(), capture_error_mode=CAPTURE_MODE):
                self._forward_backward(i, [j])
            pieces.append(g)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode=CAPTURE_MODE):
            self._optimizer_step(i)
        return (pieces, g2, loss)

    MAX_SEQ_GRAPHS = 8

    # ---- several steps behind one replay -------------------------------------------------------
    def steps(self, seq: Sequence[int]) -> List[torch.Tensor]:
        """The steps on batches[seq[0]], batches[seq[1]], ... in this order; returns their losses.  With `use_graph` on one
        rank the whole sequence is ONE captured graph (per distinct sequence): what is saved is the gap between two replays
        (measured at the ZINC batch of 128 with four steps per graph: 4 - 22 us per step depending on how far ahead the host
        runs), which a training loop over a fixed epoch order need not pay once per step.  Otherwise (eager, data parallel) the steps run one by one."""
        seq = tuple(int(i) for i in seq)
        if not self.use_graph or self.world > 1 or self.staged is not None or len(seq) < 2:
            return [self.step(i) for i in seq]
        key = ('seq',) + seq
        if key not in self._graphs:
            # (a loop that reshuffles its batch order every epoch asks for a new sequence each time; every captured
            # graph holds device memory of its own -- keep the most recent MAX_SEQ_GRAPHS of them)
            seqs = [k for k in self._graphs if isinstance(k, tuple)]
            while len(seqs) >= self.MAX_SEQ_GRAPHS:
                del self._graphs[seqs.pop(0)]
            for i in dict.fromkeys(seq):                       # warm-up and the one-step graphs (they own the warm-up logic)
                if i not in self._graphs:
                    self._graphs[i] = self._capture(i)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                losses = [self._eager(i) for i in seq]
            self._graphs[key] = (g, losses)
        g, losses = self._graphs[key]
        ops.weights_changed()
        g.replay()
        return losses

    # ---- the step ----------------------------------------------------------------------------
    def step(self, i: int) -> torch.Tensor:
        if not self.use_graph:
            return self._eager(i)
        if i not in self._graphs:
            self._graphs[i] = self._capture(i)
        pieces, g2, loss = self._graphs[i]
        ops.weights_changed()        # a replay runs no Python: caches of packed weights / folded BatchNorm must not survive it
        if g2 is None:
            pieces[0].replay()
            return loss
        n_local = self._n_local(i)
        if self.staged is None:
            pieces[0].replay()
            self.bucket.all_reduce_mean(n_local=n_local)   # the ONE collective of the step (RCCL over xGMI)
        else:
            for j, g in enumerate(pieces):
                g.replay()
                self.bucket.reduce_chunk(j, n_local)        # on RCCL's stream, while the next piece replays
            self.bucket.finish()
        g2.replay()
        return loss
