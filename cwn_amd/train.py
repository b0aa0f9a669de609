"""One optimisation step of a cell-complex model on a GPU-resident batch, as the reference's
training loop runs it (exp/train_utils.py:57-75: zero_grad, forward, loss, backward, step), laid
out for MI355X:

  * all gradients live in ONE flat fp32 buffer (dist.FlatGradBucket): zeroing them is one fill,
    the weight-gradient kernels accumulate straight into it (ops.ACCUMULATE_INTO_GRAD), and the
    data-parallel collective of a step is a single RCCL all-reduce of that buffer;
  * the optimiser is Adam over ONE flat parameter buffer (`FlatAdam`, cwn_adam_f32: one launch
    for the whole model; torch's fused multi-tensor Adam needs 8 x 22 us for the 265 tensors);
  * the whole step -- plan reuse, forward, backward, optimiser -- is captured once per distinct batch
    in a hipGraph and replayed (world size 1), or as two graphs around the eager all-reduce.

Loss functions follow exp/train_utils.py:10-13 (L1 for 'regression', MSE, BCE-with-logits, CE).
"""
import os
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import ops
import torch.distributed as dist

from . import _ffi
from .dist import FlatGradBucket

# 'thread_local': a capture is only invalidated by calls of the capturing thread, not by another
# thread of the process touching the runtime meanwhile (the RCCL watchdog of torch.distributed)
CAPTURE_MODE = 'thread_local'

_LOSSES: Dict[str, Callable] = {
    'regression': torch.nn.L1Loss(),                 # reg_criterion
    'mse_regression': torch.nn.MSELoss(),            # msereg_criterion
    'bin_classification': torch.nn.BCEWithLogitsLoss(),
    'classification': torch.nn.CrossEntropyLoss(),
}


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
        off = 0
        with torch.no_grad():
            for p in bucket.params:
                n = p.numel()
                view = self.flat_p[off:off + n].view_as(p)
                view.copy_(p.data)
                p.data = view
                off += n
        self.exp_avg = torch.zeros_like(g)
        self.exp_avg_sq = torch.zeros_like(g)
        self.t = torch.zeros(1, dtype=torch.int32, device=g.device)
        # what TrainStep snapshots around its warm-up
        self.param_groups = [{'params': list(bucket.params)}]
        self.state = {}

    def state_tensors(self) -> List[torch.Tensor]:
        return [self.exp_avg, self.exp_avg_sq, self.t]

    def zero_grad(self, set_to_none: bool = False):
        self.bucket.zero_()

    @torch.no_grad()
    def step(self):
        self.t.add_(1)
        g = self.bucket.flat
        _ffi.check(_ffi.lib().cwn_adam_f32(
            self.flat_p.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            g.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
            self.t.data_ptr(), _ffi.stream_ptr(g.device)), 'cwn_adam_f32')


class TrainStep:
    """step(i) runs one optimisation step on batches[i] and returns the (detached) loss.

    `batches` are ComplexBatch objects already on the device; their input features are restored
    before every step because the models overwrite them layer by layer (set_xs, as the reference
    does).  With `use_graph` the first call of step(i) for each i captures, later calls replay."""

    def __init__(self, model: torch.nn.Module, batches: Sequence, task_type: str = 'regression',
                 lr: float = 1e-3, use_graph: bool = True, optimizer: Optional[torch.optim.Optimizer] = None,
                 rebuild_plans: bool = True):
        if task_type not in _LOSSES:
            raise NotImplementedError('Training on task type {} not yet supported.'.format(task_type))
        self.model, self.batches = model.train(), list(batches)
        self.loss_fn = _LOSSES[task_type]
        self.task_type = task_type
        self.bucket = FlatGradBucket(model.parameters())
        self.opt = optimizer or FlatAdam(self.bucket, lr=lr)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if os.environ.get('CWN_TRAIN_TWO_GRAPH') == '1':     # debugging: the data-parallel form on one rank
            self.world = max(self.world, 2)
        self.use_graph = use_graph
        self.inputs = [[None if c.x is None else c.x.clone() for c in self._cochains(b)]
                       for b in self.batches]
        self._graphs: Dict[int, tuple] = {}
        self._warm = False
        # a new batch needs its adjacency plans (forward + transposed) built: part of the step
        # unless the caller trains on a fixed set of batches and says so
        self.rebuild_plans = rebuild_plans
        for b in self.batches:
            b.prepare(backward=True)

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

    def _forward_backward(self, i: int) -> torch.Tensor:
        b = self._restore(i)
        if self.rebuild_plans:
            b.forget_plans().prepare(backward=True)
        self.bucket.zero_()
        pred = self.model(b)
        y = b.y.view(-1,) if self.task_type == 'classification' else b.y.view(pred.shape).to(pred.dtype)
        loss = self.loss_fn(pred, y)
        with ops.accumulate_into_grad():        # gradients are views into self.bucket: kernels add in place
            loss.backward()
        self._restore(i)                          # drop the references to the autograd graph
        return loss.detach()

    def _eager(self, i: int) -> torch.Tensor:
        loss = self._forward_backward(i)
        if self.world > 1:
            self.bucket.all_reduce_mean(n_local=self.batches[i].num_complexes)
        self.opt.step()
        return loss

    def _capture(self, i: int):
        if not self._warm:
            # side-stream warm-up of every batch, once (allocator pools, lazy optimiser state,
            # plan caches) -- with the model / optimiser state put back afterwards, so that a
            # graphed run takes exactly the steps an eager run takes
            keep = [t.detach().clone() for t in self._state_tensors()]
            n_before = len(keep)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for k in range(len(self.batches)):
                    self._eager(k)
            torch.cuda.current_stream().wait_stream(s)
            with torch.no_grad():
                now = self._state_tensors()
                for t, old in zip(now[:n_before], keep):
                    t.copy_(old)
                for t in now[n_before:]:          # optimiser state created by the warm-up
                    t.zero_()
            torch.cuda.synchronize()
            self._warm = True
        g1 = torch.cuda.CUDAGraph()
        if self.world == 1:
            with torch.cuda.graph(g1, capture_error_mode=CAPTURE_MODE):
                loss = self._eager(i)
            return (g1, None, loss)
        with torch.cuda.graph(g1, capture_error_mode=CAPTURE_MODE):
            loss = self._forward_backward(i)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode=CAPTURE_MODE):
            self.opt.step()
        return (g1, g2, loss)

    # ---- the step ----------------------------------------------------------------------------
    def step(self, i: int) -> torch.Tensor:
        if not self.use_graph:
            return self._eager(i)
        if i not in self._graphs:
            self._graphs[i] = self._capture(i)
        g1, g2, loss = self._graphs[i]
        g1.replay()
        if g2 is not None:
            self.bucket.all_reduce_mean(n_local=self.batches[i].num_complexes)   # the ONE collective of the step (RCCL over xGMI)
            g2.replay()
        return loss
