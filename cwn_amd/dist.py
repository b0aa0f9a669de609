"""Multi-GPU: one process per GPU, complexes sharded across ranks, ONE collective per step.

A ComplexBatch is a disjoint union (every index is offset per complex, data/complex.py:148-169),
so adjacency is block-diagonal and no message crosses complexes: rank r of R takes complexes
r, r+R, ... builds its own batch, and forward / backward need no exchange.  The only
collective of a training step is the gradient all-reduce, done as a single flat fp32 bucket
(~1.7 M parameters = 6.8 MB for the ZINC model): one RCCL ring all-reduce over xGMI instead of
one per parameter tensor.  For pure propagate throughput (bench.py) there is no communication at
all: replicas over disjoint shards.  The reference has no distributed code.

Works with any torch.distributed backend ("nccl" is RCCL on ROCm; "gloo" in the CPU tests).
"""
import os
from typing import Iterable, List, Optional, Sequence, TypeVar

import torch
import torch.distributed as dist

T = TypeVar('T')

# The data-parallel form on ONE rank (CWN_FORCE_DP=1, or set by a test): TrainStep cuts its backward and brackets the
# collectives with graphs as it does for world > 1, and the bucket issues its all-reduces to a process group of one rank --
# RCCL reduces the buffer with itself.  Everything a first multi-GPU run exercises except the xGMI wire: the library load,
# the communicator, RCCL's stream ordering against the graph replays, "no collective inside a capture".
FORCE_DP = os.environ.get('CWN_FORCE_DP') == '1'


def force_dp() -> bool:
    return FORCE_DP


def init_from_env(backend: Optional[str] = None) -> (int, int):
    """(rank, world) from the torchrun environment; initialises the process group when world > 1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            local = int(os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, **kw)
    return rank, world


def shard(items: Sequence[T], rank: int, world: int) -> List[T]:
    """Round-robin shard of a list of complexes: rank r gets items r, r+world, ..."""
    return list(items[rank::world])


class FlatGradBucket:
    """All gradients of a parameter set in one contiguous fp32 buffer.  Gradients are views into the bucket, so
    there is no pack / unpack copy around the collective.

    Plain use: `all_reduce_mean()` after the backward -- the single collective of a data-parallel step.

    Staged use (`stage_of` given): the parameters are laid out by STAGE, the stage whose gradients are complete
    first (the last layers of the network) at the front, and the buffer is reduced chunk by chunk while the
    backward of the earlier layers still runs: `reduce_chunk(0, w)` as soon as the backward has left the last
    stage, ..., `reduce_chunk(S - 1, w)` after the backward, then `finish()`.  xGMI rings are per-link bound
    (~150 GB/s per link): a 6.8 MB bucket is ~0.1 ms of wire time that this hides behind the backward kernels."""

    def __init__(self, params: Iterable[torch.nn.Parameter], stage_of: Optional[dict] = None,
                 n_stages: int = 1):
        """`stage_of`: {id(parameter): stage in [0, n_stages)} (StagedBackward.stages); missing = stage 0."""
        ps = [p for p in params if p.requires_grad]
        if not ps:
            raise ValueError('no trainable parameters')
        if stage_of is None:
            n_stages, stages = 1, [0] * len(ps)
        else:
            stages = [int(stage_of.get(id(p), 0)) for p in ps]
            if min(stages) < 0 or max(stages) >= n_stages:
                raise ValueError('stage_of: stages must lie in [0, n_stages)')
        # latest stage first (stable within a stage): the chunks are reduced front to back
        order = sorted(range(len(ps)), key=lambda i: -stages[i])
        self.params = [ps[i] for i in order]
        dev, dt = self.params[0].device, self.params[0].dtype
        # every parameter starts on a 16-byte boundary (the kernels read weights and write gradients as float4:
        # a scalar such as a GIN eps would otherwise misalign everything behind it); the pad elements stay zero
        pad4 = lambda n: (n + 3) // 4 * 4
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += pad4(p.numel())
        # one extra element behind the gradients carries the rank's sample count through the SAME
        # all-reduce (weighted mean over ranks with unequal shards, still one collective)
        self._buf = torch.zeros(total + 1, dtype=dt, device=dev)
        self.flat = self._buf[:total]
        ends = [0] * n_stages
        for i, p, off in zip(order, self.params, self.offsets):
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            ends[n_stages - 1 - stages[i]] = off + pad4(n)
        for c in range(1, n_stages):                # an empty stage: a zero-length chunk
            ends[c] = max(ends[c], ends[c - 1])
        # chunk c = the gradients of stage n_stages - 1 - c; the last one also carries the count element
        self.chunks = [(0 if c == 0 else ends[c - 1], ends[c]) for c in range(n_stages)]
        self._pending: list = []

    @property
    def n_stages(self) -> int:
        return len(self.chunks)

    def chunk(self, c: int) -> torch.Tensor:
        lo, hi = self.chunks[c]
        return self.flat[lo:hi]

    def zero_(self):
        self.flat.zero_()

    @staticmethod
    def _world(group) -> int:
        """Ranks the collectives run over: 1 = none is issued.  A group of ONE rank counts as two under FORCE_DP (the
        all-reduce of the buffer with itself is issued)."""
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        w = dist.get_world_size(group)
        return max(w, 2) if FORCE_DP else w

    def all_reduce_mean(self, group=None, async_op: bool = False, n_local: Optional[int] = None):
        """Mean of the per-rank gradients, weighted by `n_local` (the number of samples the rank's loss
        averaged over; None = equal weights): sum_r n_r g_r / sum_r n_r, the gradient of the mean loss
        over the GLOBAL batch, which is what the single-process reference computes."""
        world = self._world(group)
        if world == 1:
            return None
        w = self._weight(n_local)
        self.flat.mul_(w)
        self._buf[-1:] = w
        work = dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work, world
        self._divide()
        return None

    def _weight(self, n_local):
        """The rank's weight as something `mul_` takes: a float, or -- a static batch, whose sample count lives on the device
        (cwn_amd/static_graph.py) -- a one-element tensor of the bucket's dtype (no host sync)."""
        if torch.is_tensor(n_local):
            return n_local.reshape(1).to(dtype=self._buf.dtype, device=self._buf.device)
        return float(1 if n_local is None else n_local)

    def _divide(self) -> None:
        """flat /= the summed weights.  A step in which NO rank held a sample (tensor weights may be 0: the empty tail of a
        static epoch) leaves the zeros it summed: the divisor is clamped, the optimizer skips that step (global_count())."""
        self.flat.div_(self._buf[-1].clamp(min=torch.finfo(self._buf.dtype).tiny))

    def global_count(self) -> torch.Tensor:
        """The summed weights of the last reduction (one element, on the device): the samples of the GLOBAL batch."""
        return self._buf[-1:]

    def finish(self, handle=None):
        """Complete an async all-reduce started with all_reduce_mean(async_op=True), or every chunk
        started with reduce_chunk()."""
        if handle is not None:
            work, world = handle
            work.wait()
            self._divide()
            return
        if self._pending:
            for work in self._pending:
                work.wait()
            self._pending = []
            self._divide()

    def reduce_chunk(self, c: int, n_local: Optional[int] = None, group=None) -> None:
        """Start the (asynchronous) weighted all-reduce of chunk c; every rank calls it for c = 0 .. S-1 in
        order, with the same `n_local` for every chunk of a step, then `finish()`.  On the RCCL backend the
        collective runs on the process group's own stream behind an event of the current one, so kernels
        launched afterwards (the backward of the earlier stages) overlap with it."""
        if self._world(group) == 1:
            return
        w = self._weight(n_local)
        lo, hi = self.chunks[c]
        last = c == len(self.chunks) - 1
        if last:
            self._buf[-1:] = w
            # the tail chunk goes with the count element even when it holds no gradient
            t = self._buf[lo:]
            t[:-1].mul_(w)
        else:
            if hi == lo:
                return
            t = self.flat[lo:hi]
            t.mul_(w)
        self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))


def _upstream_leaves(roots, barrier):
    """Leaf tensors (AccumulateGrad variables) reachable from the autograd nodes `roots` without walking
    past a node of `barrier` (a barrier node itself is not expanded unless it is a root)."""
    seen, leaves, stack = set(), [], [n for n in roots if n is not None]
    first = set(stack)
    while stack:
        n = stack.pop()
        if n in seen:
            continue
        seen.add(n)
        if hasattr(n, 'variable'):
            leaves.append(n.variable)
            continue
        if n in barrier and n not in first:
            continue
        stack.extend(f for f, _ in n.next_functions if f is not None)
    return leaves


class StagedBackward:
    """The backward of one loss in S pieces, cut at the outputs of S - 1 modules of the network (its
    message-passing layers), so that a caller can act between the pieces -- start the all-reduce of the
    gradients that are already complete.

    `cut_modules` in FORWARD order.  Usage per step:  `begin()`; forward; `loss`; then for j = 0 .. S-1
    `piece(j, loss)`: piece 0 runs from the loss to the outputs of the last cut module, piece j from there to
    the next cut, the last piece to the leaves.  Each piece differentiates with torch.autograd.grad w.r.t. the
    cut tensors and the parameters of its stage only: the engine then executes exactly the nodes between the
    two cuts, each once (backward(inputs=non-leaf) would also execute the producer of the cut tensor, and run
    it a second time in the next piece).

    A cut is valid when every path from the loss to an earlier parameter passes through the cut tensors: a
    layered network without skip connections around whole layers.  Jumping knowledge is NOT such a network --
    the loss reads every layer's output directly, so a layer's parameters are reached in two pieces -- and
    `stages()` returns None for it (the caller keeps the one-collective step and says so, train.py).
    `stages()` checks validity on the autograd graph and returns the stage of every parameter."""

    def __init__(self, cut_modules: Sequence[torch.nn.Module]):
        self.cut_modules = list(cut_modules)
        self._rec = None
        self._cuts: List[List[torch.Tensor]] = []
        self._grads: list = []
        self._handles = [m.register_forward_hook(self._hook(k)) for k, m in enumerate(self.cut_modules)]

    @property
    def n_stages(self) -> int:
        return len(self.cut_modules) + 1

    def _hook(self, k):
        def fn(module, args, out):
            if self._rec is not None:
                outs = out if isinstance(out, (list, tuple)) else [out]
                self._rec[k] = [t for t in outs if torch.is_tensor(t) and t.requires_grad]
        return fn

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def begin(self):
        self._rec = {}

    def _end(self):
        """Cut tensor lists in backward order (the last module's outputs first)."""
        if self._rec is not None:
            rec, self._rec = self._rec, None
            if len(rec) != len(self.cut_modules):
                raise RuntimeError('StagedBackward: a cut module did not run in this forward')
            self._cuts = [rec[k] for k in reversed(range(len(self.cut_modules)))]
        return self._cuts

    def stages(self, loss: torch.Tensor, params: Sequence[torch.nn.Parameter], default: Optional[int] = 0) -> Optional[dict]:
        """{id(parameter): stage} from the autograd graph of `loss` (stage S-1 = complete after piece 0, ...,
        stage 0 = complete after the last piece; a parameter the loss does not reach: `default`, or left out
        when that is None), or None when the cuts are not valid for this network (a parameter is reached in
        two pieces)."""
        cuts = self._end()
        S = len(cuts) + 1
        ids = {id(p) for p in params}
        found = {}
        for j in range(S):
            roots = [loss.grad_fn] if j == 0 else [t.grad_fn for t in cuts[j - 1]]
            barrier = {t.grad_fn for t in cuts[j]} if j < S - 1 else set()
            for v in _upstream_leaves(roots, barrier):
                if id(v) not in ids:
                    continue
                st = S - 1 - j
                if found.setdefault(id(v), st) != st:
                    return None
        return {id(p): found.get(id(p), default) for p in params if default is not None or id(p) in found}

    def piece(self, j: int, loss: torch.Tensor, stage_params: Sequence[torch.nn.Parameter]) -> None:
        """Piece j of the backward; gradients of `stage_params` (the parameters of stage S-1-j) are added to
        their `.grad` -- by the kernels themselves inside ops.accumulate_into_grad(), here otherwise."""
        cuts = self._end()
        S = len(cuts) + 1
        if j == 0:
            roots, seeds = [loss], None
        else:
            pairs = [(t, g) for t, g in zip(cuts[j - 1], self._grads) if g is not None]
            roots, seeds = [t for t, _ in pairs], [g for _, g in pairs]
        cut = list(cuts[j]) if j < S - 1 else []
        ins = cut + list(stage_params)
        gs = torch.autograd.grad(roots, ins, grad_outputs=seeds, allow_unused=True) if roots and ins else [None] * len(ins)
        for p, g in zip(stage_params, gs[len(cut):]):
            if g is not None:
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
        self._grads = list(gs[:len(cut)])
        if j == S - 1:
            self._cuts, self._grads = [], []


def sum_across_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item())
