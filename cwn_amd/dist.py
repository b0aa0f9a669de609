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
    """All gradients of a parameter set in one contiguous fp32 buffer; `all_reduce_mean()` is the
    single collective of a data-parallel step.  Gradients are views into the bucket, so there is
    no pack / unpack copy around the collective."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        # one extra element behind the gradients carries the rank's sample count through the SAME
        # all-reduce (weighted mean over ranks with unequal shards, still one collective)
        self._buf = torch.zeros(total + 1, dtype=dt, device=dev)
        self.flat = self._buf[:total]
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None, async_op: bool = False, n_local: Optional[int] = None):
        """Mean of the per-rank gradients, weighted by `n_local` (the number of samples the rank's loss
        averaged over; None = equal weights): sum_r n_r g_r / sum_r n_r, the gradient of the mean loss
        over the GLOBAL batch, which is what the single-process reference computes."""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        world = dist.get_world_size(group)
        if world == 1:
            return None
        w = float(1 if n_local is None else n_local)
        self.flat.mul_(w)
        self._buf[-1] = w
        work = dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work, world
        self.flat.div_(self._buf[-1])
        return None

    def finish(self, handle):
        """Complete an async all-reduce started with all_reduce_mean(async_op=True)."""
        if handle is not None:
            work, world = handle
            work.wait()
            self.flat.div_(self._buf[-1])


def sum_across_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item())
