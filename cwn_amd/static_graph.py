"""One captured hipGraph for batches of DIFFERENT shapes (propagate scope of a stack of SparseCIN layers).

A hipGraph bakes pointers, grid sizes and kernel arguments in.  The complex-blocked layer kernel
(csrc/cwn_layer.hip) needs none of a batch's sizes in its arguments: which cells and entries a
workgroup touches is written in the item table, and the table lives in device memory.  So the graph
is captured ONCE over capacity-sized buffers -- features [cap_d, F], indices [2, cap] with the second
row at a fixed offset, an item table with a fixed region per set -- and a new batch is served by
copying its tensors and its table into those buffers and replaying.  Records past the batch's own are
empty (flags 0, no task): their workgroups leave at once.  Real ZINC training
(exp/train_utils.py:35: every batch has its own cell and entry counts) replays instead of launching
eagerly; a handful of capacity buckets covers a dataset.
"""
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _ffi, ops
from .blockplan import ITEM_INTS, LDS_BYTES, ItemTable, gemm_rows_cap, lds_bytes
from .complex import ComplexBatch

CAPTURE_MODE = 'thread_local'


class StaticPropagate:
    """propagate scope of `convs` (SparseCINConv modules with the coboundary message; no autograd) over
    static buffers.  caps: cells per dimension, upper entries per dimension (dims 0, 1), boundary entries
    (dims 1, 2), items per set (sets of dims 0 and 1)."""

    def __init__(self, convs: Sequence[torch.nn.Module], F: int, cap_cells: Sequence[int], cap_up: Sequence[int],
                 cap_b: Sequence[int], cap_items: Sequence[int], device):
        self.convs, self.F, self.dev = list(convs), F, torch.device(device)
        self.cap_cells, self.cap_up, self.cap_b, self.cap_items = list(cap_cells), list(cap_up), list(cap_b), list(cap_items)
        dev = self.dev
        L = len(self.convs)
        # inputs of every layer (the propagate scope takes each layer's own input features)
        self.x = [[torch.zeros(n, F, device=dev) for n in self.cap_cells] for _ in range(L)]
        self.up_index = [torch.zeros(2, max(e, 1), dtype=torch.long, device=dev) for e in self.cap_up]
        self.up_shared = [torch.zeros(max(e, 1), dtype=torch.long, device=dev) for e in self.cap_up]
        self.b_index = [None] + [torch.zeros(2, max(e, 1), dtype=torch.long, device=dev) for e in self.cap_b]
        n_items = sum(self.cap_items)
        self.items = torch.zeros(n_items, ITEM_INTS, dtype=torch.int32, device=dev)
        self.set_start = [0, self.cap_items[0]]
        # one launch = one LDS size for every batch this graph will serve: the full row cap, and for the boundary
        # sources what is left of the 160 KiB (F = 64: 256 staged rows leave room for ~170 source rows, not 256 --
        # the first form asked for 186 KB and failed at the first replay)
        cap = gemm_rows_cap(F)
        src_cap = min(cap, (LDS_BYTES - lds_bytes(F, cap, 0)) // (F * 4))
        if src_cap < 16 or _ffi.lib().cwn_layer_fused_lds_bytes(F, cap, src_cap) == 0:
            raise ValueError(f'StaticPropagate: no LDS split for feature width {F}')
        self.table = ItemTable(np.zeros((n_items, ITEM_INTS), dtype=np.int32), self.set_start, cap, src_cap,
                               list(self.cap_cells), list(self.cap_up) + [0], [0] + list(self.cap_b), dev)
        self.table.items = self.items          # the launch reads THIS buffer; `load` rewrites it
        self.launches: List[ops.LayerLaunch] = []
        for l, conv in enumerate(self.convs):
            dims = []
            for d in range(3):
                lvl = conv.mp_levels[d]
                D = ops.LayerDim(x=self.x[l][d], eps1=lvl.eps1, eps2=lvl.eps2)
                if d < 2:
                    lin = lvl.msg_up_nn[1]
                    D.up_index, D.up_shared = self.up_index[d], self.up_shared[d]
                    D.msg_w_packed, D.msg_bias = ops.pack_layer_weight(lin.weight), lin.bias
                if d > 0:
                    D.b_index = self.b_index[d]
                dims.append(D)
            self.launches.append(ops.LayerLaunch(dims, self.table))
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.outs: Optional[List[List[torch.Tensor]]] = None
        self.n_cells = [0, 0, 0]
        # the packed weights above are baked into the launches (and later the graph): inference over fixed weights
        self._weights = [(conv.mp_levels[d].msg_up_nn[1].weight, _ffi.tver(conv.mp_levels[d].msg_up_nn[1].weight))
                         for conv in self.convs for d in range(2)]
        self._epoch = ops.STATE_EPOCH

    # ---- a batch into the static buffers ------------------------------------------------------------
    def load(self, batch: ComplexBatch, feats: Sequence[Sequence[torch.Tensor]]) -> None:
        """Copy `batch`'s indices, its item table and the per-layer input features (feats[l][d]) into the
        static buffers.  Raises when the batch exceeds a capacity (pick a larger bucket)."""
        plan = batch.block_plan()
        table = plan.items(self.F, [True, True, False]) if plan is not None else None
        if table is None:
            raise ValueError('the batch has no item table for this feature width (hub complexes?)')
        if table.max_rows > self.table.max_rows or table.max_src > self.table.max_src:
            raise ValueError(f'an item of this batch needs {table.max_rows} staged / {table.max_src} source rows; the '
                             f'captured launch holds {self.table.max_rows} / {self.table.max_src}')
        cnt = [table.set_start[1], table.n_items - table.set_start[1]]
        if any(c > cap for c, cap in zip(cnt, self.cap_items)):
            raise ValueError(f'items per set {cnt} exceed the capacity {self.cap_items}')
        host = torch.zeros(sum(self.cap_items), ITEM_INTS, dtype=torch.int32)
        src = table.items.cpu()
        host[:cnt[0]] = src[:cnt[0]]
        host[self.cap_items[0]:self.cap_items[0] + cnt[1]] = src[cnt[0]:]
        self.items.copy_(host, non_blocking=True)
        for d in range(3):
            c = batch.cochains[d]
            n = c.num_cells
            if n > self.cap_cells[d]:
                raise ValueError(f'{n} cells of dimension {d} exceed the capacity {self.cap_cells[d]}')
            self.n_cells[d] = n
            for l in range(len(self.convs)):
                self.x[l][d][:n].copy_(feats[l][d])
            if d < 2:
                e = c.upper_index.size(1)
                if e > self.cap_up[d]:
                    raise ValueError(f'{e} upper entries of dimension {d} exceed the capacity {self.cap_up[d]}')
                self.up_index[d][:, :e].copy_(c.upper_index)          # row 1 stays at offset cap: [2, cap] layout
                self.up_shared[d][:e].copy_(c.shared_coboundaries)
            if d > 0:
                e = c.boundary_index.size(1)
                if e > self.cap_b[d - 1]:
                    raise ValueError(f'{e} boundary entries of dimension {d} exceed the capacity {self.cap_b[d - 1]}')
                self.b_index[d][:, :e].copy_(c.boundary_index)
        self.table.csr_key = None

    # ---- run ----------------------------------------------------------------------------------------
    def _run(self) -> List[List[torch.Tensor]]:
        outs = []
        for l, launch in enumerate(self.launches):
            mode = _ffi.LAYER_CSR_STORE if l == 0 else _ffi.LAYER_CSR_LOAD      # the layers share the indices
            outs.append(launch.run(self.x[l], mode))
        return outs

    def replay(self) -> List[List[torch.Tensor]]:
        """[layer][out_up_0, out_b_0, out_up_1, ...] restricted to the loaded batch's cells.  The first
        call captures the graph; every later call (any batch that fits) replays it."""
        if self._epoch != ops.STATE_EPOCH or any(_ffi.tver(w) != v for w, v in self._weights):
            raise RuntimeError('StaticPropagate: the layer weights changed since this object packed them; build a new one')
        with torch.no_grad():
            if self.graph is None:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._run()                       # warm-up outside the capture
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
                    self.outs = self._run()
            self.graph.replay()
        return [[o[:self.n_cells[i // 2]] for i, o in enumerate(lo)] for lo in self.outs]


# ------------------------------------------------------------------------------------------------------------------------------
# Round 4: whole forwards and whole training steps over batches never seen before
# ------------------------------------------------------------------------------------------------------------------------------
# StaticPropagate above replays the propagate scope for a new batch after the HOST has cut and uploaded its item table.
# The classes below capture a model's full forward / a full optimisation step ONCE over a cwn_amd.static_batch.StaticBatch:
# the collate, the per-batch tables, the item tables and every row count are device-side (static_batch.py), so a step on a
# batch of the reference's shuffled epoch (data/data_loading.py:84-111, exp/train_utils.py:35-75) is a graph replay and
# nothing else -- with StaticBatch.set_epoch not even the batch's complex numbers cross the bus per step.  A StaticBatch of
# S slots puts S consecutive steps behind one replay: the three fill launches and the gap between two replays are paid
# once per S steps.
from .static_batch import StaticBatch            # noqa: E402
from .train import TrainStep                     # noqa: E402


def refuse_unsupported_layers(model: torch.nn.Module, who: str, static: Optional['StaticBatch'] = None, training: bool = False) -> None:
    """A static batch in mode 'blocked' carries, per slot, what SparseCINConv's blocked kernels read (item tables, the collated
    CSR of the boundary adjacencies for the backward) -- not a CSR plan of the UPPER / LOWER adjacencies, which the streaming
    aggregation of other layers (CINppConv, CINConv, OrientedConv) reads: those layers are refused there.  In mode 'csr'
    (round 5) the fill rebuilds those plans on the device (cwn_csr_desc.e_dev) and every layer's streaming path runs inside the
    captured graph -- except, in TRAINING, CINConv / EdgeCINConv: their per-entry BatchNorm takes its statistics with framework
    reductions over the capacity rows of a slot, which would count rows the batch does not have."""
    from .layers import CINConv, CINppConv, EdgeCINConv, OrientedConv
    if static is not None and static.mode == 'csr':
        kinds = (CINConv, EdgeCINConv) if training else ()
        why = 'their per-entry BatchNorm(train) statistics are framework reductions over the capacity rows of a slot'
    else:
        kinds = (CINppConv, CINConv, EdgeCINConv, OrientedConv)
        why = "their aggregation needs a CSR plan of the upper / lower adjacency: build the StaticBatch with mode='csr'"

    def served(m) -> bool:
        # round 6: a CIN++ layer as the reference's molecular models run it (lower stream off, no co-boundary stream) takes the
        # blocked launches of SparseCINConv (third output written by the layer kernel): what a 'blocked' static batch carries
        return (isinstance(m, CINppConv) and (static is None or static.mode != 'csr')
                and all(not lvl.use_down_msg and lvl.update_coboundaries_nn is None for lvl in m.mp_levels)
                and all(lvl._up_kind() == 'cat_linear_relu' for lvl in m.mp_levels))
    bad = sorted({type(m).__name__ for m in model.modules() if kinds and isinstance(m, kinds) and not served(m)})
    if bad:
        raise NotImplementedError(f'{who}: {", ".join(bad)} layers are not served by this static batch ({why}); or use collated '
                                  'batches with model(batch) / TrainStep')


class StaticForward:
    """`model(batch)` (eval, no autograd) for every batch a StaticBatch holds, as one captured graph:
        fill (tables + collate + item tables, all slots) -> per slot: front -> L x (layer launch + update launch) -> head.
    `run(idx)` = set_batch + replay -> predictions of those complexes; after `static.set_epoch(batches)` every `replay()`
    takes the next S batches of the epoch.  The graph holds the packed forms of the model's weights: it is re-captured when
    a parameter (or, through ops.STATE_EPOCH, a raw-pointer writer such as a TrainStep) has changed them."""

    def __init__(self, model: torch.nn.Module, static: StaticBatch):
        refuse_unsupported_layers(model, 'StaticForward', static)
        self.model, self.sb = model, static
        # one captured graph per number of slots it runs (round 6): S for the body of an epoch, a power of two below it for its
        # tail -- an epoch of 65 batches at S = 16 used to end with one batch and FIFTEEN empty slots, each a full sequence
        # of launches over zero rows (~0.1 ms at the molhiv sizes)
        self._graphs = {}                 # n_slots -> (graph, outs, stamp)

    # (the graph / outputs of the full replay: what round 5's callers read)
    graph = property(lambda self: self._graphs.get(self.sb.S, (None, None, None))[0])
    outs = property(lambda self: self._graphs.get(self.sb.S, (None, None, None))[1])

    def _state(self):
        # (the list of the model's tensors is walked out of the module tree once per STRUCT_EPOCH -- torch's registration hooks
        #  move it when any module / parameter / buffer is (re)registered --, their version counters on every call: walking the
        #  tree cost 0.75 ms per replay, more than the replay of an epoch's one-slot tail)
        if getattr(self, '_tensors_epoch', None) != ops.STRUCT_EPOCH:
            self._tensors = list(self.model.parameters()) + list(self.model.buffers())
            self._tensors_epoch = ops.STRUCT_EPOCH
        return (ops.STATE_EPOCH, ops.STRUCT_EPOCH) + tuple(_ffi.tver(t) for t in self._tensors)

    def _run(self, n_slots: Optional[int] = None) -> List[torch.Tensor]:
        n = self.sb.S if n_slots is None else int(n_slots)
        self.sb.fill(n)
        outs = []
        for slot in self.sb.slots[:n]:
            slot.restore()
            with slot.dynamic():
                outs.append(self.model(slot.batch))
            slot.restore()
        return outs

    def eager(self) -> List[torch.Tensor]:
        """The same forwards as ordinary launches (what a replay must reproduce bit for bit)."""
        with torch.no_grad():
            return self._run()

    def slots_for(self, n_batches: int) -> int:
        """The slots the replay that serves `n_batches` remaining batches runs: S, or the power of two that holds a shorter tail."""
        S = self.sb.S
        if n_batches >= S:
            return S
        n = 1
        while n < n_batches:
            n <<= 1
        return min(n, S)

    def replay(self, n_slots: Optional[int] = None) -> List[torch.Tensor]:
        """Per slot the predictions [capacity, out]; rows past a batch's complexes are not meaningful.  n_slots < S: the first
        n_slots slots only, taking the next n_slots batches of the epoch (the tail of an epoch: slots_for)."""
        if self.model.training:
            raise RuntimeError('StaticForward: model.eval() first (training-mode layers have no static inference form)')
        n = self.sb.S if n_slots is None else int(n_slots)
        if not (1 <= n <= self.sb.S):
            raise ValueError(f'1 .. {self.sb.S} slots')
        hit = self._graphs.get(n)
        state = self._state()
        if hit is None or hit[2] != state:
            with torch.no_grad():
                cur = self.sb.cursor.clone()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(2):            # warm-up outside the capture: packed weights, prepared launches, item tables
                        self._run(n)
                        self.sb.cursor.copy_(cur)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
                    outs = self._run(n)
            hit = self._graphs[n] = (graph, outs, self._state())
        hit[0].replay()
        return hit[1]

    def run_epoch(self, batches: Sequence[np.ndarray]) -> List[torch.Tensor]:
        """Predictions of every batch of `batches` (views of the replays' outputs: valid until the next replay of the same
        length overwrites them -- clone what must live longer; the LAST replay's are safe until the next call)."""
        n_full = self.sb.set_epoch(batches)
        res: List[torch.Tensor] = []
        done = 0
        while done < len(batches):
            n = self.slots_for(len(batches) - done)
            outs = self.replay(n)
            for j in range(min(n, len(batches) - done)):
                res.append(outs[j][:len(batches[done + j])].clone())
            done += n
        return res

    def run(self, idx: Sequence[int]) -> torch.Tensor:
        """Predictions of the complexes `idx` (one batch, slot 0; the other slots run empty batches)."""
        self.sb.set_batch(idx)
        return self.replay()[0][:len(idx)]

    def run_many(self, batches: Sequence[Sequence[int]]) -> List[torch.Tensor]:
        self.sb.set_batches(batches)
        outs = self.replay()
        return [outs[j][:len(idx)] for j, idx in enumerate(batches)]


class StaticTrainStep(TrainStep):
    """Optimisation steps (exp/train_utils.py:57-75: zero_grad, forward, loss, backward, Adam) on whatever batches the
    StaticBatch holds next, captured ONCE: `step()` replays S steps, one per slot, for every S batches of an epoch.  A slot
    whose batch is empty (an epoch that is not a multiple of S) leaves the model and the optimizer untouched.  World size 1."""

    def __init__(self, model: torch.nn.Module, static: StaticBatch, task_type: str = 'regression', lr: float = 1e-3,
                 use_graph: bool = True, optimizer=None, share: Optional[TrainStep] = None):
        refuse_unsupported_layers(model, 'StaticTrainStep', static, training=True)
        self.sb = static
        static.build_backward = True                  # (mode 'csr': the fill also builds the transposed plans)
        static.fill()                                 # the buffers hold real batches from here on (warm-up)
        super().__init__(model, [sl.batch for sl in static.slots], task_type=task_type, lr=lr, use_graph=use_graph,
                         optimizer=optimizer, rebuild_plans=False, stages=1, share=share)
        # (world > 1: step() runs the slots of a fill as TrainStep's per-step graph(forward + backward) -> all-reduce -> graph(Adam);
        #  the backward is not cut into pieces here -- stages = 1: one collective per step behind the backward)
        # the inputs ARE the static buffers (TrainStep keeps clones: the collate writes through raw pointers)
        self.inputs = [list(sl.inputs) for sl in static.slots]
        self._actives = [static.tables[j, static.o_sizes + 3: static.o_sizes + 4] for j in range(static.S)]
        self._global_active = torch.ones(1, dtype=torch.int64, device=static.device)

    def _forward_backward(self, i: int, pieces=None):
        if i == 0:
            self.sb.fill(getattr(self, '_n_fill', None))          # (None: all slots; a tail replay: the slots it runs)
        with self.sb.slots[i].dynamic():
            return super()._forward_backward(i, pieces)

    def slots_for(self, n_batches: int) -> int:
        """The slots the replay that serves `n_batches` remaining batches runs: S, or the power of two that holds a shorter tail
        (round 6: an epoch that is not a multiple of S ends with a shorter captured sequence, not with empty slots -- an empty
        slot of a training step is still ~55 launches)."""
        S = self.sb.S
        if n_batches >= S or self.world > 1 or self.staged is not None:
            return S
        n = 1
        while n < n_batches:
            n <<= 1
        return min(n, S)

    def _loss(self, b) -> torch.Tensor:
        """The criterion over the complexes that EXIST: only the fused form knows the device-side count (a framework criterion
        would average over the capacity rows of a short batch) -- anything else is refused, loudly."""
        from .train import fused_loss
        pred = self.model(b)
        if self.task_type == 'classification':
            loss = fused_loss(self.task_type, pred, b.y.view(-1))         # (cross-entropy over the complexes that exist: CWN_LOSS_CE)
        else:
            loss = fused_loss(self.task_type, pred, b.y.view(pred.shape).to(pred.dtype))
        if loss is None:
            raise NotImplementedError('StaticTrainStep: predictions / targets the fused criterion does not take (float32 CUDA '
                                      'tensors of one shape)')
        return loss

    # ---- data parallel: the sample counts live on the device --------------------------------------------------------------------
    def _n_local(self, i: int):
        return self._actives[i] if self.world > 1 else self.batches[i].num_complexes

    def _count_at_begin(self) -> bool:
        return self.world == 1                        # (under data parallelism a step is real iff ANY rank holds a sample)

    def _before_optimizer(self, i: int) -> None:
        if self.world > 1 and hasattr(self.opt, 'active') and self.opt.active is not None:
            # the samples of the GLOBAL batch, summed by the all-reduce (no process group -- a forced data-parallel form on
            # one rank: its own)
            self._global_active.copy_(self.bucket.global_count() if self.bucket._world(None) > 1 else self._actives[i])

    def _eager(self, i: int) -> torch.Tensor:
        if hasattr(self.opt, 'active'):
            # FlatAdam: an empty batch's step changes nothing -- empty on EVERY rank under data parallelism (one that holds no
            # sample itself still takes the step the others take: the parameters stay in step)
            self.opt.active = self._global_active if self.world > 1 else self._actives[i]
        try:
            return super()._eager(i)
        finally:
            if hasattr(self.opt, 'active'):
                self.opt.active = None

    def _optimizer_step(self, i: int) -> None:
        # (also the data-parallel form's captured graph(Adam), which does not come through _eager: without `active` set
        #  while it is recorded, an empty tail slot would take an unconditional Adam step on a zero gradient -- ADVICE r4)
        had = getattr(self.opt, 'active', None)
        if hasattr(self.opt, 'active') and had is None:
            self.opt.active = self._global_active if self.world > 1 else self._actives[i]
        try:
            super()._optimizer_step(i)
        finally:
            if hasattr(self.opt, 'active'):
                self.opt.active = had

    def _capture(self, i: int):
        # the warm-up steps of the capture consume batches: put the cursor back so that the first replay takes the batches
        # the caller expects
        cur = self.sb.cursor.clone()
        res = super()._capture(i)
        self.sb.cursor.copy_(cur)
        return res

    def step(self, i: int = 0, n_slots: Optional[int] = None) -> List[torch.Tensor]:
        """S steps, one per slot, on the next S batches (set_epoch) / the batches of set_batches.  Returns the loss tensors of
        the captured steps (overwritten by the next replay; NaN for an empty batch).  n_slots < S (slots_for): the first n_slots
        slots only, on the next n_slots batches -- the tail of an epoch."""
        S = self.sb.S
        if S == 1:
            return [super().step(0)]
        if self.world > 1 or self.staged is not None:
            # data parallel: no collective is captured (train.TrainStep: graph(forward + backward [pieces]) -> all-reduce ->
            # graph(Adam) per step), so the S steps of a fill are S replays of those graphs; slot 0's first piece holds the fill
            return [TrainStep.step(self, j) for j in range(S)]
        n = S if n_slots is None else int(n_slots)
        if not (1 <= n <= S):
            raise ValueError(f'1 .. {S} slots')
        key = ('seq',) + tuple(range(n))
        if key not in self._graphs:
            cur = self.sb.cursor.clone()
            for j in range(n):                        # warm-up + the one-step graphs TrainStep.steps builds on
                if j not in self._graphs:
                    self._graphs[j] = self._capture(j)
            g = torch.cuda.CUDAGraph()
            self._n_fill = n
            try:
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    losses = [self._eager(j) for j in range(n)]
            finally:
                self._n_fill = None
            self._graphs[key] = (g, losses)
            self.sb.cursor.copy_(cur)
        g, losses = self._graphs[key]
        ops.weights_changed()
        g.replay()
        return losses

    def run_epoch(self, batches: Sequence[np.ndarray], keep_losses: bool = True) -> List[Optional[torch.Tensor]]:
        """One optimisation step per batch of `batches`, S at a time and a shorter replay for the tail; the losses in order
        (clones) or None."""
        self.sb.set_epoch(batches)
        out: List[Optional[torch.Tensor]] = [None] * len(batches)
        k = 0
        while k < len(batches):
            n = self.slots_for(len(batches) - k)
            losses = self.step(n_slots=n)
            if keep_losses:
                for j in range(min(n, len(batches) - k)):
                    out[k + j] = losses[j].clone()
            k += n
        return out

    def step_on(self, batches: Sequence[Sequence[int]]) -> List[torch.Tensor]:
        """Steps on the given batches (<= S index lists; the remaining slots run empty)."""
        self.sb.set_batches(batches)
        return self.step()[:len(batches)]


class StaticRouter:
    """Two static batches over one packed dataset -- mode 'blocked' (the complex-blocked launches: the fast path, a complex
    must fit one workgroup) and mode 'csr' (device-built CSR plans: everything fits) -- and the split of an epoch's batches
    between them: a batch every complex of which fits the blocked tables goes there, the others (ogbg-molhiv draws one of its
    120 - 220-atom molecules into ~one batch in seven at batch 512) to the streaming path.  Both run captured graphs; the
    host's work per epoch is this split (numpy over the per-complex sizes) and two permutation uploads."""

    def __init__(self, packed, batch_size: int, slots: int = 8, caps: Optional[dict] = None, variant: Optional[int] = None):
        self.blocked = StaticBatch(packed, batch_size, caps=caps, variant=variant, slots=slots, mode='blocked')
        self.csr = StaticBatch(packed, batch_size, caps=caps, slots=slots, mode='csr')
        self.S = int(slots)

    def split(self, batches: Sequence[np.ndarray]):
        """(indices of the batches the blocked path takes, indices of the others); raises for a batch neither takes."""
        fit = self.blocked.fits(batches)
        rest = [i for i, f in enumerate(fit) if not f]
        if rest:
            ok = self.csr.fits([batches[i] for i in rest])
            if not ok.all():
                bad = [rest[k] for k, f in enumerate(ok) if not f]
                raise ValueError(f'batches {bad[:8]} exceed the capacities of the static buffers (or hold fewer than two cells of a '
                                 'dimension): build the router with larger `caps`')
        return [i for i, f in enumerate(fit) if f], rest

    def set_epoch(self, batches: Sequence[np.ndarray]):
        a, b = self.split(batches)
        na = self.blocked.set_epoch([batches[i] for i in a]) if a else 0
        nb = self.csr.set_epoch([batches[i] for i in b]) if b else 0
        return a, b, na, nb


class RoutedForward:
    """model(batch) in eval mode for every batch of an epoch through a StaticRouter: run_epoch(batches) -> predictions in the
    order of `batches`.

    regroup (default; round 6): in eval mode a complex's prediction does not depend on the batch it sits in (BatchNorm uses its
    running statistics, every kernel of the path works complex by complex or row by row) -- so ONE molecule beyond a workgroup
    need not send the 511 others of its batch to the streaming path.  The batch keeps the blocked path for the complexes that
    fit; the ones that do not are pooled over the whole epoch into a few small batches of a csr-mode static batch of their own
    (capacities from THEIR sizes), and the predictions are put back in place.  ogbg-molhiv-like sizes at batch 512: a third of
    the batches hold such a molecule; 21 whole batches on the streaming path became one or two pooled ones.  Within the
    gate of what the per-batch launches give (the two paths share the layer kernel's arithmetic bit for bit, not the update
    networks').  Batches that exceed the blocked buffers' CAPACITIES go to the streaming path whole, as before."""

    def __init__(self, model: torch.nn.Module, router: StaticRouter, regroup: bool = True, pool_batch: int = 32):
        self.router = router
        self.fa, self.fb = StaticForward(model, router.blocked), StaticForward(model, router.csr)
        # capture now, over the EMPTY batches the buffers hold: the item tables the model asks for are cut, and
        # StaticBatch.fits() -- the router's test -- knows what one workgroup holds for this model
        self.fa.replay()
        self.fb.replay()
        self.mask = router.blocked.fit_mask()
        self.fbig = None
        big = np.nonzero(~self.mask)[0]
        if regroup and big.size:
            packed = router.blocked.packed
            self.big = StaticBatch(packed, int(min(pool_batch, max(1, big.size))), slots=1, mode='csr', indices=big)
            self.fbig = StaticForward(model, self.big)
            self.fbig.replay()

    def _pool(self, pos: np.ndarray, ids: Optional[np.ndarray] = None) -> List[np.ndarray]:
        """Contiguous ranges of the pooled list, in order, each a batch the pool's buffers hold (its capacities are a statistical
        bound over the complexes that do not fit a workgroup: a range beyond them is halved)."""
        ids = self._pool_ids if ids is None else ids
        todo, out = [pos[k: k + self.big.B] for k in range(0, pos.size, self.big.B)][::-1], []
        while todo:
            c = todo.pop()
            if c.size == 0:
                continue
            if c.size == 1 or bool(self.big.fits([ids[c]])[0]):
                out.append(c)
            else:
                todo += [c[c.size // 2:], c[: c.size // 2]]          # (popped first half first: the order of the list stays)
        return out

    def run_epoch(self, batches: Sequence[np.ndarray]) -> List[torch.Tensor]:
        batches = [np.asarray(b, dtype=np.int64) for b in batches]
        if self.fbig is None:
            return self._run_whole(batches)
        sb = self.router.blocked
        cap_ok, single_ok = sb.fits_detail(batches)
        # per unit of work (batch, positions, complexes): a whole batch on the blocked path, the fitting part of a batch there
        # (a part of a batch within the capacities is within them too), a whole batch on the streaming path; the complexes of
        # the split batches that do not fit: pooled
        a_units, b_units, pooled = [], [], []
        for i, idx in enumerate(batches):
            if cap_ok[i] and single_ok[i]:
                a_units.append((i, None, idx))
                continue
            m = self.mask[idx]
            if not cap_ok[i] or int(m.sum()) < 2:
                b_units.append((i, None, idx))                       # capacity, not the size of a complex (or nothing left)
            else:
                a_units.append((i, np.nonzero(m)[0], idx[m]))
                pooled.append((i, np.nonzero(~m)[0], idx[~m]))
        # every position list of the epoch in ONE upload, before the first replay (an upload per batch would put a host
        # synchronisation between the replays)
        dev = sb.device
        pos_host = [u[1] for u in a_units if u[1] is not None]
        owner = np.concatenate([np.full(q[2].size, q[0], dtype=np.int64) for q in pooled]) if pooled else np.zeros(0, np.int64)
        where = np.concatenate([q[1] for q in pooled]) if pooled else np.zeros(0, np.int64)
        flat = np.concatenate(pos_host + [where]) if (pos_host or pooled) else np.zeros(0, np.int64)
        flat_dev = None
        if flat.size:          # (as StaticBatch._upload: the host -> device copy on a side stream, the compute stream behind its event)
            side = self.__dict__.setdefault('_side', torch.cuda.Stream(device=dev))
            with torch.cuda.stream(side):
                flat_dev = torch.as_tensor(flat, device=dev)
                up = torch.cuda.Event()
                up.record(side)
            torch.cuda.current_stream(dev).wait_event(up)
            flat_dev.record_stream(torch.cuda.current_stream(dev))
        offs, o = {}, 0
        for u in a_units:
            if u[1] is not None:
                offs[u[0]] = (o, o + u[1].size)
                o += u[1].size
        where_dev = flat_dev[o:] if pooled else None
        out: List[Optional[torch.Tensor]] = [None] * len(batches)

        def blank(i, rows):
            if out[i] is None:
                out[i] = torch.empty((len(batches[i]),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
            return out[i]

        # every upload of the epoch (the three permutations) before the first replay: an upload between replays waits for them
        plan = []
        for units, sf, static in ((a_units, self.fa, sb), (b_units, self.fb, self.router.csr)):
            if units:
                plan.append((units, sf, static, static.set_epoch([u[2] for u in units])))
        chunks, n_big = [], 0
        if pooled:
            ids = np.concatenate([q[2] for q in pooled])
            chunks = self._pool(np.arange(ids.size), ids)            # (contiguous ranges of the pooled list)
            n_big = self.big.set_epoch([ids[c] for c in chunks])
        for units, sf, static, n_rep in plan:
            k = 0
            while k < len(units):
                n = sf.slots_for(len(units) - k)                     # (S, or a shorter replay for the tail of the epoch)
                outs = sf.replay(n)
                for j in range(min(n, len(units) - k)):
                    i, pos, idx = units[k + j]
                    rows = outs[j][:len(idx)]
                    if pos is None:
                        out[i] = rows.clone()
                    else:
                        lo, hi = offs[i]
                        blank(i, rows).index_copy_(0, flat_dev[lo:hi], rows)
                k += n
        for r in range(n_big):
            rows = self.fbig.replay()[0]
            c0, c1 = int(chunks[r][0]), int(chunks[r][-1]) + 1
            # (the pooled list is ordered by batch: one indexed copy per batch the chunk touches)
            s_ = c0
            while s_ < c1:
                e_ = s_
                while e_ < c1 and owner[e_] == owner[s_]:
                    e_ += 1
                blank(int(owner[s_]), rows).index_copy_(0, where_dev[s_:e_], rows[s_ - c0: e_ - c0])
                s_ = e_
        return out

    def _run_whole(self, batches: Sequence[np.ndarray]) -> List[torch.Tensor]:
        a, b, na, nb = self.router.set_epoch(batches)
        out: List[Optional[torch.Tensor]] = [None] * len(batches)
        for order, sf in ((a, self.fa), (b, self.fb)):
            k = 0
            while k < len(order):
                n = sf.slots_for(len(order) - k)
                outs = sf.replay(n)
                for j in range(min(n, len(order) - k)):
                    out[order[k + j]] = outs[j][:len(batches[order[k + j]])].clone()
                k += n
        return out


class RoutedTrainStep:
    """The optimisation steps of an epoch (exp/train_utils.py:35-75) through a StaticRouter: one StaticTrainStep per static
    batch over ONE model, one flat gradient and one Adam state (TrainStep(share=)).  run_epoch(batches) -> the losses in the
    order of `batches`.  The ORDER of the steps inside an epoch is the router's (the blocked path's batches S at a time,
    then the streaming path's), not the loader's: every batch is visited exactly once, as in the reference's shuffled epoch."""

    def __init__(self, model: torch.nn.Module, router: StaticRouter, task_type: str = 'regression', lr: float = 1e-3):
        self.router = router
        self.ta = StaticTrainStep(model, router.blocked, task_type=task_type, lr=lr)
        self.tb = StaticTrainStep(model, router.csr, task_type=task_type, lr=lr, share=self.ta)
        self.opt = self.ta.opt
        # capture now, over the EMPTY batches the buffers hold (an empty batch's step changes nothing): the forward and
        # backward item tables are cut, and StaticBatch.fits() -- the router's test -- knows what one workgroup holds
        self.ta.step()
        self.tb.step()

    def run_epoch(self, batches: Sequence[np.ndarray], keep_losses: bool = True) -> List[Optional[torch.Tensor]]:
        a, b, na, nb = self.router.set_epoch(batches)
        out: List[Optional[torch.Tensor]] = [None] * len(batches)
        for order, ts in ((a, self.ta), (b, self.tb)):
            k = 0
            while k < len(order):
                n = ts.slots_for(len(order) - k)                 # (S, or a shorter captured sequence for the tail of the epoch)
                losses = ts.step(n_slots=n)
                if keep_losses:
                    for j in range(min(n, len(order) - k)):
                        out[order[k + j]] = losses[j].clone()
                k += n
        return out
