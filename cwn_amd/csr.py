"""Adjacency plans: the per-batch int32 CSR structures every aggregation kernel reads.

A COO index of the reference (`upper_index`, `lower_index`, `boundary_index`: int64 `[2, E]`,
row 0 = source cell j, row 1 = destination cell i under flow source_to_target,
mp/cell_mp.py:210) is converted ONCE per batch into a destination-sorted CSR
(`rowptr`, `col`, `perm` [, `aux`]) by the HIP kernels of csrc/cwn_csr.hip and then reused by all
layers and by the backward pass (which needs the transposed structures, built lazily and in one
batched call as well).
"""
import weakref
from typing import List, Optional, Sequence

import threading

import torch

from . import _ffi

LONG_ROW = 64             # = CWN_LONG_ROW (include/cwn_hip.h)
LONG_PARTS = 8            # = CWN_LONG_PARTS
VALIDATE_INDICES = True   # one host sync per batched build; turned off inside stream capture


class Adjacency:
    """Destination-sorted CSR of one COO index.

    rowptr[i]:rowptr[i+1]  CSR positions whose destination is i (stable in entry order)
    col[p]                 source row of position p            (= index[0][perm[p]])
    perm[p]                original entry id of position p
    aux[p]                 second gather index of position p   (= aux_index[perm[p]]), optional
    """

    def __init__(self, key: torch.Tensor, val: torch.Tensor, n_dst: int, n_val: int,
                 aux_index: Optional[torch.Tensor] = None, n_aux: int = 0):
        _ffi.require_gpu(key, 'index')
        assert key.dtype == torch.long and val.dtype == torch.long
        assert key.dim() == 1 and val.shape == key.shape
        self.key, self.val, self.aux_index = key.contiguous(), val.contiguous(), aux_index
        if aux_index is not None:
            assert aux_index.dtype == torch.long and aux_index.shape == key.shape
            self.aux_index = aux_index.contiguous()
        self.n_entries = int(key.numel())
        self.n_dst, self.n_val, self.n_aux = int(n_dst), int(n_val), int(n_aux)
        dev = key.device
        self.device = dev
        self.rowptr = torch.empty(self.n_dst + 1, dtype=torch.int32, device=dev)
        self.col = torch.empty(self.n_entries, dtype=torch.int32, device=dev)
        self.perm = torch.empty(self.n_entries, dtype=torch.int32, device=dev)
        self.aux = (torch.empty(self.n_entries, dtype=torch.int32, device=dev)
                    if aux_index is not None else None)
        # rows with more than LONG_ROW entries (hubs), listed by the build for the aggregation
        # kernel's whole-workgroup path; no row can be that long when E <= LONG_ROW
        self.long_cap = self.n_entries // LONG_ROW + 1
        self.long_rows = self.n_long = None
        if self.n_entries > LONG_ROW:       # LONG_PARTS sub-lists + their lengths in one buffer
            buf = torch.empty(LONG_PARTS * (self.long_cap + 1), dtype=torch.int32, device=dev)
            self.n_long, self.long_rows = buf[:LONG_PARTS], buf[LONG_PARTS:]
        self.built = False
        self.ready = None     # torch.cuda.Event when the plan was built on a side stream
        self._t_src: Optional['Adjacency'] = None
        self._t_aux: Optional['Adjacency'] = None
        self._counts: Optional[torch.Tensor] = None

    # ---- construction ------------------------------------------------------------------
    @classmethod
    def from_index(cls, index: torch.Tensor, n_dst: int, n_src: int,
                   aux_index: Optional[torch.Tensor] = None, n_aux: int = 0,
                   build: bool = True) -> 'Adjacency':
        """`index` is a reference-style `[2, E]` LongTensor (mp/cell_mp.py:158-160)."""
        # NB: no strong reference to `index` itself is kept (only row views), so the plan cache's
        # weakref on the caller's tensor can evict the plan when the batch dies
        adj = cls(index[1], index[0], n_dst, n_src, aux_index, n_aux)
        if build:
            build_many([adj])
        return adj

    def _desc(self) -> _ffi.CsrDesc:
        return _ffi.CsrDesc(
            key=self.key.data_ptr(), val=self.val.data_ptr(), aux=_ffi.ptr(self.aux_index),
            n_entries=self.n_entries, n_dst=self.n_dst, n_val=self.n_val, n_aux=self.n_aux,
            rowptr=self.rowptr.data_ptr(), col=self.col.data_ptr(), perm=self.perm.data_ptr(),
            aux_out=_ffi.ptr(self.aux), long_rows=_ffi.ptr(self.long_rows),
            n_long=_ffi.ptr(self.n_long),
            # a static buffer (cwn_amd/static_batch.py, mode 'csr'): n_entries is its capacity, the batch's own count is in
            # device memory at the address the static batch TAGGED this plan with -- never looked up by value (an unrelated
            # plan whose entry count happens to equal a mapped capacity must not pick up another tensor's live count)
            e_dev=getattr(self, 'e_dev_ptr', None))

    # ---- transposes for the backward pass ----------------------------------------------
    def transposes(self) -> List['Adjacency']:
        """The not-yet-built transposed structures this adjacency's backward needs."""
        todo = []
        if self._t_src is None:
            # keyed on the SOURCE cell: col = destination, aux = aux of the same entry
            self._t_src = Adjacency(self.val, self.key, self.n_val, self.n_dst, self.aux_index,
                                    self.n_aux)
            todo.append(self._t_src)
        if self.aux_index is not None and self._t_aux is None:
            # keyed on the AUX cell (shared coboundary / boundary): col = destination, aux = source
            self._t_aux = Adjacency(self.aux_index, self.key, self.n_aux, self.n_dst, self.val,
                                    self.n_val)
            todo.append(self._t_aux)
        return todo

    def _ensure_transposes(self) -> None:
        self.transposes()
        build_many([a for a in (self._t_src, self._t_aux) if a is not None])

    @property
    def t_src(self) -> 'Adjacency':
        if self._t_src is None or not self._t_src.built:
            self._ensure_transposes()
        return self._t_src

    @property
    def t_aux(self) -> 'Adjacency':
        assert self.aux_index is not None
        if self._t_aux is None or not self._t_aux.built:
            self._ensure_transposes()
        return self._t_aux

    def long_row_list(self) -> torch.Tensor:
        """Rows with more than LONG_ROW entries (host sync; for tests and diagnostics)."""
        if self.long_rows is None:
            return torch.empty(0, dtype=torch.long, device=self.device)
        n = self.n_long.tolist()
        lists = self.long_rows.view(LONG_PARTS, self.long_cap)
        return torch.cat([lists[p, :n[p]] for p in range(LONG_PARTS)]).long()

    @property
    def counts(self) -> torch.Tensor:
        """Entries per destination row, float32 [n_dst, 1], clamped to >= 1 (mean reduce)."""
        if self._counts is None:
            self._counts = (self.rowptr[1:] - self.rowptr[:-1]).clamp(min=1).to(torch.float32).unsqueeze(1)
        return self._counts


_err_flags = {}


def _err_flag(dev) -> torch.Tensor:
    """One sticky int32 error word per device (zeroed once, and again after it has been reported):
    no per-call fill kernel in front of every plan build."""
    key = (dev.type, dev.index)
    f = _err_flags.get(key)
    if f is None:
        f = torch.zeros(1, dtype=torch.int32, device=dev)
        _err_flags[key] = f
    return f


_side_streams = {}


def side_stream(dev) -> torch.cuda.Stream:
    key = (dev.type, dev.index)
    s = _side_streams.get(key)
    if s is None:
        s = torch.cuda.Stream(device=dev)
        _side_streams[key] = s
    return s


def wait_ready(adjs) -> None:
    """Make the current stream wait for plans that were built on the side stream (no-op otherwise)."""
    for a in adjs:
        if a is not None and a.ready is not None:
            torch.cuda.current_stream(a.device).wait_event(a.ready)
            a.ready = None


def build_many(adjs: Sequence[Adjacency], overlap: bool = False, validate: bool = True, force: bool = False) -> None:
    """Build any number of adjacencies with batched C-ABI calls (<= CSR_MAX_DESCS per call; each call is
    one launch for small inputs, a fixed sequence of 5-7 launches otherwise, whatever the number of
    index tensors).  `overlap=True` enqueues the build on a side stream and tags every plan with
    an event that the first aggregation using it waits on, so the integer work runs underneath
    whatever dense work the caller issues next (the layer-0 message GEMMs).  `force`: plans that are built already are
    built AGAIN (a static batch's buffers hold another batch now: cwn_amd/static_batch.py, mode 'csr')."""
    adjs = [a for a in adjs if force or not a.built]
    if not adjs:
        return
    if force:
        for a in adjs:
            a._counts = None          # (row counts of the batch the buffers held before: reduce='mean' divides by them)
    dev = adjs[0].device
    if overlap:
        main, side = torch.cuda.current_stream(dev), side_stream(dev)
        side.wait_stream(main)          # the index tensors were produced on the main stream
        with torch.cuda.stream(side):
            build_many(adjs, overlap=False, validate=False)   # no host sync; see check_errors()
            ev = torch.cuda.Event()
            ev.record(side)
        for a in adjs:
            a.ready = ev
            for t in (a.rowptr, a.col, a.perm, a.aux, a.n_long):
                if t is not None:
                    t.record_stream(main)
        return
    L = _ffi.lib()
    capturing = torch.cuda.is_current_stream_capturing()
    err = _err_flag(dev)
    for i in range(0, len(adjs), _ffi.CSR_MAX_DESCS):
        chunk = adjs[i:i + _ffi.CSR_MAX_DESCS]
        arr = (_ffi.CsrDesc * len(chunk))(*[a._desc() for a in chunk])
        nbytes = L.cwn_csr_workspace_bytes(arr, len(chunk))
        ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
        _ffi.check(L.cwn_csr_build(arr, len(chunk), ws.data_ptr(), nbytes, err.data_ptr(),
                                   _ffi.stream_ptr(dev)), 'cwn_csr_build')
        for a in chunk:
            a.built = True
    if VALIDATE_INDICES and validate and not capturing:
        check_errors(dev)


def _describe(flag: int) -> str:
    what = [n for b, n in ((1, 'destination index'), (2, 'source index'), (4, 'shared (co)boundary index'),
                           (8, 'an index outside its complex (batch not block-diagonal, or a stale item table)'),
                           (16, 'a complex beyond what one workgroup holds reached the device-side item-table build '
                                '(static_batch.StaticBatch.fits() tells which batches a static batch takes)'),
                           (32, 'a batch beyond the capacity of its static buffers was dropped (its slot ran as an empty batch; '
                                'static_batch.StaticBatch.fits() / larger `caps`)'))
            if flag & b]
    return 'index out of range in adjacency: ' + ', '.join(what)


class _Deferred(threading.local):       # per thread: nesting depth of deferred_checks, devices with a check pending
    def __init__(self):
        self.depth, self.pending = 0, []


_deferred = _Deferred()


class deferred_checks:
    """Inside this block `check_errors` only notes the device; the block's end reads the sticky error word ONCE (round 5: a
    model's forward wraps itself in one -- the embedding front's range check was a device sync between the first launch of
    a forward and all the others, 0.9 ms per eager forward at the ZINC batch against 0.11 ms of kernels; the kernels skip
    what is out of range, so running on is safe).  The IndexError is raised by the same call as before -- the model's
    forward -- after its launches are enqueued; nothing is read when the block is left by an exception."""

    def __enter__(self):
        _deferred.depth += 1
        return self

    def __exit__(self, exc_type, exc, tb):
        _deferred.depth -= 1
        if _deferred.depth == 0:
            pending, _deferred.pending = _deferred.pending, []
            if exc_type is None:
                for dev in pending:
                    check_errors(dev)
        return False


def check_errors(dev) -> None:
    """Raise the IndexError of any out-of-range index seen by plan builds that skipped the host
    sync (overlap mode / stream capture).  One device sync."""
    dev = torch.device(dev)
    if _deferred.depth > 0:
        if dev not in _deferred.pending:
            _deferred.pending.append(dev)
        return
    err = _err_flag(dev)
    flag = int(err.item())
    if flag:
        global ERROR_EPOCH
        ERROR_EPOCH += 1
        err.zero_()
        raise IndexError(_describe(flag))


ERROR_EPOCH = 0      # moves with every raised IndexError: "these indices were checked" marks taken before it are void


# ---- cache keyed on the identity of the reference-style index tensor ---------------------------
_cache = {}


_defer = threading.local()


class deferred_builds:
    """Inside this context `cached_adjacency` hands out plans WITHOUT building them (round 6): the caller knows that the launches it
    is about to describe may never read them -- the training forward through the complex-blocked launch, whose backward is the
    owner-form launch over its own item table (cwn_amd/layers.py propagate_all) -- and every reader of a plan builds it on
    demand (`ensure_built`: ops.run_aggregate, the streaming backward).  Two CSR builds per training step at ZINC-128."""

    def __enter__(self):
        self.prev = getattr(_defer, 'on', False)
        _defer.on = True
        return self

    def __exit__(self, *exc):
        _defer.on = self.prev
        return False


def ensure_built(adj: Optional['Adjacency']) -> Optional['Adjacency']:
    """`adj`, built (a plan handed out under `deferred_builds` and wanted after all)."""
    if adj is not None and not adj.built:
        build_many([adj])
    return adj


def cached_adjacency(index: torch.Tensor, n_dst: int, n_src: int,
                     aux_index: Optional[torch.Tensor] = None, n_aux: int = 0,
                     build: bool = True) -> Adjacency:
    """propagate() is called with the same index tensors by every layer (mp/molec_models.py:110);
    convert each one once.  Keyed on tensor identity + version counter, evicted when the tensor
    dies.  A plan that carries a shared-cell (aux) index also serves requests without one."""
    build = build and not getattr(_defer, 'on', False)
    key = id(index)
    hit = _cache.get(key)
    ver = (_ffi.tver(index), n_dst, n_src)
    if hit is not None and hit[0] == ver and hit[1]() is index:
        adj = hit[2]
        if aux_index is None or (adj.aux_index is not None and adj.n_aux == n_aux and
                                 (adj.aux_index is aux_index or
                                  adj.aux_index.data_ptr() == aux_index.data_ptr())):
            if build and not adj.built:
                build_many([adj])
            return adj
    adj = Adjacency.from_index(index, n_dst, n_src, aux_index, n_aux, build=build)
    _cache[key] = (ver, weakref.ref(index, lambda _r, k=key: _cache.pop(k, None)), adj)
    return adj
