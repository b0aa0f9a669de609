"""A batch of FIXED CAPACITY that a captured hipGraph refills on the device: the reference's shuffled epoch
(data/data_loading.py:84-111 -> exp/train_utils.py:35-75: every step sees a batch it has never seen) on the graph path.

A hipGraph bakes pointers, grid sizes and kernel arguments in, and round 3's fast paths therefore needed the batch known
in advance (one graph per distinct batch).  Here everything a batch changes lives in DEVICE memory:

  * the arrays of the batch are capacity-sized buffers (features `[cap_d, width]`, indices `[2, cap]` with the second row
    at a fixed offset, `[cap]` vectors), filled by the collate launch from the HBM-resident packed dataset;
  * the segment tables of the collate -- and with them `ptr` / `__slices__`, the number of cells per dimension and the
    number of complexes -- are computed ON THE DEVICE from the dataset's per-complex metadata and the batch's complex
    numbers (cwn_collate_tables); with the `cursor` the complex numbers are read from a permutation uploaded once per epoch,
    so that a step needs NOTHING from the host but the replay;
  * the item tables of the complex-blocked launches are cut on the device from those tables (cwn_layer_items_build_dev,
    cwn_layer_bwd_items_build_dev) into fixed regions, empty records behind the batch's own;
  * the CSR plans the model's front needs (boundary adjacencies and their transposes) are the concatenation of
    per-complex CSRs kept in the packed dataset: part of the collate launch, no cwn_csr_build;
  * every row-count-carrying kernel argument of the dense / norm / weight-gradient / readout launches becomes a
    capacity, the actual count being read from the tables (`m_dev`, include/cwn_hip.h "device-side row counts").

`StaticBatch.batch` is a ComplexBatch over those buffers: the model code runs on it unchanged (it sees the capacities as
its sizes) inside `StaticBatch.dynamic()`, which maps each capacity to the device address of the actual count.
cwn_amd/static_graph.py captures a forward / a training step over it once.
"""
from typing import Dict, List, Optional, Sequence

import ctypes as C
import os
import numpy as np
import torch

from . import _ffi, csr
from .blockplan import BWD_LDS_BYTES, BwdItemTable, ITEM_INTS, ItemTable, LDS_BYTES, gemm_rows_cap, lds_bytes
from .complex import CochainBatch, ComplexBatch
from .packed import _CSR_KEYS, PackedComplexes

_TWO_ROW = ('upper_index', 'lower_index', 'boundary_index')
_ADD_ROW = dict(PackedComplexes._ADD_ROW, b_col=2, bt_col=0)


class _NoArray:
    """Stands where a CSR array of an upper adjacency would be: present (the layer code only asks `is None`), unusable."""

    def __getattr__(self, name):
        raise _ffi.CwnError('a static batch has no CSR of its upper adjacencies: the complex-blocked launches are the only path '
                            '(cwn_amd/static_batch.py); use PackedComplexes.collate for this model / configuration')


class _PlanOnlyAdjacency:
    """What the layer code receives for an UPPER adjacency of a static batch: sizes only.  The blocked launches read the
    int64 index tensors through the item tables; a code path that wants this adjacency's CSR (the streaming backward, a
    generic message hook) is not available on a static batch and fails here, loudly."""
    built = True
    ready = None
    long_rows = n_long = None
    long_cap = 1

    def __init__(self, index: torch.Tensor, n_dst: int, n_val: int, aux_index: Optional[torch.Tensor], n_aux: int):
        self.key, self.val, self.aux_index = index[1], index[0], aux_index
        self.n_entries = int(index.size(1))
        self.n_dst, self.n_val, self.n_aux = int(n_dst), int(n_val), int(n_aux)
        self.device = index.device
        self._t_src = self._t_aux = None

    def _no(self, what):
        raise _ffi.CwnError(f'a static batch has no CSR of its upper adjacencies ({what}): the complex-blocked launches are the '
                            'only path (cwn_amd/static_batch.py); use PackedComplexes.collate for this model / configuration')

    rowptr = property(lambda self: self._no('rowptr'))
    col = property(lambda self: self._no('col'))
    perm = property(lambda self: self._no('perm'))
    aux = _NoArray()                  # (`Stream.validate` asks whether the plan was built with a shared-cell index: it was)
    t_src = property(lambda self: self._no('t_src'))
    t_aux = property(lambda self: self._no('t_aux'))

    def transposes(self):
        return []


class _CollatedAdjacency(csr.Adjacency):
    """The destination-sorted CSR of a boundary adjacency of a static batch: arrays written by the collate launch (the
    concatenation of the complexes' own CSRs), not by cwn_csr_build.  No `perm` (nothing on these paths reads one)."""

    def __init__(self, index: torch.Tensor, key_row: int, n_dst: int, n_val: int, rowptr: torch.Tensor, col: torch.Tensor):
        self.key, self.val, self.aux_index = index[key_row], index[1 - key_row], None
        self.n_entries = int(index.size(1))
        self.n_dst, self.n_val, self.n_aux = int(n_dst), int(n_val), 0
        self.device = index.device
        self.rowptr, self.col, self.perm, self.aux = rowptr, col, None, None
        self.long_cap = self.n_entries // csr.LONG_ROW + 1
        self.long_rows = self.n_long = None          # (a cell's boundary is a handful of entries: no long rows)
        self.built, self.ready = True, None
        self._t_src = self._t_aux = None
        self._counts = None

    def transposes(self):
        return []


class StaticBlockPlan:
    """The BlockPlan (cwn_amd/blockplan.py) of one slot of a static batch: item tables in fixed device buffers, rebuilt by the
    device builders after every fill.  Duck-types what the layer / model code asks a BlockPlan."""

    def __init__(self, owner: 'StaticBatch', slot: int):
        self.owner, self.slot = owner, int(slot)
        self.n_dims = owner.D
        self.C = owner.B
        self.device = owner.device
        self.cell_ptr = [np.array([0, owner.cap_cells[d]], dtype=np.int64) for d in range(owner.D)]
        self.up_ptr = [(np.array([0, owner.cap_key(d, 'upper_index')]) if owner.k_of(d, 'upper_index') >= 0 else None)
                       for d in range(owner.D)]
        self.b_ptr = [(np.array([0, owner.cap_key(d, 'boundary_index')]) if owner.k_of(d, 'boundary_index') >= 0 else None)
                      for d in range(owner.D)]
        self.validated = True           # (index VALUES are checked by every blocked launch; the sticky word is read by the caller)

    @property
    def variant(self) -> int:
        return self.owner.variant

    @property
    def group(self) -> int:
        return self.owner.group

    # ---- what the model code asks -------------------------------------------------------------------------------
    def cell_ptr_device(self, d: int, device) -> torch.Tensor:
        return self.owner.seg_view(d, self.slot)

    def forget_csr(self) -> None:
        for fam in self.owner._families.values():
            t = fam[self.slot]
            if hasattr(t, 'csr_key'):
                t.csr_key = None

    def at_least(self, F: int, has_up, has_b=None) -> int:
        return self.owner._n_sets(has_up) * -(-self.C // self.group)

    def items_mixed(self, F, has_up, has_b=None):
        return None

    def _norm_key(self, has_up, has_b):
        if has_b is None:
            has_b = [p is not None for p in self.b_ptr]
        return (tuple(bool(h) for h in has_up),
                tuple(bool(h) and self.b_ptr[d] is not None for d, h in enumerate(has_b)))

    def items(self, F: int, has_up, has_b=None, variant: int = 0, allow_big: bool = False) -> Optional[ItemTable]:
        if allow_big:
            return None                        # no BIG records
        # (one form per static batch, chosen when it was built: whatever form the caller's heuristics ask for gets this table)
        hu, hb = self._norm_key(has_up, has_b)
        fam = self.owner.family('fwd', int(F), hu, hb)
        return None if fam is None else fam[self.slot]

    def bwd_items(self, F: int, has_up, has_b=None) -> Optional[BwdItemTable]:
        hu, hb = self._norm_key(has_up, has_b)
        fam = self.owner.family('bwd', int(F), hu, hb)
        return None if fam is None else fam[self.slot]


class StaticComplexBatch(ComplexBatch):
    """The ComplexBatch over one slot of a StaticBatch: its plans are not built per batch (the collate launch writes them)."""

    def prepare(self, *args, **kwargs):
        return self

    def forget_plans(self):
        return self


class StaticSlot:
    """One batch of a StaticBatch: the ComplexBatch the model code runs on, its plan, its views of the buffers."""

    def __init__(self, owner: 'StaticBatch', j: int, batch: StaticComplexBatch, bufs: Dict, plan: StaticBlockPlan):
        self.owner, self.j, self.batch, self.bufs, self.plan = owner, j, batch, bufs, plan
        self.inputs = [bufs.get((d, 'x')) for d in range(owner.D)]

    def restore(self) -> None:
        """The raw input features back into the container (the models overwrite them layer by layer: set_xs)."""
        for d in range(self.owner.D):
            self.batch.cochains[d]._x = self.inputs[d]

    def size_ptr(self, k: int) -> int:
        return self.owner.size_ptr(k, self.j)

    def sizes(self) -> List[int]:
        """[cells of dims 0..2, complexes] of the batch in this slot (host sync: tests, diagnostics)."""
        o = self.owner
        return o.tables[self.j, o.o_sizes: o.o_sizes + 4].tolist()

    def dynamic(self) -> _ffi.dynamic_rows:
        """Context manager: inside, every launch whose row count is one of the capacities reads this slot's actual count
        from its tables (the three cell counts are consecutive int64: cwn_embed_front_f32's n_dev)."""
        o = self.owner
        m = {o.cap_cells[d]: self.size_ptr(d) for d in range(min(o.D, 3))}
        m[o.B] = self.size_ptr(3)
        # entry counts of the index keys (cwn_csr_desc.e_dev of a plan built over this slot's buffers)
        for k, (d, key, pk) in enumerate(o.packed._klist):
            if key in _TWO_ROW:
                m[o._caps[k]] = self.size_ptr(8 + k)
        return _ffi.dynamic_rows(m)


class StaticBatch:
    """Capacity-sized device buffers for `slots` batches of `batch_size` complexes of `packed` (a PackedComplexes built with
    with_csr=True), and the launches that fill them (`fill`): the batch's tables, its arrays, its item tables -- for ALL
    slots at once, so that a graph holding S steps pays for three small launches per S steps, not per step.

    caps: {'cells': [per dimension], (d, key): elements} overrides; by default every capacity is what a batch of the
    dataset's sizes needs with a wide margin (mean x B + 6 sigma sqrt(B), at most the sum of the B largest), and
    `fits(batches)` tells the caller which batches the buffers hold."""

    def __init__(self, packed: PackedComplexes, batch_size: int, caps: Optional[dict] = None, variant: Optional[int] = None,
                 group: Optional[int] = None, indices: Optional[Sequence[int]] = None, slots: int = 1, mode: str = 'blocked'):
        """mode 'blocked' (default): the complex-blocked launches of SparseCINConv are the only path -- item tables cut on the
        device, no CSR of the upper adjacencies; a complex beyond one workgroup does not fit.  mode 'csr' (round 5): every
        adjacency of a slot gets a REAL destination-sorted CSR plan, rebuilt by the fill from the slot's int64 entries with
        the entry count read from the tables (cwn_csr_desc.e_dev) -- the streaming path (grouped GEMM + cwn_aggregate_f32)
        then runs inside the captured graph: hub complexes (REDDIT-like clique lifts), CINppConv / OrientedConv layers,
        molecules beyond a workgroup.  No item tables in this mode (the layers take their CSR path)."""
        if mode not in ('blocked', 'csr'):
            raise ValueError("mode 'blocked' or 'csr'")
        self.mode = mode
        self.build_backward = False          # (StaticTrainStep: the fill also builds the transposed plans)
        self._slot_long: Dict = {}           # slot -> the collated plans whose long-row lists the fill writes (mode 'csr')
        self._cap_cols = None                # (_check_capacity: the distinct size columns a batch can exceed)
        if not packed.with_csr:
            raise ValueError('StaticBatch needs a PackedComplexes built with with_csr=True')
        if packed.device.type != 'cuda':
            raise _ffi.CwnError('StaticBatch needs the packed dataset on the GPU')
        if variant not in (None, 0, 1):
            raise ValueError('variant 0 (one 16-wave workgroup per CU), 1 (the two-per-CU form) or None (chosen when the first '
                             'item table is cut: _pick_variant)')
        self.packed, self.B, self.S = packed, int(batch_size), int(slots)
        if self.S < 1 or self.S > 64:
            raise ValueError('1 .. 64 slots')
        self.device = packed.device
        self.D = packed.max_dim + 1
        self.K = len(packed._klist)
        # the form of the complex-blocked launches and the complexes per item of the device-side cut.  Measured (round 5,
        # tools/sweep_static_group.py; propagate scope over never-seen batches, share of the fixed-batch replay whose table the
        # host builder packs): ONE complex per item in every case (ZINC-512: two-per-CU 0.98 / 0.94 at 1 / 2 complexes per
        # item, 16-wave 0.78 / 0.80 at 2 / 4; molhiv-512: 0.81 / 0.79 / 0.75 at 1 / 2 / 3 against 0.65 / 0.70 / 0.70 / 0.62 at
        # 1 / 2 / 3 / 4), and the two-per-CU form as soon as a batch has more items than the chip has CUs (ZINC-256: 1.01
        # against 0.94; ZINC-2048: 0.90; molhiv-2048: 0.63) -- until round 5 the default was the 16-wave form with B / 128
        # complexes per item (molhiv-512: 0.62).
        self._variant_arg = None if variant is None else int(variant)
        self.variant = 0 if variant is None else int(variant)       # (None: settled by the first forward table, _pick_variant)
        self._variant_settled = variant is not None
        self.group = int(group) if group is not None else 1
        B, D, K, S = self.B, self.D, self.K, self.S
        dev = self.device
        meta = packed._meta if indices is None else packed._meta[np.asarray(indices, dtype=np.int64)]
        # ---- capacities ---------------------------------------------------------------------------------------------
        def cap_of(col: np.ndarray) -> int:
            col = np.asarray(col, dtype=np.float64)
            top = float(np.sort(col)[-min(B, col.size):].sum()) + (B - min(B, col.size)) * float(col.max(initial=0))
            stat = B * float(col.mean()) + 6.0 * float(col.std()) * np.sqrt(B) + 8
            return int(max(1, np.ceil(min(top, stat))))
        caps = dict(caps or {})
        self.cap_cells = list(caps.get('cells', [cap_of(meta[:, 3 * d]) for d in range(D)]))
        # pairwise distinct, and distinct from the number of complexes: a row count then identifies what it counts
        # (_ffi.dynamic_rows)
        used = {B}
        for d in range(D):
            while self.cap_cells[d] in used:
                self.cap_cells[d] += 1
            used.add(self.cap_cells[d])
        self._caps: List[int] = []
        for k, (d, key, pk) in enumerate(packed._klist):
            if key == 'x':
                c = self.cap_cells[d] * pk.width
            elif key == 'b_rowptr':
                c = self.cap_cells[d]
            elif key == 'bt_rowptr':
                c = self.cap_cells[d - 1]
            elif d < 0:
                c = B * max(1, int(np.max(meta[:, 3 * D + k], initial=1)))
            else:
                c = int(caps.get((d, key), cap_of(meta[:, 3 * D + k])))
            self._caps.append(int(c))
        # the entry capacities of the index keys: pairwise distinct and distinct from the cell capacities -- an Adjacency's
        # capacity then identifies the size word its actual count lives in (dynamic(): _ffi.dynamic_rows)
        for k, (d, key, pk) in enumerate(packed._klist):
            if key in _TWO_ROW:
                while self._caps[k] in used:
                    self._caps[k] += 1
                used.add(self._caps[k])
        # (an index and its shared-cell vector have one entry each per adjacency entry: one capacity)
        for d in range(D):
            for key, aux in (('upper_index', 'shared_coboundaries'), ('lower_index', 'shared_boundaries')):
                ki, ka = self.k_of(d, key), self.k_of(d, aux)
                if ki >= 0 and ka >= 0:
                    self._caps[ka] = self._caps[ki]
        for d in range(1, D):            # the CSR columns cover the boundary entries one to one
            kb = self.k_of(d, 'boundary_index')
            if kb >= 0:
                for name in ('b_col', 'bt_col'):
                    self._caps[self.k_of(d, name)] = self._caps[kb]
        # ---- tables -------------------------------------------------------------------------------------------------
        L = _ffi.lib()
        self.n_tab = int(L.cwn_collate_tables_len(D, K, B))
        self.tables = torch.zeros(S, self.n_tab, dtype=torch.int64, device=dev)
        self.o_src = K * (B + 1)
        self.o_off = self.o_src + K * B
        self.o_seg = self.o_off + D * 5 * B
        self.o_sizes = self.o_seg + D * (B + 1)
        self.meta = packed.meta_device()
        # what cwn_collate_guard compares a slot's scanned totals with: a batch beyond them runs as an EMPTY batch
        self.caps_dev = torch.tensor(list(self.cap_cells) + list(self._caps), dtype=torch.int64, device=dev)
        # the batch numbers the fills read: room for an epoch over the whole dataset (+ a replay's worth of spare slots) from the
        # start -- a captured fill holds this buffer's address and length, so it must not move once a step has been captured
        # (reserve_epoch asks for more BEFORE the first fill).  Fill number j takes the batches at the device cursor, which it
        # advances by S: set_batches / set_epoch upload and rewind.
        self.n_batches = -(-(max(S, -(-packed.num // B)) + S) // S) * S
        self.idx = torch.full((self.n_batches * B,), -1, dtype=torch.int64, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self.use_cursor = True
        self.fill_id = 0
        # ---- buffers ([S, ...]) + the collate launch's descriptors (slot 0's pointers + the slot strides) --------------
        base = self.tables.data_ptr()
        self.bufs: Dict = {}
        descs, slot_bytes = [], []
        for k, (d, key, pk) in enumerate(packed._klist):
            cap = self._caps[k]
            add_ptr, dst_off = None, 0
            if key == 'x':
                out = torch.zeros(S, self.cap_cells[d], pk.width, dtype=pk.data.dtype, device=dev)
            elif key in _TWO_ROW:
                out = torch.zeros(S, 2, cap, dtype=pk.data.dtype, device=dev)
            elif key in ('b_rowptr', 'bt_rowptr'):
                out = torch.zeros(S, cap + 1, dtype=torch.int32, device=dev)      # rowptr[0] = 0 is never rewritten
                dst_off = 4
                add_ptr = base + 8 * (self.k_of(d, 'b_col' if key == 'b_rowptr' else 'bt_col') * (B + 1))
            else:
                out = torch.zeros(S, cap, dtype=pk.data.dtype, device=dev)
            row = _ADD_ROW.get(key)
            if row is not None:
                add_ptr = base + 8 * (self.o_off + (d * 5 + row) * B)
            self.bufs[(d, key)] = out
            two = pk.rows == 2
            descs.append(_ffi.CollateDesc(
                src=pk.data.data_ptr(), dst=out.data_ptr() + dst_off, dst_start=base + 8 * (k * (B + 1)),
                src_start=base + 8 * (self.o_src + k * B), add=add_ptr,
                src_row_stride=pk.data.size(-1) if two else 0, dst_row_stride=cap if two else 0,
                n_rows=pk.rows, op=pk.op))
            slot_bytes.append(out[0].numel() * out.element_size())
        for d in range(D):
            out = torch.zeros(S, self.cap_cells[d], dtype=torch.int64, device=dev)
            self.bufs[(d, 'batch')] = out
            descs.append(_ffi.CollateDesc(src=None, dst=out.data_ptr(), dst_start=base + 8 * (self.o_seg + d * (B + 1)),
                                          src_start=None, add=None, src_row_stride=0, dst_row_stride=0, n_rows=1,
                                          op=_ffi.COLLATE_SEGID64))
            slot_bytes.append(out[0].numel() * 8)
        self._descs = []
        for i in range(0, len(descs), _ffi.MAX_COLLATE_DESCS):
            part = descs[i:i + _ffi.MAX_COLLATE_DESCS]
            self._descs.append(((_ffi.CollateDesc * len(part))(*part), (C.c_int64 * len(part))(*slot_bytes[i:i + len(part)])))
        # ---- the slots: a ComplexBatch over slot j's views, its plans ---------------------------------------------------
        self._families: Dict = {}
        self._adjs = []
        self._slot_adjs: Dict[int, list] = {}      # mode 'csr': the upper / lower adjacencies of every slot (rebuilt by every fill)
        self.slots: List[StaticSlot] = [self._make_slot(j) for j in range(S)]
        # (slot 0 under the names a one-slot caller uses)
        self.batch, self.plan = self.slots[0].batch, self.slots[0].plan

    def _make_slot(self, j: int) -> StaticSlot:
        D, B = self.D, self.B
        bufs = {k: v[j] for k, v in self.bufs.items()}
        cochains = [CochainBatch(d) for d in range(D)]
        y = None
        for (d, key), out in bufs.items():
            if d < 0:
                y = out
            elif key == 'x':
                cochains[d]._x = out
            elif key not in _CSR_KEYS:
                setattr(cochains[d], key, out)
        for d, cb in enumerate(cochains):
            cb.ptr = None
            cb.__num_cells__ = self.cap_cells[d]
            cb.__num_cells_up__ = self.cap_cells[d + 1] if d + 1 < D else 0
            if d > 0:
                cb.__num_cells_down__ = self.cap_cells[d - 1]
            cb.__num_cochains__ = B
            cb.__slices__ = {}
        batch = StaticComplexBatch(*cochains, y=y, num_complexes=B, dimension=D - 1)
        # plans: CSR of the boundary adjacencies (collated), sizes-only stand-ins for the upper ones
        for d in range(D):
            cb = cochains[d]
            if d > 0 and cb.boundary_index is not None and self.k_of(d, 'b_col') >= 0:
                bi = cb.boundary_index
                adj = _CollatedAdjacency(bi, 1, self.cap_cells[d], self.cap_cells[d - 1], bufs[(d, 'b_rowptr')], bufs[(d, 'b_col')])
                adj._t_src = _CollatedAdjacency(bi, 0, self.cap_cells[d - 1], self.cap_cells[d], bufs[(d, 'bt_rowptr')],
                                                bufs[(d, 'bt_col')])
                if self.mode == 'csr':
                    # the TRANSPOSE has hub rows on REDDIT-like complexes (a vertex of degree 300 is the boundary of 300 edges):
                    # its long-row lists, which the collate does not bring, are written by the fill (cwn_csr_long_rows) when a
                    # training step reads this plan -- without them the streaming backward walked every hub row with one lane
                    # group (round 6: 54 against 37 us per backward aggregation launch)
                    t = adj._t_src
                    t.long_rows = torch.zeros(csr.LONG_PARTS, t.long_cap, dtype=torch.int32, device=bi.device)
                    t.n_long = torch.zeros(csr.LONG_PARTS, dtype=torch.int32, device=bi.device)
                    t.rows_dev_ptr = self.size_ptr(d - 1, j)
                    self._slot_long.setdefault(j, []).append(t)
                self._register(bi, adj)
            if cb.upper_index is not None and d + 1 < D:
                ui = cb.upper_index
                if self.mode == 'csr':
                    adj = csr.Adjacency(ui[1], ui[0], self.cap_cells[d], self.cap_cells[d], cb.shared_coboundaries, self.cap_cells[d + 1])
                    adj.e_dev_ptr = self.size_ptr(8 + self.k_of(d, 'upper_index'), j)       # the slot's live entry count
                    self._slot_adjs.setdefault(j, []).append(adj)
                    self._register(ui, adj)
                else:
                    self._register(ui, _PlanOnlyAdjacency(ui, self.cap_cells[d], self.cap_cells[d], cb.shared_coboundaries,
                                                          self.cap_cells[d + 1]))
            if self.mode == 'csr' and d > 0 and getattr(cb, 'lower_index', None) is not None:
                li = cb.lower_index
                adj = csr.Adjacency(li[1], li[0], self.cap_cells[d], self.cap_cells[d], cb.shared_boundaries, self.cap_cells[d - 1])
                adj.e_dev_ptr = self.size_ptr(8 + self.k_of(d, 'lower_index'), j)
                self._slot_adjs.setdefault(j, []).append(adj)
                self._register(li, adj)
        plan = StaticBlockPlan(self, j)
        some = next(iter(bufs.values()))
        batch._block_plan = (some.device, plan)          # (the device as the tensors spell it: Complex.block_plan compares)
        return StaticSlot(self, j, batch, bufs, plan)

    # ---- layout helpers --------------------------------------------------------------------------------------------
    def k_of(self, d: int, key: str) -> int:
        return self.packed.key_index(d, key)

    def cap_key(self, d: int, key: str) -> int:
        k = self.k_of(d, key)
        return self._caps[k] if k >= 0 else 0

    def seg_view(self, d: int, slot: int = 0) -> torch.Tensor:
        """`ptr` of dimension d: int64 [B + 1], the cells of complex c are rows seg[c] .. seg[c + 1]."""
        o = self.o_seg + d * (self.B + 1)
        return self.tables[slot, o: o + self.B + 1]

    def dst_view(self, k: int, slot: int = 0) -> torch.Tensor:
        """`__slices__` of key k: int64 [B + 1]."""
        return self.tables[slot, k * (self.B + 1): (k + 1) * (self.B + 1)]

    def size_ptr(self, j: int, slot: int = 0) -> int:
        return self.tables.data_ptr() + 8 * (slot * self.n_tab + self.o_sizes + j)

    def sizes(self, slot: int = 0) -> List[int]:
        return self.slots[slot].sizes()

    def dynamic(self) -> _ffi.dynamic_rows:
        return self.slots[0].dynamic()

    def _register(self, index: torch.Tensor, adj) -> None:
        import weakref
        key = id(index)
        csr._cache[key] = ((_ffi.tver(index), adj.n_dst, adj.n_val), weakref.ref(index, lambda _r, k=key: csr._cache.pop(k, None)), adj)
        self._adjs.append((index, adj))          # (the view tensors the cache is keyed on stay alive with the batch)

    def _n_sets(self, has_up) -> int:
        n, d = 0, 0
        while d < self.D:
            step = 1
            if has_up[d] and d + 1 < self.D and not has_up[d + 1] and d + 2 >= self.D:
                step = 2
            n += 1
            d += step
        return n

    # ---- item tables: one family = the tables of all slots for one (kind, width, streams), cut by ONE launch ------------
    def _sizes_dev(self, has_up, has_b) -> _ffi.LayerSizesDev:
        s = _ffi.LayerSizesDev(n_complexes=self.size_ptr(3), cap_complexes=self.B, n_dims=self.D, n_slots=self.S,
                               table_slot_stride=self.n_tab)
        for d in range(self.D):
            s.has_up[d] = 1 if has_up[d] else 0
            s.cell_ptr[d] = self.seg_view(d).data_ptr()
            k = self.k_of(d, 'upper_index')
            if k >= 0:
                s.up_ptr[d] = self.dst_view(k).data_ptr()
            k = self.k_of(d, 'boundary_index')
            if k >= 0 and has_b[d]:
                s.b_ptr[d] = self.dst_view(k).data_ptr()
        return s

    def family(self, kind: str, F: int, has_up, has_b):
        key = (kind, F, has_up, has_b)
        if key not in self._families:
            fam = self._make_family(kind, F, has_up, has_b)
            self._families[key] = fam
            if fam is not None:
                self._launch_family(key, self.S)       # cut for the batches the buffers hold now
        return self._families[key]

    def _pick_variant(self, F: int, has_up, has_b) -> None:
        """variant=None: the two-per-CU form when a batch has more items than the chip has CUs (B > 128: two sets of items per
        complex) AND every complex of the dataset fits its smaller workgroup (80 KiB of LDS: ~30 atoms at width 128) -- one
        form per static batch, settled by the first forward table that is cut."""
        if self._variant_settled:
            return
        self._variant_settled = True
        if self.B <= 128 or F not in (64, 128):
            return
        from .blockplan import single_fit_forward
        meta, D = self.packed._meta, self.D
        col = lambda d, key: (meta[:, 3 * D + self.k_of(d, key)] if self.k_of(d, key) >= 0 else None)
        ok = single_fit_forward([meta[:, 3 * d] for d in range(D)], [col(d, 'upper_index') for d in range(D)],
                                [col(d, 'boundary_index') for d in range(D)], F, has_up, has_b, 1,
                                128 if F == 64 else 80, 128 if F == 64 else 48)
        if bool(ok.all()):
            self.variant = 1

    def _make_family(self, kind: str, F: int, has_up, has_b):
        if self.mode == 'csr':
            return None                        # (no item tables: the layers take their CSR path)
        if kind == 'fwd':
            self._pick_variant(F, has_up, has_b)
        if F not in (64, 128) or any(has_up[d] and (d + 1 >= self.D or self.k_of(d, 'upper_index') < 0) for d in range(self.D)):
            return None
        S, B, dev = self.S, self.B, self.device
        n_sets = self._n_sets(has_up)
        n_items = n_sets * B                                   # a region of B records per set always suffices
        cap_up = [self.cap_key(d, 'upper_index') if has_up[d] else 0 for d in range(self.D)]
        cap_b = [self.cap_key(d, 'boundary_index') if (d > 0 and has_b[d]) else 0 for d in range(self.D)]
        if kind == 'fwd':
            if self.variant == 0:
                # one launch = one LDS layout: the full row cap, the boundary sources take what is left of the 160 KiB
                cap = gemm_rows_cap(F)
                src_cap = min(cap, (LDS_BYTES - lds_bytes(F, cap, 0)) // (F * 4))
                if src_cap < 16 or _ffi.lib().cwn_layer_fused_lds_bytes(F, cap, src_cap) == 0:
                    return None
                dyn_lds = 0
            else:
                cap = 128 if F == 64 else 80                   # = CWN_LAYER_W8_GEMM_ROWS / _SOURCE_ROWS / _LDS_BYTES
                src_cap = 128 if F == 64 else 48
                dyn_lds = 80 * 1024
            store = torch.zeros(S, n_items, ITEM_INTS, dtype=torch.int32, device=dev)
            fam = []
            for j in range(S):
                t = ItemTable(np.zeros((0, ITEM_INTS), dtype=np.int32), [s_ * B for s_ in range(n_sets)], cap, src_cap,
                              list(self.cap_cells), cap_up, cap_b, None, variant=self.variant, lds_bytes=dyn_lds)
                t.items, t.n_items, t.device = store[j], n_items, dev
                fam.append(t)
        else:
            store = torch.zeros(S, n_items, _ffi.LAYER_BWD_ITEM_INTS, dtype=torch.int32, device=dev)
            fam = []
            for j in range(S):
                t = BwdItemTable(np.zeros((0, _ffi.LAYER_BWD_ITEM_INTS), dtype=np.int32), BWD_LDS_BYTES, list(self.cap_cells), cap_up,
                                 cap_b, None)
                t.items, t.n_items, t.device = store[j], n_items, dev
                fam.append(t)
        fam[0].store, fam[0].sizes, fam[0].F = store, self._sizes_dev(has_up, has_b), F
        return fam

    def _launch_family(self, key, n_slots: int) -> None:
        fam = self._families.get(key)
        if fam is None:
            return
        t0 = fam[0]
        t0.sizes.n_slots = int(n_slots)
        err, s = csr._err_flag(self.device).data_ptr(), _ffi.stream_ptr(self.device)
        if key[0] == 'fwd':
            _ffi.check(_ffi.lib().cwn_layer_items_build_dev(t0.sizes, t0.F, t0.c_plan(with_cache=False), self.group, err, s),
                       'cwn_layer_items_build_dev')
            for t in fam[:n_slots]:
                t.csr_key = None                 # (the per-item CSR cache belongs to the previous batch's entries)
        else:
            _ffi.check(_ffi.lib().cwn_layer_bwd_items_build_dev(t0.sizes, t0.F, t0.c_plan(), self.group, err, s),
                       'cwn_layer_bwd_items_build_dev')

    def fit_mask(self) -> np.ndarray:
        """bool per complex of the dataset: every item table cut so far takes it as one item (numpy restatement of the device
        builders' test: blockplan.single_fit_forward / _backward)."""
        from .blockplan import single_fit_backward, single_fit_forward
        meta, D = self.packed._meta, self.D
        cells = [meta[:, 3 * d] for d in range(D)]
        col = lambda d, key: (meta[:, 3 * D + self.k_of(d, key)] if self.k_of(d, key) >= 0 else None)
        up_len = [col(d, 'upper_index') for d in range(D)]
        b_len = [col(d, 'boundary_index') for d in range(D)]
        # (cached per set of tables cut so far: a router asks once per epoch and batch list)
        stamp = tuple(sorted((repr(k), fam is not None, self.variant) for k, fam in self._families.items()))
        hit = getattr(self, '_fit_mask_cache', None)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        ok = np.ones(meta.shape[0], dtype=bool)
        for key, fam in self._families.items():
            if fam is None:
                continue
            kind, F, hu, hb = key
            if kind == 'fwd':
                ok &= single_fit_forward(cells, up_len, b_len, F, hu, hb, self.variant, fam[0].max_rows, fam[0].max_src)
            else:
                ok &= single_fit_backward(cells, up_len, b_len, F, hu, hb)
        self._fit_mask_cache = (stamp, ok)
        return ok

    # ---- which batches fit ------------------------------------------------------------------------------------------
    def fits(self, batches: Sequence[np.ndarray]) -> np.ndarray:
        """bool per batch (index arrays): every array of the batch within its capacity, every complex within what one
        workgroup of the item tables cut so far holds, and at least two cells of every dimension the dataset has
        (BatchNorm in training mode needs them; the reference raises below two)."""
        cap_ok, single_ok = self.fits_detail(batches)
        return cap_ok & single_ok

    def fits_detail(self, batches: Sequence[np.ndarray]):
        """(capacities + the two-cell rule hold, every complex fits one workgroup): the two halves of `fits`, bool per batch --
        a router that may take the complexes beyond a workgroup OUT of a batch (static_graph.RoutedForward) asks which half failed."""
        meta, D, K = self.packed._meta, self.D, self.K
        cap_ok = np.ones(len(batches), dtype=bool)
        single_ok = np.ones(len(batches), dtype=bool)
        caps_cells = np.asarray(self.cap_cells, dtype=np.int64)
        caps_keys = np.asarray(self._caps, dtype=np.int64)
        single = self.fit_mask()
        for i, idx in enumerate(batches):
            idx = np.asarray(idx, dtype=np.int64)
            if idx.size == 0 or idx.size > self.B:
                cap_ok[i] = False
                continue
            m = meta[idx]
            cells = m[:, 0:3 * D:3].sum(axis=0)
            lens = m[:, 3 * D:3 * D + K].sum(axis=0)
            cap_ok[i] = bool((cells <= caps_cells).all() and (lens <= caps_keys).all() and (cells >= 2).all())
            single_ok[i] = bool(single[idx].all())
        return cap_ok, single_ok

    # ---- filling ------------------------------------------------------------------------------------------------------
    def _host_perm(self, batches: Sequence[np.ndarray], n: int) -> np.ndarray:
        host = np.full((n, self.B), -1, dtype=np.int64)
        for j, idx in enumerate(batches):
            idx = np.asarray(idx, dtype=np.int64)
            if idx.size > self.B or idx.size == 0:
                raise ValueError(f'batch {j}: 1 .. {self.B} complexes')
            if idx.min() < 0 or idx.max() >= self.packed.num:
                raise IndexError(f'batch {j}: complex numbers outside 0 .. {self.packed.num - 1}')
            host[j, :idx.size] = idx
        self._check_capacity(host[:len(batches)])
        return host

    def _check_capacity(self, host: np.ndarray) -> None:
        """Every batch within the capacity of every buffer (the default capacities are a statistical bound, mean x B + 6 sigma
        sqrt(B): a size-sorted or bucketed batch order can exceed them).  Host work per epoch, inside a training loop's critical
        path: the size columns that are copies of one another (an index and its shared-cell vector, the boundary entries and
        their CSR columns ...) are checked once, against the smallest of their capacities -- ~7 gathers over the epoch's
        complex numbers.  The device repeats the test per fill (cwn_collate_guard: an oversize batch runs as an empty one and
        sets a sticky bit) for callers that write `idx` themselves."""
        if self._cap_cols is None:
            D, K = self.D, self.K
            meta = self.packed._meta
            take = np.concatenate([meta[:, 0:3 * D:3], meta[:, 3 * D:3 * D + K]], axis=1)          # [num, D + K]
            caps = np.asarray(list(self.cap_cells) + list(self._caps), dtype=np.int64)
            groups = {}
            for c in range(D + K):
                groups.setdefault(take[:, c].tobytes(), []).append(c)
            cols = []
            for members in groups.values():
                col = np.concatenate([take[:, members[0]], [0]]).astype(np.int64)                  # (row -1: no complex)
                if int(np.sort(col)[-min(self.B, col.size):].sum()) <= int(caps[members].min()):
                    continue                                                                       # no batch can exceed it
                cols.append((np.ascontiguousarray(col), int(caps[members].min()), members[int(np.argmin(caps[members]))]))
            self._cap_cols = cols
        D = self.D
        for col, cap, c in self._cap_cols:
            tot = col[host].sum(axis=1)
            bad = np.nonzero(tot > cap)[0]
            if bad.size:
                j = int(bad[0])
                what = f'cells of dimension {c}' if c < D else 'elements of {1!r} (dimension {0})'.format(*self.packed._klist[c - D][:2])
                raise ValueError(f'batch {j}: {int(tot[j])} {what} exceed the capacity {cap} of this StaticBatch '
                                 f'({bad.size} such batch(es)); build it with larger `caps`, or route these batches through '
                                 'PackedComplexes.collate (StaticBatch.fits() tells which)')

    def set_batches(self, batches: Sequence[Sequence[int]]) -> None:
        """The complexes of the NEXT fill, one index list per slot (at most `slots`; the other slots get empty batches): an
        epoch of one replay."""
        if len(batches) > self.S or len(batches) == 0:
            raise ValueError(f'1 .. {self.S} batches')
        self._upload(self._host_perm(batches, self.S).reshape(-1))
        self.cursor.zero_()

    def set_batch(self, idx: Sequence[int]) -> None:
        self.set_batches([idx])

    def _upload(self, host: np.ndarray) -> None:
        """The complex numbers of the next fills into `idx`, WITHOUT stopping the host behind the GPU (round 6).  A pageable copy
        on the compute stream waits for everything in front of it -- the whole previous epoch -- so the host prepared epoch
        e + 1 only after epoch e had finished (~1 ms of host work per epoch outside the GPU's shadow).  Here the host -> device
        copy runs on a SIDE stream into one of two staging buffers (the host waits for the copy alone), and the compute stream
        takes it over with a device -> device copy behind an event: in stream order behind the replays that still read the
        previous numbers.  (A pinned, non-blocking copy on the compute stream was tried first: it goes through the DMA engine,
        and the queue switch cost the short replays of the propagate scope 0.4 ms per epoch.)"""
        n = int(host.size)
        if os.environ.get('CWN_STATIC_BLOCKING_UPLOAD') == '1':       # (A/B: the pageable copy of rounds 4 - 5)
            self.idx[:n].copy_(torch.from_numpy(host))
            return
        st = self.__dict__.get('_up')
        if st is None or st['bufs'][0].numel() < max(n, self.idx.numel()):
            st = self._up = {'stream': torch.cuda.Stream(device=self.device), 'k': 0, 'ev': [None, None],
                             'bufs': [torch.empty(max(n, self.idx.numel()), dtype=torch.int64, device=self.device) for _ in range(2)]}
        k = st['k']
        st['k'] = k ^ 1
        side, buf = st['stream'], st['bufs'][k]
        with torch.cuda.stream(side):
            if st['ev'][k] is not None:
                side.wait_event(st['ev'][k])             # (the copy that read this buffer two uploads ago)
            buf[:n].copy_(torch.from_numpy(host))
            up = torch.cuda.Event()
            up.record(side)
        main = torch.cuda.current_stream(self.device)
        main.wait_event(up)
        self.idx[:n].copy_(buf[:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(main)
        st['ev'][k] = ev

    def reserve_epoch(self, n_batches: int) -> None:
        """Size the permutation buffer for epochs of up to n_batches batches (before the first fill / capture: a captured
        fill holds the buffer's address and its length)."""
        n = -(-max(1, int(n_batches)) // self.S) * self.S              # whole replays of S steps
        if n <= self.n_batches:
            return
        if self.fill_id > 0:
            raise RuntimeError(f'reserve_epoch({n_batches}) after the first fill: a captured step reads the buffer of {self.n_batches} '
                               'batches it was captured with -- ask before building StaticForward / StaticTrainStep')
        self.n_batches = n
        self.idx = torch.full((n * self.B,), -1, dtype=torch.int64, device=self.device)
        self.cursor.zero_()

    def set_epoch(self, batches: Sequence[np.ndarray]) -> int:
        """Upload the complex numbers of a whole epoch's batches (one host -> device copy); fill number j after this call takes
        batches j S .. j S + S - 1 (the device cursor advances by itself: a replayed step needs nothing from the host).  Batches
        past the epoch's own are empty (their steps change nothing).  Returns the number of fills the epoch takes."""
        self.reserve_epoch(len(batches))
        if len(batches) > self.n_batches:
            raise RuntimeError(f'set_epoch: {len(batches)} batches, the permutation buffer a captured step reads holds '
                               f'{self.n_batches}: reserve_epoch(n) before the first fill')
        self._upload(self._host_perm(batches, self.n_batches).reshape(-1))
        self.cursor.zero_()
        return -(-len(batches) // self.S)

    def rewind(self, j: int = 0) -> None:
        self.cursor.fill_(int(j))

    def fill(self, n_slots: Optional[int] = None) -> None:
        """The tables (cwn_collate_tables), the arrays (cwn_collate_slots) and the item tables asked for so far
        (cwn_layer_items_build_dev / _bwd_) of the first n_slots slots (default: all) -- one launch each whatever the number
        of slots.  Graph-capturable, no host sync; an item table asked for later is cut at that moment."""
        n = self.S if n_slots is None else int(n_slots)
        if not (1 <= n <= self.S):
            raise ValueError(f'1 .. {self.S} slots')
        L = _ffi.lib()
        s = _ffi.stream_ptr(self.device)
        err = csr._err_flag(self.device).data_ptr()
        cur = self.cursor.data_ptr()
        _ffi.check(L.cwn_collate_tables(self.meta.data_ptr(), self.packed.num, self.D, self.K, self.idx.data_ptr(), self.B,
                                        self.n_batches, cur, n, self.n_tab, self.tables.data_ptr(), err, s), 'cwn_collate_tables')
        _ffi.check(L.cwn_collate_guard(self.tables.data_ptr(), self.D, self.K, self.B, n, self.n_tab, self.caps_dev.data_ptr(), err, s),
                   'cwn_collate_guard')
        for i, (arr, strides) in enumerate(self._descs):
            _ffi.check(L.cwn_collate_slots(arr, len(arr), self.B, n, self.n_tab, strides, cur if i == len(self._descs) - 1 else None, s),
                       'cwn_collate_slots')
        self.fill_id += 1
        for key in self._families:
            self._launch_family(key, n)
        if self.mode == 'csr':
            # the CSR plans of the upper / lower adjacencies (and, for a training step, their transposes) of ALL slots in
            # batched cwn_csr_build calls (<= 8 plans each: one launch sequence per eight plans, not per slot) over the
            # capacity-sized entries, every plan with the address of ITS slot's live entry count (cwn_csr_desc.e_dev)
            todo = []
            for j in range(n):
                for adj in self._slot_adjs.get(j, []):
                    todo.append(adj)
                    if self.build_backward:
                        adj.transposes()
                        for t in (adj._t_src, adj._t_aux):
                            if t is not None:
                                t.e_dev_ptr = adj.e_dev_ptr
                                todo.append(t)
            if todo:
                csr.build_many(todo, validate=False, force=True)
            if self.build_backward:
                lists = [t for j in range(n) for t in self._slot_long.get(j, [])]
                for i in range(0, len(lists), _ffi.CSR_MAX_DESCS):
                    chunk = lists[i:i + _ffi.CSR_MAX_DESCS]
                    arr = (_ffi.LongRowsDesc * len(chunk))(*[
                        _ffi.LongRowsDesc(rowptr=t.rowptr.data_ptr(), n_rows=t.n_dst, m_dev=t.rows_dev_ptr, long_rows=t.long_rows.data_ptr(),
                                          n_long=t.n_long.data_ptr(), long_cap=t.long_cap) for t in chunk])
                    _ffi.check(L.cwn_csr_long_rows(arr, len(chunk), s), 'cwn_csr_long_rows')

    # ---- the reference tables (tests) -----------------------------------------------------------------------------------
    def host_tables(self, idx: Sequence[int]) -> np.ndarray:
        """What cwn_collate_tables must write for the complexes `idx` (numpy restatement: the test's checker)."""
        B, D, K = self.B, self.D, self.K
        meta = self.packed._meta
        idx = np.asarray(idx, dtype=np.int64)
        m = np.zeros((B, meta.shape[1]), dtype=np.int64)
        m[:idx.size] = meta[idx]
        out = np.zeros(self.n_tab, dtype=np.int64)
        cs = np.cumsum(m[:, :3 * D + K], axis=0)
        dst = np.zeros((K, B + 1), dtype=np.int64)
        dst[:, 1:] = cs[:, 3 * D:].T
        out[:K * (B + 1)] = dst.reshape(-1)
        out[self.o_src: self.o_src + K * B] = m[:, 3 * D + K: 3 * D + 2 * K].T.reshape(-1)
        cnt = m[:, :3 * D].T.reshape(D, 3, B)
        end = cs[:, :3 * D].T.reshape(D, 3, B)
        off = end - cnt
        out[self.o_off: self.o_off + D * 5 * B] = off[:, (0, 0, 1, 0, 2), :].reshape(-1)
        seg = np.concatenate([off[:, 0, :], end[:, 0, -1:]], axis=1)
        out[self.o_seg: self.o_seg + D * (B + 1)] = seg.reshape(-1)
        for d in range(min(D, 3)):
            out[self.o_sizes + d] = end[d, 0, -1]
        out[self.o_sizes + 3] = idx.size
        out[self.o_sizes + 8: self.o_sizes + 8 + K] = dst[:, -1]
        return out
