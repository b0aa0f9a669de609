"""The first form of PackedComplexes.collate's host half (one small numpy call per key and dimension), kept as
the independent restatement the vectorised one in cwn_amd/packed.py is checked against on the CPU: same output
tensors (shapes, dtypes), same per-launch descriptors, same table CONTENTS behind every descriptor."""
from typing import List

import numpy as np
import torch

from cwn_amd.complex import CochainBatch


def prepare(self, idx):
    """-> (cochains, y, tables (one int64 array), plan): plan entries are
    (packed or None, out tensor, offset of dst_start, offset of src_start or None, offset of add or None, total)."""
    idx = np.asarray(idx, dtype=np.int64)
    B = int(idx.size)
    dimension = int(self.dims[idx].max())
    dev = self.device
    tables: List[np.ndarray] = []
    plan = []
    cur = 0

    def table(arr):
        nonlocal cur
        off = cur
        tables.append(np.ascontiguousarray(arr, dtype=np.int64).reshape(-1))
        cur += tables[-1].size
        return off

    def excl(v):
        return np.concatenate([[0], np.cumsum(v)[:-1]]).astype(np.int64)

    cochains = []
    for d in range(dimension + 1):
        n_sel = self.n_cells[d, idx]
        off_here, off_down, off_up = excl(n_sel), excl(self.n_down[d, idx]), excl(self.n_up[d, idx])
        cb = CochainBatch(d)
        cb.__num_cells_list__ = n_sel.tolist()
        cb.__slices__ = {}
        for key, pk in self.keys[d].items():
            if not pk.has[idx].any():
                continue
            lens = pk.length[idx]
            total = int(lens.sum())
            dst_start = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            if key == 'x':
                out = torch.empty(total // pk.width, pk.width, dtype=pk.data.dtype, device=dev)
            elif pk.rows == 2:
                out = torch.empty(2, total, dtype=pk.data.dtype, device=dev)
            else:
                out = torch.empty(total, dtype=pk.data.dtype, device=dev)
            add = None
            if key in ('upper_index', 'lower_index'):
                add = np.stack([off_here, off_here])
            elif key == 'shared_boundaries':
                add = off_down[None]
            elif key == 'shared_coboundaries':
                add = off_up[None]
            elif key == 'boundary_index':
                add = np.stack([off_down, off_here])
            plan.append((pk, out, table(dst_start), table(pk.start[idx]),
                         None if add is None else table(add), total))
            if key != 'x':
                cb.__slices__[key] = dst_start.tolist()
            if key == 'x':
                cb._x = out
            else:
                setattr(cb, key, out)
        if self.has_cells[d, idx].any():
            total = int(n_sel.sum())
            out = torch.empty(total, dtype=torch.int64, device=dev)
            plan.append((None, out, table(np.concatenate([[0], np.cumsum(n_sel)])), None, None, total))
            cb.batch = out
            cb.ptr = [0] + np.cumsum(n_sel[self.has_cells[d, idx]]).tolist()
        cb.__num_cells__ = int(n_sel.sum())
        cb.__num_cells_up__ = int(self.n_up[d, idx].sum())
        if d > 0:
            cb.__num_cells_down__ = int(self.n_down[d, idx].sum())
        cb.__num_cochains__ = B
        cochains.append(cb)
    y = None
    if self.y is not None:
        lens = self.y.length[idx]
        y = torch.empty(int(lens.sum()), dtype=self.y.data.dtype, device=dev)
        plan.append((self.y, y, table(np.concatenate([[0], np.cumsum(lens)])), table(self.y.start[idx]),
                     None, int(lens.sum())))
    return cochains, y, np.concatenate(tables), plan
