"""Graph -> cell complex lifting through the C-ABI host routines (`cwn_lift_*`, csrc/cwn_lift.cpp):
the native counterpart of the reference's `compute_ring_2complex` / `compute_clique_complex_with_gudhi`
(data/utils.py:224-272, 400-498), which need graph-tool and gudhi.  Produces `cwn_amd.complex.Complex`
objects with exactly the tensors of the Python restatement in `cwn_amd/synthetic.py`
(tests/test_lifting.py compares the two and the reference's expected tensors).  Host CPU code: this
is input preparation, not the GPU hot path."""
import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi
from .complex import Cochain, Complex

RING, CLIQUE = 0, 1
(EDGES, CELLS2_PTR, CELLS2_VERTS, UP0, COB0, UP1, COB1, DOWN1, BND1, DOWN2, BND2, BINDEX1,
 BINDEX2) = range(13)


class Lift:
    """RAII wrapper of a cwn_lift_t."""

    def __init__(self, kind: int, n: int, edges: Sequence[Tuple[int, int]], max_k: int = 6,
                 include_down: bool = False):
        L = _ffi.lib()
        e = np.ascontiguousarray(np.asarray(list(edges), dtype=np.int64).reshape(-1, 2))
        self._L = L
        self._h = L.cwn_lift_create(kind, n, e.ctypes.data_as(C.c_void_p), e.shape[0], max_k,
                                    int(include_down))
        if not self._h:
            raise ValueError('invalid graph for lifting (vertex out of range or self loop)')

    def array(self, which: int) -> torch.Tensor:
        n = self._L.cwn_lift_size(self._h, which)
        out = torch.empty(max(n, 0), dtype=torch.long)
        _ffi.check(self._L.cwn_lift_copy(self._h, which, out.data_ptr()), 'cwn_lift_copy')
        return out

    def index(self, which: int) -> Optional[torch.Tensor]:
        a = self.array(which)
        return a.view(2, -1) if a.numel() else None

    def vector(self, which: int) -> Optional[torch.Tensor]:
        a = self.array(which)
        return a if a.numel() else None

    def cells2(self):
        ptr, verts = self.array(CELLS2_PTR).tolist(), self.array(CELLS2_VERTS).tolist()
        return [tuple(verts[ptr[i]:ptr[i + 1]]) for i in range(len(ptr) - 1)]

    def __del__(self):
        if getattr(self, '_h', None):
            self._L.cwn_lift_destroy(self._h)
            self._h = None


def _complex(L: Lift, n: int, vx, ex, cx2, include_down_adj: bool, y) -> Complex:
    edges = L.array(EDGES).view(-1, 2)
    E, n2 = edges.size(0), L.array(CELLS2_PTR).numel() - 1
    dim = 2 if n2 else (1 if E else 0)
    cochains = [Cochain(dim=0, x=vx, upper_index=L.index(UP0), shared_coboundaries=L.vector(COB0),
                        num_cells_up=E if dim >= 1 else 0, num_cells=n)]
    if dim >= 1:
        cochains.append(Cochain(
            dim=1, x=ex, upper_index=L.index(UP1), shared_coboundaries=L.vector(COB1),
            lower_index=L.index(DOWN1) if include_down_adj else None,
            shared_boundaries=L.vector(BND1) if include_down_adj else None,
            boundary_index=L.index(BINDEX1), num_cells=E, num_cells_down=n,
            num_cells_up=n2 if dim >= 2 else 0))
    if dim >= 2:
        cochains.append(Cochain(
            dim=2, x=cx2, lower_index=L.index(DOWN2) if include_down_adj else None,
            shared_boundaries=L.vector(BND2) if include_down_adj else None,
            boundary_index=L.index(BINDEX2), num_cells=n2, num_cells_down=E, num_cells_up=0))
    return Complex(*cochains, y=y, dimension=dim)


def ring_lift(n: int, bonds: Sequence[Tuple[int, int]], vx: torch.Tensor,
              ex: Optional[torch.Tensor] = None, max_k: int = 6, include_down_adj: bool = False,
              y: Optional[torch.Tensor] = None, rx: Optional[torch.Tensor] = None) -> Complex:
    """data/utils.py:400-498.  `ex` rows follow the sorted (u < v) edge order."""
    return _complex(Lift(RING, n, bonds, max_k, include_down_adj), n, vx, ex, rx, include_down_adj, y)


def clique_lift(n: int, edges: Sequence[Tuple[int, int]], vx: torch.Tensor, max_dim: int = 2,
                init_method: str = 'sum', y: Optional[torch.Tensor] = None,
                include_down_adj: bool = False) -> Complex:
    """data/utils.py:224-272 (expansion_dim <= 2); higher-cell features = reduce of the vertices'
    features (construct_features, :141-156)."""
    if max_dim > 2:
        raise NotImplementedError('clique lift up to dimension 2')
    L = Lift(CLIQUE, n, edges, 0, include_down_adj) if max_dim >= 2 else Lift(RING, n, edges, 0, include_down_adj)
    red = (lambda f: f.sum(1)) if init_method in ('sum', 'add') else (lambda f: f.mean(1))
    ev = L.array(EDGES).view(-1, 2)
    ex = red(vx[ev]) if ev.numel() else None
    tris = L.cells2()
    tx = red(vx[torch.tensor(tris, dtype=torch.long)]) if tris else None
    return _complex(L, n, vx, ex, tx, include_down_adj, y)


def induced_cycles(n: int, bonds: Sequence[Tuple[int, int]], max_k: int):
    """Chordless cycles with 3..max_k vertices, each in cyclic order from its smallest vertex."""
    return Lift(RING, n, bonds, max_k, False).cells2()
