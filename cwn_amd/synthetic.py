"""Synthetic ring-lifted molecular complexes (no dataset is available offline).

`zinc_like_complexes` draws ZINC-shaped molecules (18-30 atoms, mean ~23; a tree backbone with 1-4
rings of size 5/6, fused pairs with probability 0.3; atom type in [0,28), bond type in [0,4),
data/datasets/zinc.py:29-30) and lifts each to a 2-complex with the ring lift restated from the
reference (data/utils.py:400-498: vertices, edges in lexicographic (u<v) order, induced cycles of
length <= max_ring as 2-cells; adjacency construction as build_adj, :103-138; boundary_index as
generate_cochain, :177-221).  Ring ids are canonical here (sorted by sorted vertex tuple); the
reference leaves them in Python-set order (:300-330).

Host-side numpy; it produces the INPUT of the hot path and is not on it.
"""
import itertools
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .complex import Cochain, Complex, ComplexBatch


# ------------------------------------------------------------------------------------------------
# graphs
# ------------------------------------------------------------------------------------------------
def random_molecule(rng: np.random.Generator, n_lo: int = 18, n_hi: int = 30,
                    ring_probs=(0.2, 0.4, 0.3, 0.1)) -> Tuple[int, List[Tuple[int, int]]]:
    """A connected molecule-like graph: (num_atoms, sorted list of bonds (u < v))."""
    target = int(rng.integers(n_lo, n_hi + 1))
    n_rings = int(rng.choice(len(ring_probs), p=ring_probs)) + 1
    deg: List[int] = []
    bonds = set()
    ring_bonds: List[Tuple[int, int]] = []      # bonds that belong to exactly one ring so far

    def new_atom() -> int:
        deg.append(0)
        return len(deg) - 1

    def bond(u: int, v: int):
        bonds.add((min(u, v), max(u, v)))
        deg[u] += 1
        deg[v] += 1

    def add_cycle(path: List[int]):
        for a, b in zip(path, path[1:] + path[:1]):
            if (min(a, b), max(a, b)) not in bonds:
                bond(a, b)
                ring_bonds.append((min(a, b), max(a, b)))

    for r in range(n_rings):
        k = 5 if rng.random() < 0.3 else 6
        fusable = [b for b in ring_bonds if deg[b[0]] <= 3 and deg[b[1]] <= 3]
        if r > 0 and fusable and rng.random() < 0.3:
            u, v = fusable[int(rng.integers(len(fusable)))]
            ring_bonds.remove((u, v))
            add_cycle([u] + [new_atom() for _ in range(k - 2)] + [v])
        else:
            cyc = [new_atom() for _ in range(k)]
            add_cycle(cyc)
            if r > 0:      # link the new ring to the molecule through 0-2 chain atoms
                free = [a for a in range(cyc[0]) if deg[a] < 4]
                prev = free[int(rng.integers(len(free)))]
                for _ in range(int(rng.integers(0, 3))):
                    a = new_atom()
                    bond(prev, a)
                    prev = a
                bond(prev, cyc[0])
    while len(deg) < target:
        free = [a for a in range(len(deg)) if deg[a] < 4]
        # prefer chain ends so side chains look like chains, not stars
        w = np.array([2.0 if deg[a] == 1 else 1.0 for a in free])
        prev = free[int(rng.choice(len(free), p=w / w.sum()))]
        bond(prev, new_atom())
    return len(deg), sorted(bonds)


def induced_cycles(n: int, bonds: Sequence[Tuple[int, int]], max_k: int) -> List[Tuple[int, ...]]:
    """All chordless cycles with 3..max_k vertices, each as a vertex tuple in cyclic order starting
    at its smallest vertex, sorted canonically (what graph-tool's induced subgraph isomorphism
    finds in data/utils.py:300-330; cross-checked against a networkx enumerator by the reference,
    data/helper_test.py:68-99)."""
    adj = [set() for _ in range(n)]
    for u, v in bonds:
        adj[u].add(v)
        adj[v].add(u)
    found = {}

    def extend(path: List[int]):
        s, last = path[0], path[-1]
        for w in adj[last]:
            if w == s:
                continue
            if w < s or w in path:
                continue
            # chordless: w may touch the path only at `last` (and at s when it closes the cycle)
            touches = adj[w].intersection(path)
            closing = s in touches
            if touches - {last, s}:
                continue
            if closing and len(path) + 1 >= 3:
                cyc = path + [w]
                key = tuple(sorted(cyc))
                if key not in found:
                    found[key] = tuple(cyc) if cyc[1] < cyc[-1] else tuple([cyc[0]] + cyc[:0:-1])
                continue   # a vertex adjacent to s cannot be an interior vertex of a longer cycle
            if len(path) + 1 < max_k:
                extend(path + [w])

    for s in range(n):
        for w in adj[s]:
            if w > s:
                extend([s, w])
    return [found[k] for k in sorted(found)]


# ------------------------------------------------------------------------------------------------
# ring lift
# ------------------------------------------------------------------------------------------------
def ring_lift(n: int, bonds: Sequence[Tuple[int, int]], vx: torch.Tensor,
              ex: Optional[torch.Tensor] = None, max_k: int = 6, include_down_adj: bool = False,
              y: Optional[torch.Tensor] = None, rx: Optional[torch.Tensor] = None) -> Complex:
    """compute_ring_2complex (data/utils.py:400-498) restated: returns a cwn_amd Complex."""
    bonds = sorted((min(u, v), max(u, v)) for u, v in bonds)
    edge_id = {b: i for i, b in enumerate(bonds)}
    rings = induced_cycles(n, bonds, max_k)
    ring_edges = []
    for ring in rings:
        es = sorted(tuple(sorted((ring[i], ring[(i + 1) % len(ring)]))) for i in range(len(ring)))
        ring_edges.append([edge_id[e] for e in es])          # get_ring_boundaries (:355-367)

    def pairs(groups):
        idx, shared = [], []
        for gid, members in enumerate(groups):
            for a, b in itertools.combinations(members, 2):
                idx += [(a, b), (b, a)]
                shared += [gid, gid]
        return idx, shared

    def lower(members_of_cell, n_lower):
        """cells sharing a boundary cell: for each lower cell, all pairs of its cofaces."""
        cof = [[] for _ in range(n_lower)]
        for cid, members in enumerate(members_of_cell):
            for m in members:
                cof[m].append(cid)
        return pairs(cof)

    def idx_tensor(lst):
        return torch.tensor(lst, dtype=torch.long).t().contiguous() if lst else None

    def vec_tensor(lst):
        return torch.tensor(lst, dtype=torch.long) if lst else None

    dim = 2 if rings else (1 if bonds else 0)
    v_up, v_cob = pairs([list(b) for b in bonds])
    cochains = [Cochain(dim=0, x=vx, upper_index=idx_tensor(v_up), shared_coboundaries=vec_tensor(v_cob),
                        num_cells_up=len(bonds) if dim >= 1 else 0, num_cells=n)]
    if dim >= 1:
        e_up, e_cob = pairs(ring_edges)
        e_down, e_bnd = lower([list(b) for b in bonds], n) if include_down_adj else ([], [])
        b_index = torch.tensor([[v for b in bonds for v in b],
                                [i for i in range(len(bonds)) for _ in range(2)]], dtype=torch.long)
        cochains.append(Cochain(dim=1, x=ex, upper_index=idx_tensor(e_up),
                                shared_coboundaries=vec_tensor(e_cob), lower_index=idx_tensor(e_down),
                                shared_boundaries=vec_tensor(e_bnd), boundary_index=b_index,
                                num_cells=len(bonds), num_cells_down=n,
                                num_cells_up=len(rings) if dim >= 2 else 0))
    if dim >= 2:
        r_down, r_bnd = lower(ring_edges, len(bonds)) if include_down_adj else ([], [])
        b_index = torch.tensor([[e for es in ring_edges for e in es],
                                [i for i, es in enumerate(ring_edges) for _ in es]], dtype=torch.long)
        cochains.append(Cochain(dim=2, x=rx, lower_index=idx_tensor(r_down),
                                shared_boundaries=vec_tensor(r_bnd), boundary_index=b_index,
                                num_cells=len(rings), num_cells_down=len(bonds), num_cells_up=0))
    return Complex(*cochains, y=y, dimension=dim)


# ------------------------------------------------------------------------------------------------
# batches
# ------------------------------------------------------------------------------------------------
def zinc_like_complexes(num: int, seed: int = 0, max_ring: int = 6, n_lo: int = 18, n_hi: int = 30,
                        atom_types: int = 28, bond_types: int = 4,
                        include_down_adj: bool = False, size_dist: str = 'uniform') -> List[Complex]:
    """`num` ZINC-shaped ring-lifted complexes with integer atom / bond types as [N,1] floats
    (the form EmbedSparseCIN expects, mp/molec_models.py:95-99).  size_dist: 'uniform' on [n_lo, n_hi] (SURVEY.md 8d's
    generator: 18 - 30), or 'zinc': the published size statistics of the ZINC-12k subset the reference trains on (9 - 37 heavy
    atoms, mean 23.2, standard deviation ~4.6: a clipped normal -- about 2 % of the molecules have more than 32 atoms (what a workgroup of the layer kernel held at width 128 until round 4), the
    most one workgroup of the 16-wave layer kernel holds at width 128)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num):
        if size_dist == 'zinc':
            k = int(np.clip(np.rint(rng.normal(23.2, 4.6)), 9, 37))
            n, bonds = random_molecule(rng, k, k)
        else:
            n, bonds = random_molecule(rng, n_lo, n_hi)
        vx = torch.from_numpy(rng.integers(0, atom_types, size=(n, 1))).float()
        ex = torch.from_numpy(rng.integers(0, bond_types, size=(len(bonds), 1))).float()
        y = torch.from_numpy(rng.standard_normal(1).astype(np.float32))
        out.append(ring_lift(n, bonds, vx, ex, max_k=max_ring, include_down_adj=include_down_adj, y=y))
    return out


def zinc_like_batch(batch_size: int = 128, seed: int = 0, max_ring: int = 6, device=None,
                    **kw) -> ComplexBatch:
    b = ComplexBatch.from_complex_list(zinc_like_complexes(batch_size, seed, max_ring, **kw), max_dim=2)
    return b.to(device) if device is not None else b


def batch_stats(batch: ComplexBatch) -> dict:
    """Cell and adjacency-entry counts (the N_d, E_up, B of SURVEY.md §8)."""
    s = {}
    for d in range(batch.dimension + 1):
        c = batch.cochains[d]
        s[f'N{d}'] = int(c.num_cells)
        s[f'E_up{d}'] = int(c.upper_index.size(1)) if c.upper_index is not None else 0
        s[f'E_down{d}'] = int(c.lower_index.size(1)) if c.lower_index is not None else 0
        s[f'B{d}'] = int(c.boundary_index.size(1)) if c.boundary_index is not None else 0
    s['cells'] = sum(s[f'N{d}'] for d in range(batch.dimension + 1))
    return s


# ------------------------------------------------------------------------------------------------
# molhiv-like (BASELINE config 3) and REDDIT-like clique complexes (config 5)
# ------------------------------------------------------------------------------------------------
def molhiv_like_complexes(num: int, seed: int = 0, max_ring: int = 6, n_lo: int = 10, n_hi: int = 60,
                          atom_dims=(119, 4, 12, 12, 10, 6, 6, 2, 2), bond_dims=(5, 6, 2), tail: float = 0.0,
                          tail_lo: int = 120, tail_hi: int = 220) -> List[Complex]:
    """ogbg-molhiv-shaped inputs: 10-60 atoms, 9 integer atom-feature columns, 3 integer bond-feature
    columns (OGB convention), ring lift with max_ring.  `tail`: the probability of a molecule from the dataset's heavy tail
    instead -- ogbg-molhiv's published statistics (41 127 molecules, 25.5 atoms on average, the largest 222) put a few
    molecules of 120 - 220 atoms in every 10 000: complexes beyond what one workgroup of the blocked layer kernel holds, which
    the generator of SURVEY.md 8d (10 - 60 atoms) never produces (`molhiv_real_tail`, tail = 5e-4)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num):
        lo_, hi_ = (tail_lo, tail_hi) if (tail > 0.0 and rng.random() < tail) else (n_lo, n_hi)
        n, bonds = random_molecule(rng, lo_, hi_, ring_probs=(0.25, 0.35, 0.25, 0.15))
        vx = torch.from_numpy(np.stack([rng.integers(0, d, size=n) for d in atom_dims], 1))
        ex = torch.from_numpy(np.stack([rng.integers(0, d, size=len(bonds)) for d in bond_dims], 1))
        y = torch.from_numpy(rng.integers(0, 2, size=(1, 1)).astype(np.float32))
        out.append(ring_lift(n, bonds, vx, ex, max_k=max_ring, y=y))
    return out


def preferential_attachment_graph(rng: np.random.Generator, n: int, m: int = 2, hubs: int = 2,
                                  hub_degree: int = 100) -> List[Tuple[int, int]]:
    """A REDDIT-like discussion graph: preferential attachment (each new vertex links to `m`
    earlier ones, probability ~ degree) with `hubs` vertices forced to degree >= hub_degree."""
    edges = set()
    targets = [0, 1]
    edges.add((0, 1))
    for v in range(2, n):
        picks = set()
        while len(picks) < min(m, v):
            picks.add(int(targets[int(rng.integers(len(targets)))]))
        for u in picks:
            edges.add((min(u, v), max(u, v)))
            targets += [u, v]
    deg = np.zeros(n, dtype=np.int64)
    for u, v in edges:
        deg[u] += 1
        deg[v] += 1
    for h in np.argsort(-deg)[:hubs]:
        need = hub_degree - int(deg[h])
        if need > 0:
            others = [int(v) for v in rng.permutation(n) if v != h and (min(h, v), max(h, v)) not in edges]
            for v in others[:need]:
                edges.add((min(int(h), v), max(int(h), v)))
    return sorted(edges)


def clique_lift(n: int, edges: Sequence[Tuple[int, int]], vx: torch.Tensor, max_dim: int = 2,
                init_method: str = 'sum', y: Optional[torch.Tensor] = None,
                include_down_adj: bool = False) -> Complex:
    """compute_clique_complex_with_gudhi (data/utils.py:224-272) restated for expansion_dim <= 2:
    vertices, edges in lexicographic (u<v) order, triangles in lexicographic order (the order of a
    gudhi SimplexTree traversal); upper adjacencies as build_adj (:103-138); boundaries of a
    simplex in itertools.combinations order (:40-42); features of higher cells = reduce of their
    vertices' features (construct_features, :141-156)."""
    edges = sorted((min(u, v), max(u, v)) for u, v in edges)
    edge_id = {e: i for i, e in enumerate(edges)}
    nbrs = [set() for _ in range(n)]
    for u, v in edges:
        nbrs[u].add(v)
        nbrs[v].add(u)
    tris = []
    if max_dim >= 2:
        for u, v in edges:
            for w in sorted(nbrs[u] & nbrs[v]):
                if w > v:
                    tris.append((u, v, w))
        tris.sort()
    tri_edges = [[edge_id[(a, b)], edge_id[(a, c)], edge_id[(b, c)]] for a, b, c in tris]

    def pairs(groups):
        idx, shared = [], []
        for gid, members in enumerate(groups):
            for a, b in itertools.combinations(members, 2):
                idx += [(a, b), (b, a)]
                shared += [gid, gid]
        return idx, shared

    def idx_tensor(lst):
        return torch.tensor(lst, dtype=torch.long).t().contiguous() if lst else None

    def vec_tensor(lst):
        return torch.tensor(lst, dtype=torch.long) if lst else None

    def reduce_feats(groups):
        if not groups:
            return None
        g = torch.tensor(groups, dtype=torch.long)
        f = vx[g]                                    # [cells, k, F]
        return f.sum(1) if init_method in ('sum', 'add') else f.mean(1)

    def lower(members_of_cell, n_lower):
        cof = [[] for _ in range(n_lower)]
        for cid, members in enumerate(members_of_cell):
            for m in members:
                cof[m].append(cid)
        return pairs(cof)

    dim = 2 if tris else (1 if edges else 0)
    v_up, v_cob = pairs([list(e) for e in edges])
    cochains = [Cochain(dim=0, x=vx, upper_index=idx_tensor(v_up), shared_coboundaries=vec_tensor(v_cob),
                        num_cells=n, num_cells_up=len(edges) if dim >= 1 else 0)]
    if dim >= 1:
        e_up, e_cob = pairs(tri_edges)
        e_down, e_bnd = lower([list(e) for e in edges], n) if include_down_adj else ([], [])
        b_index = torch.tensor([[v for e in edges for v in e],
                                [i for i in range(len(edges)) for _ in range(2)]], dtype=torch.long)
        cochains.append(Cochain(dim=1, x=reduce_feats([list(e) for e in edges]), upper_index=idx_tensor(e_up),
                                shared_coboundaries=vec_tensor(e_cob), lower_index=idx_tensor(e_down),
                                shared_boundaries=vec_tensor(e_bnd), boundary_index=b_index,
                                num_cells=len(edges), num_cells_down=n,
                                num_cells_up=len(tris) if dim >= 2 else 0))
    if dim >= 2:
        t_down, t_bnd = lower(tri_edges, len(edges)) if include_down_adj else ([], [])
        b_index = torch.tensor([[e for es in tri_edges for e in es],
                                [i for i in range(len(tris)) for _ in range(3)]], dtype=torch.long)
        cochains.append(Cochain(dim=2, x=reduce_feats([list(t) for t in tris]), lower_index=idx_tensor(t_down),
                                shared_boundaries=vec_tensor(t_bnd), boundary_index=b_index,
                                num_cells=len(tris), num_cells_down=len(edges), num_cells_up=0))
    return Complex(*cochains, y=y, dimension=dim)


def reddit_like_complexes(num: int = 32, seed: int = 0, n_lo: int = 200, n_hi: int = 1000,
                          init_method: str = 'mean') -> List[Complex]:
    """REDDIT-BINARY-shaped inputs (exp/scripts/mpsn-redditb.sh): large irregular graphs with 1-3
    hubs of degree >= 100, a constant scalar vertex feature, clique lift to dimension 2,
    higher-cell features by `init_method` (mean -> all ones)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num):
        n = int(rng.integers(n_lo, n_hi + 1))
        edges = preferential_attachment_graph(rng, n, m=int(rng.integers(1, 4)), hubs=int(rng.integers(1, 4)),
                                              hub_degree=int(rng.integers(100, 200)))
        vx = torch.ones(n, 1)
        y = torch.from_numpy(rng.integers(0, 2, size=1))
        out.append(clique_lift(n, edges, vx, max_dim=2, init_method=init_method, y=y))
    return out
