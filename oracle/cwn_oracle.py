"""CPU ORACLE for the cellular message-passing hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A plain-PyTorch (CPU, fp32) restatement of the reference algorithm for
`CochainMessagePassing.propagate` and the layers built on it.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module, and only as
the checker / the timed CPU baseline.  Nothing under `cwn_amd/` imports it.

Parity status: PINNED.  `oracle/gen_golden.py` runs the reference's own `mp/cell_mp.py`,
`mp/layers.py`, `data/complex.py` (imported from /root/reference behind `oracle/refshim`) and
writes inputs + outputs to `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every
function here against those files and against the hand-computed expected tensors the reference's
own tests hold (mp/test_cell_mp.py, mp/test_layers.py, data/test_batching.py).  The arithmetic of
the scatter itself lives in torch-scatter==2.0.5 (pyG_install.sh:4), absent from the tree; its
published semantics (sum / mean = sum / max(count,1) / max with empty rows = 0) are restated in
`scatter_rows`.  `reduce='mean'|'max'` are pinned by no reference test: PARITY UNPINNED for those
two modes.

Data model (deliberately not the product's classes): a complex is a dict
    {'dimension': int, 'y': Tensor|None,
     'cochains': [ {'dim', 'x', 'upper_index', 'lower_index', 'shared_boundaries',
                    'shared_coboundaries', 'boundary_index', 'num_cells', 'num_cells_up',
                    'num_cells_down', 'batch'} , ...]}
with absent things stored as None.  All citations are relative to /root/reference.
"""
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

COCHAIN_KEYS = ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries',
                'boundary_index', 'y')


# --------------------------------------------------------------------------------------------
# K1 / K2: the two primitive ops
# --------------------------------------------------------------------------------------------
def lift(src: Tensor, rows: Tensor) -> Tensor:
    """mp/cell_mp.py:195-198  `src.index_select(node_dim, index[dim])`."""
    return src.index_select(0, rows)


def scatter_rows(msg: Tensor, dst: Tensor, n_dst: int, reduce: str = 'add') -> Tensor:
    """mp/cell_mp.py:439-440 -> torch_scatter.scatter(msg, dst, dim=-2, dim_size=n_dst, reduce).

    torch-scatter 2.0.5 semantics: rows that receive no message are zero for every reduce
    (mp/test_cell_mp.py:114-134 pins this for 'add'); duplicates count with multiplicity
    (mp/test_cell_mp.py:179-269); 'mean' divides by max(count, 1).
    """
    out = torch.zeros(n_dst, msg.size(1), dtype=msg.dtype)
    if reduce in ('add', 'sum'):
        return out.index_add_(0, dst, msg)
    if reduce == 'mean':
        out.index_add_(0, dst, msg)
        cnt = torch.bincount(dst, minlength=n_dst).clamp(min=1).to(msg.dtype)
        return out / cnt.unsqueeze(1)
    if reduce in ('max', 'min'):
        full = dst.unsqueeze(1).expand_as(msg)
        return out.scatter_reduce_(0, full, msg, 'amax' if reduce == 'max' else 'amin',
                                   include_self=False)
    raise ValueError(reduce)


def csr_from_coo(index: Tensor, n_dst: int):
    """Destination-sorted CSR of a COO `[2,E]` index (row 0 = source j, row 1 = destination i for
    flow source_to_target, mp/cell_mp.py:210).  Stable in the original entry order, so the
    summation order inside a segment equals the sequential index_add_ order.
    Returns int32 (rowptr[n_dst+1], col[E], perm[E]) with col = index[0][perm]."""
    src = index[0].numpy()
    dst = index[1].numpy()
    perm = np.argsort(dst, kind='stable')
    counts = np.bincount(dst, minlength=n_dst)
    rowptr = np.zeros(n_dst + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return (torch.from_numpy(rowptr.astype(np.int32)), torch.from_numpy(src[perm].astype(np.int32)),
            torch.from_numpy(perm.astype(np.int32)))


# --------------------------------------------------------------------------------------------
# a1-a7: propagate
# --------------------------------------------------------------------------------------------
def _identity_msg(x_j: Tensor, attr: Optional[Tensor]) -> Tensor:
    return x_j


def _check_index(index):
    """mp/cell_mp.py:153-184."""
    if index is None:
        return
    if not isinstance(index, Tensor):
        raise ValueError('only LongTensor [2, E] indices are supported')
    assert index.dtype == torch.long and index.dim() == 2 and index.size(0) == 2


def propagate(x: Tensor,
              up_index: Optional[Tensor], down_index: Optional[Tensor],
              boundary_index: Optional[Tensor],
              up_attr: Optional[Tensor] = None, down_attr: Optional[Tensor] = None,
              boundary_attr: Optional[Tensor] = None,
              message_up: Callable = _identity_msg, message_down: Callable = _identity_msg,
              message_boundary: Callable = lambda x_j: x_j,
              aggr_up: str = 'add', aggr_down: str = 'add', aggr_boundary: str = 'add',
              use_down_msg: bool = True, use_boundary_msg: bool = True,
              up_msg_size: Optional[int] = None, down_msg_size: Optional[int] = None,
              boundary_msg_size: Optional[int] = None):
    """mp/cell_mp.py:357-392 with __collect__ (:209-282) and update (:511-524) folded in.

    For each present adjacency: x_j = source rows gathered through index[0]; the hook builds the
    message; messages are scatter-reduced onto index[1] with dim_size = x.size(0).  The boundary
    stream gathers from `boundary_attr` (features of dim d-1, :228-236) and runs only when
    use_boundary_msg and boundary_attr is given (:381).  The down stream runs only when
    use_down_msg and down_index is given (:376).  Absent streams come back as zeros of the declared
    message width (:517-522)."""
    for idx in (up_index, down_index, boundary_index):
        _check_index(idx)
    n = x.size(0)
    up_out = down_out = boundary_out = None
    if up_index is not None:
        up_out = scatter_rows(message_up(lift(x, up_index[0]), up_attr), up_index[1], n, aggr_up)
    if use_down_msg and down_index is not None:
        down_out = scatter_rows(message_down(lift(x, down_index[0]), down_attr), down_index[1], n,
                                aggr_down)
    if use_boundary_msg and boundary_attr is not None:
        boundary_out = scatter_rows(message_boundary(lift(boundary_attr, boundary_index[0])),
                                    boundary_index[1], n, aggr_boundary)
    if boundary_msg_size is None:
        boundary_msg_size = down_msg_size
    if up_out is None:
        up_out = torch.zeros(n, up_msg_size, dtype=x.dtype)
    if down_out is None:
        down_out = torch.zeros(n, down_msg_size, dtype=x.dtype)
    if boundary_out is None:
        boundary_out = torch.zeros(n, boundary_msg_size, dtype=x.dtype)
    return up_out, down_out, boundary_out


# --------------------------------------------------------------------------------------------
# a10: cochain parameters (the K3 / K4 attribute gathers)
# --------------------------------------------------------------------------------------------
def cochain_params(cx: Dict, dim: int, max_dim: int = 2, include_top_features: bool = True,
                   include_down_features: bool = True,
                   include_boundary_features: bool = True) -> Dict:
    """data/complex.py:548-602."""
    cochains = cx['cochains']
    if dim >= len(cochains):
        raise NotImplementedError(f'Dim {dim} is not present in the complex')
    c = cochains[dim]
    has_up = dim + 1 < len(cochains)
    out = dict(x=c['x'], up_index=None, down_index=None, up_attr=None, down_attr=None,
               boundary_attr=None, boundary_index=None)
    if c['upper_index'] is not None and has_up:
        out['up_index'] = c['upper_index']
        xu = cochains[dim + 1]['x']
        if xu is not None and (dim < max_dim or include_top_features):
            out['up_attr'] = xu.index_select(0, c['shared_coboundaries'])
    if include_down_features and c['lower_index'] is not None:
        out['down_index'] = c['lower_index']
        if dim > 0 and cochains[dim - 1]['x'] is not None:
            out['down_attr'] = cochains[dim - 1]['x'].index_select(0, c['shared_boundaries'])
    if include_boundary_features and c['boundary_index'] is not None:
        out['boundary_index'] = c['boundary_index']
        if dim > 0 and cochains[dim - 1]['x'] is not None:
            out['boundary_attr'] = cochains[dim - 1]['x']
    return out


def all_cochain_params(cx: Dict, max_dim: int = 2, **kw) -> List[Dict]:
    """data/complex.py:604-626."""
    top = min(max_dim, cx['dimension'])
    return [cochain_params(cx, d, max_dim=max_dim, **kw) for d in range(top + 1)]


# --------------------------------------------------------------------------------------------
# a14: batching (integer index layout, bit-exact)
# --------------------------------------------------------------------------------------------
def _num_cells(c: Dict) -> Optional[int]:
    """data/complex.py:182-192."""
    if c.get('num_cells') is not None:
        return int(c['num_cells'])
    if c['x'] is not None:
        return int(c['x'].size(0))
    if c['boundary_index'] is not None:
        return int(c['boundary_index'][1].max()) + 1
    return None


def empty_cochain(dim: int) -> Dict:
    c = {k: None for k in COCHAIN_KEYS}
    c.update(dim=dim, num_cells=None, num_cells_up=None, num_cells_down=None, batch=None)
    return c


def consolidate(cx: Dict) -> Dict:
    """data/complex.py:518-537: fill num_cells_up / num_cells_down from the neighbouring dims."""
    cs = cx['cochains']
    for d, c in enumerate(cs):
        if d + 1 < len(cs) and c.get('num_cells_up') is None:
            c['num_cells_up'] = _num_cells(cs[d + 1])
        if d > 0 and c.get('num_cells_down') is None:
            c['num_cells_down'] = _num_cells(cs[d - 1])
    return cx


def batch_cochains(cochain_list: Sequence[Dict], dim: int) -> Dict:
    """data/complex.py:323-458 (+ the increments of :148-169).

    Per key, concatenate the per-complex tensors, first adding the running offset:
      upper_index / lower_index      += cells of this dim seen so far
      shared_boundaries              += cells of dim-1 seen so far
      shared_coboundaries            += cells of dim+1 seen so far
      boundary_index                 += [[cells of dim-1 so far], [cells of this dim so far]]
    Index-like keys concatenate along the last axis, everything else along axis 0.  `batch`
    numbers the complexes that have cells of this dim."""
    index_keys = ('upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries',
                  'boundary_index')
    parts = {k: [] for k in COCHAIN_KEYS}
    batch_vec = []
    off_here = off_down = off_up = 0
    for i, c in enumerate(cochain_list):
        n_here = _num_cells(c)
        for k in COCHAIN_KEYS:
            item = c.get(k)
            if item is None:
                continue
            if k in ('upper_index', 'lower_index'):
                item = item + off_here
            elif k == 'shared_boundaries':
                item = item + off_down
            elif k == 'shared_coboundaries':
                item = item + off_up
            elif k == 'boundary_index':
                item = item + torch.tensor([[off_down], [off_here]], dtype=item.dtype)
            parts[k].append(item)
        if n_here is not None:
            batch_vec.append(torch.full((n_here,), i, dtype=torch.long))
        # increments (data/complex.py:148-169): None counts as 0
        off_here += n_here or 0
        off_down += (c.get('num_cells_down') or 0) if dim > 0 else 0
        off_up += c.get('num_cells_up') or 0
    out = empty_cochain(dim)
    for k in COCHAIN_KEYS:
        if parts[k]:
            out[k] = torch.cat(parts[k], dim=-1 if k in index_keys else 0).contiguous()
    out['batch'] = torch.cat(batch_vec) if batch_vec else None
    out['num_cells'] = off_here
    out['num_cells_down'] = off_down if dim > 0 else None
    out['num_cells_up'] = off_up
    return out


def batch_complexes(complexes: Sequence[Dict], max_dim: int = 2) -> Dict:
    """data/complex.py:690-728.  A complex lacking some dim contributes an empty cochain that
    still carries num_cells_down (so later boundary indices are offset correctly, :709-716)."""
    dimension = min(max(c['dimension'] for c in complexes), max_dim)
    per_dim = [[] for _ in range(dimension + 1)]
    labels = []
    for cx in complexes:
        cs = cx['cochains']
        for d in range(dimension + 1):
            if d < len(cs) and d <= cx['dimension']:
                per_dim[d].append(cs[d])
            else:
                e = empty_cochain(d)
                if 0 <= d - 1 < len(cs):
                    e['num_cells_down'] = _num_cells(cs[d - 1])
                per_dim[d].append(e)
        labels.append(cx.get('y'))
    y = torch.cat(labels, 0) if all(l is not None for l in labels) else None
    out = {'dimension': dimension, 'y': y, 'num_complexes': len(complexes),
           'cochains': [batch_cochains(lst, d) for d, lst in enumerate(per_dim)]}
    # Complex._consolidate (data/complex.py:518-537) overwrites nothing that is already set.
    return out


# --------------------------------------------------------------------------------------------
# layers (functional, explicit weights)
# --------------------------------------------------------------------------------------------
def _bn(h: Tensor, p: Dict, prefix: str, training: bool, eps: float = 1e-5) -> Tensor:
    """torch.nn.BatchNorm1d forward: batch statistics (biased variance) in training mode,
    running statistics in eval mode."""
    w, b = p[prefix + '.weight'], p[prefix + '.bias']
    if training:
        mean, var = h.mean(0), h.var(0, unbiased=False)
    else:
        mean, var = p[prefix + '.running_mean'], p[prefix + '.running_var']
    return (h - mean) / torch.sqrt(var + eps) * w + b


def _lin(h: Tensor, p: Dict, prefix: str) -> Tensor:
    return h @ p[prefix + '.weight'].t() + p[prefix + '.bias']


def _mlp2(h: Tensor, p: Dict, prefix: str, training: bool, norm: str) -> Tensor:
    """mp/layers.py:303-321: Linear -> norm -> ReLU -> Linear -> norm -> ReLU (Sequential ids
    0..5)."""
    h = _lin(h, p, prefix + '.0')
    if norm == 'bn':
        h = _bn(h, p, prefix + '.1', training)
    h = torch.relu(h)
    h = _lin(h, p, prefix + '.3')
    if norm == 'bn':
        h = _bn(h, p, prefix + '.4', training)
    return torch.relu(h)


def sparse_cin_cochain_conv(p: Dict, prm: Dict, use_coboundaries: bool, training: bool = False,
                            norm: str = 'bn') -> Tensor:
    """SparseCINCochainConv.forward, mp/layers.py:184-214 + the default sub-networks of
    SparseCINConv (:286-325).  `p` is that level's state_dict (keys such as
    'msg_up_nn.1.weight', 'update_up_nn.0.weight', 'combine_nn.0.weight', 'eps1').
    `prm` is a cochain_params dict."""
    x = prm['x']
    width = x.size(1)

    def msg_up(x_j, attr):
        if not use_coboundaries:
            return x_j                                    # lambda xs: xs[0]   (:295)
        # Catter -> Linear(2F, F) -> ReLU                 (:290-293)
        return torch.relu(_lin(torch.cat([x_j, attr], dim=-1), p, 'msg_up_nn.1'))

    out_up, _, out_b = propagate(
        x, prm['up_index'], prm['down_index'], prm['boundary_index'],
        up_attr=prm['up_attr'], boundary_attr=prm['boundary_attr'],
        message_up=msg_up, use_down_msg=False,
        up_msg_size=width, down_msg_size=width, boundary_msg_size=width)
    out_up = out_up + (1 + p['eps1']) * x
    out_b = out_b + (1 + p['eps2']) * x
    out_up = _mlp2(out_up, p, 'update_up_nn', training, norm)
    out_b = _mlp2(out_b, p, 'update_boundaries_nn', training, norm)
    h = _lin(torch.cat([out_up, out_b], dim=-1), p, 'combine_nn.0')
    if norm == 'bn':
        h = _bn(h, p, 'combine_nn.1', training)
    return torch.relu(h)


def cinpp_cochain_conv(p: Dict, prm: Dict, use_coboundaries: bool, training: bool = False,
                       norm: str = 'bn') -> Tensor:
    """CINppCochainConv.forward, mp/layers.py:243-260, with the default sub-networks of CINppConv (:366-416).  The
    class inherits use_down_msg=False (:167-168) and its forward passes no down_attr (:244-247): the lower stream is
    zeros, so out_down is the self term (1 + eps2) x alone; the boundary stream takes eps3."""
    x = prm['x']
    width = x.size(1)

    def msg_up(x_j, attr):
        if not use_coboundaries:
            return x_j
        return torch.relu(_lin(torch.cat([x_j, attr], dim=-1), p, 'msg_up_nn.1'))

    out_up, out_down, out_b = propagate(
        x, prm['up_index'], prm['down_index'], prm['boundary_index'],
        up_attr=prm['up_attr'], boundary_attr=prm['boundary_attr'],
        message_up=msg_up, use_down_msg=False,
        up_msg_size=width, down_msg_size=width, boundary_msg_size=width)
    out_up = out_up + (1 + p['eps1']) * x
    out_down = out_down + (1 + p['eps2']) * x
    out_b = out_b + (1 + p['eps3']) * x
    out_up = _mlp2(out_up, p, 'update_up_nn', training, norm)
    out_down = _mlp2(out_down, p, 'update_down_nn', training, norm)
    out_b = _mlp2(out_b, p, 'update_boundaries_nn', training, norm)
    h = _lin(torch.cat([out_up, out_down, out_b], dim=-1), p, 'combine_nn.0')
    if norm == 'bn':
        h = _bn(h, p, 'combine_nn.1', training)
    return torch.relu(h)


def _level_state(state: Dict, level: int) -> Dict:
    pre = f'mp_levels.{level}.'
    return {k[len(pre):]: v for k, v in state.items() if k.startswith(pre)}


def sparse_cin_conv(state: Dict, params: List[Dict], use_coboundaries: bool,
                    training: bool = False, norm: str = 'bn', start_to_process: int = 0, conv: str = 'sparse_cin'):
    """SparseCINConv.forward, mp/layers.py:333-342 (`conv='cinpp'`: CINppConv.forward, :418-427)."""
    level = cinpp_cochain_conv if conv == 'cinpp' else sparse_cin_cochain_conv
    outs = []
    for d, prm in enumerate(params):
        if d < start_to_process:
            outs.append(prm['x'])
        else:
            outs.append(level(_level_state(state, d), prm, use_coboundaries, training, norm))
    return outs


def cin_cochain_conv(prm: Dict, msg_up_nn: Callable, msg_down_nn: Callable, update_nn: Callable,
                     eps: Tensor) -> Tensor:
    """CINCochainConv.forward, mp/layers.py:78-103 (use_boundary_msg=False)."""
    x = prm['x']

    def mu(x_j, attr):
        return msg_up_nn(x_j if attr is None else torch.cat([x_j, attr], dim=-1))

    def md(x_j, attr):
        return msg_down_nn(torch.cat([x_j, attr], dim=-1))

    # the message width is whatever the message nets emit; absent streams take the declared size
    width = x.size(1)
    up, down, _ = propagate(x, prm['up_index'], prm['down_index'], None,
                            up_attr=prm['up_attr'], down_attr=prm['down_attr'],
                            message_up=mu, message_down=md, use_boundary_msg=False,
                            up_msg_size=width, down_msg_size=width)
    up = up + (1 + eps) * x
    down = down + (1 + eps) * x
    return update_nn(up + down)


def dummy_cochain_mp(prm: Dict, use_boundary_msg: bool = False, use_down_msg: bool = True) -> Tensor:
    """DummyCochainMessagePassing.forward, mp/layers.py:14-40: messages are x_j + attr."""
    x = prm['x']
    w = x.size(1)
    up, down, bnd = propagate(
        x, prm['up_index'], prm['down_index'], prm['boundary_index'],
        up_attr=prm['up_attr'], down_attr=prm['down_attr'], boundary_attr=prm['boundary_attr'],
        message_up=lambda xj, a: xj + a, message_down=lambda xj, a: xj + a,
        use_down_msg=use_down_msg, use_boundary_msg=use_boundary_msg,
        up_msg_size=w, down_msg_size=w, boundary_msg_size=w)
    return x + up + down + bnd


def init_reduce(boundary_x: Tensor, boundary_index: Tensor, reduce: str = 'add') -> Tensor:
    """InitReduceConv.forward, mp/layers.py:484-487: dim_size = max(destination) + 1."""
    n = int(boundary_index[1].max()) + 1
    return scatter_rows(lift(boundary_x, boundary_index[0]), boundary_index[1], n, reduce)


def oriented_conv_messages(x: Tensor, up_index, down_index, up_orient, down_orient, orient=True):
    """OrientedConv.propagate with message x_j * attr, mp/layers.py:448-470."""
    mul = (lambda xj, a: xj * a) if orient else (lambda xj, a: xj)
    w = x.size(1)
    up, down, _ = propagate(x, up_index, down_index, None,
                            up_attr=None if up_orient is None else up_orient.view(-1, 1),
                            down_attr=None if down_orient is None else down_orient.view(-1, 1),
                            message_up=mul, message_down=mul, use_boundary_msg=False,
                            up_msg_size=w, down_msg_size=w)
    return up, down


def pool_complex(xs: List[Tensor], batches: List[Tensor], num_complexes: int, max_dim: int,
                 readout: str = 'sum') -> Tensor:
    """mp/nn.py:50-60: per-dimension global add / mean pool into [max_dim+1, B, H]."""
    out = torch.zeros(max_dim + 1, num_complexes, xs[0].size(-1), dtype=xs[0].dtype)
    for d, x in enumerate(xs):
        out[d] = scatter_rows(x, batches[d], num_complexes, 'add' if readout == 'sum' else 'mean')
    return out


def embed_ve_with_reduce(v_emb: Tensor, e_emb: Optional[Tensor], params: List[Dict],
                         reduce: str = 'add') -> List[Tensor]:
    """EmbedVEWithReduce.forward, mp/layers.py:516-543 with Embedding tables given as matrices."""
    v = params[0]
    vx = v_emb.index_select(0, v['x'].squeeze(1).long())
    out = [vx]
    if len(params) < 2:
        return out
    e = params[1]
    reduced_ex = init_reduce(vx, e['boundary_index'], reduce)
    ex = reduced_ex
    if e['x'] is not None:
        ex = e_emb.index_select(0, e['x'].squeeze(1).long())
    out.append(ex)
    if len(params) == 3:
        out.append(init_reduce(reduced_ex, params[2]['boundary_index'], reduce) / 2.)
    return out


def _sum_embedding(state: Dict, prefix: str, x: Tensor) -> Tensor:
    """OGB Atom/BondEncoder convention (third-party, absent from the tree: PARITY UNPINNED for this
    front-end): sum over integer feature columns of one Embedding table per column."""
    x = x.long()
    return sum(state[f'{prefix}.{i}.weight'].index_select(0, x[:, i]) for i in range(x.size(1)))


def sparse_cin_model_forward(state: Dict, cx: Dict, num_layers: int, max_dim: int = 2,
                             use_coboundaries: bool = True, readout: str = 'sum',
                             final_readout: str = 'sum', training: bool = False, norm: str = 'bn',
                             jump_mode: Optional[str] = None, embed: Optional[str] = 'zinc',
                             init_reduce_mode: str = 'add', readout_dims=(0, 1, 2), conv: str = 'sparse_cin',
                             drop_edge_up: bool = False, dropout: Optional[Dict] = None, drop_position: str = 'lin2'):
    """SparseCIN.forward (mp/models.py:195-260), EmbedSparseCIN.forward (mp/molec_models.py:90-160)
    and OGBEmbedSparseCIN.forward (mp/molec_models.py:281-350) with dropout off and jump_mode in
    {None, 'cat', 'max'}.  `embed`: None (features used as they are), 'zinc' (one Embedding per dimension
    0/1) or 'ogb' (sum of per-column embeddings).  `conv='cinpp'`: EmbedCINpp / OGBEmbedCINpp (mp/molec_models.py:167-199,
    355-384: the same forward over CINppConv layers).  `drop_edge_up` (with max_dim 1): EmbedSparseCINNoRings
    (mp/molec_models.py:386-503: the edges' upper adjacency is removed from every layer's parameters, :456-457, 471-472).
    `dropout`: the MULTIPLIERS (0 or 1 / (1 - p), a tensor per application) F.dropout would have drawn, handed in by the caller so
    that a training forward with active dropout can be checked: keys ('in', d) -- the input features of dimension d
    (mp/molec_models.py:104-106, 290-292), ('conv', l, d) -- the output of conv layer l, OGBEmbedSparseCIN only (:297-300),
    ('head', k) -- the head's single application at `drop_position` (:129-146 / :334-346): before lin1 of readout dimension k
    ('lin1'), on relu(lin1) of readout dimension k ('final_readout'), or ('head',) on the summed hidden vector (else).
    Returns (out, per-layer / pooled tensors)."""
    dropout = dropout or {}
    cx = {'dimension': cx['dimension'], 'y': cx.get('y'), 'num_complexes': cx.get('num_complexes'),
          'cochains': [dict(c) for c in cx['cochains']]}
    partial = {}
    def layer_params():
        params = all_cochain_params(cx, max_dim=max_dim, include_down_features=False)
        if drop_edge_up and len(params) > 1:
            params[1]['up_index'] = None
        return params

    if embed is not None:
        params = layer_params()
        if embed == 'zinc':
            xs = embed_ve_with_reduce(state['v_embed_init.weight'], state.get('e_embed_init.weight'),
                                      params, init_reduce_mode)
        else:
            vx = _sum_embedding(state, 'v_embed_init.atom_embedding_list', params[0]['x'])
            xs = [vx]
            if len(params) >= 2:
                reduced = init_reduce(vx, params[1]['boundary_index'], init_reduce_mode)
                ex = reduced
                if params[1]['x'] is not None:
                    ex = _sum_embedding(state, 'e_embed_init.bond_embedding_list', params[1]['x'])
                xs.append(ex)
                if len(params) == 3:
                    xs.append(init_reduce(reduced, params[2]['boundary_index'], init_reduce_mode) / 2.)
        xs = [x * dropout[('in', d)] if ('in', d) in dropout else x for d, x in enumerate(xs)]
        for d, x in enumerate(xs):
            cx['cochains'][d]['x'] = x
    jump = None
    for l in range(num_layers):
        params = layer_params()
        pre = f'convs.{l}.'
        lstate = {k[len(pre):]: v for k, v in state.items() if k.startswith(pre)}
        xs = sparse_cin_conv(lstate, params, use_coboundaries, training, norm, conv=conv)
        xs = [x * dropout[('conv', l, d)] if ('conv', l, d) in dropout else x for d, x in enumerate(xs)]
        for d, x in enumerate(xs):
            cx['cochains'][d]['x'] = x
            partial[f'layer{l}_{d}'] = x
        if jump_mode is not None:
            jump = [[] for _ in xs] if jump is None else jump
            for d, x in enumerate(xs):
                jump[d].append(x)
    if jump_mode == 'cat':
        xs = [torch.cat(j, dim=-1) for j in jump]
    elif jump_mode == 'max':
        # torch_geometric.nn.JumpingKnowledge('max') (torch_geometric is not in /root/reference nor in this
        # image: its published forward is `torch.stack(xs, dim=-1).max(dim=-1)[0]`; parity unpinned for this mode)
        xs = [torch.stack(j, dim=-1).max(dim=-1)[0] for j in jump]
    elif jump_mode is not None:
        raise NotImplementedError(jump_mode)
    nb = cx.get('num_complexes') or int(cx['cochains'][0]['batch'].max()) + 1
    pooled = pool_complex(xs, [cx['cochains'][d]['batch'] for d in range(len(xs))], nb, max_dim, readout)
    dims = [d for d in readout_dims if d <= max_dim]
    for k, d in enumerate(dims):
        partial[f'pool_{k}'] = pooled[d]
    hs = []
    for k, d in enumerate(dims):
        pin = pooled[d]
        if drop_position == 'lin1' and ('head', k) in dropout:
            pin = pin * dropout[('head', k)]
        h = pin @ state[f'lin1s.{d}.weight'].t()
        if f'lin1s.{d}.bias' in state:
            h = h + state[f'lin1s.{d}.bias']
        h = torch.relu(h)
        if drop_position == 'final_readout' and ('head', k) in dropout:
            h = h * dropout[('head', k)]
        hs.append(h)
    h = torch.stack(hs, 0)
    h = h.sum(0) if final_readout == 'sum' else h.mean(0)
    if drop_position not in ('lin1', 'final_readout') and ('head',) in dropout:
        h = h * dropout[('head',)]
    return _lin(h, state, 'lin2'), partial


def embed_sparse_cin_forward(state: Dict, cx: Dict, num_layers: int, max_dim: int = 2,
                             use_coboundaries: bool = True, readout: str = 'sum',
                             final_readout: str = 'sum', training: bool = False,
                             norm: str = 'bn', init_reduce_mode: str = 'add',
                             embed_edge: bool = True, readout_dims=(0, 1, 2)):
    """EmbedSparseCIN.forward, mp/molec_models.py:90-160 (see sparse_cin_model_forward)."""
    return sparse_cin_model_forward(state, cx, num_layers, max_dim, use_coboundaries, readout,
                                    final_readout, training, norm, None, 'zinc', init_reduce_mode,
                                    readout_dims)


# --------------------------------------------------------------------------------------------
# algorithmic byte count
# --------------------------------------------------------------------------------------------
def propagate_algorithmic_bytes(F: int, n_cells: int, e_up: int = 0, e_down: int = 0, b: int = 0,
                                coboundary: bool = False, down_attr: bool = False,
                                streams: int = 2) -> int:
    """E_up(16+4F) + [cob] E_up(8+4F) + B(16+4F) + E_down(16+4F) [+ E_down(8+4F)] + 4F N streams."""
    t = e_up * (16 + 4 * F) + b * (16 + 4 * F) + e_down * (16 + 4 * F)
    if coboundary:
        t += e_up * (8 + 4 * F)
    if down_attr:
        t += e_down * (8 + 4 * F)
    return t + 4 * F * n_cells * streams
