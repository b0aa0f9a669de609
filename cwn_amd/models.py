"""Callers of the hot path: the molecular SparseCIN model and the mp/nn.py helpers.

`EmbedSparseCIN` mirrors the reference's mp/molec_models.py:12-163 (same constructor arguments,
parameter names and forward(data, include_partial)) so its state_dict loads unchanged; it exists
here because BASELINE's configs name it ("ZINC ring-lift, 4-layer SparseCIN") and the parity tests
pin the whole stack against golden vectors of the reference model.  `SparseCIN` mirrors
mp/models.py:120-260 for non-embedded inputs (REDDIT-like config).  The readout
(`pool_complex`, mp/nn.py:50-60) runs on the same segmented-reduce kernel as propagate.
"""
from typing import List

import weakref

import torch
import torch.nn.functional as F
from torch.nn import BatchNorm1d as BN, Embedding, Identity, LayerNorm as LN, Linear

from . import _ffi, layers, ops
from .layers import reset as reset_net
from .complex import ComplexBatch
from .csr import cached_adjacency, deferred_checks
from .layers import CINConv, EdgeCINConv, EmbedVEWithReduce, InitReduceConv, SparseCINConv


_HEAD_CACHE = layers.ByModule()      # model -> {id(plan): (ops.HeadLaunch, the batch's BlockPlan, head signature, readout dims)}


def _one_check(forward):
    """A model's forward reads the device's error word once, at its end (csr.deferred_checks), instead of once per validated
    launch in the middle of its launches."""
    import functools

    @functools.wraps(forward)
    def wrapped(self, data, *args, **kwargs):
        with deferred_checks():
            return forward(self, data, *args, **kwargs)
    return wrapped


def get_nonlinearity(nonlinearity, return_module=True):
    """mp/nn.py:7-28."""
    table = {'relu': (torch.nn.ReLU, F.relu), 'elu': (torch.nn.ELU, F.elu),
             'id': (torch.nn.Identity, lambda x: x), 'sigmoid': (torch.nn.Sigmoid, torch.sigmoid),
             'tanh': (torch.nn.Tanh, torch.tanh)}
    if nonlinearity not in table:
        raise NotImplementedError(f'Nonlinearity {nonlinearity} is not currently supported.')
    return table[nonlinearity][0 if return_module else 1]


def get_graph_norm(norm):
    """mp/nn.py:39-47."""
    if norm == 'bn':
        return BN
    if norm == 'ln':
        return LN
    if norm == 'id':
        return Identity
    raise ValueError(f'Graph Normalisation {norm} not currently supported')


def _batch_adjacency(batch: torch.Tensor, size: int, n_rows: int, build: bool = True):
    """CSR keyed on the batch vector (cells of complex b = segment b), cached on the tensor."""
    ids = getattr(batch, '_cwn_ids', None)
    if ids is None or ids.numel() != batch.numel():
        ids = torch.arange(batch.numel(), device=batch.device)
        batch._cwn_ids, batch._cwn_index = ids, torch.stack([ids, batch])
    return cached_adjacency(batch._cwn_index, size, n_rows, build=build)


def global_pool(x: torch.Tensor, batch: torch.Tensor, size: int, mean: bool = False) -> torch.Tensor:
    """global_add_pool / global_mean_pool (K11) as a segmented reduce keyed on the batch vector."""
    return ops.aggregate(_batch_adjacency(batch, size, x.size(0)), size, x,
                         reduce='mean' if mean else 'add')


def pool_complex_list(xs: List[torch.Tensor], data: ComplexBatch, max_dim: int, readout_type: str):
    """Per-dimension readout of all dimensions in ONE segmented-reduce launch; absent dimensions
    give zero rows (mp/nn.py:55-56)."""
    if readout_type not in ('sum', 'mean'):
        raise NotImplementedError(f'Readout {readout_type} is not currently supported.')
    batch_size = data.num_complexes
    if batch_size is None:
        batch_size = int(data.cochains[0].batch.max()) + 1
    red = 'mean' if readout_type == 'mean' else 'add'
    # the plans of all dimensions -- and, when a backward will follow, their transposes -- in ONE batched build (they
    # were four separate launches of ~10 us in every training step)
    adjs = [_batch_adjacency(data.cochains[i].batch, batch_size, xs[i].size(0), build=False) for i in range(len(xs))]
    todo = [a for a in adjs if not a.built]
    if torch.is_grad_enabled() and any(x.requires_grad for x in xs):
        for a in adjs:
            a.transposes()
            todo += [t for t in (a._t_src, a._t_aux) if t is not None and not t.built]
    if todo:
        from .csr import build_many
        build_many(todo)
    streams = [ops.Stream(adj=adjs[i], n_dst=batch_size, width=int(xs[i].size(1)), A=xs[i], reduce=red)
               for i in range(len(xs))]
    pooled = ops.aggregate_many(streams)
    for _ in range(len(xs), max_dim + 1):
        pooled.append(torch.zeros_like(pooled[0]))
    return pooled


def pool_complex(xs: List[torch.Tensor], data: ComplexBatch, max_dim: int, readout_type: str):
    """mp/nn.py:50-60 -> [max_dim+1, num_complexes, H]; rows of absent dimensions stay zero.
    The batch size comes from the container instead of `batch.max() + 1` (a device sync)."""
    return torch.stack(pool_complex_list(xs, data, max_dim, readout_type), dim=0)


ATOM_DIMS = (119, 4, 12, 12, 10, 6, 6, 2, 2)   # OGB convention (third-party)
BOND_DIMS = (5, 6, 2)


class AtomEncoder(torch.nn.Module):
    """Sum of one Embedding per integer atom-feature column (the published OGB AtomEncoder form;
    `ogb` is absent here, parameter names follow it so its state_dicts load)."""

    def __init__(self, emb_dim, dims=ATOM_DIMS):
        super().__init__()
        self.atom_embedding_list = torch.nn.ModuleList(Embedding(d, emb_dim) for d in dims)
        for emb in self.atom_embedding_list:
            torch.nn.init.xavier_uniform_(emb.weight.data)

    def reset_parameters(self):
        for emb in self.atom_embedding_list:
            torch.nn.init.xavier_uniform_(emb.weight.data)

    def forward(self, x):
        return sum(self.atom_embedding_list[i](x[:, i]) for i in range(x.shape[1]))


class BondEncoder(torch.nn.Module):
    def __init__(self, emb_dim, dims=BOND_DIMS):
        super().__init__()
        self.bond_embedding_list = torch.nn.ModuleList(Embedding(d, emb_dim) for d in dims)
        for emb in self.bond_embedding_list:
            torch.nn.init.xavier_uniform_(emb.weight.data)

    def reset_parameters(self):
        for emb in self.bond_embedding_list:
            torch.nn.init.xavier_uniform_(emb.weight.data)

    def forward(self, edge_attr):
        return sum(self.bond_embedding_list[i](edge_attr[:, i]) for i in range(edge_attr.shape[1]))


class _SparseCINStack(torch.nn.Module):
    """What SparseCIN (mp/models.py:112-260), EmbedSparseCIN (mp/molec_models.py:12-163) and
    OGBEmbedSparseCIN (mp/molec_models.py:201-352) share: L x SparseCINConv, optional JK-cat,
    per-dimension readout, lin1s, final readout, lin2.  Subclasses provide the input front."""

    def _build(self, first_dim, out_size, num_layers, hidden, dropout_rate, max_dim, jump_mode,
               nonlinearity, readout, train_eps, final_hidden_multiplier, readout_dims,
               final_readout, apply_dropout_before, use_coboundaries, graph_norm):
        self.max_dim = max_dim
        self.readout_dims = (tuple(d for d in readout_dims if d <= max_dim)
                             if readout_dims is not None else list(range(max_dim + 1)))
        self.final_readout = final_readout
        self.dropout_rate = dropout_rate
        self.apply_dropout_before = apply_dropout_before
        self.jump_mode = jump_mode
        # torch_geometric's JumpingKnowledge as the reference constructs it (mp/models.py:351: no `channels`,
        # so 'lstm' cannot be built there either): 'cat' and the elementwise 'max' over the layers
        if jump_mode not in (None, 'cat', 'max'):
            raise NotImplementedError("jump_mode must be None, 'cat' or 'max'")
        self.nonlinearity = nonlinearity
        self.readout = readout
        self.graph_norm = get_graph_norm(graph_norm)
        act_module = get_nonlinearity(nonlinearity, return_module=True)
        self.convs = torch.nn.ModuleList()
        for i in range(num_layers):
            layer_dim = first_dim if i == 0 else hidden
            self.convs.append(self._make_conv(layer_dim, hidden, act_module, train_eps, use_coboundaries))
        self.lin1s = torch.nn.ModuleList()
        for _ in range(max_dim + 1):
            if jump_mode == 'cat':   # no bias: an absent level contributes exactly zero
                self.lin1s.append(Linear(num_layers * hidden, final_hidden_multiplier * hidden, bias=False))
            else:
                self.lin1s.append(Linear(hidden, final_hidden_multiplier * hidden))
        self.lin2 = Linear(final_hidden_multiplier * hidden, out_size)

    conv_dropout = False       # OGBEmbedSparseCIN drops out after every conv (:298-300)

    def _edit_params(self, params):
        return params              # (EmbedSparseCINNoRings drops the edges' upper adjacency)

    def _make_conv(self, layer_dim, hidden, act_module, train_eps, use_coboundaries):
        return SparseCINConv(
            up_msg_size=layer_dim, down_msg_size=layer_dim, boundary_msg_size=layer_dim,
            passed_msg_boundaries_nn=None, passed_msg_up_nn=None, passed_update_up_nn=None,
            passed_update_boundaries_nn=None, train_eps=train_eps, max_dim=self.max_dim,
            hidden=hidden, act_module=act_module, layer_dim=layer_dim,
            graph_norm=self.graph_norm, use_coboundaries=use_coboundaries)

    def _convs_and_head(self, data: ComplexBatch, include_partial: bool, res: dict):
        act = get_nonlinearity(self.nonlinearity, return_module=False)
        jump_xs, xs = None, None
        if torch.is_grad_enabled() and layers.BLOCKED_TRAIN_FORWARD and layers.BLOCKED_LAYER:
            # training forward through the blocked layer kernel: the message weights of all layers packed in one launch
            ws = [lvl.msg_up_nn[1].weight for conv in self.convs          # (CIN++ layers too: round 6)
                  for lvl in getattr(conv, 'mp_levels', [])
                  if getattr(lvl, '_up_kind', lambda: None)() == 'cat_linear_relu' and lvl.msg_up_nn[1].weight.is_cuda]
            if ws:
                ops.pack_layer_weights_many(ws, transposed=bool(ops.BLOCKED_BACKWARD))
        if torch.is_grad_enabled() and ops.STAGE_KERNEL and layers.FUSED_DENSE_TRAINING:
            # ... and the blocks of the update / combine Linear layers (cwn_dense_stage_f32 streams them pre-split)
            sw = []
            for conv in self.convs:
                if isinstance(conv, layers.CINppConv):
                    continue                  # (packs its own blocks per forward, layers.CINppConv._dense_train)
                for lvl in getattr(conv, 'mp_levels', []):
                    for net in (getattr(lvl, 'update_up_nn', None), getattr(lvl, 'update_boundaries_nn', None),
                                getattr(lvl, 'combine_nn', None)):
                        st = layers._mlp_stages(net) if net is not None else None
                        for lin, _ in (st or []):
                            if lin.weight.is_cuda and lin.weight.requires_grad:
                                sw.append(lin.weight)
            if sw:
                ops.pack_stage_weights_many(sw)
        for c, conv in enumerate(self.convs):
            params = self._edit_params(data.get_all_cochain_params(max_dim=self.max_dim, include_down_features=False))
            if self.conv_dropout and self.training and self.dropout_rate > 0 and isinstance(conv, SparseCINConv):
                # (the layer's own last launch applies it: layers.SparseCINConv.forward(out_dropout=))
                xs = conv(*params, start_to_process=0, out_dropout=self.dropout_rate)
            else:
                xs = conv(*params, start_to_process=0)
                if self.conv_dropout:
                    xs = [ops.dropout(x, self.dropout_rate, self.training) for x in xs]
            data.set_xs(xs)
            if include_partial:
                for k in range(len(xs)):
                    res[f'layer{c}_{k}'] = xs[k]
            if self.jump_mode is not None:
                if jump_xs is None:
                    jump_xs = [[] for _ in xs]
                for i, x in enumerate(xs):
                    jump_xs[i] += [x]
        if self.jump_mode == 'cat':
            # (the fused head reads the layers' outputs block by block: the concatenation of mp/models.py:222-232 is only
            #  written when that head does not apply)
            xs = [list(j) if len(j) > 1 else j[0] for j in jump_xs]
        elif self.jump_mode == 'max':
            xs = [torch.stack(j, dim=-1).max(dim=-1)[0] for j in jump_xs]
        fused = self._head_fused(xs, data, include_partial, res)
        if fused is None and self.jump_mode == 'cat':
            xs = [torch.cat(x, dim=-1) if isinstance(x, list) else x for x in xs]
        if fused is not None:
            if include_partial:
                res['out'] = fused
                return fused, res
            return fused
        pooled = pool_complex_list(xs, data, self.max_dim, self.readout)
        xs = [pooled[i] for i in self.readout_dims]
        if include_partial:
            for k in range(len(xs)):
                res[f'pool_{k}'] = xs[k]
        if self.final_readout not in ('mean', 'sum'):
            raise NotImplementedError
        lins = [self.lin1s[d] for d in self.readout_dims]
        dense_head = (self.nonlinearity == 'relu' and xs[0].is_cuda
                      and max(l.in_features for l in lins + [self.lin2]) <= ops.GEMM_MAX_K)
        if self.apply_dropout_before == 'lin1':
            xs = [ops.dropout(x, self.dropout_rate, self.training) for x in xs]
        if dense_head:     # lin1s of all dimensions (+ReLU) in ONE grouped MFMA launch
            new_xs = ops.gemm_many([ops.Gemm(X=x, W=l.weight, bias=l.bias, relu=True)
                                    for x, l in zip(xs, lins)])
        else:
            new_xs = [act(l(x)) for x, l in zip(xs, lins)]
        if self.apply_dropout_before == 'final_readout':
            new_xs = [ops.dropout(x, self.dropout_rate, self.training) for x in new_xs]
        x = new_xs[0]
        for t in new_xs[1:]:
            x = x + t
        if self.final_readout == 'mean':
            x = x / len(new_xs)
        if self.apply_dropout_before not in ['lin1', 'final_readout']:
            x = ops.dropout(x, self.dropout_rate, self.training)
        if dense_head:
            x, = ops.gemm_many([ops.Gemm(X=x, W=self.lin2.weight, bias=self.lin2.bias)])
        else:
            x = self.lin2(x)
        if include_partial:
            res['out'] = x
            return x, res
        return x

    def _head_fused(self, xs, data: ComplexBatch, include_partial: bool, res: dict):
        """Readout, lin1s (+ReLU), final readout and lin2 in ONE launch, one workgroup per complex (ops.head,
        csrc/cwn_ends.hip) -- 5 launches / 43 us of a 167 us forward at the ZINC batch of 128 before.  Inference
        (no autograd, no active dropout) on a batch that carries the collate's per-complex tables; None otherwise."""
        first = lambda x: x[0] if isinstance(x, list) else x
        width = lambda x: sum(int(b.size(1)) for b in x) if isinstance(x, list) else int(x.size(1))
        # the prepared launch of (this model, this batch): ops.HeadLaunch -- inference without side outputs
        plain = not include_partial and not torch.is_grad_enabled() and not (self.training and self.dropout_rate > 0)
        sig = (self.readout, self.final_readout, tuple(self.readout_dims), self.nonlinearity, self.jump_mode, ops.FUSED_ENDS)
        plan = data.block_plan() if plain else None
        ent = _HEAD_CACHE.get(self, {}).get(id(plan)) if plan is not None else None
        tried = False
        if ent is not None and ent[2] == sig and ent[1] is plan and ent[0].current():
            tried = True
            out = ent[0].run([xs[d] if d < len(xs) else None for d in ent[3]])
            if out is not None:
                return out
        if not ops.FUSED_ENDS or self.nonlinearity != 'relu' or not xs or not first(xs[0]).is_cuda:
            return None
        if self.readout not in ('sum', 'mean') or self.final_readout not in ('sum', 'mean'):
            return None
        # the head's dropout (`apply_dropout_before`, mp/molec_models.py:129-146) rides inside the launch (cwn_dropout)
        drop_p = float(self.dropout_rate) if (self.training and self.dropout_rate > 0) else 0.0
        if drop_p >= 1.0:
            return None
        drop_pos = {'lin1': _ffi.HEAD_DROP_LIN1, 'final_readout': _ffi.HEAD_DROP_FINAL}.get(self.apply_dropout_before, _ffi.HEAD_DROP_LIN2)
        rd = list(self.readout_dims)
        if not rd or len(rd) > 3:
            return None
        lins = [self.lin1s[d] for d in rd]
        train = torch.is_grad_enabled() and (any(p.requires_grad for l in lins + [self.lin2] for p in l.parameters())
                                             or any(b.requires_grad for x in xs for b in (x if isinstance(x, list) else [x])))
        if train and (not ops.FUSED_HEAD_TRAINING or any(d >= len(xs) for d in rd) or self.lin2.bias is None
                      or len({l.bias is None for l in lins}) != 1):
            return None              # (absent dimensions: the unfused autograd path; bias-free lin1s -- jump_mode 'cat' -- ride along)
        plan = data.block_plan()
        if plan is None or data.num_complexes is None or plan.C != data.num_complexes:
            return None
        K, H2 = lins[0].in_features, lins[0].out_features
        if K % 4 != 0 or K > 2048 or H2 % 4 != 0 or H2 > 512 or any(l.in_features != K or l.out_features != H2 for l in lins):
            return None
        if any(first(x).dtype != torch.float32 or first(x).dim() != 2 or width(x) != K for x in xs):
            return None
        if any(isinstance(x, list) and (len(x) > _ffi.HEAD_MAX_PARTS or any(b.size(1) != x[0].size(1) or b.size(1) % 4 for b in x))
               for x in xs):
            return None
        dev = first(xs[0]).device
        hx = [xs[d] if d < len(xs) and d < plan.n_dims and first(xs[d]).size(0) == int(plan.cell_ptr[d][-1]) else None for d in rd]
        if any(d < len(xs) and h is None for d, h in zip(rd, hx)):
            return None                    # a feature matrix that is not the batch's own rows
        ptrs = [plan.cell_ptr_device(d, dev) if h is not None else None for d, h in zip(rd, hx)]
        if train:
            out, pooled = ops.head_train(hx, ptrs, plan.C, [l.weight for l in lins], [l.bias for l in lins],
                                         self.lin2.weight, self.lin2.bias, mean_readout=self.readout == 'mean',
                                         mean_final=self.final_readout == 'mean', drop_p=drop_p, drop_pos=drop_pos)
            if include_partial:
                for k in range(len(rd)):
                    res[f'pool_{k}'] = pooled[k]
            return out
        if plain and not tried and not _ffi.DYN_ROWS:
            launch = ops.HeadLaunch(hx, ptrs, plan.C, [l.weight for l in lins], [l.bias for l in lins], self.lin2.weight,
                                    self.lin2.bias, self.readout == 'mean', self.final_readout == 'mean')
            cache = _HEAD_CACHE.setdefault(self, {})
            if len(cache) >= 16:
                cache.clear()
            cache[id(plan)] = (launch, plan, sig, rd)
            out = launch.run(hx)
            if out is not None:
                return out
        got = ops.head(hx, ptrs, plan.C, [l.weight for l in lins], [l.bias for l in lins], self.lin2.weight, self.lin2.bias,
                       mean_readout=self.readout == 'mean', mean_final=self.final_readout == 'mean',
                       want_pooled=include_partial, drop=ops.dropout_record(dev, drop_p, tag=('head', drop_pos)) if drop_p > 0 else None, drop_pos=drop_pos)
        if include_partial:
            out, pooled = got
            for k in range(len(rd)):
                res[f'pool_{k}'] = pooled[k]
            return out
        return got

    def __repr__(self):
        return self.__class__.__name__


class SparseCIN(_SparseCINStack):
    """mp/models.py:112-260: features are used as given (e.g. REDDIT: constant scalar features)."""

    def __init__(self, num_input_features, num_classes, num_layers, hidden, dropout_rate: float = 0.5,
                 max_dim: int = 2, jump_mode=None, nonlinearity='relu', readout='sum', train_eps=False,
                 final_hidden_multiplier: int = 2, use_coboundaries=False, readout_dims=(0, 1, 2),
                 final_readout='sum', apply_dropout_before='lin2', graph_norm='bn'):
        super().__init__()
        self._build(num_input_features, num_classes, num_layers, hidden, dropout_rate, max_dim,
                    jump_mode, nonlinearity, readout, train_eps, final_hidden_multiplier, readout_dims,
                    final_readout, apply_dropout_before, use_coboundaries, graph_norm)

    @_one_check
    def forward(self, data: ComplexBatch, include_partial=False):
        return self._convs_and_head(data, include_partial, {})


class EmbedSparseCIN(_SparseCINStack):
    """mp/molec_models.py:12-163 (ZINC: integer atom / bond types embedded, rings initialised by
    reduction)."""

    def __init__(self, atom_types, bond_types, out_size, num_layers, hidden,
                 dropout_rate: float = 0.5, max_dim: int = 2, jump_mode=None, nonlinearity='relu',
                 readout='sum', train_eps=False, final_hidden_multiplier: int = 2,
                 readout_dims=(0, 1, 2), final_readout='sum', apply_dropout_before='lin2',
                 init_reduce='sum', embed_edge=False, embed_dim=None, use_coboundaries=False,
                 graph_norm='bn'):
        super().__init__()
        if embed_dim is None:
            embed_dim = hidden
        self.v_embed_init = Embedding(atom_types, embed_dim)
        self.e_embed_init = Embedding(bond_types, embed_dim) if embed_edge else None
        self.reduce_init = InitReduceConv(reduce=init_reduce)
        self.init_conv = EmbedVEWithReduce(self.v_embed_init, self.e_embed_init, self.reduce_init)
        self._build(embed_dim, out_size, num_layers, hidden, dropout_rate, max_dim, jump_mode,
                    nonlinearity, readout, train_eps, final_hidden_multiplier, readout_dims,
                    final_readout, apply_dropout_before, use_coboundaries, graph_norm)

    @_one_check
    def forward(self, data: ComplexBatch, include_partial=False):
        assert data.cochains[0].x.size(-1) == 1
        if 1 in data.cochains and data.cochains[1].x is not None:
            assert data.cochains[1].x.size(-1) == 1
        params = self._edit_params(data.get_all_cochain_params(max_dim=self.max_dim, include_down_features=False))
        xs = list(self.init_conv(*params))
        xs = [ops.dropout(x, self.dropout_rate, self.training) for x in xs]
        data.set_xs(xs)
        return self._convs_and_head(data, include_partial, {})


class EmbedSparseCINNoRings(EmbedSparseCIN):
    """mp/molec_models.py:386-503: the same stack on vertices and edges only -- max_dim 1, and the edges' upper adjacency (the
    rings they bound) dropped from every layer's parameters (:456-457, 471-472)."""

    def __init__(self, atom_types, bond_types, out_size, num_layers, hidden, dropout_rate: float = 0.5, nonlinearity='relu',
                 readout='sum', train_eps=False, final_hidden_multiplier: int = 2, final_readout='sum',
                 apply_dropout_before='lin2', init_reduce='sum', embed_edge=False, embed_dim=None, use_coboundaries=False,
                 graph_norm='bn'):
        super().__init__(atom_types, bond_types, out_size, num_layers, hidden, dropout_rate=dropout_rate, max_dim=1,
                         jump_mode=None, nonlinearity=nonlinearity, readout=readout, train_eps=train_eps,
                         final_hidden_multiplier=final_hidden_multiplier, readout_dims=(0, 1), final_readout=final_readout,
                         apply_dropout_before=apply_dropout_before, init_reduce=init_reduce, embed_edge=embed_edge,
                         embed_dim=embed_dim, use_coboundaries=use_coboundaries, graph_norm=graph_norm)

    def _edit_params(self, params):
        if len(params) > 1:
            params[1].up_index = None
        return params


class OGBEmbedSparseCIN(_SparseCINStack):
    """mp/molec_models.py:201-352 (ogbg-mol*: OGB atom / bond encoders, dropout after each conv)."""
    conv_dropout = True

    def __init__(self, out_size, num_layers, hidden, dropout_rate: float = 0.5,
                 indropout_rate: float = 0.0, max_dim: int = 2, jump_mode=None, nonlinearity='relu',
                 readout='sum', train_eps=False, final_hidden_multiplier: int = 2,
                 readout_dims=(0, 1, 2), final_readout='sum', apply_dropout_before='lin2',
                 init_reduce='sum', embed_edge=False, embed_dim=None, use_coboundaries=False,
                 graph_norm='bn'):
        super().__init__()
        from .layers import OGBEmbedVEWithReduce
        if embed_dim is None:
            embed_dim = hidden
        self.v_embed_init = AtomEncoder(embed_dim)
        self.e_embed_init = BondEncoder(embed_dim) if embed_edge else None
        self.reduce_init = InitReduceConv(reduce=init_reduce)
        self.init_conv = OGBEmbedVEWithReduce(self.v_embed_init, self.e_embed_init, self.reduce_init)
        self.in_dropout_rate = indropout_rate
        self._build(embed_dim, out_size, num_layers, hidden, dropout_rate, max_dim, jump_mode,
                    nonlinearity, readout, train_eps, final_hidden_multiplier, readout_dims,
                    final_readout, apply_dropout_before, use_coboundaries, graph_norm)

    @_one_check
    def forward(self, data: ComplexBatch, include_partial=False):
        params = data.get_all_cochain_params(max_dim=self.max_dim, include_down_features=False)
        xs = list(self.init_conv(*params))
        xs = [ops.dropout(x, self.in_dropout_rate, self.training) for x in xs]
        data.set_xs(xs)
        return self._convs_and_head(data, include_partial, {})


class _CINppLayers:
    """The one difference of the CIN++ models (mp/molec_models.py:185-199, 370-384): every layer is a CINppConv."""

    def _make_conv(self, layer_dim, hidden, act_module, train_eps, use_coboundaries):
        return layers.CINppConv(
            up_msg_size=layer_dim, down_msg_size=layer_dim, boundary_msg_size=layer_dim,
            passed_msg_boundaries_nn=None, passed_msg_up_nn=None, passed_msg_down_nn=None, passed_update_up_nn=None,
            passed_update_down_nn=None, passed_update_boundaries_nn=None, train_eps=train_eps, max_dim=self.max_dim,
            hidden=hidden, act_module=act_module, layer_dim=layer_dim, graph_norm=self.graph_norm,
            use_coboundaries=use_coboundaries)


class CINpp(_CINppLayers, SparseCIN):
    """mp/models.py:259-284: the SparseCIN stack over CINppConv layers (features used as given)."""


class EmbedCINpp(_CINppLayers, EmbedSparseCIN):
    """mp/molec_models.py:167-199: EmbedSparseCIN with CINppConv layers.  As in the reference the lower stream stays off
    (the forward asks for include_down_features=False and CINppCochainConv.forward passes no down_attr); the
    state_dict is interchangeable (tests/golden/embed_cinpp.npz holds the reference's)."""


class OGBEmbedCINpp(_CINppLayers, OGBEmbedSparseCIN):
    """mp/molec_models.py:355-384."""


# ------------------------------------------------------------------------------------------------
# CIN0 / EdgeCIN0 (mp/models.py:12-109, 286-420): the dense-CIN stacks -- every adjacency entry carries a message network
# Linear(2F -> F) -> act -> BatchNorm (layers.CINConv / EdgeCINConv: the fused per-entry form, training included), one update
# network per dimension, the per-dimension readouts summed, lin1 -> act -> dropout -> lin2
# ------------------------------------------------------------------------------------------------
class _CIN0Stack(torch.nn.Module):
    max_dim: int

    @staticmethod
    def _entry_net(k_in, width, act):
        return torch.nn.Sequential(Linear(k_in, width), act(), BN(width))

    @staticmethod
    def _update_net(k_in, hidden, act):
        return torch.nn.Sequential(Linear(k_in, hidden), act(), Linear(hidden, hidden), act(), BN(hidden))

    def _finish_init(self, num_classes, num_layers, hidden, dropout_rate, jump_mode, nonlinearity, readout):
        if jump_mode not in (None, 'cat', 'max'):
            raise NotImplementedError("jump_mode must be None, 'cat' or 'max'")
        self.dropout_rate, self.jump_mode, self.nonlinearity, self.readout = dropout_rate, jump_mode, nonlinearity, readout
        self.lin1 = Linear(num_layers * hidden if jump_mode == 'cat' else hidden, hidden)
        self.lin2 = Linear(hidden, num_classes)

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()

    def _after_conv(self, data: ComplexBatch, c: int, xs):
        data.set_xs(xs)

    def _params(self, data: ComplexBatch):
        return data.get_all_cochain_params(max_dim=self.max_dim)

    @_one_check
    def forward(self, data: ComplexBatch):
        act = get_nonlinearity(self.nonlinearity, return_module=False)
        xs, kept = None, None
        for c, conv in enumerate(self.convs):
            xs = conv(*self._params(data))
            self._after_conv(data, c, xs)
            if self.jump_mode is not None:
                kept = [[] for _ in xs] if kept is None else kept
                for i, x in enumerate(xs):
                    kept[i].append(x)
        if self.jump_mode == 'cat':
            xs = [torch.cat(k, dim=-1) for k in kept]
        elif self.jump_mode == 'max':
            xs = [torch.stack(k, dim=-1).max(dim=-1)[0] for k in kept]
        pooled = pool_complex_list(xs, data, self.max_dim, self.readout)     # absent dimensions: zero rows (mp/models.py:71-74)
        x = pooled[0]
        for t in pooled[1:]:
            x = x + t
        x = act(self.lin1(x))
        x = ops.dropout(x, self.dropout_rate, self.training)
        return self.lin2(x)

    def __repr__(self):
        return self.__class__.__name__


class CIN0(_CIN0Stack):
    """mp/models.py:12-109: a cellular GIN over upper AND lower adjacencies (CINConv), all dimensions."""

    def __init__(self, num_input_features, num_classes, num_layers, hidden, dropout_rate: float = 0.5, max_dim: int = 2,
                 jump_mode=None, nonlinearity='relu', readout='sum'):
        super().__init__()
        self.max_dim = max_dim
        act = get_nonlinearity(nonlinearity, return_module=True)
        self.convs = torch.nn.ModuleList()
        for i in range(num_layers):
            w = num_input_features if i == 0 else hidden
            update, up, down = self._update_net(w, hidden, act), self._entry_net(2 * w, w, act), self._entry_net(2 * w, w, act)
            self.convs.append(CINConv(w, w, up, down, update, train_eps=False, max_dim=max_dim))
        self._finish_init(num_classes, num_layers, hidden, dropout_rate, jump_mode, nonlinearity, readout)


class EdgeCIN0(_CIN0Stack):
    """mp/models.py:286-420: CIN0 up to the edges; the two-cell features ride along as the edges' upper attributes
    (`include_top_features`) and are updated by a network of their own between layers (`update_top_features`)."""

    def __init__(self, num_input_features, num_classes, num_layers, hidden, dropout_rate: float = 0.5, jump_mode=None,
                 nonlinearity='relu', include_top_features=True, update_top_features=True, readout='sum'):
        super().__init__()
        self.max_dim = 1
        self.include_top_features = include_top_features
        self.update_top_features = include_top_features and update_top_features
        act = get_nonlinearity(nonlinearity, return_module=True)
        self.convs = torch.nn.ModuleList()
        self.update_top_nns = torch.nn.ModuleList()
        for i in range(num_layers):
            w = num_input_features if i == 0 else hidden
            v_update, e_update = self._update_net(w, hidden, act), self._update_net(w, hidden, act)
            v_up, e_down = self._entry_net(2 * w, w, act), self._entry_net(2 * w, w, act)
            e_up = self._entry_net(2 * w if include_top_features else w, w, act)
            self.convs.append(EdgeCINConv(w, w, v_up, e_down, e_up, v_update, e_update, train_eps=False))
            if self.update_top_features and i < num_layers - 1:
                self.update_top_nns.append(self._update_net(w, hidden, act))
        self._finish_init(num_classes, num_layers, hidden, dropout_rate, jump_mode, nonlinearity, readout)

    def reset_parameters(self):
        super().reset_parameters()
        for net in self.update_top_nns:
            reset_net(net)

    def _params(self, data: ComplexBatch):
        return data.get_all_cochain_params(max_dim=self.max_dim, include_top_features=self.include_top_features)

    def _after_conv(self, data: ComplexBatch, c: int, xs):
        # (not behind the last layer; only where the batch has two-cells: mp/models.py:388-394)
        if self.update_top_features and c < len(self.convs) - 1 and 2 in data.cochains:
            data.set_xs(list(xs) + [self.update_top_nns[c](data.cochains[2].x)])
        else:
            data.set_xs(xs)


class Dummy(torch.nn.Module):
    """mp/models.py:422-473: parameter-free layers (DummyCellularMessagePassing), the per-dimension readouts summed, one Linear."""

    def __init__(self, num_input_features, num_classes, num_layers, max_dim: int = 2, readout='sum'):
        super().__init__()
        self.max_dim, self.readout = max_dim, readout
        self.convs = torch.nn.ModuleList(layers.DummyCellularMessagePassing(max_dim=max_dim) for _ in range(num_layers))
        self.lin = Linear(num_input_features, num_classes)

    def reset_parameters(self):
        self.lin.reset_parameters()

    @_one_check
    def forward(self, data: ComplexBatch):
        xs = None
        for conv in self.convs:
            xs = conv(*data.get_all_cochain_params())
            data.set_xs(xs)
        pooled = pool_complex_list(xs, data, self.max_dim, self.readout)
        x = pooled[0]
        for t in pooled[1:]:
            x = x + t
        return self.lin(x)

    def __repr__(self):
        return self.__class__.__name__


class EdgeOrient(torch.nn.Module):
    """mp/models.py:476-546: edge signals under a choice of edge orientations -- L x OrientedConv (bias-free update maps: the
    layers stay equivariant), |.| for invariance, readout over the edges of each complex, lin1 -> ReLU -> dropout -> lin2.
    `data` is a CochainBatch of edges with `upper_orient` / `lower_orient`."""

    def __init__(self, num_input_features, num_classes, num_layers, hidden, dropout_rate: float = 0.0, jump_mode=None,
                 nonlinearity='id', readout='sum', fully_invar=False):
        super().__init__()
        self.max_dim = 1
        self.fully_invar, self.dropout_rate, self.jump_mode = fully_invar, dropout_rate, jump_mode
        self.nonlinearity, self.readout = nonlinearity, readout
        if readout not in ('sum', 'mean'):
            raise NotImplementedError(f'Readout {readout} is not currently supported.')
        self.convs = torch.nn.ModuleList()
        for i in range(num_layers):
            w = num_input_features if i == 0 else hidden
            self.convs.append(layers.OrientedConv(
                dim=1, up_msg_size=w, down_msg_size=w, update_up_nn=Linear(w, hidden, bias=False),
                update_down_nn=Linear(w, hidden, bias=False), update_nn=Linear(w, hidden, bias=False),
                act_fn=get_nonlinearity(nonlinearity, return_module=False), orient=not fully_invar))
        self.lin1 = Linear(hidden, hidden)
        self.lin2 = Linear(hidden, num_classes)

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()

    @_one_check
    def forward(self, data, include_partial=False):
        if self.fully_invar:
            data.x = torch.abs(data.x)
        x = data.x
        for conv in self.convs:
            x = conv(data)
            data.x = x
        cell_pred = x
        if not self.fully_invar:
            x = torch.abs(x)
        n = getattr(data, 'num_cochains', None)
        if n is None:
            n = int(data.batch.max()) + 1
        x = global_pool(x, data.batch, int(n), mean=self.readout == 'mean')
        x = torch.relu(self.lin1(x))
        x = ops.dropout(x, self.dropout_rate, self.training)
        x = self.lin2(x)
        return (x, cell_pred) if include_partial else x

    def __repr__(self):
        return self.__class__.__name__
