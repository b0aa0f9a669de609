"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (this container only).

    python oracle/gen_golden.py

Imports /root/reference's own mp/cell_mp.py, mp/layers.py, mp/molec_models.py, data/complex.py,
data/dummy_complexes.py behind the stand-ins in oracle/refshim (for the third-party packages that
are not installed: torch_geometric 1.6.3, torch_scatter 2.0.5, torch_sparse 0.6.8, ogb), first
re-checks the hand-computed expectations of the reference's own tests (so the stand-in scatter is
pinned), then writes inputs + outputs as data.  Nothing of the reference's source is written
anywhere; the fixtures are tensors only.  TEST INFRASTRUCTURE -- never imported by cwn_amd/.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.path[:0] = [os.path.join(HERE, 'refshim'), REF]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mp.cell_mp import CochainMessagePassing  # noqa: E402
from mp.layers import (SparseCINConv, CINConv, InitReduceConv,  # noqa: E402
                       DummyCellularMessagePassing, OrientedConv)
from mp.molec_models import EmbedSparseCIN  # noqa: E402
from mp.nn import get_nonlinearity, get_graph_norm  # noqa: E402
from data.complex import ComplexBatch  # noqa: E402
import data.dummy_complexes as dc  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
NAMES = ['house', 'bridged', 'fullstop', 'colon', 'square', 'square_dot', 'kite', 'pyramid',
         'filled_square', 'molecular']
# same orders as data/dummy_complexes.py:28-42 (names only -- the builders stay in the reference)
TESTING_LIST = ['fullstop', 'pyramid', 'house', 'kite', 'square', 'square_dot', 'square', 'fullstop',
                'house', 'kite', 'pyramid', 'bridged', 'square_dot', 'colon', 'filled_square',
                'molecular', 'fullstop', 'colon', 'bridged', 'colon', 'fullstop', 'fullstop', 'colon']
MOL_LIST = ['house', 'kite', 'square', 'fullstop', 'bridged', 'square_dot', 'square',
            'filled_square', 'colon', 'bridged', 'kite', 'square_dot', 'colon', 'molecular',
            'bridged', 'filled_square', 'molecular', 'fullstop', 'colon']
KEYS = ['x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries',
        'boundary_index', 'y', 'batch']


def get(name):
    return getattr(dc, f'get_{name}_complex')()


def np_(t):
    return t.detach().cpu().numpy().copy()


def dump_complex(out, prefix, cx):
    out[f'{prefix}/dimension'] = np.int64(cx.dimension)
    if cx.y is not None:
        out[f'{prefix}/y'] = np_(cx.y)
    for d in range(cx.dimension + 1):
        c = cx.cochains[d]
        for k in KEYS:
            v = c[k]
            if v is not None and torch.is_tensor(v):
                out[f'{prefix}/{d}/{k}'] = np_(v)
        for k, v in (('num_cells', c.num_cells), ('num_cells_up', c.num_cells_up),
                     ('num_cells_down', c.num_cells_down)):
            if v is not None:
                out[f'{prefix}/{d}/{k}'] = np.int64(v)


def dump_params(out, prefix, prm):
    if prm.x is not None:
        out[f'{prefix}/x'] = np_(prm.x)
    for k, v in (('up_index', prm.up_index), ('down_index', prm.down_index),
                 ('boundary_index', prm.boundary_index), ('up_attr', prm.kwargs['up_attr']),
                 ('down_attr', prm.kwargs['down_attr']),
                 ('boundary_attr', prm.kwargs['boundary_attr'])):
        if v is not None:
            out[f'{prefix}/{k}'] = np_(v)


def run_base(prm, **ctor):
    ctor = dict(dict(up_msg_size=prm.x.size(1), down_msg_size=prm.x.size(1)), **ctor)
    cmp = CochainMessagePassing(**ctor)
    return cmp.propagate(prm.up_index, prm.down_index, prm.boundary_index, x=prm.x,
                         up_attr=prm.kwargs['up_attr'], down_attr=prm.kwargs['down_attr'],
                         boundary_attr=prm.kwargs['boundary_attr'])


def check_reference_known_answers():
    """The hand-computed expectations the reference's own tests carry; they pin the stand-in
    scatter before anything is written."""
    e = get('house').get_cochain_params(dim=1)
    up, down, bnd = run_base(e)
    assert up.flatten().tolist() == [0, 0, 11, 0, 9, 8]            # mp/test_cell_mp.py:31
    assert down.flatten().tolist() == [6, 10, 17, 9, 13, 10]       # :28
    assert bnd.flatten().tolist() == [3, 5, 7, 5, 9, 8]            # :34
    v = get('house').get_cochain_params(dim=0)
    assert run_base(v)[0].flatten().tolist() == [6, 4, 11, 9, 7]   # :55
    t = get('house').get_cochain_params(dim=2)
    assert run_base(t)[2].flatten().tolist() == [14]               # :87
    eb = get('bridged').get_cochain_params(dim=1)
    assert run_base(eb)[0].flatten().tolist() == [24, 22, 20, 18, 22, 20]   # :202-207
    tb = get('bridged').get_cochain_params(dim=2)
    _, d2, b2 = run_base(tb)
    assert d2.flatten().tolist() == [10, 8, 6] and b2.flatten().tolist() == [16, 16, 10]  # :237-247
    h = get('house')
    prms = [h.get_cochain_params(dim=d) for d in range(3)]
    vx, ex, tx = DummyCellularMessagePassing().forward(*prms)
    assert vx.flatten().tolist() == [12, 9, 25, 25, 23]            # mp/test_layers.py:20
    assert ex.flatten().tolist() == [10, 20, 47, 22, 42, 37]       # :23
    assert tx.flatten().tolist() == [1]                            # :26
    vx, ex, tx = DummyCellularMessagePassing(use_boundary_msg=True, use_down_msg=False).forward(*prms)
    assert ex.flatten().tolist() == [4, 7, 23, 9, 25, 24] and tx.flatten().tolist() == [15]  # :42-45
    m = get('molecular')
    prms = [m.get_cochain_params(dim=d) for d in range(3)]
    vx, ex, rx = DummyCellularMessagePassing(use_boundary_msg=True, use_down_msg=True).forward(*prms)
    assert vx.flatten().tolist() == [12, 24, 24, 15, 25, 31, 47, 24]          # :58
    assert ex.flatten().tolist() == [35, 79, 41, 27, 66, 70, 92, 82, 53]     # :62
    assert rx.flatten().tolist() == [15, 33]                                 # :68
    conv = InitReduceConv(reduce='add')
    hp = [h.get_cochain_params(dim=d) for d in range(3)]
    assert conv(hp[0].x, hp[1].boundary_index).flatten().tolist() == [3, 5, 7, 5, 9, 8]  # :144
    assert conv(hp[1].x, hp[2].boundary_index).flatten().tolist() == [14]               # :148
    print('reference known-answer tests reproduced')


def randomize_features(batch, F, gen):
    for d in range(batch.dimension + 1):
        n = batch.cochains[d].num_cells
        batch.cochains[d].x = torch.randn(n, F, generator=gen)


def save(name, out):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **out)
    print(f'{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB')


def state_np(module, prefix):
    return {f'{prefix}/{k}': np_(v) for k, v in module.state_dict().items()}


def main():
    torch.manual_seed(0)
    os.makedirs(GOLD, exist_ok=True)
    check_reference_known_answers()

    # 1. the hand-built complexes as data --------------------------------------------------
    out = {}
    for n in NAMES:
        dump_complex(out, n, get(n))
    out['lists/testing'] = np.array(TESTING_LIST)
    out['lists/mol'] = np.array(MOL_LIST)
    save('dummy_complexes.npz', out)

    # 2. base-class propagate + dummy layer on every complex / dim, original integer features
    out = {}
    for n in NAMES:
        cx = get(n)
        for d in range(cx.dimension + 1):
            prm = cx.get_cochain_params(dim=d)
            dump_params(out, f'{n}/{d}/params', prm)
            up, down, bnd = run_base(prm)
            out[f'{n}/{d}/up'], out[f'{n}/{d}/down'], out[f'{n}/{d}/boundary'] = map(np_, (up, down, bnd))
        prms = [cx.get_cochain_params(dim=d) for d in range(min(cx.dimension, 2) + 1)]
        for ub in (False, True):
            for ud in (False, True):
                outs = DummyCellularMessagePassing(use_boundary_msg=ub, use_down_msg=ud).forward(*prms)
                for d, o in enumerate(outs):
                    out[f'{n}/dummy_b{int(ub)}_d{int(ud)}/{d}'] = np_(o)
    save('propagate_known_answer.npz', out)

    # 3. batching: integer layouts + params --------------------------------------------------
    out = {}
    for lname, lst, md in (('testing', TESTING_LIST, 2), ('testing3', TESTING_LIST, 3),
                           ('mol', MOL_LIST, 2), ('pair', ['house', 'square'], 2),
                           ('nodes_only', ['fullstop', 'colon'], 2)):
        b = ComplexBatch.from_complex_list([get(n) for n in lst], max_dim=md)
        out[f'{lname}/names'] = np.array(lst)
        out[f'{lname}/max_dim'] = np.int64(md)
        dump_complex(out, f'{lname}/batch', b)
        for kw_name, kw in (('full', {}), ('nodown', dict(include_down_features=False))):
            for d, prm in enumerate(b.get_all_cochain_params(max_dim=md, **kw)):
                dump_params(out, f'{lname}/params_{kw_name}/{d}', prm)
    save('batching.npz', out)

    # 4. propagate on random features, every aggregation --------------------------------------
    out = {}
    gen = torch.Generator().manual_seed(1234)
    for F in (1, 3, 8, 64, 128):
        b = ComplexBatch.from_complex_list([get(n) for n in TESTING_LIST], max_dim=2)
        randomize_features(b, F, gen)
        for d, prm in enumerate(b.get_all_cochain_params(max_dim=2)):
            dump_params(out, f'F{F}/{d}/params', prm)
            for aggr in ('add', 'mean', 'max'):
                up, down, bnd = run_base(prm, aggr_up=aggr, aggr_down=aggr, aggr_boundary=aggr)
                out[f'F{F}/{d}/{aggr}/up'], out[f'F{F}/{d}/{aggr}/down'], out[f'F{F}/{d}/{aggr}/boundary'] = \
                    map(np_, (up, down, bnd))
            # flags off -> zero streams of the declared widths (mp/cell_mp.py:517-522)
            up, down, bnd = run_base(prm, up_msg_size=F, down_msg_size=5, boundary_msg_size=7,
                                     use_down_msg=False, use_boundary_msg=False)
            out[f'F{F}/{d}/flags_off/down_shape'] = np.array(down.shape)
            out[f'F{F}/{d}/flags_off/boundary_shape'] = np.array(bnd.shape)
        prms = b.get_all_cochain_params(max_dim=2)
        outs = DummyCellularMessagePassing(input_dim=F, use_boundary_msg=True, use_down_msg=True).forward(*prms)
        for d, o in enumerate(outs):
            out[f'F{F}/dummy/{d}'] = np_(o)
    save('propagate_random.npz', out)

    # 5. SparseCINConv forward + backward ------------------------------------------------------
    out = {}
    gen = torch.Generator().manual_seed(99)
    for tag, lst, F, H, cob, norm in (('mol_cob_bn', MOL_LIST, 8, 16, True, 'bn'),
                                      ('mol_nocob_bn', MOL_LIST, 8, 16, False, 'bn'),
                                      ('test_cob_id', TESTING_LIST, 8, 8, True, 'id'),
                                      ('mol_cob_bn_64', MOL_LIST, 64, 64, True, 'bn')):
        torch.manual_seed(7)
        b = ComplexBatch.from_complex_list([get(n) for n in lst], max_dim=2)
        randomize_features(b, F, gen)
        conv = SparseCINConv(up_msg_size=F, down_msg_size=F, boundary_msg_size=F,
                             passed_msg_up_nn=None, passed_msg_boundaries_nn=None,
                             passed_update_up_nn=None, passed_update_boundaries_nn=None,
                             train_eps=True, max_dim=2, hidden=H,
                             act_module=get_nonlinearity('relu', return_module=True), layer_dim=F,
                             graph_norm=get_graph_norm(norm), use_coboundaries=cob)
        with torch.no_grad():  # non-trivial eps and BN buffers
            for lvl in conv.mp_levels:
                lvl.eps1.fill_(0.25)
                lvl.eps2.fill_(-0.5)
                for m in lvl.modules():
                    if isinstance(m, torch.nn.BatchNorm1d):
                        m.running_mean.normal_(generator=gen)
                        m.running_var.uniform_(0.5, 1.5, generator=gen)
        out[f'{tag}/meta'] = np.array([F, H, int(cob), int(norm == 'bn')])
        out[f'{tag}/names'] = np.array(lst)
        out.update(state_np(conv, f'{tag}/state'))
        for d in range(3):
            out[f'{tag}/x/{d}'] = np_(b.cochains[d].x)
        conv.eval()
        with torch.no_grad():
            outs = conv(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
        for d, o in enumerate(outs):
            out[f'{tag}/eval/{d}'] = np_(o)
        conv.train()
        xs = [b.cochains[d].x.clone().requires_grad_(True) for d in range(3)]
        for d in range(3):
            b.cochains[d]._Cochain__x = xs[d]
        outs = conv(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
        ws = [torch.randn(o.shape, generator=gen) for o in outs]
        loss = sum((o * w).sum() for o, w in zip(outs, ws))
        loss.backward()
        for d, o in enumerate(outs):
            out[f'{tag}/train/{d}'] = np_(o)
            out[f'{tag}/train_w/{d}'] = np_(ws[d])
            out[f'{tag}/train_gx/{d}'] = np_(xs[d].grad)
        for k, p in conv.named_parameters():
            if p.grad is not None:
                out[f'{tag}/train_grad/{k}'] = np_(p.grad)
    save('sparse_cin_conv.npz', out)

    # 6. CINConv (up + down, per-message nets) and OrientedConv messages ----------------------
    out = {}
    gen = torch.Generator().manual_seed(5)
    torch.manual_seed(11)
    F = 8
    b = ComplexBatch.from_complex_list([get(n) for n in TESTING_LIST], max_dim=2)
    randomize_features(b, F, gen)
    msg_up = torch.nn.Sequential(torch.nn.Linear(2 * F, F), torch.nn.ReLU())
    msg_down = torch.nn.Sequential(torch.nn.Linear(2 * F, F), torch.nn.ReLU())
    upd = torch.nn.Sequential(torch.nn.Linear(F, 12), torch.nn.ReLU())
    conv = CINConv(F, F, msg_up, msg_down, upd, eps=0.1, train_eps=False, max_dim=2)
    out.update(state_np(conv, 'cin/state'))
    prms = b.get_all_cochain_params(max_dim=2)
    # top dim has no up_attr and lowest no down_attr: reference CINCochainConv.message_down cats
    # None -> only dims where both exist can run; dim 1 has up+down with attrs in this batch.
    for d in range(3):
        out[f'cin/x/{d}'] = np_(b.cochains[d].x)
    with torch.no_grad():
        o1 = conv.mp_levels[1].forward(prms[1])
    out['cin/out/1'] = np_(o1)
    e = prms[1]
    orient_up = torch.where(torch.rand(e.up_index.size(1), generator=gen) > 0.5, 1.0, -1.0)
    orient_dn = torch.where(torch.rand(e.down_index.size(1), generator=gen) > 0.5, 1.0, -1.0)
    oc = OrientedConv(1, F, F, update_up_nn=None, update_down_nn=None, update_nn=None, act_fn=None)
    up, down, _ = oc.propagate(e.up_index, e.down_index, None, x=e.x,
                               up_attr=orient_up.view(-1, 1), down_attr=orient_dn.view(-1, 1))
    out['orient/up_orient'], out['orient/down_orient'] = np_(orient_up), np_(orient_dn)
    out['orient/up'], out['orient/down'] = np_(up), np_(down)
    save('cin_conv.npz', out)

    # 7. the whole EmbedSparseCIN stack (ZINC model shape, small widths) ----------------------
    out = {}
    gen = torch.Generator().manual_seed(3)
    for tag, H, L in (('h16_l2', 16, 2), ('h32_l4', 32, 4)):
        torch.manual_seed(21)
        cxs = [get(n) for n in MOL_LIST]
        for cx in cxs:  # integer atom / bond types like ZINC (data/datasets/zinc.py:29-30)
            cx.cochains[0]._Cochain__x = torch.randint(0, 28, (cx.cochains[0].num_cells, 1), generator=gen).float()
            if cx.dimension >= 1:
                cx.cochains[1]._Cochain__x = torch.randint(0, 4, (cx.cochains[1].num_cells, 1), generator=gen).float()
            if cx.dimension >= 2:
                cx.cochains[2]._Cochain__x = None
        b = ComplexBatch.from_complex_list(cxs, max_dim=2)
        model = EmbedSparseCIN(28, 4, 1, L, H, dropout_rate=0.0, max_dim=2, jump_mode=None,
                               nonlinearity='relu', readout='sum', train_eps=False,
                               final_hidden_multiplier=2, final_readout='sum',
                               apply_dropout_before='lin2', init_reduce='sum', embed_edge=True,
                               use_coboundaries=True, graph_norm='bn')
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.normal_(generator=gen)
                    m.running_var.uniform_(0.5, 1.5, generator=gen)
        out[f'{tag}/meta'] = np.array([H, L])
        out.update(state_np(model, f'{tag}/state'))
        out[f'{tag}/v_types'] = np_(b.cochains[0].x)
        out[f'{tag}/e_types'] = np_(b.cochains[1].x)
        for mode in ('eval', 'train'):
            model.train(mode == 'train')
            bb = ComplexBatch.from_complex_list(cxs, max_dim=2)
            with torch.no_grad():
                y, res = model(bb, include_partial=True)
            out[f'{tag}/{mode}/out'] = np_(y)
            for k, v in res.items():
                out[f'{tag}/{mode}/{k}'] = np_(v)
    save('embed_sparse_cin.npz', out)


def edge_and_oriented():
    """Round 3: EdgeCINConv (mp/layers.py:127-150) as EdgeCIN0 builds and calls it (mp/models.py:311-341, 388-390:
    max_dim 1, top features included) and a FULL OrientedConv.forward (mp/layers.py:430-470: propagate with the
    orientation messages, three update networks, activation) on the edges of a batched complex."""
    from mp.layers import EdgeCINConv
    from data.complex import Cochain
    out = {}
    gen = torch.Generator().manual_seed(9)
    torch.manual_seed(13)
    F, Hd = 8, 12
    b = ComplexBatch.from_complex_list([get(n) for n in TESTING_LIST], max_dim=2)
    randomize_features(b, F, gen)

    def msg_net(k):
        return torch.nn.Sequential(torch.nn.Linear(k, F), torch.nn.ReLU(), torch.nn.BatchNorm1d(F))

    def upd_net():
        return torch.nn.Sequential(torch.nn.Linear(F, Hd), torch.nn.ReLU(), torch.nn.Linear(Hd, Hd), torch.nn.ReLU(),
                                   torch.nn.BatchNorm1d(Hd))
    conv = EdgeCINConv(F, F, msg_net(2 * F), msg_net(2 * F), msg_net(2 * F), upd_net(), upd_net(), eps=0.2, train_eps=False)
    with torch.no_grad():
        for m in conv.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(generator=gen)
                m.running_var.uniform_(0.5, 1.5, generator=gen)
    conv.eval()
    out.update(state_np(conv, 'edge_cin/state'))
    for d in range(3):
        out[f'edge_cin/x/{d}'] = np_(b.cochains[d].x)
    prms = b.get_all_cochain_params(max_dim=1, include_top_features=True)
    assert len(prms) == 2 and prms[1].kwargs['up_attr'] is not None and prms[0].down_index is None
    with torch.no_grad():
        outs = conv(*prms)
    for d, o in enumerate(outs):
        out[f'edge_cin/out/{d}'] = np_(o)

    e = b.get_cochain_params(dim=1, max_dim=2)
    orient_up = torch.where(torch.rand(e.up_index.size(1), generator=gen) > 0.5, 1.0, -1.0)
    orient_dn = torch.where(torch.rand(e.down_index.size(1), generator=gen) > 0.5, 1.0, -1.0)
    oc = OrientedConv(1, F, F, update_up_nn=torch.nn.Linear(F, Hd), update_down_nn=torch.nn.Linear(F, Hd),
                      update_nn=torch.nn.Linear(F, Hd), act_fn=torch.tanh)
    out.update(state_np(oc, 'oriented/state'))
    cochain = Cochain(dim=1, x=e.x, upper_index=e.up_index, lower_index=e.down_index, upper_orient=orient_up,
                      lower_orient=orient_dn)
    with torch.no_grad():
        y = oc(cochain)
    out['oriented/x'], out['oriented/upper_index'], out['oriented/lower_index'] = np_(e.x), np_(e.up_index), np_(e.down_index)
    out['oriented/upper_orient'], out['oriented/lower_orient'], out['oriented/out'] = np_(orient_up), np_(orient_dn), np_(y)
    save('edge_oriented.npz', out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'edge_oriented':
        edge_and_oriented()         # the round-3 fixture alone (the others stay byte-identical)
    elif len(sys.argv) > 1 and sys.argv[1] in ('cinpp', 'cin0', 'no_rings', 'dropout'):
        pass                        # a round-4 / round-5 fixture alone (written at the end of this file)
    else:
        main()
        edge_and_oriented()


def extra_models():
    """SparseCIN (mp/models.py, REDDIT-style: F=1 inputs, no coboundaries, norm id, JK cat) and
    OGBEmbedSparseCIN (mp/molec_models.py, molhiv-style: mean readout) -> sparse_cin_models.npz."""
    from mp.models import SparseCIN
    from mp.molec_models import OGBEmbedSparseCIN
    out = {}
    gen = torch.Generator().manual_seed(8)
    torch.manual_seed(31)
    b = ComplexBatch.from_complex_list([get(n) for n in TESTING_LIST], max_dim=2)
    model = SparseCIN(1, 2, 3, 16, dropout_rate=0.0, max_dim=2, jump_mode='cat', nonlinearity='relu',
                      readout='sum', use_coboundaries=False, graph_norm='id', final_readout='sum')
    model.eval()
    out.update(state_np(model, 'reddit/state'))
    for d in range(3):
        out[f'reddit/x/{d}'] = np_(b.cochains[d].x)
    with torch.no_grad():
        y, res = model(b, include_partial=True)
    out['reddit/out'] = np_(y)
    for k, v in res.items():
        out[f'reddit/{k}'] = np_(v)

    torch.manual_seed(32)
    cxs = [get(n) for n in MOL_LIST]
    from ogb.graphproppred.mol_encoder import ATOM_DIMS, BOND_DIMS
    for cx in cxs:
        n0 = cx.cochains[0].num_cells
        cx.cochains[0]._Cochain__x = torch.stack([torch.randint(0, d, (n0,), generator=gen) for d in ATOM_DIMS], 1)
        if cx.dimension >= 1:
            n1 = cx.cochains[1].num_cells
            cx.cochains[1]._Cochain__x = torch.stack([torch.randint(0, d, (n1,), generator=gen) for d in BOND_DIMS], 1)
        if cx.dimension >= 2:
            cx.cochains[2]._Cochain__x = None
    b = ComplexBatch.from_complex_list(cxs, max_dim=2)
    model = OGBEmbedSparseCIN(1, 2, 16, dropout_rate=0.0, indropout_rate=0.0, max_dim=2, jump_mode=None,
                              nonlinearity='relu', readout='mean', final_readout='sum', init_reduce='sum',
                              embed_edge=True, use_coboundaries=True, graph_norm='bn')
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(generator=gen)
                m.running_var.uniform_(0.5, 1.5, generator=gen)
    model.eval()
    out.update(state_np(model, 'molhiv/state'))
    out['molhiv/v_feats'] = np_(b.cochains[0].x)
    out['molhiv/e_feats'] = np_(b.cochains[1].x)
    with torch.no_grad():
        y, res = model(b, include_partial=True)
    out['molhiv/out'] = np_(y)
    for k, v in res.items():
        out[f'molhiv/{k}'] = np_(v)
    save('sparse_cin_models.npz', out)


def cinpp_models():
    """Round 4: CINppConv as the reference's EmbedCINpp / OGBEmbedCINpp build and call it (mp/molec_models.py:167-199,
    355-384; mp/layers.py:216-260, 344-427: the lower stream stays off -- the forward passes no down_attr and the models ask
    for include_down_features=False) -> embed_cinpp.npz: state, inputs, every layer's outputs and the prediction, eval and
    training mode."""
    from mp.molec_models import EmbedCINpp, OGBEmbedCINpp
    from ogb.graphproppred.mol_encoder import ATOM_DIMS, BOND_DIMS
    out = {}
    gen = torch.Generator().manual_seed(17)
    for tag, H, L in (('h16_l2', 16, 2), ('h64_l2', 64, 2)):
        torch.manual_seed(41)
        cxs = [get(n) for n in MOL_LIST]
        for cx in cxs:
            cx.cochains[0]._Cochain__x = torch.randint(0, 28, (cx.cochains[0].num_cells, 1), generator=gen).float()
            if cx.dimension >= 1:
                cx.cochains[1]._Cochain__x = torch.randint(0, 4, (cx.cochains[1].num_cells, 1), generator=gen).float()
            if cx.dimension >= 2:
                cx.cochains[2]._Cochain__x = None
        b = ComplexBatch.from_complex_list(cxs, max_dim=2)
        model = EmbedCINpp(28, 4, 1, L, H, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum',
                           train_eps=True, final_hidden_multiplier=2, final_readout='sum', apply_dropout_before='lin2',
                           init_reduce='sum', embed_edge=True, use_coboundaries=True, graph_norm='bn')
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.normal_(generator=gen)
                    m.running_var.uniform_(0.5, 1.5, generator=gen)
            for n, p in model.named_parameters():
                if '.eps' in n:                       # distinct eps1 / eps2 / eps3: a mix-up of the three shows
                    p.copy_(torch.rand(1, generator=gen) * 0.5)
        out[f'{tag}/meta'] = np.array([H, L])
        out.update(state_np(model, f'{tag}/state'))
        out[f'{tag}/v_types'] = np_(b.cochains[0].x)
        out[f'{tag}/e_types'] = np_(b.cochains[1].x)
        for mode in ('eval', 'train'):
            model.train(mode == 'train')
            bb = ComplexBatch.from_complex_list(cxs, max_dim=2)
            with torch.no_grad():
                y, res = model(bb, include_partial=True)
            out[f'{tag}/{mode}/out'] = np_(y)
            for k, v in res.items():
                out[f'{tag}/{mode}/{k}'] = np_(v)
    # the OGB front on the same layers
    torch.manual_seed(42)
    cxs = [get(n) for n in MOL_LIST]
    for cx in cxs:
        n0 = cx.cochains[0].num_cells
        cx.cochains[0]._Cochain__x = torch.stack([torch.randint(0, d, (n0,), generator=gen) for d in ATOM_DIMS], 1)
        if cx.dimension >= 1:
            n1 = cx.cochains[1].num_cells
            cx.cochains[1]._Cochain__x = torch.stack([torch.randint(0, d, (n1,), generator=gen) for d in BOND_DIMS], 1)
        if cx.dimension >= 2:
            cx.cochains[2]._Cochain__x = None
    b = ComplexBatch.from_complex_list(cxs, max_dim=2)
    model = OGBEmbedCINpp(1, 2, 16, dropout_rate=0.0, indropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu',
                          readout='mean', final_readout='sum', init_reduce='sum', embed_edge=True, use_coboundaries=True,
                          graph_norm='bn')
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(generator=gen)
                m.running_var.uniform_(0.5, 1.5, generator=gen)
    model.eval()
    out.update(state_np(model, 'ogb/state'))
    out['ogb/v_feats'] = np_(b.cochains[0].x)
    out['ogb/e_feats'] = np_(b.cochains[1].x)
    with torch.no_grad():
        y, res = model(b, include_partial=True)
    out['ogb/out'] = np_(y)
    for k, v in res.items():
        out[f'ogb/{k}'] = np_(v)
    # mp/models.py:259-284: CINpp, the SparseCIN stack (features as given, JK cat) over CINppConv layers; no coboundary
    # features in the messages (msg networks = first operand), eval and training mode
    from mp.models import CINpp
    torch.manual_seed(43)
    b = ComplexBatch.from_complex_list([get(n) for n in TESTING_LIST], max_dim=2)
    model = CINpp(1, 2, 2, 16, dropout_rate=0.0, max_dim=2, jump_mode='cat', nonlinearity='relu', readout='sum',
                  train_eps=True, use_coboundaries=False, graph_norm='bn', final_readout='sum')
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(generator=gen)
                m.running_var.uniform_(0.5, 1.5, generator=gen)
        for n, p in model.named_parameters():
            if '.eps' in n:
                p.copy_(torch.rand(1, generator=gen) * 0.5)
    out.update(state_np(model, 'plain/state'))
    for d in range(3):
        out[f'plain/x/{d}'] = np_(b.cochains[d].x)
    for mode in ('eval', 'train'):
        model.train(mode == 'train')
        bb = ComplexBatch.from_complex_list([get(n) for n in TESTING_LIST], max_dim=2)
        with torch.no_grad():
            y, res = model(bb, include_partial=True)
        out[f'plain/{mode}/out'] = np_(y)
        for k, v in res.items():
            out[f'plain/{mode}/{k}'] = np_(v)
    save('embed_cinpp.npz', out)


def no_rings_model():
    """Round 4: EmbedSparseCINNoRings (mp/molec_models.py:386-503) on the molecule list -> no_rings.npz."""
    from mp.molec_models import EmbedSparseCINNoRings
    out = {}
    gen = torch.Generator().manual_seed(29)
    torch.manual_seed(61)
    cxs = [get(n) for n in MOL_LIST]
    for cx in cxs:
        cx.cochains[0]._Cochain__x = torch.randint(0, 28, (cx.cochains[0].num_cells, 1), generator=gen).float()
        if cx.dimension >= 1:
            cx.cochains[1]._Cochain__x = torch.randint(0, 4, (cx.cochains[1].num_cells, 1), generator=gen).float()
        if cx.dimension >= 2:
            cx.cochains[2]._Cochain__x = None
    model = EmbedSparseCINNoRings(28, 4, 1, 2, 16, dropout_rate=0.0, nonlinearity='relu', readout='sum', train_eps=False,
                                  final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True,
                                  use_coboundaries=True, graph_norm='bn')
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(generator=gen)
                m.running_var.uniform_(0.5, 1.5, generator=gen)
    out.update(state_np(model, 'state'))
    b = ComplexBatch.from_complex_list(cxs, max_dim=2)
    out['v_types'], out['e_types'] = np_(b.cochains[0].x), np_(b.cochains[1].x)
    for mode in ('eval', 'train'):
        model.train(mode == 'train')
        with torch.no_grad():
            out[f'{mode}/out'] = np_(model(ComplexBatch.from_complex_list(cxs, max_dim=2)))
    save('no_rings.npz', out)


def cin0_models():
    """Round 4: CIN0 and EdgeCIN0 (mp/models.py:12-109, 286-420) on the testing batch -> cin0_models.npz: state, inputs, the
    prediction in eval and training mode (BatchNorm over the adjacency ENTRIES in the message networks)."""
    from mp.models import CIN0, EdgeCIN0
    out = {}
    gen = torch.Generator().manual_seed(23)
    F = 8

    def run(tag, model, max_dim):
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.normal_(generator=gen)
                    m.running_var.uniform_(0.5, 1.5, generator=gen)
        out.update(state_np(model, f'{tag}/state'))
        b = ComplexBatch.from_complex_list([get(n) for n in TESTING_LIST], max_dim=2)
        randomize_features(b, F, torch.Generator().manual_seed(5))
        for d in range(3):
            out[f'{tag}/x/{d}'] = np_(b.cochains[d].x)
        for mode in ('eval', 'train'):
            model.train(mode == 'train')
            bb = ComplexBatch.from_complex_list([get(n) for n in TESTING_LIST], max_dim=2)
            randomize_features(bb, F, torch.Generator().manual_seed(5))
            with torch.no_grad():
                out[f'{tag}/{mode}/out'] = np_(model(bb))

    torch.manual_seed(51)
    run('cin0', CIN0(F, 3, 2, 12, dropout_rate=0.0, max_dim=2, jump_mode='cat', nonlinearity='relu', readout='sum'), 2)
    torch.manual_seed(52)
    run('edge', EdgeCIN0(F, 3, 3, 12, dropout_rate=0.0, jump_mode=None, nonlinearity='relu', include_top_features=True,
                         update_top_features=True, readout='mean'), 1)
    torch.manual_seed(53)
    run('edge_notop', EdgeCIN0(F, 3, 2, 12, dropout_rate=0.0, jump_mode=None, nonlinearity='relu', include_top_features=False,
                               readout='sum'), 1)
    # Dummy (mp/models.py:422-473) on the testing batch, features as the dummy complexes carry them
    from mp.models import Dummy, EdgeOrient
    from data.complex import Cochain, CochainBatch
    torch.manual_seed(54)
    model = Dummy(1, 3, 2, max_dim=2, readout='sum')
    out.update(state_np(model, 'dummy/state'))
    b = ComplexBatch.from_complex_list([get(n) for n in TESTING_LIST], max_dim=2)
    for d in range(3):
        out[f'dummy/x/{d}'] = np_(b.cochains[d].x)
    with torch.no_grad():
        out['dummy/out'] = np_(model(b))
    # EdgeOrient (mp/models.py:476-546) on a CochainBatch of edges with random orientations: both forms
    edges = []
    for name in ('house', 'kite', 'pyramid', 'bridged', 'square_dot'):
        e = get(name).cochains[1]
        n_up = 0 if e.upper_index is None else e.upper_index.size(1)
        n_dn = 0 if e.lower_index is None else e.lower_index.size(1)
        c = Cochain(dim=1, x=torch.randn(e.num_cells, F, generator=gen),
                    upper_index=e.upper_index if e.upper_index is not None else torch.zeros(2, 0, dtype=torch.long),
                    lower_index=e.lower_index if e.lower_index is not None else torch.zeros(2, 0, dtype=torch.long),
                    upper_orient=torch.where(torch.rand(n_up, generator=gen) > 0.5, 1.0, -1.0),
                    lower_orient=torch.where(torch.rand(n_dn, generator=gen) > 0.5, 1.0, -1.0))
        edges.append(c)
    for k, c in enumerate(edges):
        for key in ('x', 'upper_index', 'lower_index', 'upper_orient', 'lower_orient'):
            out[f'orient/edges/{k}/{key}'] = np_(getattr(c, key))
    out['orient/n'] = np.int64(len(edges))
    for tag, invar, act in (('orient', False, 'id'), ('orient_invar', True, 'relu')):
        torch.manual_seed(55)
        model = EdgeOrient(F, 2, 2, 12, dropout_rate=0.0, nonlinearity=act, readout='sum', fully_invar=invar)
        out.update(state_np(model, f'{tag}/state'))
        data = CochainBatch.from_cochain_list([Cochain(dim=1, **{key: getattr(c, key).clone() for key in
                                               ('x', 'upper_index', 'lower_index', 'upper_orient', 'lower_orient')}) for c in edges])
        with torch.no_grad():
            y, cells = model(data, include_partial=True)
        out[f'{tag}/out'], out[f'{tag}/cells'] = np_(y), np_(cells)
    save('cin0_models.npz', out)


def dropout_models():
    """Round 5: WHERE the reference drops out (mp/molec_models.py:104-106 input features, :298-300 after every conv layer of
    OGBEmbedSparseCIN, :129-146 / :334-346 the head's `apply_dropout_before` position) -> dropout.npz.  The reference's models
    run in TRAINING mode with dropout 0.5 (exp/scripts/cwn-molhiv.sh:11-14) while `F.dropout` in mp.molec_models is replaced by
    a function that draws a Bernoulli keep-mask from a seeded generator, RECORDS the multipliers (0 or 1 / (1 - p)) and applies
    them -- F.dropout's own arithmetic with the randomness handed in.  The fixture holds the state, the inputs, the recorded
    multipliers in call order and the training-mode outputs (BatchNorm batch statistics); a checker that applies the same
    multipliers at the same places must reproduce the outputs."""
    import mp.molec_models as MM
    from mp.molec_models import OGBEmbedSparseCIN
    from ogb.graphproppred.mol_encoder import ATOM_DIMS, BOND_DIMS
    out = {}
    real = MM.F.dropout
    for tag, pos, indrop in (('ogb_lin2', 'lin2', 0.0), ('ogb_lin1', 'lin1', 0.25), ('ogb_final', 'final_readout', 0.0)):
        gen = torch.Generator().manual_seed(77)
        torch.manual_seed(83)
        cxs = [get(n) for n in MOL_LIST]
        for cx in cxs:
            n0 = cx.cochains[0].num_cells
            cx.cochains[0]._Cochain__x = torch.stack([torch.randint(0, d, (n0,), generator=gen) for d in ATOM_DIMS], 1)
            if cx.dimension >= 1:
                n1 = cx.cochains[1].num_cells
                cx.cochains[1]._Cochain__x = torch.stack([torch.randint(0, d, (n1,), generator=gen) for d in BOND_DIMS], 1)
            if cx.dimension >= 2:
                cx.cochains[2]._Cochain__x = None
        b = ComplexBatch.from_complex_list(cxs, max_dim=2)
        model = OGBEmbedSparseCIN(1, 2, 16, dropout_rate=0.5, indropout_rate=indrop, max_dim=2, jump_mode=None,
                                  nonlinearity='relu', readout='mean', final_readout='sum', apply_dropout_before=pos,
                                  init_reduce='sum', embed_edge=True, use_coboundaries=True, graph_norm='bn')
        model.train()
        out.update(state_np(model, f'{tag}/state'))
        out[f'{tag}/v_feats'] = np_(b.cochains[0].x)
        out[f'{tag}/e_feats'] = np_(b.cochains[1].x)
        calls = []

        def recorded(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                calls.append(None)
                return x
            m = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype) / (1.0 - p)
            calls.append(m)
            return x * m
        MM.F.dropout = recorded
        try:
            with torch.no_grad():
                y, res = model(b, include_partial=True)
        finally:
            MM.F.dropout = real
        out[f'{tag}/n_calls'] = np.int64(len(calls))
        for k, m in enumerate(calls):
            if m is not None:
                out[f'{tag}/mult/{k}'] = np_(m)
        out[f'{tag}/out'] = np_(y)
        for k, v in res.items():
            out[f'{tag}/{k}'] = np_(v)
    save('dropout.npz', out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'no_rings':
        no_rings_model()
    elif len(sys.argv) > 1 and sys.argv[1] == 'dropout':
        dropout_models()
    elif len(sys.argv) > 1 and sys.argv[1] == 'cin0':
        cin0_models()
    elif len(sys.argv) > 1 and sys.argv[1] == 'cinpp':
        cinpp_models()
    elif os.environ.get('CWN_GOLDEN_EXTRA', '1') == '1':
        extra_models()
        if len(sys.argv) == 1:
            cinpp_models()
            cin0_models()
            no_rings_model()
            dropout_models()
