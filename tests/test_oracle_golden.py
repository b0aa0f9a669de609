"""The oracle (oracle/cwn_oracle.py) against (a) the hand-computed expectations the reference's
own tests hold and (b) golden vectors produced by running the reference itself
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import cwn_oracle as O
from tests._golden import load, T, complex_dict, dummy_complex, params_dict, state_dict

NAMES = ['house', 'bridged', 'fullstop', 'colon', 'square', 'square_dot', 'kite', 'pyramid',
         'filled_square', 'molecular']


def base(prm, **kw):
    w = prm['x'].size(1)
    kw = dict(dict(up_msg_size=w, down_msg_size=w), **kw)
    return O.propagate(prm['x'], prm['up_index'], prm['down_index'], prm['boundary_index'],
                       up_attr=prm['up_attr'], down_attr=prm['down_attr'],
                       boundary_attr=prm['boundary_attr'], **kw)


# ---- (a) known answers written in the reference's tests -------------------------------------
def test_house_edges_known_answer():            # mp/test_cell_mp.py:13-35
    up, down, bnd = base(O.cochain_params(dummy_complex('house'), 1))
    assert up.flatten().tolist() == [0, 0, 11, 0, 9, 8]
    assert down.flatten().tolist() == [6, 10, 17, 9, 13, 10]
    assert bnd.flatten().tolist() == [3, 5, 7, 5, 9, 8]


def test_house_vertices_known_answer():         # mp/test_cell_mp.py:38-62
    up, down, bnd = base(O.cochain_params(dummy_complex('house'), 0))
    assert up.flatten().tolist() == [6, 4, 11, 9, 7]
    assert torch.equal(down, torch.zeros(5, 1)) and torch.equal(bnd, torch.zeros(5, 1))


def test_house_two_cell_known_answer():         # mp/test_cell_mp.py:65-88
    up, down, bnd = base(O.cochain_params(dummy_complex('house'), 2))
    assert torch.equal(up, torch.zeros(1, 1)) and torch.equal(down, torch.zeros(1, 1))
    assert bnd.flatten().tolist() == [14]


def test_two_triangles_known_answer():          # mp/test_cell_mp.py:91-111
    x = torch.tensor([[32.], [17.]])
    up, down, _ = O.propagate(x, None, torch.tensor([[0, 1], [1, 0]]), None,
                              down_attr=torch.tensor([[1], [1]]), up_msg_size=1, down_msg_size=1)
    assert (up + down).flatten().tolist() == [17, 32]


def test_isolated_and_empty():                  # mp/test_cell_mp.py:114-176
    prm = O.cochain_params(dummy_complex('square_dot'), 0)
    up, down, _ = base(prm)
    assert up[4].item() == 0 and all(up[i].item() != 0 for i in range(4))
    x = torch.tensor([[1.]])
    up, _, _ = O.propagate(x, torch.empty(2, 0, dtype=torch.long), None, None, up_msg_size=1, down_msg_size=1)
    assert torch.equal(up, torch.zeros(1, 1))
    up, _, _ = O.propagate(x, None, None, None, up_msg_size=1, down_msg_size=1)
    assert torch.equal(up, torch.zeros(1, 1))


def test_bridged_multiplicity_known_answer():   # mp/test_cell_mp.py:179-247
    up, _, _ = base(O.cochain_params(dummy_complex('bridged'), 1))
    assert up.flatten().tolist() == [24, 22, 20, 18, 22, 20]
    _, down, bnd = base(O.cochain_params(dummy_complex('bridged'), 2))
    assert down.flatten().tolist() == [10, 8, 6] and bnd.flatten().tolist() == [16, 16, 10]


def test_dummy_layer_known_answers():           # mp/test_layers.py:11-69
    h = dummy_complex('house')
    prms = [O.cochain_params(h, d) for d in range(3)]
    assert O.dummy_cochain_mp(prms[0]).flatten().tolist() == [12, 9, 25, 25, 23]
    assert O.dummy_cochain_mp(prms[1]).flatten().tolist() == [10, 20, 47, 22, 42, 37]
    assert O.dummy_cochain_mp(prms[2]).flatten().tolist() == [1]
    assert O.dummy_cochain_mp(prms[1], True, False).flatten().tolist() == [4, 7, 23, 9, 25, 24]
    assert O.dummy_cochain_mp(prms[2], True, False).flatten().tolist() == [15]
    m = dummy_complex('molecular')
    prms = [O.cochain_params(m, d) for d in range(3)]
    assert O.dummy_cochain_mp(prms[0], True, True).flatten().tolist() == [12, 24, 24, 15, 25, 31, 47, 24]
    assert O.dummy_cochain_mp(prms[1], True, True).flatten().tolist() == [35, 79, 41, 27, 66, 70, 92, 82, 53]
    assert O.dummy_cochain_mp(prms[2], True, True).flatten().tolist() == [15, 33]


def test_init_reduce_known_answer():            # mp/test_layers.py:135-149
    h = dummy_complex('house')
    p = [O.cochain_params(h, d) for d in range(3)]
    assert O.init_reduce(p[0]['x'], p[1]['boundary_index']).flatten().tolist() == [3, 5, 7, 5, 9, 8]
    assert O.init_reduce(p[1]['x'], p[2]['boundary_index']).flatten().tolist() == [14]


def test_house_params_known_answer():           # data/test_data.py:6-54
    h = dummy_complex('house')
    v, e = O.cochain_params(h, 0), O.cochain_params(h, 1)
    assert v['up_attr'].flatten().tolist() == [1, 1, 4, 4, 2, 2, 3, 3, 6, 6, 5, 5]
    assert e['up_attr'].flatten().tolist() == [1] * 6
    assert e['down_attr'].flatten().tolist() == [2, 2, 1, 1, 3, 3, 3, 3, 4, 4, 4, 4, 3, 3, 4, 4, 5, 5]


# ---- (b) golden vectors from the live reference ------------------------------------------------
@pytest.mark.parametrize('name', NAMES)
def test_propagate_golden_every_complex(name):
    g = load('propagate_known_answer.npz')
    cx = dummy_complex(name)
    for d in range(cx['dimension'] + 1):
        prm = O.cochain_params(cx, d)
        ref = params_dict(g, f'{name}/{d}/params')
        for k, v in ref.items():
            assert (v is None) == (prm[k] is None), (name, d, k)
            if v is not None:
                assert torch.equal(v, prm[k]), (name, d, k)
        up, down, bnd = base(prm)
        assert torch.equal(up, T(g[f'{name}/{d}/up']))
        assert torch.equal(down, T(g[f'{name}/{d}/down']))
        assert torch.equal(bnd, T(g[f'{name}/{d}/boundary']))
    for ub in (0, 1):
        for ud in (0, 1):
            for d in range(min(cx['dimension'], 2) + 1):
                o = O.dummy_cochain_mp(O.cochain_params(cx, d), bool(ub), bool(ud))
                assert torch.equal(o, T(g[f'{name}/dummy_b{ub}_d{ud}/{d}']))


@pytest.mark.parametrize('lname', ['testing', 'testing3', 'mol', 'pair', 'nodes_only'])
def test_batching_golden_integer_exact(lname):
    g = load('batching.npz')
    names = [str(n) for n in g[f'{lname}/names']]
    md = int(g[f'{lname}/max_dim'])
    got = O.batch_complexes([dummy_complex(n) for n in names], max_dim=md)
    ref = complex_dict(g, f'{lname}/batch')
    assert got['dimension'] == ref['dimension']
    assert torch.equal(got['y'], ref['y'])
    for d in range(ref['dimension'] + 1):
        for k in ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries',
                  'boundary_index', 'y', 'batch'):
            a, b = got['cochains'][d][k], ref['cochains'][d][k]
            assert (a is None) == (b is None), (d, k)
            if a is not None:
                assert a.dtype == b.dtype and torch.equal(a, b), (d, k)
        assert got['cochains'][d]['num_cells'] == ref['cochains'][d]['num_cells']
    for kw_name, kw in (('full', {}), ('nodown', dict(include_down_features=False))):
        for d, prm in enumerate(O.all_cochain_params(got, max_dim=md, **kw)):
            refp = params_dict(g, f'{lname}/params_{kw_name}/{d}')
            for k, v in refp.items():
                assert (v is None) == (prm[k] is None), (d, k)
                if v is not None:
                    assert torch.equal(v, prm[k]), (d, k)


@pytest.mark.parametrize('F', [1, 3, 8, 64, 128])
def test_propagate_random_golden(F):
    g = load('propagate_random.npz')
    for d in range(3):
        prm = params_dict(g, f'F{F}/{d}/params')
        for aggr in ('add', 'mean', 'max'):
            up, down, bnd = base(prm, aggr_up=aggr, aggr_down=aggr, aggr_boundary=aggr)
            for got, key in ((up, 'up'), (down, 'down'), (bnd, 'boundary')):
                torch.testing.assert_close(got, T(g[f'F{F}/{d}/{aggr}/{key}']), rtol=0, atol=1e-6)
        _, down, bnd = base(prm, up_msg_size=F, down_msg_size=5, boundary_msg_size=7,
                            use_down_msg=False, use_boundary_msg=False)
        assert list(down.shape) == g[f'F{F}/{d}/flags_off/down_shape'].tolist()
        assert list(bnd.shape) == g[f'F{F}/{d}/flags_off/boundary_shape'].tolist()
        o = O.dummy_cochain_mp(prm, True, True)
        torch.testing.assert_close(o, T(g[f'F{F}/dummy/{d}']), rtol=0, atol=1e-5)


@pytest.mark.parametrize('tag', ['mol_cob_bn', 'mol_nocob_bn', 'test_cob_id', 'mol_cob_bn_64'])
def test_sparse_cin_conv_golden(tag):
    g = load('sparse_cin_conv.npz')
    F, H, cob, bn = g[f'{tag}/meta'].tolist()
    names = [str(n) for n in g[f'{tag}/names']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    for d in range(3):
        cx['cochains'][d]['x'] = T(g[f'{tag}/x/{d}'])
    state = state_dict(g, f'{tag}/state')
    prms = O.all_cochain_params(cx, max_dim=2, include_down_features=False)
    for mode in ('eval', 'train'):
        outs = O.sparse_cin_conv(state, prms, bool(cob), training=(mode == 'train'),
                                 norm='bn' if bn else 'id')
        for d, o in enumerate(outs):
            torch.testing.assert_close(o, T(g[f'{tag}/{mode}/{d}']), rtol=1e-5, atol=1e-5)


def test_sparse_cin_conv_backward_golden():
    """Gradients of the oracle (autograd through the restatement) equal the reference's."""
    tag = 'mol_cob_bn'
    g = load('sparse_cin_conv.npz')
    names = [str(n) for n in g[f'{tag}/names']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    xs = [T(g[f'{tag}/x/{d}']).clone().requires_grad_(True) for d in range(3)]
    for d in range(3):
        cx['cochains'][d]['x'] = xs[d]
    state = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v)
             for k, v in state_dict(g, f'{tag}/state').items()}
    outs = O.sparse_cin_conv(state, O.all_cochain_params(cx, max_dim=2, include_down_features=False),
                             True, training=True)
    sum((o * T(g[f'{tag}/train_w/{d}'])).sum() for d, o in enumerate(outs)).backward()
    for d in range(3):
        torch.testing.assert_close(xs[d].grad, T(g[f'{tag}/train_gx/{d}']), rtol=1e-4, atol=1e-5)
    pre = f'{tag}/train_grad/'
    n = 0
    for k, v in g.items():
        if k.startswith(pre):
            torch.testing.assert_close(state[k[len(pre):]].grad, T(v), rtol=1e-4, atol=2e-5)
            n += 1
    assert n > 20


def test_cin_conv_and_orient_golden():
    g = load('cin_conv.npz')
    names = [str(n) for n in load('dummy_complexes.npz')['lists/testing']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    for d in range(3):
        cx['cochains'][d]['x'] = T(g[f'cin/x/{d}'])
    st = state_dict(g, 'cin/state')
    p = {k[len('mp_levels.1.'):]: v for k, v in st.items() if k.startswith('mp_levels.1.')}
    lin = lambda pre: (lambda h: h @ p[pre + '.weight'].t() + p[pre + '.bias'])
    prm = O.cochain_params(cx, 1)
    out = O.cin_cochain_conv(prm, lambda h: torch.relu(lin('msg_up_nn.0')(h)),
                             lambda h: torch.relu(lin('msg_down_nn.0')(h)),
                             lambda h: torch.relu(lin('update_nn.0')(h)), p['eps'])
    torch.testing.assert_close(out, T(g['cin/out/1']), rtol=1e-5, atol=1e-5)
    up, down = O.oriented_conv_messages(prm['x'], prm['up_index'], prm['down_index'],
                                        T(g['orient/up_orient']), T(g['orient/down_orient']))
    torch.testing.assert_close(up, T(g['orient/up']), rtol=0, atol=1e-6)
    torch.testing.assert_close(down, T(g['orient/down']), rtol=0, atol=1e-6)


def _bn_eval(p, pre):
    return lambda h: (h - p[pre + '.running_mean']) / torch.sqrt(p[pre + '.running_var'] + 1e-5) * p[pre + '.weight'] + p[pre + '.bias']


def test_edge_cin_conv_and_full_oriented_conv_golden():
    """Round 3: EdgeCINConv as EdgeCIN0 calls it (mp/layers.py:127-150, mp/models.py:388-390:
    max_dim 1, top features as up_attr of the edges, no lower adjacency on the vertices) and a full
    OrientedConv.forward (mp/layers.py:441-452), oracle vs the reference's outputs."""
    g = load('edge_oriented.npz')
    names = [str(n) for n in load('dummy_complexes.npz')['lists/testing']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    for d in range(3):
        cx['cochains'][d]['x'] = T(g[f'edge_cin/x/{d}'])
    st = state_dict(g, 'edge_cin/state')
    prms = O.all_cochain_params(cx, 1, include_top_features=True)
    assert len(prms) == 2
    for d in range(2):
        p = {k[len(f'mp_levels.{d}.'):]: v for k, v in st.items() if k.startswith(f'mp_levels.{d}.')}
        lin = lambda pre, p=p: (lambda h: h @ p[pre + '.weight'].t() + p[pre + '.bias'])
        msg = lambda pre, p=p: (lambda h: _bn_eval(p, pre + '.2')(torch.relu(lin(pre + '.0', p)(h))))
        upd = lambda h, p=p: _bn_eval(p, 'update_nn.4')(torch.relu(lin('update_nn.2', p)(torch.relu(lin('update_nn.0', p)(h)))))
        out = O.cin_cochain_conv(prms[d], msg('msg_up_nn'), msg('msg_down_nn') if d == 1 else (lambda h: None), upd, p['eps'])
        torch.testing.assert_close(out, T(g[f'edge_cin/out/{d}']), rtol=1e-5, atol=1e-5)
    so = state_dict(g, 'oriented/state')
    x = T(g['oriented/x'])
    up, down = O.oriented_conv_messages(x, T(g['oriented/upper_index']), T(g['oriented/lower_index']),
                                        T(g['oriented/upper_orient']), T(g['oriented/lower_orient']))
    aff = lambda pre, h: h @ so[pre + '.weight'].t() + so[pre + '.bias']
    y = torch.tanh(aff('update_nn', x) + aff('update_up_nn', up) + aff('update_down_nn', down))
    torch.testing.assert_close(y, T(g['oriented/out']), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('tag', ['h16_l2', 'h32_l4'])
def test_embed_sparse_cin_golden(tag):
    g = load('embed_sparse_cin.npz')
    H, L = g[f'{tag}/meta'].tolist()
    names = [str(n) for n in load('dummy_complexes.npz')['lists/mol']]
    cxs = [dummy_complex(n) for n in names]
    cx = O.batch_complexes(cxs, max_dim=2)
    cx['cochains'][0]['x'] = T(g[f'{tag}/v_types'])
    cx['cochains'][1]['x'] = T(g[f'{tag}/e_types'])
    cx['cochains'][2]['x'] = None
    state = state_dict(g, f'{tag}/state')
    for mode in ('eval', 'train'):
        y, partial = O.embed_sparse_cin_forward(state, cx, L, training=(mode == 'train'))
        for k, v in partial.items():
            torch.testing.assert_close(v, T(g[f'{tag}/{mode}/{k}']), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(y, T(g[f'{tag}/{mode}/out']), rtol=1e-4, atol=1e-4)


def test_csr_from_coo_matches_scatter():
    g = load('propagate_random.npz')
    prm = params_dict(g, 'F8/1/params')
    idx = prm['up_index']
    n = prm['x'].size(0)
    rowptr, col, perm = O.csr_from_coo(idx, n)
    assert rowptr[-1].item() == idx.size(1) and rowptr.dtype == torch.int32
    assert torch.equal(idx[0][perm.long()].int(), col)
    dst_sorted = idx[1][perm.long()]
    assert torch.all(dst_sorted[1:] >= dst_sorted[:-1])
    # stability: inside one destination the original order is kept
    same = dst_sorted[1:] == dst_sorted[:-1]
    assert torch.all(perm[1:][same] > perm[:-1][same])


def test_extra_models_golden():
    """SparseCIN (REDDIT-style) and OGBEmbedSparseCIN (molhiv-style) stacks vs the live reference."""
    g = load('sparse_cin_models.npz')
    names = [str(n) for n in load('dummy_complexes.npz')['lists/testing']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    for d in range(3):
        cx['cochains'][d]['x'] = T(g[f'reddit/x/{d}'])
    y, partial = O.sparse_cin_model_forward(state_dict(g, 'reddit/state'), cx, 3, use_coboundaries=False,
                                            norm='id', jump_mode='cat', embed=None)
    for k, v in partial.items():
        torch.testing.assert_close(v, T(g[f'reddit/{k}']), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y, T(g['reddit/out']), rtol=1e-4, atol=1e-4)

    names = [str(n) for n in load('dummy_complexes.npz')['lists/mol']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    cx['cochains'][0]['x'], cx['cochains'][1]['x'] = T(g['molhiv/v_feats']), T(g['molhiv/e_feats'])
    cx['cochains'][2]['x'] = None
    y, partial = O.sparse_cin_model_forward(state_dict(g, 'molhiv/state'), cx, 2, readout='mean', embed='ogb')
    for k, v in partial.items():
        torch.testing.assert_close(v, T(g[f'molhiv/{k}']), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y, T(g['molhiv/out']), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('tag', ['h16_l2', 'h64_l2'])
def test_embed_cinpp_golden(tag):
    """The oracle's CIN++ layer (cinpp_cochain_conv: mp/layers.py:243-260 -- three eps, the lower stream off) inside the
    EmbedCINpp forward (mp/molec_models.py:167-199) against the reference's own outputs, eval and training mode."""
    g = load('embed_cinpp.npz')
    H, L = g[f'{tag}/meta'].tolist()
    names = [str(n) for n in load('dummy_complexes.npz')['lists/mol']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    cx['cochains'][0]['x'], cx['cochains'][1]['x'] = T(g[f'{tag}/v_types']), T(g[f'{tag}/e_types'])
    cx['cochains'][2]['x'] = None
    state = state_dict(g, f'{tag}/state')
    assert len({float(state[f'convs.0.mp_levels.1.eps{k}']) for k in (1, 2, 3)}) == 3      # a mix-up of the three would show
    for mode in ('eval', 'train'):
        y, partial = O.sparse_cin_model_forward(state, cx, L, training=(mode == 'train'), conv='cinpp')
        for k, v in partial.items():
            torch.testing.assert_close(v, T(g[f'{tag}/{mode}/{k}']), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(y, T(g[f'{tag}/{mode}/out']), rtol=1e-4, atol=1e-4)


def test_ogb_embed_cinpp_golden():
    g = load('embed_cinpp.npz')
    names = [str(n) for n in load('dummy_complexes.npz')['lists/mol']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    cx['cochains'][0]['x'], cx['cochains'][1]['x'] = T(g['ogb/v_feats']), T(g['ogb/e_feats'])
    cx['cochains'][2]['x'] = None
    y, partial = O.sparse_cin_model_forward(state_dict(g, 'ogb/state'), cx, 2, readout='mean', embed='ogb', conv='cinpp')
    for k, v in partial.items():
        torch.testing.assert_close(v, T(g[f'ogb/{k}']), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y, T(g['ogb/out']), rtol=1e-4, atol=1e-4)


def test_plain_cinpp_golden():
    """CINpp (mp/models.py:259-284: features as given, messages without coboundary features, JK cat) -- oracle vs the reference."""
    g = load('embed_cinpp.npz')
    names = [str(n) for n in load('dummy_complexes.npz')['lists/testing']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    for d in range(3):
        cx['cochains'][d]['x'] = T(g[f'plain/x/{d}'])
    for mode in ('eval', 'train'):
        y, partial = O.sparse_cin_model_forward(state_dict(g, 'plain/state'), cx, 2, use_coboundaries=False, norm='bn',
                                                jump_mode='cat', embed=None, training=(mode == 'train'), conv='cinpp')
        for k, v in partial.items():
            torch.testing.assert_close(v, T(g[f'plain/{mode}/{k}']), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(y, T(g[f'plain/{mode}/out']), rtol=1e-4, atol=1e-4)


def test_embed_sparse_cin_no_rings_golden():
    """EmbedSparseCINNoRings (mp/molec_models.py:386-503: vertices and edges only, the edges' upper adjacency dropped) -- the
    oracle's forward with max_dim 1 / drop_edge_up against the reference's outputs, eval and training mode."""
    g = load('no_rings.npz')
    names = [str(n) for n in load('dummy_complexes.npz')['lists/mol']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    cx['cochains'][0]['x'], cx['cochains'][1]['x'] = T(g['v_types']), T(g['e_types'])
    cx['cochains'][2]['x'] = None
    for mode in ('eval', 'train'):
        y, _ = O.sparse_cin_model_forward(state_dict(g, 'state'), cx, 2, max_dim=1, training=(mode == 'train'), embed='zinc',
                                          readout_dims=(0, 1), drop_edge_up=True)
        torch.testing.assert_close(y, T(g[f'{mode}/out']), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('tag,pos', [('ogb_lin2', 'lin2'), ('ogb_lin1', 'lin1'), ('ogb_final', 'final_readout')])
def test_dropout_placement_golden(tag, pos):
    """WHERE the reference drops out (round 5): OGBEmbedSparseCIN in TRAINING mode with dropout 0.5 (exp/scripts/cwn-molhiv.sh),
    F.dropout's randomness recorded by the generating script (oracle/gen_golden.py dropout) -- call 0..2 the input features
    (mp/molec_models.py:290-292), 3..8 the outputs of the two conv layers (:297-300), then the head's position (:334-346).
    The oracle with the same multipliers at the same places reproduces every layer output, the pooled vectors and the
    prediction."""
    g = load('dropout.npz')
    names = [str(n) for n in load('dummy_complexes.npz')['lists/mol']]
    cx = O.batch_complexes([dummy_complex(n) for n in names], max_dim=2)
    cx['cochains'][0]['x'], cx['cochains'][1]['x'] = T(g[f'{tag}/v_feats']), T(g[f'{tag}/e_feats'])
    cx['cochains'][2]['x'] = None
    mult = lambda k: T(g[f'{tag}/mult/{k}']) if f'{tag}/mult/{k}' in g else None
    drop = {}
    for d in range(3):
        if mult(d) is not None:
            drop[('in', d)] = mult(d)
    for l in range(2):
        for d in range(3):
            drop[('conv', l, d)] = mult(3 + 3 * l + d)
    n = int(g[f'{tag}/n_calls'])
    if pos == 'lin1':
        assert n == 12
        for k in range(3):
            drop[('head', k)] = mult(9 + k)
    elif pos == 'final_readout':
        assert n == 10
        for k in range(3):
            drop[('head', k)] = mult(9)[k]
    else:
        assert n == 10
        drop[('head',)] = mult(9)
    assert (pos == 'lin1') == (('in', 0) in drop)          # (only that variant has an input dropout rate)
    y, partial = O.sparse_cin_model_forward(state_dict(g, f'{tag}/state'), cx, 2, readout='mean', embed='ogb', training=True,
                                            dropout=drop, drop_position=pos)
    for k, v in partial.items():
        torch.testing.assert_close(v, T(g[f'{tag}/{k}']), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y, T(g[f'{tag}/out']), rtol=1e-4, atol=1e-4)
    # ... and the multipliers matter: without them the prediction differs
    y0, _ = O.sparse_cin_model_forward(state_dict(g, f'{tag}/state'), cx, 2, readout='mean', embed='ogb', training=True)
    assert float((y0 - y).abs().max()) > 1e-3
