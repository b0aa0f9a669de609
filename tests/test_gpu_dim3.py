"""Complexes of dimension 3 (clique lifts with expansion_dim = 3: data/utils.py:224-272; the reference's dummy dataset and its
PROTEINS tests run max_dim = 3, data/datasets/dummy.py:12, data/test_batching.py:628-634) through the product on the GPU:
the conv layer and the SparseCIN stack with max_dim = 3 against the float64 oracle.  The blocked launches hold dimensions
0 .. 2 (CWN_LAYER_MAX_DIMS); what these batches take is whatever path the library chooses for them -- the gate is the same."""
import itertools

import numpy as np
import pytest
import torch

from oracle import cwn_oracle as O
from tests._product import gate, to_double

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def cpu(t):
    return None if t is None else t.detach().cpu()


def _complexes(num, seed, F):
    """small dense graphs: enough 4-cliques that dimension 3 has cells, lifted to dimension 3 with lower adjacencies"""
    from cwn_amd import lifting
    rng = np.random.default_rng(seed)
    out = []
    for i in range(num):
        n = int(rng.integers(6, 12))
        edges = [(u, v) for u, v in itertools.combinations(range(n), 2) if rng.random() < 0.6]
        vx = torch.from_numpy(rng.standard_normal((n, F))).float()
        out.append(lifting.clique_lift(n, edges, vx, max_dim=3, init_method='mean', include_down_adj=True, y=torch.tensor([i % 2])))
    return out


def _oracle_cx(b):
    return {'dimension': b.dimension, 'y': None, 'num_complexes': b.num_complexes, 'cochains': [
        {k: cpu(b.cochains[d][k]) for k in ('x', 'upper_index', 'lower_index', 'shared_boundaries',
                                            'shared_coboundaries', 'boundary_index', 'y', 'batch')}
        for d in range(b.dimension + 1)]}


@pytest.mark.parametrize('use_cob', [True, False])
def test_sparse_cin_conv_over_four_dimensions_vs_float64_oracle(use_cob):
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.layers import SparseCINConv
    F = 32
    cxs = _complexes(12, 3, F)
    assert sum(c.dimension == 3 for c in cxs) >= 6
    b = ComplexBatch.from_complex_list(cxs, max_dim=3)
    assert b.dimension == 3 and b.cochains[3].num_cells > 10
    torch.manual_seed(5)
    conv = SparseCINConv(F, F, F, None, None, None, None, max_dim=3, hidden=F, eps=0.25, train_eps=False,
                         act_module=torch.nn.ReLU, layer_dim=F, use_coboundaries=use_cob).eval()
    with torch.no_grad():            # BatchNorm with non-trivial running statistics
        for m in conv.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
    state = {k: v.clone() for k, v in conv.state_dict().items()}
    ocx = _oracle_cx(b)
    for c in ocx['cochains']:
        c['x'] = c['x'].double()
    want = O.sparse_cin_conv(to_double(state), O.all_cochain_params(ocx, max_dim=3, include_down_features=False), use_cob)
    conv = conv.to(DEV)
    with torch.no_grad():
        got = conv(*b.to(DEV).get_all_cochain_params(max_dim=3, include_down_features=False))
    assert len(got) == len(want) == 4
    for d in range(4):
        gate(got[d], want[d], f'SparseCINConv max_dim 3 (coboundaries {use_cob}) dimension {d}')


@pytest.mark.parametrize('rd', [(0, 1, 2, 3), (0, 1, 2)])
def test_sparse_cin_stack_over_four_dimensions_vs_float64_oracle(rd):
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import SparseCIN
    F = 8
    cxs = _complexes(10, 7, F)
    b = ComplexBatch.from_complex_list(cxs, max_dim=3)
    assert b.dimension == 3
    torch.manual_seed(1)
    model = SparseCIN(F, 2, 3, 32, dropout_rate=0.0, max_dim=3, jump_mode='cat', readout='sum', use_coboundaries=True,
                      graph_norm='bn', readout_dims=rd).eval()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    ocx = _oracle_cx(b)
    for c in ocx['cochains']:
        c['x'] = c['x'].double()
    ref, rpart = O.sparse_cin_model_forward(to_double(state), ocx, 3, max_dim=3, use_coboundaries=True, norm='bn', jump_mode='cat',
                                            embed=None, readout_dims=rd)
    model = model.to(DEV)
    with torch.no_grad():
        y, res = model(b.to(DEV), include_partial=True)
    for k, v in rpart.items():
        gate(res[k], v, f'SparseCIN max_dim 3 {k}')
    gate(y, ref, 'SparseCIN max_dim 3 prediction')
