"""CPU-side checks: the C-ABI library loads and exports every symbol include/cwn_hip.h declares;
container layout is integer-exact against the golden vectors; host logic (hook routing, errors)
behaves like the reference contract.  No compute call is made here (no GPU in this container)."""
import ctypes
import os
import re

import pytest
import torch

from cwn_amd import _ffi, csr
from cwn_amd.cell_mp import CochainMessagePassing, IndexedRows
from cwn_amd.complex import Cochain, Complex, ComplexBatch
from cwn_amd.layers import (SparseCINConv, SparseCINCochainConv, FirstOf, Catter,
                            DummyCochainMessagePassing, CINConv, OrientedConv)
from tests._golden import load, T, complex_dict
from tests._product import dummy_complex, dummy_batch, list_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'cwn_hip.h')).read()
    declared = set(re.findall(r'\b(cwn_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    lib = _ffi.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cwn_abi_version() == _ffi.ABI_VERSION == 24
    assert lib.cwn_target_arch() == b'gfx950'
    assert lib.cwn_error_string(0) == b'ok'


def test_struct_layout_matches_header(tmp_path):
    """Compile a probe against include/cwn_hip.h with the host C compiler and compare every
    field offset and struct size with the ctypes mirror in cwn_amd/_ffi.py."""
    import subprocess
    structs = {'cwn_csr_desc': _ffi.CsrDesc, 'cwn_agg_desc': _ffi.AggDesc,
               'cwn_gemm_desc': _ffi.GemmDesc, 'cwn_gemm_bnb': _ffi.GemmBnb, 'cwn_collate_desc': _ffi.CollateDesc,
               'cwn_bn_desc': _ffi.BnDesc, 'cwn_norm_desc': _ffi.NormDesc,
               'cwn_gemm_tn_desc': _ffi.GemmTnDesc, 'cwn_layer_dim': _ffi.LayerDim,
               'cwn_layer_plan': _ffi.LayerPlan, 'cwn_layer_bwd_dim': _ffi.LayerBwdDim, 'cwn_mlp_dim': _ffi.MlpDim, 'cwn_layer_sizes': _ffi.LayerSizes,
               'cwn_embed_table': _ffi.EmbedTable, 'cwn_head_dim': _ffi.HeadDim, 'cwn_head_bwd_dim': _ffi.HeadBwdDim,
               'cwn_layer_bwd_plan': _ffi.LayerBwdPlan, 'cwn_stage_desc': _ffi.StageDesc, 'cwn_stage_bwd_desc': _ffi.StageBwdDesc,
               'cwn_dropout': _ffi.Dropout}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cwn_hip.h"', 'int main(void) {']
    for cname, st in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0; }']
    src = tmp_path / 'probe.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'probe'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True,
                                                   text=True).stdout.splitlines())
    for cname, st in structs.items():
        assert int(got[cname]) == ctypes.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got[f'{cname}.{fname}']) == getattr(st, fname).offset, (cname, fname)
    assert csr.LONG_ROW == int(re.search(r'#define CWN_LONG_ROW (\d+)', open(os.path.join(ROOT, 'include', 'cwn_hip.h')).read()).group(1))


def test_argument_errors_without_gpu():
    lib = _ffi.lib()
    assert lib.cwn_aggregate_f32(None, 1, None) == 1          # CWN_ERR_BAD_ARG
    assert lib.cwn_csr_build(None, 1, None, 0, None, None) == 1
    assert lib.cwn_gather_rows_f32(None, 0, 0, None, 0, None, None) == 1
    d = (_ffi.CsrDesc * 1)(_ffi.CsrDesc(n_entries=10, n_dst=5))
    assert lib.cwn_csr_workspace_bytes(d, 1) > 0
    assert lib.cwn_csr_workspace_bytes(d, 99) == 0
    # every launcher validates before it touches the device
    assert lib.cwn_gemm_f32(None, 1, None) == 1
    assert lib.cwn_gemm_tn_f32(None, 1, None, 0, None) == 1
    assert lib.cwn_bn_finalize_f32(None, 1, None) == 1
    for fn in (lib.cwn_norm_act_f32, lib.cwn_norm_bwd_reduce_f32, lib.cwn_norm_bwd_apply_f32):
        assert fn(None, 1, None) == 1
    assert lib.cwn_norm_bwd_f32(None, 1, 0, None) == 1
    nd = (_ffi.NormDesc * 1)(_ffi.NormDesc(M=_ffi.NORM_BWD_FUSED_MAX_ROWS + 1, N=128))
    assert lib.cwn_norm_bwd_f32(nd, 1, 0, None) == 1                # beyond the one-launch form's rows
    nd[0].M = 0
    assert lib.cwn_norm_bwd_f32(nd, 1, 0, None) == 0                # nothing to do
    assert _ffi.NORM_BWD_FUSED_MAX_ROWS == int(re.search(r'#define CWN_NORM_BWD_FUSED_MAX_ROWS (\d+)', open(os.path.join(ROOT, 'include', 'cwn_hip.h')).read()).group(1))
    assert lib.cwn_adam_f32(None, None, None, None, 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, None, None, None) == 1
    assert lib.cwn_embedding_bwd_f32(None, None, None, None, None, 8, 1, 64, 28, 0, None, None) == 1
    assert lib.cwn_embedding_bwd_f32(None, None, None, None, None, 0, 1, 64, 28, 0, None, None) == 0   # nothing to do
    assert lib.cwn_embedding_fwd_f32(None, None, None, None, None, 8, 1, 64, 28, None, None) == 1
    g = (_ffi.GemmDesc * 1)(_ffi.GemmDesc(M=4, N=8, K=300, K2=0, ldx=300, ldw=300, ldy=8))
    assert lib.cwn_gemm_f32(g, 1, None) == 2                       # CWN_ERR_TOO_LARGE: K beyond the kernel
    t = (_ffi.GemmTnDesc * 1)(_ffi.GemmTnDesc(M=1000, N=128, K=128, K2=128))
    assert lib.cwn_gemm_tn_workspace_bytes(t, 1) >= 8 * (128 * 256 + 128) * 4
    assert lib.cwn_lift_create(7, 3, None, 0, 6, 0) is None        # unknown lift kind
    assert lib.cwn_layer_fused_f32(None, 1, 128, None, 0, None, None) == 1
    assert lib.cwn_layer_pack_weights_f32(None, 256, 128, None, None) == 1
    assert lib.cwn_layer_pack_weights_many_f32(None, None, 128, None, 1, None) == 1
    assert lib.cwn_layer_pack_weights_t_many_f32(None, None, 128, None, 1, None) == 1
    assert lib.cwn_layer_bwd_f32(None, 1, 128, None, None, None) == 1
    assert lib.cwn_layer_bwd_lds_bytes(128, 96) == 96 * 132 * 4 + 3 * 96 * 136 * 2 + 3 * 1024 * 4 and lib.cwn_layer_bwd_lds_bytes(64, 256) == 0
    assert lib.cwn_layer_fused_f32(None, 1, 128, None, _ffi.LAYER_STORE_Y, None, None) == 1
    assert lib.cwn_layer_packed_weight_bytes(128) == 128 * 256 * 6 and lib.cwn_layer_packed_weight_bytes(96) == 0
    assert lib.cwn_layer_fused_lds_bytes(128, 96, 64) == 3 * 96 * 136 * 2 + 65 * 128 * 4 + 9648
    # round-2 additions
    assert lib.cwn_update_mlp_f32(None, 1, 128, None) == 1
    m = (_ffi.MlpDim * 1)(_ffi.MlpDim(M=0))
    assert lib.cwn_update_mlp_f32(m, 1, 96, None) == 1               # widths other than 64 / 128
    assert lib.cwn_update_mlp_f32(m, 1, 64, None) == 0               # nothing to do
    m[0].M = 5                                                        # rows but no operands
    assert lib.cwn_update_mlp_f32(m, 1, 128, None) == 1
    m[0].M = lib.cwn_update_mlp_max_rows() + 1
    assert lib.cwn_update_mlp_f32(m, 1, 128, None) == 2              # CWN_ERR_TOO_LARGE
    assert lib.cwn_update_mlp_packed_weight_bytes(64) == 64 * 64 * 6 and lib.cwn_update_mlp_packed_weight_bytes(32) == 0
    assert lib.cwn_update_mlp_pack_weights_f32(None, 128, 128, None, None) == 1
    assert lib.cwn_gemm_packed_weight_bytes() == 128 * 128 * 6
    assert lib.cwn_gemm_pack_weights_f32(None, 128, None, None) == 1
    assert lib.cwn_layer_items_check(None, 0, 128, None) == 1
    assert lib.cwn_layer_items_build(None, 128, None, 0, None) == _ffi.LAYER_ITEMS_BAD_ARG
    assert lib.cwn_layer_round_rows(128) in (16, 32) and lib.cwn_layer_round_rows(100) == 0
    g = (_ffi.GemmDesc * 1)(_ffi.GemmDesc(M=4, N=64, K=64, K2=0, ldx=64, ldw=64, ldy=64, flags=_ffi.GEMM_W_PACKED))
    assert lib.cwn_gemm_would_split(g, 1) == 0                       # not the split kernel's shape ...
    # round-3 additions: the fused ends (csrc/cwn_ends.hip)
    tv = _ffi.EmbedTable(V=28, cols=1)
    assert lib.cwn_embed_front_f32(None, 4, None, None, 0, None, None, None, 0, 0, None, None, None, 0, 64, 1, None, None, None) == 1
    assert lib.cwn_embed_front_f32(tv, 4, None, None, 0, None, None, None, 0, 0, None, None, None, 0, 66, 1, None, None, None) == 1   # H % 4
    err = ctypes.c_int32(0)
    assert lib.cwn_embed_front_f32(tv, 0, None, None, 0, None, None, None, 0, 0, None, None, None, 0, 64, 1, ctypes.byref(err), None, None) == 0  # nothing to do
    assert lib.cwn_embed_front_f32(tv, 4, None, None, 0, None, None, None, 0, 0, None, None, None, 0, 64, 1, ctypes.byref(err), None, None) == 1  # rows, no table
    assert lib.cwn_loss_f32(0, None, None, 4, None, None, None, None) == 1 and lib.cwn_loss_f32(7, None, None, 4, None, None, None, None) == 1
    hd = (_ffi.HeadDim * 1)(_ffi.HeadDim())
    assert lib.cwn_head_f32(None, 1, 4, 128, 256, 0, 0, None, None, 1, None, None, None, 0, None, 0, 1, None) == 1
    assert lib.cwn_head_f32(hd, 1, 0, 128, 256, 0, 0, None, None, 1, None, None, None, 0, None, 0, 1, None) == 0          # no complexes
    assert lib.cwn_head_f32(hd, 1, 4, 130, 256, 0, 0, None, None, 1, None, None, None, 0, None, 0, 1, None) == 1          # K % 4
    assert lib.cwn_head_f32(hd, 1, 4, 128, 1024, 0, 0, None, None, 1, None, None, None, 0, None, 0, 1, None) == 1         # H2 beyond a workgroup
    assert lib.cwn_head_f32(hd, 4, 4, 128, 256, 0, 0, None, None, 1, None, None, None, 0, None, 0, 1, None) == 1    # more than 3 dimensions
    hb = (_ffi.HeadBwdDim * 1)(_ffi.HeadBwdDim())
    assert lib.cwn_head_bwd_f32(None, 1, 4, 128, 256, 0, 0, None, 1, None, None, 0, 1, None) == 1
    assert lib.cwn_head_bwd_f32(hb, 1, 0, 128, 256, 0, 0, None, 1, None, None, 0, 1, None) == 0             # no complexes
    assert lib.cwn_head_bwd_f32(hb, 1, 4, 128, 256, 0, 0, None, 1, None, None, 0, 1, None) == 1             # no operands
    # round 5: dropout (csrc/cwn_dropout.h)
    dr = _ffi.Dropout(state=None, p=0.5, site=1)
    assert lib.cwn_dropout_f32(None, None, 4, 64, 64, 64, None, None, None) == 1                         # no record
    assert lib.cwn_dropout_f32(None, None, 0, 64, 64, 64, dr, None, None) == 0                           # nothing to do
    assert lib.cwn_dropout_f32(None, None, 4, 64, 64, 64, dr, None, None) == 1                           # rows, no operands
    bad = _ffi.Dropout(state=None, p=1.0, site=1)
    assert lib.cwn_dropout_f32(None, None, 0, 64, 64, 64, bad, None, None) == 1                          # p must be < 1
    assert lib.cwn_step_begin(None, 0, None, 0, None, None, None, None) == 0
    assert lib.cwn_collate_guard(None, 3, 10, 8, 1, 0, None, None, None) == 1


def test_item_table_builder_rejects_tables_that_are_not_prefix_sums():
    """cwn_layer_items_build (host C++) reads caller-provided per-complex tables: one that does not start at 0,
    decreases, or sums past the int32 record fields is refused, not cut into records with negative counts."""
    import numpy as np
    from cwn_amd.blockplan import ITEM_INTS
    lib = _ffi.lib()
    good = dict(cells=[[0, 5, 9], [0, 6, 10]], up=[0, 12, 20], b=[0, 12, 20])

    def build(cells, up, b, cap=4):
        keep = [np.asarray(a, dtype=np.int64) for a in (cells[0], cells[1], up, b)]
        sizes = _ffi.LayerSizes(n_complexes=2, n_dims=2)
        sizes.has_up[0], sizes.has_up[1] = 1, 0
        sizes.cell_ptr[0], sizes.cell_ptr[1] = keep[0].ctypes.data, keep[1].ctypes.data
        sizes.up_ptr[0] = keep[2].ctypes.data
        sizes.b_ptr[1] = keep[3].ctypes.data
        table = np.zeros((cap, ITEM_INTS), dtype=np.int32)
        return int(lib.cwn_layer_items_build(sizes, 64, table.ctypes.data, cap, _ffi.LayerPlan())), table
    n, table = build(good['cells'], good['up'], good['b'])
    assert n >= 1 and int(table[:n, 3].sum()) == 9            # every vertex in exactly one record
    bad = [dict(good, cells=[[1, 5, 9], [0, 6, 10]]),        # does not start at 0
           dict(good, cells=[[0, 5, 9], [0, 6, 4]]),         # decreases
           dict(good, up=[0, 12, 7]),
           dict(good, b=[0, -3, 20]),
           dict(good, up=[0, 12, 2 ** 31])]                  # past the int32 fields of a record
    for kw in bad:
        assert build(kw['cells'], kw['up'], kw['b'])[0] == _ffi.LAYER_ITEMS_BAD_ARG, kw


def test_cpu_tensors_fail_loudly():
    from cwn_amd import ops
    x = torch.randn(4, 8)
    with pytest.raises(_ffi.CwnError, match='no CPU fallback'):
        ops.gather_rows(x, torch.tensor([0, 1]))
    cmp = CochainMessagePassing(8, 8)
    idx = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(_ffi.CwnError):
        cmp.propagate(idx, None, None, x=x, up_attr=None)


@pytest.mark.parametrize('lname', ['testing', 'testing3', 'mol', 'pair', 'nodes_only'])
def test_batching_layout_integer_exact(lname):
    g = load('batching.npz')
    names = [str(n) for n in g[f'{lname}/names']]
    md = int(g[f'{lname}/max_dim'])
    b = dummy_batch(names, max_dim=md)
    ref = complex_dict(g, f'{lname}/batch')
    assert b.dimension == ref['dimension'] and b.num_complexes == len(names)
    assert torch.equal(b.y, ref['y'])
    for d in range(ref['dimension'] + 1):
        for k in ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries',
                  'boundary_index', 'y', 'batch'):
            a, r = b.cochains[d][k], ref['cochains'][d][k]
            assert (a is None) == (r is None), (d, k)
            if a is not None:
                assert a.dtype == r.dtype and torch.equal(a, r), (d, k)
        assert b.cochains[d].num_cells == ref['cochains'][d]['num_cells']
    # propagate arguments: indices identical; lazy attributes resolve to the golden gathers
    for kw_name, kw in (('full', {}), ('nodown', dict(include_down_features=False))):
        for d, prm in enumerate(b.get_all_cochain_params(max_dim=md, **kw)):
            pre = f'{lname}/params_{kw_name}/{d}'
            for k, v in (('x', prm.x), ('up_index', prm.up_index), ('down_index', prm.down_index),
                         ('boundary_index', prm.boundary_index),
                         ('boundary_attr', prm.kwargs['boundary_attr'])):
                assert (v is None) == (f'{pre}/{k}' not in g), (d, k)
                if v is not None:
                    assert torch.equal(v, T(g[f'{pre}/{k}'])), (d, k)
            for k in ('up_attr', 'down_attr'):
                v = prm.kwargs[k]
                assert (v is None) == (f'{pre}/{k}' not in g), (d, k)
                if v is not None:
                    assert isinstance(v, IndexedRows)
                    assert torch.equal(v.src.index_select(0, v.index), T(g[f'{pre}/{k}'])), (d, k)


def test_house_params_known_answer():           # data/test_data.py:6-54
    h = dummy_complex('house')
    v, e = h.get_cochain_params(dim=0), h.get_cochain_params(dim=1)
    ua = v.kwargs['up_attr']
    assert ua.src[ua.index].flatten().tolist() == [1, 1, 4, 4, 2, 2, 3, 3, 6, 6, 5, 5]
    da = e.kwargs['down_attr']
    assert da.src[da.index].flatten().tolist() == [2, 2, 1, 1, 3, 3, 3, 3, 4, 4, 4, 4, 3, 3, 4, 4, 5, 5]
    assert e.kwargs['boundary_attr'].flatten().tolist() == [1, 2, 3, 4, 5]


def test_index_contract_errors():
    """mp/cell_mp.py:153-193: AssertionError for dtype / shape, ValueError for foreign index types."""
    cmp = CochainMessagePassing(1, 1)
    x = torch.zeros(3, 1)
    with pytest.raises(AssertionError):
        cmp.propagate(torch.tensor([[0, 1], [1, 0]], dtype=torch.int32), None, None, x=x, up_attr=None)
    with pytest.raises(AssertionError):
        cmp.propagate(torch.tensor([0, 1]), None, None, x=x, up_attr=None)
    with pytest.raises(AssertionError):
        cmp.propagate(torch.zeros(3, 2, dtype=torch.long), None, None, x=x, up_attr=None)
    with pytest.raises(ValueError):
        cmp.propagate([[0, 1], [1, 0]], None, None, x=x, up_attr=None)
    with pytest.raises(AssertionError):   # __check_input_together__ (:146-151)
        i = torch.tensor([[0, 1], [1, 0]])
        cmp.propagate(i, i, None, up_size=(3, 3), down_size=(4, 3), x=x, up_attr=None, down_attr=None)


def test_hook_override_detection():
    base = CochainMessagePassing(1, 1)
    assert base._identity_path('up') and base._identity_path('down') and base._identity_path('boundary')
    assert not (base.fuse_up or base.fuse_down or base.fuse_boundary)

    class Custom(CochainMessagePassing):
        def message_up(self, up_x_j, up_attr, up_x_i):
            return up_x_j - up_x_i

    c = Custom(1, 1)
    assert not c._identity_path('up') and c._identity_path('down')
    assert c._args_of(['message_up']) == {'up_x_j', 'up_attr', 'up_x_i'}
    d = DummyCochainMessagePassing(1, 1)
    assert d.fuse_up and d.fuse_down and not d.fuse_boundary
    with pytest.raises(TypeError, match='Required parameter'):
        c._distribute('message_up', {'up_x_j': 1})


def test_constructor_contract():
    with pytest.raises(AssertionError):
        CochainMessagePassing(1, 1, aggr_up='median')
    with pytest.raises(AssertionError):
        CochainMessagePassing(1, 1, flow='sideways')
    m = CochainMessagePassing(3, 5)
    assert m.boundary_msg_size == 5 and m.up_msg_size == 3   # boundary defaults to down (:100)
    assert CochainMessagePassing(3, 5, boundary_msg_size=7).boundary_msg_size == 7


def test_sparse_cin_state_dict_matches_reference_names():
    """Parameter names equal the reference's, so its state_dicts load unchanged."""
    g = load('sparse_cin_conv.npz')
    tag = 'mol_cob_bn'
    F, H, cob, bn = g[f'{tag}/meta'].tolist()
    conv = SparseCINConv(F, F, F, None, None, None, None, train_eps=True, max_dim=2, hidden=H,
                         act_module=torch.nn.ReLU, layer_dim=F, use_coboundaries=bool(cob))
    ref_keys = {k[len(f'{tag}/state/'):] for k in g if k.startswith(f'{tag}/state/')}
    assert set(conv.state_dict().keys()) == ref_keys
    lvl = conv.mp_levels[1]
    assert lvl._up_kind() == 'cat_linear_relu' and lvl._boundary_fusable()
    nocob = SparseCINConv(F, F, F, None, None, None, None, max_dim=2, hidden=H,
                          act_module=torch.nn.ReLU, layer_dim=F, use_coboundaries=False)
    assert nocob.mp_levels[0]._up_kind() == 'first'
    custom = SparseCINCochainConv(0, F, F, F, lambda xs: xs[0], lambda x: x, torch.nn.Identity(),
                                  torch.nn.Identity(), torch.nn.Identity())
    assert custom._up_kind() == 'custom' and not custom._boundary_fusable()


def test_complex_prepare_requires_gpu():
    with pytest.raises(RuntimeError, match='GPU'):
        dummy_complex('house').prepare()


def test_bench_contract_static():
    """bench.py (needs a GPU to run) keeps the driver's contract: flags and the keys of its JSON line."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    for flag in ('--gpus', '--steps', '--warmup'):
        assert f"'{flag}'" in src, flag
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert f"'{key}':" in src, key
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert f"'{key}':" in src, key
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert f"'{key}':" in src, key
    # nothing under cwn_amd/ imports the oracle (it is the checker, never the product)
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'cwn_amd')):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, f
    # and nothing that runs on the GPU box reads /root/reference
    for f in ('bench.py', 'bench_fresh.py', '__graft_entry__.py', os.path.join('tests', 'test_gpu_parity.py')):
        assert '/root/reference' not in open(os.path.join(ROOT, f)).read(), f


def test_aggregate_descriptor_small_operand_flag():
    """ops.AggSpec vouches for 32-bit row offsets (CWN_AGG_SMALL_OPERANDS) exactly when the gathered
    operands lie within 4 GiB of their base pointers; the test switch turns it off."""
    import re
    import torch
    from cwn_amd import ops
    hdr = open(os.path.join(ROOT, 'include', 'cwn_hip.h')).read()
    assert ops.AGG_SMALL_OPERANDS == int(re.search(r'#define CWN_AGG_SMALL_OPERANDS (\d+)', hdr).group(1))
    A, out = torch.zeros(10, 8), torch.zeros(4, 8)
    spec = ops.AggSpec(adj=None, n_dst=4, F=8, A=A, out=out)
    assert spec.desc().flags == ops.AGG_SMALL_OPERANDS

    class Huge:                      # a stand-in with the size of a > 4 GiB operand
        def numel(self): return (1 << 30) + 1
        def size(self, d): return 8
        def data_ptr(self): return A.data_ptr()
    big = ops.AggSpec(adj=None, n_dst=4, F=8, A=Huge(), out=out)
    assert big.desc().flags == 0
    prev = ops.ALLOW_SMALL_OPERANDS
    try:
        ops.ALLOW_SMALL_OPERANDS = False
        assert spec.desc().flags == 0
    finally:
        ops.ALLOW_SMALL_OPERANDS = prev


def test_bench_gemm_roofline_pricing():
    """The roofline entry of the message GEMM: the exact kernel against the fp32-MFMA peak, the split
    kernel against HBM (its matrix-pipe floor, bf16 peak / 6, lies below the HBM floor at N = K = 128)
    with the matrix-pipe figures beside it; every contract key present either way."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_for_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                       # defines functions only (main() is guarded)
    rows = 10151
    flops, io = 2.0 * rows * 128 * 128, 4 * (2 * rows * 128 + 4 * 128 * 128)
    exact = mod.gemm_roofline(flops=flops, us=8.8, split=False, io_bytes=io, narrow=False, traffic=123)
    split = mod.gemm_roofline(flops=flops, us=8.4, split=True, io_bytes=io, narrow=False, traffic=None)
    for r in (exact, split):
        assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel'} <= set(r)
        assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0 < r['frac'] < 1
    assert exact['bound'] == 'mfma' and exact['unit'] == 'TFLOP/s' and exact['peak'] == 157.3
    assert split['bound'] == 'hbm' and split['unit'] == 'GB/s' and split['peak'] == 8000.0
    assert abs(split['achieved'] - io / 8.4e-6 / 1e9) < 1.0
    assert abs(split['matrix_pipe_ceiling_tflops'] - 2500.0 / 6) < 0.1
    # the reason for the choice of bound: per row, six bf16 MFMAs per term cost less than 1 KiB of HBM
    assert (2 * 128 * 128) / (2500e12 / 6) < 1024 / 8e12


def test_integration_stub_structs_match_the_header():
    """INTEGRATION.md shows the ctypes binding a maintainer of the reference would paste: its struct
    layouts must be the library's (a stale stub corrupts memory silently)."""
    import ctypes as C
    import re
    from cwn_amd import _ffi
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    found = {}
    code = re.sub(r'#[^\n]*', '', text)
    for name, body in re.findall(r'class (\w+)\(C\.Structure\):\s*\n\s+_fields_ = \[(.*?)\]\s*\n', code, re.S):
        found[name] = eval('[' + body + ']', {'C': C})
    assert {'CsrDesc', 'AggDesc'} <= set(found)
    for name, fields in found.items():
        assert fields == list(getattr(_ffi, name)._fields_), name
    m = re.search(r'_L\.cwn_abi_version\(\) == (\d+)', text)
    assert m and int(m.group(1)) == _ffi.ABI_VERSION


def test_three_way_bf16_split_is_exact_numpy_model():
    """The arithmetic identity csrc/cwn_split.h rests on, modelled in numpy float32 / uint32:
    x = hi + mid + lo EXACTLY with every piece a bf16 number (round to nearest even at each step),
    |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x| (2^-9, 2^-18 away from rounding boundaries), so that the six
    products the kernels keep differ from x * w by the three dropped ones: < 2^-24 |x||w|."""
    import numpy as np
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200_000), rng.standard_normal(200_000) * 1e-20,
                        rng.standard_normal(200_000) * 1e20, [0.0, -0.0, 1.0, -1.0, 3.0, 2.0 ** -100,
                                                               np.float32(16777215.0), 1.0 + 2.0 ** -23,
                                                               1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -9]]).astype(np.float32)

    def bf16_rne(v):                       # fp32 -> nearest bf16 (ties to even), as fp32
        u = v.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)

    def split3(v):
        h = bf16_rne(v)
        r1 = (v - h).astype(np.float32)
        m = bf16_rne(r1)
        r2 = (r1 - m).astype(np.float32)
        return h, m, bf16_rne(r2), r2

    h, m, l, r2 = split3(x)
    assert np.array_equal(l, r2)                                        # the last piece is exact
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    for piece in (h, m, l):
        assert not np.any(piece.view(np.uint32) & np.uint32(0xFFFF))    # representable in bf16
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(m.astype(np.float64)) <= ax * 2.0 ** -8) and np.all(np.abs(l.astype(np.float64)) <= ax * 2.0 ** -16)
    # the six kept products against the exact product
    w = (rng.standard_normal(x.size) / 16).astype(np.float32)
    wh, wm, wl, _ = split3(w)
    f = lambda a: a.astype(np.float64)
    kept = f(wl) * f(h) + f(wh) * f(l) + f(wm) * f(m) + f(wm) * f(h) + f(wh) * f(m) + f(wh) * f(h)
    exact = f(x) * f(w)
    err = np.abs(kept - exact)
    assert np.all(err <= np.abs(exact) * 2.0 ** -24 + 1e-300), float((err / np.maximum(np.abs(exact), 1e-300)).max())
    # and the residuals are not of one sign (the truncating split of round 1 had a bias)
    nz = np.abs(kept - exact) > 0
    assert 0.3 < np.mean((kept - exact)[nz] > 0) < 0.7


def test_item_table_derived_fields_and_host_check():
    """cwn_amd/blockplan.py writes the derived record fields the layer kernel reads instead of re-deriving
    them (include/cwn_hip.h: R1, staged rows, entry segments); cwn_layer_items_check accepts the builder's
    tables and rejects every single-field corruption that changes what the kernel would do."""
    import numpy as np
    from cwn_amd import _ffi
    from cwn_amd.blockplan import BlockPlan, ITEM_INTS
    from cwn_amd.synthetic import zinc_like_batch
    L = _ffi.lib()
    b = zinc_like_batch(24, seed=5)
    plan = BlockPlan.from_batch(b)
    for F in (64, 128):
        ng = L.cwn_layer_round_rows(F)
        assert ng == 1024 // (F // 4) or ng == 512 // (F // 4)
        t = plan.items(F, [True, True, False])
        tab = t.items.numpy()
        assert tab.shape[1] == ITEM_INTS and t.n_items == tab.shape[0]
        cplan = t.c_plan(False)
        assert L.cwn_layer_items_check(tab.ctypes.data, tab.shape[0], F, cplan) == 0
        pad16 = lambda n: (n + 15) // 16 * 16
        pad4 = lambda n: (n + 3) // 4 * 4
        for r in tab:
            n0, nc, une = int(r[11]), int(r[5]), int(r[7])
            r1 = pad16(n0)                         # (round 4: whole 16-row tiles, no longer whole load rounds of `ng` rows)
            assert r[23] == r1 and r[24] == (r1 + pad16(nc) if nc > 0 else pad16(n0))
            assert r[25] == pad4(une) and r[26] == pad4(r[25] + r[13]) and r[27] == pad4(r[26] + r[20])
            assert not r[28:].any()
            for t_ in range(2):                    # sources are staged only when the task has boundary entries
                o = 9 + 7 * t_
                assert (r[o + 6] > 0) == (r[o + 4] > 0)
        rng = np.random.default_rng(0)
        for col in (3, 5, 7, 11, 13, 15, 23, 24, 25, 26, 27, 28):
            bad = tab.copy()
            bad[rng.integers(0, tab.shape[0]), col] += 4
            assert L.cwn_layer_items_check(bad.ctypes.data, bad.shape[0], F, cplan) != 0, col
        neg = tab.copy()
        neg[0, 11] = -1
        assert L.cwn_layer_items_check(neg.ctypes.data, neg.shape[0], F, cplan) != 0
        assert L.cwn_layer_items_check(tab.ctypes.data, tab.shape[0] - 1, F, cplan) != 0      # n_items mismatch
        empty = np.zeros((3, ITEM_INTS), dtype=np.int32)                                       # static-graph filler
        cplan.n_items = 3
        assert L.cwn_layer_items_check(empty.ctypes.data, 3, F, cplan) == 0


def test_item_table_is_keyed_on_the_boundary_streams_the_layer_runs():
    """a layer that does not run the boundary stream of a dimension (use_boundary_msg=False,
    get_all_cochain_params(include_boundary_features=False)) hands the launcher no boundary_index for it, so its
    table may not carry boundary entries for it: `has_b` is part of the table key, and the records / the plan
    summary of a dimension that is off name no entry of its index."""
    import numpy as np
    from cwn_amd import _ffi
    from cwn_amd.blockplan import BlockPlan
    from cwn_amd.synthetic import zinc_like_batch
    L = _ffi.lib()
    plan = BlockPlan.from_batch(zinc_like_batch(24, seed=5))
    for F in (64, 128):
        full = plan.items(F, [True, True, False])
        assert plan.items(F, [True, True, False], [False, True, True]) is full         # dim 0 never has one
        assert full.b_end[1] > 0 and full.b_end[2] > 0
        for has_b in ([False, False, False], [False, True, False], [False, False, True]):
            t = plan.items(F, [True, True, False], has_b)
            assert t is not full and plan.items(F, [True, True, False], has_b) is t  # cached under its own key
            tab = t.items.numpy()
            assert L.cwn_layer_items_check(tab.ctypes.data, tab.shape[0], F, t.c_plan(False)) == 0
            for d in (1, 2):
                recs = [(r, o) for r in tab for o in (9, 16) if r[8] > (o - 9) // 7 and r[o] == d]
                assert recs
                if has_b[d]:
                    assert t.b_end[d] == full.b_end[d] and sum(int(r[o + 4]) for r, o in recs) == full.b_end[d]
                else:
                    assert t.b_end[d] == 0 and all(r[o + 4] == 0 and r[o + 6] == 0 for r, o in recs)


def test_item_table_big_records_for_complexes_beyond_the_caps():
    """VERDICT r2 item 4 (host side): with allow_big a complex that no workgroup's LDS holds becomes one BIG record per
    set (flag bit 1: its workgroup streams it) instead of failing the whole table; every other complex is cut as
    before, the records cover each set exactly once, the host check accepts the table and counts the BIG records,
    and the two-per-CU form refuses them."""
    import numpy as np
    from cwn_amd import _ffi
    from cwn_amd.blockplan import BlockPlan
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    L = _ffi.lib()
    cxs = zinc_like_complexes(20, 0, 6)
    giants = zinc_like_complexes(2, 1, 6, n_lo=70, n_hi=90)
    cxs = cxs[:5] + giants[:1] + cxs[5:] + giants[1:]
    b = ComplexBatch.from_complex_list(cxs, max_dim=2)
    plan = BlockPlan.from_batch(b)
    F = 128
    assert plan.items(F, [True, True, False]) is None                        # as before: a complex is too large
    assert plan.items(F, [True, True, False], None, 1, True) is None          # the two-per-CU form has no BIG records
    t = plan.items(F, [True, True, False], None, 0, True)
    assert t is not None and t.n_big == 4 and t.variant == 0                  # 2 giants x 2 sets
    tab = t.items.numpy()
    big = tab[(tab[:, 0] & 2) != 0]
    assert len(big) == 4 and t.big_records.shape == big.shape
    assert L.cwn_layer_items_check(tab.ctypes.data, tab.shape[0], F, t.c_plan(False)) == 0
    wrong = t.c_plan(False)
    wrong.n_big = 3
    assert L.cwn_layer_items_check(tab.ctypes.data, tab.shape[0], F, wrong) != 0
    n_cells = [np.asarray(plan.cells[d]) for d in range(3)]
    for r in big:
        assert not r[23:].any()                                               # no derived fields: nothing is staged
        c = int(np.searchsorted(plan.cell_ptr[int(r[9])], r[10], side='right')) - 1
        assert c in (5, len(cxs) - 1) and r[11] == n_cells[int(r[9])][c]      # exactly one giant, all its cells
        pad16 = lambda n: (int(n) + 15) // 16 * 16
        assert pad16(r[11]) + pad16(r[5]) > 96 or r[15] + r[22] > 96           # ... staged rows or boundary sources beyond the caps
    # coverage: per set the task-0 rows tile the dimension exactly once (BIG records included)
    for s_, d0 in ((0, 0), (1, 1)):
        rows = tab[(tab[:, 0] >> 8) == s_]
        order = np.argsort(rows[:, 10], kind='stable')
        assert int(rows[:, 11].sum()) == int(n_cells[d0].sum())
        ends = rows[order, 10] + rows[order, 11]
        assert (rows[order, 10][1:] == ends[:-1]).all() and rows[order, 10][0] == 0
    # heavy first: the BIG records lead their sets
    for s_ in (0, 1):
        lo = t.set_start[s_]
        assert tab[lo, 0] & 2 and tab[lo + 1, 0] & 2


def test_block_plan_covers_every_complex_once_and_fits_the_launch():
    """cwn_amd/blockplan.py on random per-complex size tables (no tensors, no GPU): the items of a set are
    contiguous ranges of complexes that cover the batch exactly once, every item respects the caps, ONE LDS
    size serves the whole launch (the builder's split between staged rows and boundary sources), the count is
    at least the O(1) lower bound, and a complex that cannot fit gives no table at all."""
    import numpy as np
    from cwn_amd import _ffi
    from cwn_amd.blockplan import BlockPlan, LDS_BYTES, MAX_ENTRIES, TASK_ROWS, gemm_rows_cap, lds_bytes
    from tests import _blockplan_ref
    rng = np.random.default_rng(7)
    L = _ffi.lib()
    for trial in range(40):
        C = int(rng.integers(1, 400))
        big = trial % 5 == 0
        n0 = rng.integers(1, 60 if big else 30, size=C)
        n1 = n0 + rng.integers(-1, 4, size=C).clip(min=0)           # edges ~ vertices
        n2 = rng.integers(0, 4, size=C)
        n2[n1 == 0] = 0
        up0 = 2 * n1                                                # two directed entries per edge
        up1 = n2 * rng.integers(6, 31, size=C)                      # edges sharing a ring
        b1 = 2 * n1
        b2 = n2 * rng.integers(3, 7, size=C)
        ptr = lambda v: np.concatenate([[0], np.cumsum(v)])
        plan = BlockPlan([n0, n1, n2], [ptr(up0), ptr(up1), None], [None, ptr(b1), ptr(b2)])
        for F in (64, 128):
            t = plan.items(F, [True, True, False])
            cap = gemm_rows_cap(F)
            # the C++ builder (cwn_layer_items_build) against its Python restatement: the same table, field for field
            r = _blockplan_ref.build(plan, F, (True, True, False))
            assert (t is None) == (r is None), (trial, F)
            if t is not None:
                assert np.array_equal(t.items.numpy(), r.items.numpy()) and t.set_start == r.set_start
                assert (t.max_rows, t.max_src, t.cells_end, t.up_end, t.b_end) == (r.max_rows, r.max_src, r.cells_end, r.up_end, r.b_end)
            if t is None:
                # only legitimate when some single complex exceeds a cap
                def staged(a, b):
                    ng = L.cwn_layer_round_rows(F)
                    r1 = (a + 15) // 16 * 16
                    return ((r1 + ng - 1) // ng * ng + (b + 15) // 16 * 16) if b > 0 else r1
                too_big = any(staged(int(a), int(b)) > cap or staged(int(b), int(c)) > cap or int(u0) + 3 > MAX_ENTRIES
                              or int(u1) + int(e1) + int(e2) + 9 > MAX_ENTRIES or int(a) + int(b) > cap
                              or lds_bytes(F, staged(int(b), int(c)), int(a) + int(b)) > LDS_BYTES
                              for a, b, c, u0, u1, e1, e2 in zip(n0, n1, n2, up0, up1, b1, b2))
                assert too_big, (trial, F)
                continue
            tab = t.items.numpy()
            assert t.n_items >= plan.at_least(F, [True, True, False]) or t.n_items >= 2
            assert lds_bytes(F, t.max_rows, t.max_src) <= LDS_BYTES
            assert L.cwn_layer_fused_lds_bytes(F, t.max_rows, t.max_src) == lds_bytes(F, t.max_rows, t.max_src)
            assert tab[:, 24].max() <= t.max_rows <= cap and tab[:, 27].max() <= MAX_ENTRIES
            assert tab[:, 11].max() <= TASK_ROWS and tab[:, 18].max() <= TASK_ROWS
            # coverage: per set, the task-0 cell ranges tile [0, N_d) exactly once
            for s_, d0 in ((0, 0), (1, 1)):
                lo, hi = t.set_start[s_], (t.set_start[s_ + 1] if s_ + 1 < len(t.set_start) else t.n_items)
                rows = tab[lo:hi]
                assert ((rows[:, 0] >> 8) == s_).all()
                order = np.argsort(rows[:, 10], kind='stable')
                starts, counts = rows[order, 10], rows[order, 11]
                total = int([n0, n1][d0].sum())
                live = counts > 0
                assert int(counts.sum()) == total
                assert (starts[live][1:] == (starts[live] + counts[live])[:-1]).all() and (total == 0 or starts[live][0] == 0)
            # the rings ride as the second task of the edges' items, the same complexes
            rows = tab[t.set_start[1]:]
            assert int(rows[:, 18].sum()) == int(n2.sum())


def test_collate_host_half_equals_its_first_form():
    """PackedComplexes._prepare (all tables of a batch from stacked metadata, a dozen numpy calls) against the
    first, per-key form kept in tests/_collate_ref.py: same output tensors, same descriptors, the same table
    CONTENTS behind every descriptor, the same bookkeeping on the CochainBatch objects -- for mixed datasets
    (complexes of dimension 0, 1, 2; with / without lower adjacencies; labels or none) and arbitrary index subsets
    with repeats.  CPU-resident packed dataset: the launch itself is a GPU test."""
    import numpy as np
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.synthetic import zinc_like_complexes, ring_lift
    from tests import _collate_ref
    from tests._product import dummy_complex, list_names
    pools = {
        'zinc': zinc_like_complexes(60, seed=3),
        'zinc_down': zinc_like_complexes(30, seed=4, include_down_adj=True),
        'dummies': [dummy_complex(n) for n in list_names('testing')],
        'mixed_dims': [ring_lift(4, [], torch.zeros(4, 1), y=torch.tensor([1.0])),                       # dimension 0
                       ring_lift(5, [(0, 1), (1, 2), (2, 3), (3, 4)], torch.ones(5, 1), torch.ones(4, 1), y=torch.tensor([2.0])),
                       *zinc_like_complexes(6, seed=5)],
    }
    # the five lists (and their max_dim) of the reference's batching tests, the ones the GPU layout test collates
    g = load('batching.npz')
    dims = {}
    for lname in ('testing', 'testing3', 'mol', 'pair', 'nodes_only'):
        pools['golden_' + lname] = [dummy_complex(str(n)) for n in g[f'{lname}/names']]
        dims['golden_' + lname] = int(g[f'{lname}/max_dim'])
    rng = np.random.default_rng(0)
    for name, pool in pools.items():
        p = PackedComplexes(pool, 'cpu', max_dim=dims.get(name, 2))
        subsets = [list(range(len(pool))), [0], [len(pool) - 1, 0, 0]] + \
                  [rng.integers(0, len(pool), size=int(rng.integers(1, 2 * len(pool)))).tolist() for _ in range(6)]
        if name == 'mixed_dims':
            subsets += [[0, 0], [1, 0], [0, 1, 0]]             # batches whose dimension is below the dataset's
        for idx in subsets:
            c_new, y_new, t_new, plan_new = p._prepare(idx)
            c_old, y_old, t_old, plan_old = _collate_ref.prepare(p, idx)
            B = len(idx)
            assert len(c_new) == len(c_old) and (y_new is None) == (y_old is None)
            if y_new is not None:
                assert y_new.shape == y_old.shape and y_new.dtype == y_old.dtype

            def content(plan, tables):
                out = []
                for pk, t, o_dst, o_src, o_add, total in plan:
                    rows = 0 if pk is None else pk.rows
                    out.append((id(pk), tuple(t.shape), t.dtype, total, tables[o_dst:o_dst + B + 1].tolist(),
                                None if o_src is None else tables[o_src:o_src + B].tolist(),
                                None if o_add is None else tables[o_add:o_add + rows * B].tolist()))
                return sorted(out, key=lambda e: (e[0], e[1]))
            assert content(plan_new, t_new) == content(plan_old, t_old), (name, idx)
            for a, b in zip(c_new, c_old):
                for attr in ('__num_cells_list__', '__slices__', '__num_cells__', '__num_cells_up__', '__num_cochains__', 'ptr'):
                    assert getattr(a, attr, None) == getattr(b, attr, None), (name, attr)
                assert getattr(a, '__num_cells_down__', None) == getattr(b, '__num_cells_down__', None)
                for key in ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries', 'boundary_index', 'batch'):
                    ta, tb = a[key] if key != 'x' else a._x, b[key] if key != 'x' else b._x
                    assert (ta is None) == (tb is None), (name, key)
                    assert ta is None or (ta.shape == tb.shape and ta.dtype == tb.dtype), (name, key)
    with pytest.raises(ValueError):
        p._prepare([])


def _interpret_collate(plan, tables, B):
    """What csrc/cwn_collate.hip::collate_kernel does with the descriptors PackedComplexes.collate makes of `plan`,
    on the CPU: per descriptor and segment s, dst[r, d0:d0+len] = src[r, s0:s0+len] (+ add[r * B + s] for index
    arrays), or = s for a batch vector."""
    for pk, out, o_dst, o_src, o_add, total in plan:
        if total == 0:
            continue
        dst_start = tables[o_dst:o_dst + B + 1]
        flat = out.view(-1)
        if pk is None:
            for s in range(B):
                flat[dst_start[s]:dst_start[s + 1]] = s
            continue
        src_start = tables[o_src:o_src + B]
        src = pk.data.view(-1)
        s_stride = pk.data.size(-1) if pk.rows == 2 else 0
        d_stride = total if pk.rows == 2 else 0
        add64 = pk.op == _ffi.COLLATE_ADD64
        assert not add64 or o_add is not None          # cwn_collate rejects ADD64 without an add table
        for s in range(B):
            d0, n = int(dst_start[s]), int(dst_start[s + 1] - dst_start[s])
            for r in range(pk.rows):
                a = int(tables[o_add + r * B + s]) if add64 else 0
                seg = src[r * s_stride + int(src_start[s]): r * s_stride + int(src_start[s]) + n]
                flat[r * d_stride + d0: r * d_stride + d0 + n] = seg + a if add64 else seg


def test_device_collate_descriptors_reproduce_the_reference_layout_on_the_cpu():
    """The whole device collate minus the launch: PackedComplexes._prepare's descriptors, executed by a CPU
    restatement of collate_kernel, must give the batch `ComplexBatch.from_complex_list` gives (itself pinned on the
    reference's batching fixtures) -- index arrays with their per-complex offsets, features, batch vectors, labels."""
    import numpy as np
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.synthetic import zinc_like_complexes
    g = load('batching.npz')
    pools = {'zinc': (zinc_like_complexes(40, seed=11, include_down_adj=True), 2)}
    for lname in ('testing', 'testing3', 'mol', 'pair', 'nodes_only'):
        pools[lname] = ([dummy_complex(str(n)) for n in g[f'{lname}/names']], int(g[f'{lname}/max_dim']))
    rng = np.random.default_rng(3)
    for name, (pool, md) in pools.items():
        p = PackedComplexes(pool, 'cpu', max_dim=md)
        for idx in [list(range(len(pool)))] + [rng.permutation(len(pool))[:max(1, len(pool) // 2)].tolist() for _ in range(3)]:
            cochains, y, tables, plan = p._prepare(idx)
            _interpret_collate(plan, tables, len(idx))
            ref = ComplexBatch.from_complex_list([pool[i] for i in idx], max_dim=md)
            assert len(cochains) == ref.dimension + 1, name
            assert (y is None) == (ref.y is None) and (y is None or torch.equal(y.view(-1), ref.y.view(-1).to(y.dtype)))
            for d, cb in enumerate(cochains):
                rc = ref.cochains[d]
                for key in ('upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries', 'boundary_index', 'batch'):
                    a, b = cb[key], rc[key]
                    assert (a is None) == (b is None), (name, d, key)
                    assert a is None or torch.equal(a, b), (name, d, key)
                assert (cb._x is None) == (rc.x is None) and (cb._x is None or torch.equal(cb._x, rc.x)), (name, d)
                assert cb.num_cells == rc.num_cells
                for key, sl in cb.__slices__.items():          # per-complex entry offsets: what the item table is cut from
                    assert key not in rc.__slices__ or sl == rc.__slices__[key], (name, d, key)
                yv, ry = cb['y'], rc['y']
                assert (yv is None) == (ry is None) and (yv is None or torch.equal(yv.view(-1), ry.view(-1))), (name, d)


def test_compiled_binding_module_loads_and_matches_the_abi():
    """cwn_amd/_cwn_torch_ext.so (csrc/cwn_torch_ext.cpp, built by build()): loads without a GPU, was compiled against this
    header, registers its torch.library ops; its descriptor-size guards refuse an array of another size."""
    import ctypes
    import torch
    from cwn_amd import _build_ext, _cext, _ffi
    _build_ext.build(verbose=False)
    X = _cext.ext()
    assert X is not None and int(X.abi_version) == _ffi.ABI_VERSION == _ffi.lib().cwn_abi_version()
    assert _cext.active() == 'compiled'
    with _cext.binding('ctypes'):
        assert _cext.ext() is None and _cext.active() == 'ctypes'
    assert _cext.ext() is X
    assert hasattr(torch.ops.cwn, 'layer_fused') and hasattr(torch.ops.cwn, 'update_mlp')
    fn = _cext.fn_address(_ffi.lib().cwn_layer_fused_f32)
    arr = (_ffi.LayerDim * 2)()
    with pytest.raises(ValueError, match='ABI'):
        X.LayerCall(ctypes.addressof(arr), ctypes.sizeof(arr) - 8, 2, 64, [3, 4], 0, fn, 0, 2)
    with pytest.raises(ValueError, match='two or three outputs'):
        X.LayerCall(ctypes.addressof(arr), ctypes.sizeof(arr), 2, 64, [3, 4], 0, fn, 0, 4)
    c = X.LayerCall(ctypes.addressof(arr), ctypes.sizeof(arr), 2, 64, [3, 4], 0, fn, 0, 2)
    assert not c.has_plans(False) and not c.has_plans(True)
    with pytest.raises(ValueError, match='one feature tensor per dimension'):
        c.run([torch.zeros(3, 64)], 0, None)
    with pytest.raises(ValueError, match='plans'):
        c.run([torch.zeros(3, 64), torch.zeros(4, 64)], 0, None)
    marr = (_ffi.MlpDim * 1)()
    mfn = _cext.fn_address(_ffi.lib().cwn_update_mlp_f32)
    src = [torch.zeros(4), torch.ones(2, 2)]
    m = X.MlpCall(ctypes.addressof(marr), ctypes.sizeof(marr), 1, 64, 1 << 20, mfn, 0, src, 5, 7)
    assert m.current(5, 7) and not m.current(6, 7) and not m.current(5, 8)
    src[1].mul_(2.0)
    assert not m.current(5, 7)
    assert m.run([torch.zeros(3, 64)], [torch.zeros(3, 64)]) is None          # CPU tensors: not what the launch takes


def test_struct_epoch_moves_when_a_module_tree_changes():
    import torch
    from cwn_amd import ops
    e0 = ops.STRUCT_EPOCH
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.ReLU())
    e1 = ops.STRUCT_EPOCH
    assert e1 > e0
    net[0].weight.data.mul_(2.0)
    assert ops.STRUCT_EPOCH == e1                 # values are the version counters' business
    net[0] = torch.nn.Linear(4, 4)
    e2 = ops.STRUCT_EPOCH
    assert e2 > e1
    net[0].bias = torch.nn.Parameter(torch.zeros(4))
    assert ops.STRUCT_EPOCH > e2
    e3 = ops.STRUCT_EPOCH
    net.register_buffer('k', torch.zeros(1))
    assert ops.STRUCT_EPOCH > e3


def test_folded_batchnorm_follows_a_training_mode_forward_of_the_torch_module():
    """layers._fold_norm caches the eval-mode affine of a BatchNorm on (address, version counter) of its tensors.  torch's native
    batch_norm writes the running statistics of a training-mode forward WITHOUT moving their counters -- only the module's own
    num_batches_tracked.add_(1) moves one -- so that buffer is part of the key (round 5: tools/fuzz_round5.py found an eval
    forward folding the statistics from before a training-mode forward)."""
    import torch
    from cwn_amd.layers import _fold_norm
    bn = torch.nn.BatchNorm1d(8).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0.0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    sc1, sh1 = _fold_norm(bn, 8)
    assert _fold_norm(bn, 8)[0] is sc1                                    # cached
    v0 = bn.running_mean._version
    bn.train()
    with torch.no_grad():
        bn(torch.randn(32, 8) * 3 + 1)
    bn.eval()
    assert bn.running_mean._version == v0 or True                         # (whatever torch does with the counter ...)
    sc2, sh2 = _fold_norm(bn, 8)                                          # ... the fold follows the statistics
    want = torch.rsqrt(bn.running_var + bn.eps) * bn.weight
    assert torch.allclose(sc2, want) and torch.allclose(sh2, bn.bias - bn.running_mean * want)
    assert not torch.allclose(sc1, sc2)
    assert _fold_norm(bn.train(), 8) is None                              # batch statistics are not a fixed affine


def test_round5_host_helpers_head_split_padded_first_weight_and_narrow_shapes():
    """Pure host logic added in round 5: the head's pooling split (bounded, 1 for molecules), the zero-padded first weight of an
    update network over narrow inputs (cached per weight version), and which Linear shapes the one-launch update form takes."""
    import torch
    from cwn_amd import ops
    # pooling split: molecules never split; REDDIT-like complexes of thousands of cells: one 128-row chunk per workgroup at most
    assert ops.head_split(6800, 128) == 1 and ops.head_split(28000, 512) == 1 and ops.head_split(0, 0) == 1
    p = ops.head_split(73000, 32)
    assert 8 <= p <= 32 and p <= 73000 / 32 / 128 + 1
    assert ops.head_split(10 ** 7, 2) == 32
    # the padded first weight
    w = torch.nn.Parameter(torch.randn(64, 3))
    pad = ops._mlp_first_weight(w, 64)
    assert pad.shape == (64, 64) and torch.equal(pad[:, :3], w.detach()) and float(pad[:, 3:].abs().max()) == 0.0
    assert ops._mlp_first_weight(w, 64) is pad                              # cached
    with torch.no_grad():
        w.mul_(2.0)
    pad2 = ops._mlp_first_weight(w, 64)
    assert pad2 is not pad and torch.equal(pad2[:, :3], w.detach())         # ... per version
    full = torch.nn.Parameter(torch.randn(64, 64))
    assert ops._mlp_first_weight(full, 64) is full
    # shapes of the one-launch update form
    def dim(w_in, F, rows=5, bad=None):
        lin = lambda i, o: torch.nn.Linear(i, o)
        lins = [lin(w_in, F), lin(F, F), lin(w_in, F), lin(F, F), lin(2 * F, F)]
        if bad is not None:
            lins[bad] = lin(F + 1, F)
        return ops.MlpDim(x_up=torch.zeros(rows, w_in), x_b=torch.zeros(rows, w_in), linears=lins, folds=[(None, None)] * 5)
    assert ops.update_mlp_applies([dim(64, 64)]) and ops.update_mlp_applies([dim(1, 64)]) and ops.update_mlp_applies([dim(20, 128), dim(20, 128)])
    assert not ops.update_mlp_applies([dim(64, 96)])                        # widths other than 64 / 128
    assert not ops.update_mlp_applies([dim(130, 128)])                      # wider than the networks
    assert not ops.update_mlp_applies([dim(64, 64, bad=1)]) and not ops.update_mlp_applies([dim(64, 64, bad=4)])
    assert ops.mlp_width([dim(1, 128)]) == 128


def test_deferred_checks_read_the_error_word_once_and_not_after_an_exception(monkeypatch):
    """csr.deferred_checks: inside the block check_errors only notes the device; the block's end reads the word once per device;
    a block left by an exception reads nothing; nesting reads at the outermost end."""
    import torch
    from cwn_amd import csr
    reads = []

    class Word:
        def __init__(self, v):
            self.v = v

        def item(self):
            reads.append(self.v)
            return self.v

        def zero_(self):
            self.v = 0
    word = Word(0)
    monkeypatch.setattr(csr, '_err_flag', lambda dev: word)
    dev = torch.device('cpu')
    csr.check_errors(dev)
    assert reads == [0]
    with csr.deferred_checks():
        csr.check_errors(dev)
        with csr.deferred_checks():
            csr.check_errors(dev)
        assert reads == [0]                       # nothing read inside, nor at the inner end
    assert reads == [0, 0]                        # once at the outermost end
    with pytest.raises(RuntimeError):
        with csr.deferred_checks():
            csr.check_errors(dev)
            raise RuntimeError('the forward failed')
    assert reads == [0, 0]                        # a block left by an exception reads nothing ...
    csr.check_errors(dev)
    assert reads == [0, 0, 0]                     # ... and leaves nothing pending
    word.v = 2
    e0 = csr.ERROR_EPOCH
    with pytest.raises(IndexError, match='source index'):
        with csr.deferred_checks():
            csr.check_errors(dev)
    assert word.v == 0 and csr.ERROR_EPOCH == e0 + 1


def test_by_module_mapping_forgets_dead_modules_and_never_answers_for_a_recycled_id():
    """layers.ByModule (the prepared launches' caches: a plain dict keyed by id(module) with a weak reference per entry)."""
    import gc
    import torch
    from cwn_amd.layers import ByModule
    c = ByModule()
    m1, m2 = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    assert c.get(m1) is None and m1 not in c and len(c) == 0
    c[m1] = 'a'
    assert c[m1] == 'a' and m1 in c and m2 not in c and c.get(m2, 7) == 7
    d = c.setdefault(m2, {})
    d['k'] = 1
    assert c.setdefault(m2, {}) is d
    assert c.pop(m1) == 'a' and c.pop(m1, None) is None
    with pytest.raises(KeyError):
        c.pop(m1)
    with pytest.raises(KeyError):
        c[m1]
    c[m1] = 'b'
    c[m1] = 'c'                      # (the replaced entry's weak reference must not take the new entry with it)
    gc.collect()
    assert c[m1] == 'c'
    del c[m1]
    del m2
    gc.collect()
    assert len(c) == 0
    import copy
    net = torch.nn.Sequential(torch.nn.Linear(2, 2))
    c[net] = 'x'
    assert copy.deepcopy(net) not in c           # a copy is another module: it has no prepared launches
