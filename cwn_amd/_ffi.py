"""ctypes binding of libcwn_hip.so (C ABI in include/cwn_hip.h).

PyTorch is only the plumbing here: tensors give device pointers (`data_ptr()`), the caching
allocator gives buffers, `torch.cuda.current_stream()` gives the hipStream_t.  No torch type
crosses the boundary.  Loading fails loudly: there is NO CPU fallback for the product path.
"""
import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CWN_HIP_LIB') or os.path.join(_HERE, 'libcwn_hip.so')   # override: A/B experiments

MAX_DESCS = 8
CSR_MAX_DESCS = 16             # = CWN_CSR_MAX_DESCS
MSG_A, MSG_A_PLUS_B, MSG_A_TIMES_B, MSG_RELU_A_PLUS_B, MSG_A_MASK_RELU = range(5)
REDUCE = {'add': 0, 'sum': 0, 'mean': 1, 'max': 2}
ABI_VERSION = 24

EXPORTS = ('cwn_abi_version', 'cwn_error_string', 'cwn_target_arch', 'cwn_csr_workspace_bytes',
           'cwn_csr_build', 'cwn_csr_long_rows', 'cwn_gather_rows_f32', 'cwn_aggregate_f32', 'cwn_gemm_f32', 'cwn_gemm_would_split', 'cwn_gemm_packed_weight_bytes', 'cwn_gemm_pack_weights_f32', 'cwn_update_mlp_f32', 'cwn_update_mlp3_f32', 'cwn_update_mlp_max_rows', 'cwn_update_mlp_packed_weight_bytes', 'cwn_update_mlp_pack_weights_f32', 'cwn_update_mlp_pack_weights_many_f32', 'cwn_update_mlp_pack_weights_t_many_f32', 'cwn_update_mlp_pack_weights_both_many_f32', 'cwn_layer_pack_weights_both_many_f32', 'cwn_dense_stage_f32', 'cwn_dense_stage_ex_f32', 'cwn_dense_stage_bwd_f32', 'cwn_layer_fused_f32', 'cwn_layer_fused_lds_bytes', 'cwn_layer_variant_lds_bytes', 'cwn_layer_round_rows', 'cwn_layer_variant_round_rows', 'cwn_layer_items_check', 'cwn_layer_items_build', 'cwn_layer_pack_weights_f32', 'cwn_layer_pack_weights_many_f32', 'cwn_layer_pack_weights_t_many_f32', 'cwn_layer_bwd_f32', 'cwn_layer_bwd_lds_bytes', 'cwn_layer_bwd_items_build', 'cwn_layer_bwd_own_f32', 'cwn_layer_packed_weight_bytes', 'cwn_collate', 'cwn_collate_slots', 'cwn_collate_tables', 'cwn_collate_tables_len', 'cwn_collate_guard', 'cwn_layer_items_build_dev', 'cwn_layer_bwd_items_build_dev',
           'cwn_bn_finalize_f32', 'cwn_step_begin', 'cwn_axpy_eps_f32', 'cwn_dropout_f32', 'cwn_embed_front_bwd_f32', 'cwn_norm_act_f32', 'cwn_norm_bwd_reduce_f32', 'cwn_norm_bwd_apply_f32', 'cwn_norm_bwd_f32',
           'cwn_gemm_tn_f32', 'cwn_gemm_tn_workspace_bytes', 'cwn_adam_f32', 'cwn_loss_f32', 'cwn_loss_cols_f32', 'cwn_embedding_fwd_f32', 'cwn_embedding_bwd_f32', 'cwn_embed_front_f32', 'cwn_head_f32', 'cwn_head_bwd_f32', 'cwn_head_pool_floats', 'cwn_lift_create', 'cwn_lift_size', 'cwn_lift_copy', 'cwn_lift_destroy',
           'cwn_lift_many', 'cwn_lift_many_count', 'cwn_lift_many_lengths', 'cwn_lift_many_copy', 'cwn_lift_many_destroy')


class CsrDesc(C.Structure):
    _fields_ = [('key', C.c_void_p), ('val', C.c_void_p), ('aux', C.c_void_p),
                ('n_entries', C.c_int64), ('n_dst', C.c_int64), ('n_val', C.c_int64),
                ('n_aux', C.c_int64), ('rowptr', C.c_void_p), ('col', C.c_void_p),
                ('perm', C.c_void_p), ('aux_out', C.c_void_p),
                ('long_rows', C.c_void_p), ('n_long', C.c_void_p), ('e_dev', C.c_void_p)]


class AggDesc(C.Structure):
    _fields_ = [('rowptr', C.c_void_p), ('ia', C.c_void_p), ('ib', C.c_void_p),
                ('A', C.c_void_p), ('B', C.c_void_p), ('self_x', C.c_void_p),
                ('eps', C.c_void_p), ('self_pre', C.c_void_p), ('out', C.c_void_p),
                ('long_rows', C.c_void_p), ('n_long', C.c_void_p),
                ('n_dst', C.c_int64), ('F', C.c_int32), ('b_width', C.c_int32),
                ('msg_op', C.c_int32), ('reduce', C.c_int32),
                ('long_cap', C.c_int32), ('flags', C.c_int32),
                ('self_x2', C.c_void_p), ('eps2', C.c_void_p), ('m_dev', C.c_void_p)]


class LongRowsDesc(C.Structure):
    """cwn_long_rows_desc (include/cwn_hip.h)."""
    _fields_ = [('rowptr', C.c_void_p), ('n_rows', C.c_int64), ('m_dev', C.c_void_p), ('long_rows', C.c_void_p),
                ('n_long', C.c_void_p), ('long_cap', C.c_int64)]


AXPY_MAX_DESCS = 8             # CWN_AXPY_MAX_DESCS


class AxpyDesc(C.Structure):
    """cwn_axpy_desc (include/cwn_hip.h)."""
    _fields_ = [('y', C.c_void_p), ('x', C.c_void_p), ('eps', C.c_void_p), ('n', C.c_int64)]


class GemmBnb(C.Structure):
    """cwn_gemm_bnb (include/cwn_hip.h)."""
    _fields_ = [('z', C.c_void_p), ('dz', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p), ('mean', C.c_void_p),
                ('rstd', C.c_void_p), ('s1', C.c_void_p), ('s2', C.c_void_p), ('acc1', C.c_void_p), ('acc2', C.c_void_p),
                ('ldz', C.c_int64), ('lddz', C.c_int64), ('relu', C.c_int32), ('pad_', C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [('X', C.c_void_p), ('X2', C.c_void_p), ('W', C.c_void_p), ('bias', C.c_void_p),
                ('in_scale', C.c_void_p), ('in_shift', C.c_void_p), ('in_scale2', C.c_void_p),
                ('in_shift2', C.c_void_p), ('out_scale', C.c_void_p),
                ('out_shift', C.c_void_p), ('col_sum', C.c_void_p), ('col_sumsq', C.c_void_p),
                ('Y', C.c_void_p), ('M', C.c_int64), ('ldx', C.c_int64), ('ldx2', C.c_int64),
                ('ldw', C.c_int64), ('ldy', C.c_int64), ('N', C.c_int32), ('K', C.c_int32),
                ('K2', C.c_int32), ('relu', C.c_int32), ('in_relu', C.c_int32),
                ('w_trans', C.c_int32), ('flags', C.c_int32), ('pad_', C.c_int32), ('bnb', C.POINTER(GemmBnb)),
                ('m_dev', C.c_void_p)]


GEMM_EXACT = 1
GEMM_W_PACKED = 2         # = CWN_GEMM_W_PACKED
GEMM_ADD_OUT = 4          # = CWN_GEMM_ADD_OUT            # = CWN_GEMM_EXACT (cwn_gemm_desc.flags)


class LayerDim(C.Structure):
    """cwn_layer_dim (include/cwn_hip.h)."""
    _fields_ = [('x', C.c_void_p), ('up_index', C.c_void_p), ('up_shared', C.c_void_p),
                ('b_index', C.c_void_p), ('msg_w_packed', C.c_void_p), ('msg_bias', C.c_void_p),
                ('eps1', C.c_void_p), ('eps2', C.c_void_p), ('out_up', C.c_void_p),
                ('out_b', C.c_void_p), ('n_cells', C.c_int64), ('e_up', C.c_int64), ('n_b', C.c_int64),
                ('big_up_rowptr', C.c_void_p), ('big_up_col', C.c_void_p), ('big_up_aux', C.c_void_p),
                ('big_b_rowptr', C.c_void_p), ('big_b_col', C.c_void_p), ('big_y1', C.c_void_p), ('big_y2', C.c_void_p),
                ('out_down', C.c_void_p), ('eps3', C.c_void_p)]


class BnBwdLive(C.Structure):
    """cwn_bn_bwd_live (include/cwn_hip.h)."""
    _fields_ = [('z', C.c_void_p), ('aff', C.c_void_p), ('slots', C.c_void_p), ('ldz', C.c_int64)]


class LayerBwdDim(C.Structure):
    """cwn_layer_bwd_dim (include/cwn_hip.h)."""
    _fields_ = [('g_up', C.c_void_p), ('g_b', C.c_void_p), ('y1', C.c_void_p), ('y2', C.c_void_p), ('up_index', C.c_void_p),
                ('up_shared', C.c_void_p), ('b_index', C.c_void_p), ('wt_packed', C.c_void_p), ('eps1', C.c_void_p),
                ('eps2', C.c_void_p), ('dx', C.c_void_p), ('gy1', C.c_void_p), ('gy2', C.c_void_p),
                ('n_cells', C.c_int64), ('e_up', C.c_int64), ('n_b', C.c_int64), ('out_bn', BnBwdLive)]


class LayerPlan(C.Structure):
    """cwn_layer_plan (include/cwn_hip.h)."""
    _fields_ = [('items', C.c_void_p), ('csr_cache', C.c_void_p), ('n_items', C.c_int64),
                ('set_start', C.c_int32 * 4), ('max_gemm_rows', C.c_int32), ('max_source_rows', C.c_int32),
                ('variant', C.c_int32), ('cells_end', C.c_int64 * 3), ('up_end', C.c_int64 * 3),
                ('b_end', C.c_int64 * 3), ('n_big', C.c_int64), ('lds_bytes', C.c_int64)]


class LayerSizes(C.Structure):
    """cwn_layer_sizes (include/cwn_hip.h)."""
    _fields_ = [('n_complexes', C.c_int64), ('n_dims', C.c_int32), ('has_up', C.c_int32 * 3),
                ('allow_big', C.c_int32), ('pad_', C.c_int32),
                ('cell_ptr', C.c_void_p * 3), ('up_ptr', C.c_void_p * 3), ('b_ptr', C.c_void_p * 3),
                ('skip', C.c_void_p), ('unfit', C.c_void_p)]


class LayerSizesDev(C.Structure):
    """cwn_layer_sizes_dev (include/cwn_hip.h): the prefix sums of a batch as DEVICE arrays."""
    _fields_ = [('n_complexes', C.c_void_p), ('cap_complexes', C.c_int64), ('n_dims', C.c_int32), ('has_up', C.c_int32 * 3),
                ('cell_ptr', C.c_void_p * 3), ('up_ptr', C.c_void_p * 3), ('b_ptr', C.c_void_p * 3),
                ('n_slots', C.c_int32), ('pad_', C.c_int32), ('table_slot_stride', C.c_int64)]


ERR_BIT_UNFIT = 16        # = CWN_ERR_BIT_UNFIT
ERR_BIT_CAPACITY = 32     # = CWN_ERR_BIT_CAPACITY
LAYER_ITEMS_TOO_LARGE, LAYER_ITEMS_BAD_ARG = -1, -2
LAYER_BWD_ITEM_INTS = 16       # = CWN_LAYER_BWD_ITEM_INTS


class LayerBwdPlan(C.Structure):
    """cwn_layer_bwd_plan (include/cwn_hip.h): the item table of the owner form of the backward launch."""
    _fields_ = [('items', C.c_void_p), ('n_items', C.c_int64), ('lds_bytes', C.c_int64),
                ('cells_end', C.c_int64 * 3), ('up_end', C.c_int64 * 3), ('b_end', C.c_int64 * 3)]


class MlpDim(C.Structure):
    """cwn_mlp_dim (include/cwn_hip.h)."""
    _fields_ = [('x_up', C.c_void_p), ('x_b', C.c_void_p), ('w_packed', C.c_void_p * 6),
                ('bias', C.c_void_p * 5), ('scale', C.c_void_p * 5), ('shift', C.c_void_p * 5),
                ('y', C.c_void_p), ('M', C.c_int64), ('ldx_up', C.c_int64), ('ldx_b', C.c_int64), ('ldy', C.c_int64),
                ('m_dev', C.c_void_p), ('in_width', C.c_int32), ('pad_', C.c_int32)]


class Mlp3Dim(C.Structure):
    """cwn_mlp3_dim (include/cwn_hip.h): the three update networks + the 3F-wide combine of a CIN++ layer."""
    _fields_ = [('x', C.c_void_p * 3), ('ldx', C.c_int64 * 3), ('w_packed', C.c_void_p * 9),
                ('bias', C.c_void_p * 7), ('scale', C.c_void_p * 7), ('shift', C.c_void_p * 7),
                ('y', C.c_void_p), ('M', C.c_int64), ('ldy', C.c_int64), ('m_dev', C.c_void_p)]


BN_SLOTS = 4                   # = CWN_BN_SLOTS


class BnLive(C.Structure):
    """cwn_bn_live (include/cwn_hip.h): a BatchNorm1d(train) whose statistics are summed by the producing launch and turned
    into the affine by the consuming one."""
    _fields_ = [('slots', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('running_mean', C.c_void_p),
                ('running_var', C.c_void_p), ('num_batches_tracked', C.c_void_p), ('aff', C.c_void_p),
                ('eps', C.c_float), ('momentum', C.c_float)]


class StageDesc(C.Structure):
    """cwn_stage_desc (include/cwn_hip.h)."""
    _fields_ = [('X', C.c_void_p), ('X2', C.c_void_p), ('w_packed', C.c_void_p), ('w2_packed', C.c_void_p),
                ('bias', C.c_void_p), ('in_scale', C.c_void_p), ('in_shift', C.c_void_p), ('in_scale2', C.c_void_p),
                ('in_shift2', C.c_void_p), ('Y', C.c_void_p), ('col_sum', C.c_void_p), ('col_sumsq', C.c_void_p),
                ('M', C.c_int64), ('ldx', C.c_int64), ('ldx2', C.c_int64), ('ldy', C.c_int64),
                ('in_relu', C.c_int32), ('pad_', C.c_int32), ('m_dev', C.c_void_p), ('stat_slots', C.c_void_p),
                ('in_bn', BnLive), ('in_bn2', BnLive)]


class StageExtra(C.Structure):
    """cwn_stage_extra (include/cwn_hip.h): a third / fourth K-block of a cwn_dense_stage_ex_f32 product."""
    _fields_ = [('X', C.c_void_p), ('w_packed', C.c_void_p), ('ldx', C.c_int64), ('relu', C.c_int32), ('pad_', C.c_int32),
                ('bn', BnLive)]


class FrontBwd(C.Structure):
    """cwn_front_bwd (include/cwn_hip.h)."""
    _fields_ = [('g0', C.c_void_p), ('g1', C.c_void_p), ('g2', C.c_void_p), ('rowptr1', C.c_void_p), ('col1', C.c_void_p),
                ('rowptr2', C.c_void_p), ('col2', C.c_void_p), ('v_src', C.c_void_p), ('e_src', C.c_void_p), ('dWv', C.c_void_p),
                ('dWe', C.c_void_p), ('n0', C.c_int64), ('n1', C.c_int64), ('n0_dev', C.c_void_p), ('n1_dev', C.c_void_p),
                ('H', C.c_int32), ('Vv', C.c_int32), ('Ve', C.c_int32), ('src_f32', C.c_int32), ('halve', C.c_int32),
                ('pad_', C.c_int32)]


class StageBwdDesc(C.Structure):
    """cwn_stage_bwd_desc (include/cwn_hip.h)."""
    _fields_ = [('dy', C.c_void_p), ('z', C.c_void_p), ('dz', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('mean', C.c_void_p), ('rstd', C.c_void_p), ('s1', C.c_void_p), ('s2', C.c_void_p), ('acc1', C.c_void_p),
                ('acc2', C.c_void_p), ('wt_packed', C.c_void_p), ('wt2_packed', C.c_void_p), ('dx', C.c_void_p), ('dx2', C.c_void_p),
                ('M', C.c_int64), ('lddy', C.c_int64), ('ldz', C.c_int64), ('lddz', C.c_int64), ('lddx', C.c_int64),
                ('lddx2', C.c_int64), ('relu', C.c_int32), ('pad_', C.c_int32), ('m_dev', C.c_void_p), ('s_slots', C.c_void_p),
                ('out_bn', BnBwdLive), ('out_bn2', BnBwdLive)]


STAGE_PACK_MAX = 160           # = CWN_STAGE_PACK_MAX


class EmbedTable(C.Structure):
    """cwn_embed_table (include/cwn_hip.h)."""
    _fields_ = [('W', C.c_void_p), ('src', C.c_void_p), ('col_off', C.c_void_p), ('col_size', C.c_void_p),
                ('V', C.c_int64), ('cols', C.c_int32), ('src_is_f32', C.c_int32)]


class HeadDim(C.Structure):
    """cwn_head_dim (include/cwn_hip.h)."""
    _fields_ = [('x', C.c_void_p), ('cell_ptr', C.c_void_p), ('w1t', C.c_void_p), ('b1', C.c_void_p),
                ('pooled_out', C.c_void_p), ('h_out', C.c_void_p), ('n_cells', C.c_int64), ('ldx', C.c_int64),
                ('x_more', C.c_void_p * 7), ('n_parts', C.c_int32), ('pad_', C.c_int32)]


class HeadBwdDim(C.Structure):
    """cwn_head_bwd_dim (include/cwn_hip.h)."""
    _fields_ = [('h', C.c_void_p), ('w1', C.c_void_p), ('cell_ptr', C.c_void_p), ('dx', C.c_void_p),
                ('dh_out', C.c_void_p), ('n_cells', C.c_int64), ('lddx', C.c_int64),
                ('dx_more', C.c_void_p * 7), ('n_parts', C.c_int32), ('pad_', C.c_int32)]


ERR_BIT_BLOCK = 8         # = CWN_ERR_BIT_BLOCK
LAYER_CSR_STORE, LAYER_CSR_LOAD = 1, 2
LAYER_STORE_Y = 4          # = CWN_LAYER_STORE_Y
LAYER_CSR_SLOT_BYTES = 5264


class CollateDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('dst_start', C.c_void_p),
                ('src_start', C.c_void_p), ('add', C.c_void_p), ('src_row_stride', C.c_int64),
                ('dst_row_stride', C.c_int64), ('n_rows', C.c_int32), ('op', C.c_int32)]


class BnDesc(C.Structure):
    _fields_ = [('col_sum', C.c_void_p), ('col_sumsq', C.c_void_p), ('gamma', C.c_void_p),
                ('beta', C.c_void_p), ('running_mean', C.c_void_p), ('running_var', C.c_void_p),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p),
                ('M', C.c_int64), ('N', C.c_int32), ('eps', C.c_float), ('momentum', C.c_float),
                ('pad_', C.c_int32), ('num_batches_tracked', C.c_void_p), ('bwd_sums', C.c_void_p), ('m_dev', C.c_void_p)]


class Dropout(C.Structure):
    """cwn_dropout: state = device int64 [2] {seed, step} (NULL: off), p, site."""
    _fields_ = [('state', C.c_void_p), ('p', C.c_float), ('site', C.c_uint32)]


HEAD_DROP_NONE, HEAD_DROP_LIN1, HEAD_DROP_FINAL, HEAD_DROP_LIN2 = range(4)      # = CWN_HEAD_DROP_*
HEAD_MAX_PARTS = 8         # = CWN_HEAD_MAX_PARTS


class NormDesc(C.Structure):
    _fields_ = [('dy', C.c_void_p), ('z', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('mean', C.c_void_p), ('rstd', C.c_void_p), ('s1', C.c_void_p), ('s2', C.c_void_p),
                ('out', C.c_void_p), ('M', C.c_int64), ('lddy', C.c_int64), ('ldz', C.c_int64),
                ('ldout', C.c_int64), ('N', C.c_int32), ('relu', C.c_int32), ('acc1', C.c_void_p), ('acc2', C.c_void_p),
                ('m_dev', C.c_void_p), ('bn', BnLive), ('drop', Dropout), ('dy_out', C.c_void_p), ('lddy_out', C.c_int64)]


class GemmTnDesc(C.Structure):
    _fields_ = [('dZ', C.c_void_p), ('X', C.c_void_p), ('X2', C.c_void_p), ('in_scale', C.c_void_p),
                ('in_shift', C.c_void_p), ('in_scale2', C.c_void_p), ('in_shift2', C.c_void_p),
                ('dW', C.c_void_p), ('db', C.c_void_p), ('M', C.c_int64), ('lddz', C.c_int64),
                ('ldx', C.c_int64), ('ldx2', C.c_int64), ('lddw', C.c_int64), ('N', C.c_int32),
                ('K', C.c_int32), ('K2', C.c_int32), ('in_relu', C.c_int32), ('m_dev', C.c_void_p)]


MAX_NORM_DESCS = 16
COLLATE_COPY32, COLLATE_COPY64, COLLATE_ADD64, COLLATE_SEGID64, COLLATE_ADD32 = range(5)
MAX_COLLATE_DESCS = 32


class CwnError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library.  Raises if it has not been built (run `python -c "import
    __graft_entry__ as g; g.build()"` or `make -C cwn_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CwnError(f'{LIB_PATH} is missing: the HIP extension has not been built. '
                       'There is no CPU fallback; build it with `make -C cwn_amd/csrc`.')
    L = C.CDLL(LIB_PATH)
    L.cwn_abi_version.restype = C.c_int
    L.cwn_error_string.restype = C.c_char_p
    L.cwn_error_string.argtypes = [C.c_int]
    L.cwn_target_arch.restype = C.c_char_p
    L.cwn_csr_workspace_bytes.restype = C.c_size_t
    L.cwn_csr_workspace_bytes.argtypes = [C.POINTER(CsrDesc), C.c_int]
    L.cwn_csr_long_rows.restype = C.c_int
    L.cwn_csr_long_rows.argtypes = [C.POINTER(LongRowsDesc), C.c_int, C.c_void_p]
    L.cwn_csr_build.restype = C.c_int
    L.cwn_csr_build.argtypes = [C.POINTER(CsrDesc), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                C.c_void_p]
    L.cwn_gather_rows_f32.restype = C.c_int
    L.cwn_gather_rows_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p]
    L.cwn_aggregate_f32.restype = C.c_int
    L.cwn_aggregate_f32.argtypes = [C.POINTER(AggDesc), C.c_int, C.c_void_p]
    L.cwn_gemm_f32.restype = C.c_int
    L.cwn_gemm_f32.argtypes = [C.POINTER(GemmDesc), C.c_int, C.c_void_p]
    L.cwn_layer_fused_f32.restype = C.c_int
    L.cwn_layer_fused_f32.argtypes = [C.POINTER(LayerDim), C.c_int, C.c_int32, C.POINTER(LayerPlan), C.c_int32,
                                      C.c_void_p, C.c_void_p]
    for name in ('cwn_layer_pack_weights_many_f32', 'cwn_layer_pack_weights_t_many_f32'):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.cwn_layer_bwd_f32.restype = C.c_int
    L.cwn_layer_bwd_f32.argtypes = [C.POINTER(LayerBwdDim), C.c_int, C.c_int32, C.POINTER(LayerPlan), C.c_void_p, C.c_void_p]
    L.cwn_layer_bwd_items_build.restype = C.c_int64
    L.cwn_layer_bwd_items_build.argtypes = [C.POINTER(LayerSizes), C.c_int32, C.c_void_p, C.c_int64, C.POINTER(LayerBwdPlan)]
    L.cwn_layer_bwd_own_f32.restype = C.c_int
    L.cwn_layer_bwd_own_f32.argtypes = [C.POINTER(LayerBwdDim), C.c_int, C.c_int32, C.POINTER(LayerBwdPlan), C.c_void_p, C.c_void_p]
    L.cwn_layer_bwd_lds_bytes.restype = C.c_size_t
    L.cwn_layer_bwd_lds_bytes.argtypes = [C.c_int32, C.c_int32]
    L.cwn_layer_packed_weight_bytes.restype = C.c_size_t
    L.cwn_layer_packed_weight_bytes.argtypes = [C.c_int32]
    L.cwn_layer_pack_weights_f32.restype = C.c_int
    L.cwn_layer_pack_weights_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.cwn_layer_fused_lds_bytes.restype = C.c_size_t
    L.cwn_layer_fused_lds_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.cwn_update_mlp_f32.argtypes = [C.POINTER(MlpDim), C.c_int, C.c_int32, C.c_void_p]
    L.cwn_update_mlp3_f32.argtypes = [C.POINTER(Mlp3Dim), C.c_int, C.c_int32, C.c_void_p]
    L.cwn_update_mlp_packed_weight_bytes.restype = C.c_size_t
    L.cwn_update_mlp_packed_weight_bytes.argtypes = [C.c_int32]
    L.cwn_update_mlp_pack_weights_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.cwn_update_mlp_pack_weights_many_f32.restype = C.c_int
    L.cwn_update_mlp_pack_weights_many_f32.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_void_p),
                                                       C.c_int32, C.c_void_p]
    L.cwn_update_mlp_pack_weights_t_many_f32.restype = C.c_int
    L.cwn_update_mlp_pack_weights_t_many_f32.argtypes = L.cwn_update_mlp_pack_weights_many_f32.argtypes
    for fn in (L.cwn_update_mlp_pack_weights_both_many_f32, L.cwn_layer_pack_weights_both_many_f32):
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                       C.c_int32, C.c_void_p]
    L.cwn_dense_stage_bwd_f32.restype = C.c_int
    L.cwn_dense_stage_bwd_f32.argtypes = [C.POINTER(StageBwdDesc), C.c_int, C.c_int32, C.c_void_p]
    L.cwn_dense_stage_f32.restype = C.c_int
    L.cwn_dense_stage_f32.argtypes = [C.POINTER(StageDesc), C.c_int, C.c_int32, C.c_void_p]
    L.cwn_dense_stage_ex_f32.restype = C.c_int
    L.cwn_dense_stage_ex_f32.argtypes = [C.POINTER(StageDesc), C.POINTER(StageExtra), C.c_int, C.c_int32, C.c_void_p]
    L.cwn_update_mlp_max_rows.restype = C.c_int64
    L.cwn_update_mlp_max_rows.argtypes = []
    L.cwn_gemm_packed_weight_bytes.restype = C.c_size_t
    L.cwn_gemm_packed_weight_bytes.argtypes = []
    L.cwn_gemm_pack_weights_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.cwn_layer_round_rows.restype = C.c_int32
    L.cwn_layer_round_rows.argtypes = [C.c_int32]
    L.cwn_layer_variant_round_rows.restype = C.c_int32
    L.cwn_layer_variant_round_rows.argtypes = [C.c_int32, C.c_int32]
    L.cwn_layer_variant_lds_bytes.restype = C.c_size_t
    L.cwn_layer_variant_lds_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.cwn_layer_items_build.restype = C.c_int64
    L.cwn_layer_items_build.argtypes = [C.POINTER(LayerSizes), C.c_int32, C.c_void_p, C.c_int64, C.POINTER(LayerPlan)]
    L.cwn_layer_items_check.restype = C.c_int
    L.cwn_layer_items_check.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(LayerPlan)]
    L.cwn_gemm_would_split.restype = C.c_int
    L.cwn_gemm_would_split.argtypes = [C.POINTER(GemmDesc), C.c_int]
    L.cwn_collate.restype = C.c_int
    L.cwn_collate.argtypes = [C.POINTER(CollateDesc), C.c_int, C.c_int64, C.c_void_p]
    L.cwn_collate_tables_len.restype = C.c_size_t
    L.cwn_collate_tables_len.argtypes = [C.c_int32, C.c_int32, C.c_int64]
    L.cwn_collate_tables.restype = C.c_int
    L.cwn_collate_tables.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                     C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cwn_collate_guard.restype = C.c_int
    L.cwn_collate_guard.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cwn_collate_slots.restype = C.c_int
    L.cwn_collate_slots.argtypes = [C.POINTER(CollateDesc), C.c_int, C.c_int64, C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.c_void_p,
                                    C.c_void_p]
    L.cwn_layer_items_build_dev.restype = C.c_int
    L.cwn_layer_items_build_dev.argtypes = [C.POINTER(LayerSizesDev), C.c_int32, C.POINTER(LayerPlan), C.c_int32, C.c_void_p, C.c_void_p]
    L.cwn_layer_bwd_items_build_dev.restype = C.c_int
    L.cwn_layer_bwd_items_build_dev.argtypes = [C.POINTER(LayerSizesDev), C.c_int32, C.POINTER(LayerBwdPlan), C.c_int32, C.c_void_p,
                                                C.c_void_p]
    L.cwn_embed_front_bwd_f32.restype = C.c_int
    L.cwn_embed_front_bwd_f32.argtypes = [C.POINTER(FrontBwd), C.c_void_p]
    L.cwn_axpy_eps_f32.restype = C.c_int
    L.cwn_axpy_eps_f32.argtypes = [C.POINTER(AxpyDesc), C.c_int, C.c_void_p]
    L.cwn_step_begin.restype = C.c_int
    L.cwn_step_begin.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cwn_dropout_f32.restype = C.c_int
    L.cwn_dropout_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.POINTER(Dropout), C.c_void_p,
                                  C.c_void_p]
    L.cwn_bn_finalize_f32.restype = C.c_int
    L.cwn_bn_finalize_f32.argtypes = [C.POINTER(BnDesc), C.c_int, C.c_void_p]
    for name in ('cwn_norm_act_f32', 'cwn_norm_bwd_reduce_f32', 'cwn_norm_bwd_apply_f32'):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.POINTER(NormDesc), C.c_int, C.c_void_p]
    L.cwn_norm_bwd_f32.restype = C.c_int
    L.cwn_norm_bwd_f32.argtypes = [C.POINTER(NormDesc), C.c_int, C.c_int, C.c_void_p]
    L.cwn_gemm_tn_f32.restype = C.c_int
    L.cwn_gemm_tn_f32.argtypes = [C.POINTER(GemmTnDesc), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cwn_gemm_tn_workspace_bytes.restype = C.c_size_t
    L.cwn_gemm_tn_workspace_bytes.argtypes = [C.POINTER(GemmTnDesc), C.c_int]
    L.cwn_adam_f32.restype = C.c_int
    L.cwn_adam_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cwn_loss_f32.restype = C.c_int
    L.cwn_loss_f32.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cwn_loss_cols_f32.restype = C.c_int
    L.cwn_loss_cols_f32.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cwn_embedding_fwd_f32.restype = C.c_int
    L.cwn_embedding_fwd_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    L.cwn_embedding_bwd_f32.restype = C.c_int
    L.cwn_embedding_bwd_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.cwn_embed_front_f32.restype = C.c_int
    L.cwn_embed_front_f32.argtypes = [C.POINTER(EmbedTable), C.c_int64, C.c_void_p, C.POINTER(EmbedTable), C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cwn_head_f32.restype = C.c_int
    L.cwn_head_f32.argtypes = [C.POINTER(HeadDim), C.c_int, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(Dropout), C.c_int32, C.c_void_p,
                               C.c_int64, C.c_int32, C.c_void_p]
    L.cwn_head_pool_floats.restype = C.c_int64
    L.cwn_head_pool_floats.argtypes = [C.POINTER(HeadDim), C.c_int, C.c_int64, C.c_int32]
    L.cwn_head_bwd_f32.restype = C.c_int
    L.cwn_head_bwd_f32.argtypes = [C.POINTER(HeadBwdDim), C.c_int, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(Dropout), C.c_int32, C.c_int32, C.c_void_p]
    L.cwn_lift_create.restype = C.c_void_p
    L.cwn_lift_create.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int]
    L.cwn_lift_size.restype = C.c_int64
    L.cwn_lift_size.argtypes = [C.c_void_p, C.c_int]
    L.cwn_lift_copy.restype = C.c_int
    L.cwn_lift_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.cwn_lift_destroy.restype = None
    L.cwn_lift_destroy.argtypes = [C.c_void_p]
    L.cwn_lift_many.restype = C.c_void_p
    L.cwn_lift_many.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.cwn_lift_many_count.restype = C.c_int64
    L.cwn_lift_many_count.argtypes = [C.c_void_p]
    L.cwn_lift_many_lengths.restype = C.c_int
    L.cwn_lift_many_lengths.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.cwn_lift_many_copy.restype = C.c_int
    L.cwn_lift_many_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.cwn_lift_many_destroy.restype = None
    L.cwn_lift_many_destroy.argtypes = [C.c_void_p]
    if L.cwn_abi_version() != ABI_VERSION:
        raise CwnError(f'ABI mismatch: library {L.cwn_abi_version()} vs binding {ABI_VERSION}')
    _lib = L
    return L


def check(code: int, what: str):
    if code != 0:
        raise CwnError(f'{what}: {lib().cwn_error_string(code).decode()} (code {code})')


# ---- device-side row counts (include/cwn_hip.h, "Conventions") ------------------------------------------------------------
# A step captured ONCE over capacity-sized buffers serves batches of any shape (cwn_amd/static_graph.py): the tensors the
# model code passes around then have the CAPACITY as their row count, and the number of rows that exist lives in device
# memory.  DYN_ROWS maps a capacity to the address of that device int64; every launch wrapper below (and the few direct
# calls in ops.py / train.py) looks its descriptors' row counts up here and fills `m_dev`.  Empty outside
# `dynamic_rows(...)`: nothing changes for ordinary launches.  The capacities of a static batch are pairwise distinct
# (static_graph.StaticBatch sees to it), so a row count identifies its dimension.
DYN_ROWS = {}


class dynamic_rows:
    """Context manager: inside, a descriptor whose row count is a key of `mapping` gets m_dev = mapping[rows]."""

    def __init__(self, mapping):
        self.mapping = dict(mapping)

    def __enter__(self):
        self.prev = dict(DYN_ROWS)
        DYN_ROWS.update(self.mapping)
        return self

    def __exit__(self, *exc):
        DYN_ROWS.clear()
        DYN_ROWS.update(self.prev)
        return False


def dyn(rows) -> Optional[int]:
    """Device address of the actual row count behind the capacity `rows`, or None."""
    return DYN_ROWS.get(int(rows)) if DYN_ROWS else None


def _set_dyn(descs, field: str = 'M') -> None:
    if DYN_ROWS:
        for d in descs:
            d.m_dev = DYN_ROWS.get(int(getattr(d, field)))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(device=None) -> int:
    """The current HIP stream of `device` as an integer handle (the raw getter costs ~0.2 us, the Stream
    object ~2 us per call)."""
    if _raw_stream is not None and isinstance(device, torch.device) and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise CwnError(f'{name} is on {t.device}: cwn_amd runs on the GPU only (no CPU fallback). '
                       'Use oracle/ for CPU checks.')


def aggregate(descs: Sequence[AggDesc], device) -> None:
    """One kernel launch for up to MAX_DESCS descriptors; more are split into several calls."""
    L = lib()
    s = stream_ptr(device)
    _set_dyn(descs, 'n_dst')
    for i in range(0, len(descs), MAX_DESCS):
        chunk = descs[i:i + MAX_DESCS]
        arr = (AggDesc * len(chunk))(*chunk)
        check(L.cwn_aggregate_f32(arr, len(chunk), s), 'cwn_aggregate_f32')


def gemm(descs: Sequence[GemmDesc], device) -> None:
    """One kernel launch for up to MAX_DESCS GEMMs; more are split into several calls."""
    L = lib()
    s = stream_ptr(device)
    for i in range(0, len(descs), MAX_DESCS):
        chunk = descs[i:i + MAX_DESCS]
        arr = (GemmDesc * len(chunk))(*chunk)
        check(L.cwn_gemm_f32(arr, len(chunk), s), 'cwn_gemm_f32')


def gemm_would_split(descs: Sequence[GemmDesc]) -> bool:
    """Would cwn_gemm_f32 run this launch (<= MAX_DESCS descriptors) on the bf16-split kernel?"""
    arr = (GemmDesc * len(descs))(*descs)
    return bool(lib().cwn_gemm_would_split(arr, len(descs)))


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    out = torch.empty(idx.numel(), src.size(1), dtype=torch.float32, device=src.device)
    check(lib().cwn_gather_rows_f32(src.data_ptr(), src.size(0), src.size(1), idx.data_ptr(),
                                    idx.numel(), out.data_ptr(), stream_ptr(src.device)),
          'cwn_gather_rows_f32')
    return out


def _chunked(fn_name: str, struct, descs, device, limit: int) -> None:
    L = lib()
    s = stream_ptr(device)
    fn = getattr(L, fn_name)
    _set_dyn(descs)
    for i in range(0, len(descs), limit):
        chunk = descs[i:i + limit]
        arr = (struct * len(chunk))(*chunk)
        check(fn(arr, len(chunk), s), fn_name)


def bn_finalize(descs: Sequence[BnDesc], device) -> None:
    _chunked('cwn_bn_finalize_f32', BnDesc, descs, device, MAX_NORM_DESCS)


def norm_act(descs: Sequence[NormDesc], device) -> None:
    _chunked('cwn_norm_act_f32', NormDesc, descs, device, MAX_NORM_DESCS)


def norm_bwd_reduce(descs: Sequence[NormDesc], device) -> None:
    _chunked('cwn_norm_bwd_reduce_f32', NormDesc, descs, device, MAX_NORM_DESCS)


def norm_bwd_apply(descs: Sequence[NormDesc], device) -> None:
    _chunked('cwn_norm_bwd_apply_f32', NormDesc, descs, device, MAX_NORM_DESCS)


NORM_BWD_FUSED_MAX_ROWS = 4096     # = CWN_NORM_BWD_FUSED_MAX_ROWS


def norm_bwd(descs: Sequence[NormDesc], device, accumulate: bool) -> None:
    """Reduce + apply in one launch (matrices of at most NORM_BWD_FUSED_MAX_ROWS rows, 16-byte aligned)."""
    L = lib()
    s = stream_ptr(device)
    _set_dyn(descs)
    for i in range(0, len(descs), MAX_NORM_DESCS):
        chunk = descs[i:i + MAX_NORM_DESCS]
        arr = (NormDesc * len(chunk))(*chunk)
        check(L.cwn_norm_bwd_f32(arr, len(chunk), int(bool(accumulate)), s), 'cwn_norm_bwd_f32')


# False: the row bands of a weight gradient are added with fp32 atomics (fastest: 1.48 ms ZINC training
# step).  True: per-band partial tiles + a second launch that sums them in band order -- bit-reproducible
# weight gradients for 0.14 ms more per step (17 extra launches).
DETERMINISTIC_TN = os.environ.get('CWN_DETERMINISTIC_TN', '0') == '1'


MAX_TN_DESCS = 24          # = CWN_GEMM_TN_MAX_DESCS
# Deferred weight gradients.  Nothing reads a weight gradient before the optimizer step (or the gradient all-reduce of
# its chunk), so a caller that accumulates into buffers it owns (ops.accumulate_into_grad: cwn_amd.train.TrainStep) lets
# the launches queue up and runs them together -- `flush_tn` -- in launches of MAX_TN_DESCS: ~100 small GEMMs of a
# 4-layer model in 5 launches that fill the chip instead of 17 that each pay their own latency chain (0.31 of a
# 1.45 ms step).  The queue keeps the operand tensors alive (under stream capture a freed block would be handed out
# again before the deferred kernel has read it).
_tn_queue = None


def defer_tn(on: bool) -> None:
    """Start (True) or stop (False) queueing deferrable weight-gradient launches; stopping does NOT flush."""
    global _tn_queue
    if on and _tn_queue is None:
        _tn_queue = []
    elif not on:
        if _tn_queue:
            raise CwnError('defer_tn(False) with queued weight gradients: call flush_tn first')
        _tn_queue = None


# ... or, TN_SIDE_STREAM, launched where they arise but on a SIDE stream that forks off the caller's stream there and
# joins it again in flush_tn: the weight gradients leave the critical path of the backward (a chain of small
# latency-bound launches that leaves most of the chip idle) instead of waiting for its end -- under stream capture
# the fork / join become parallel branches of the graph.  OFF by default: measured on the graph-captured ZINC-128 step the
# forked graph replays SLOWER (1.17 -> 1.46 - 1.61 ms; every cross-branch edge of a hipGraph costs more than the overlap
# wins); kept as a switch for eager multi-stream runs (CWN_TN_SIDE=1).
TN_SIDE_STREAM = os.environ.get('CWN_TN_SIDE', '0') == '1'
# ... or (round 5, CWN_TN_SIDE=2): queued as above, but every time a whole launch's worth (MAX_TN_DESCS) has queued up it is
# issued on the side stream -- four forks and one join per ZINC step instead of one fork per stage.
TN_SIDE_CHUNKS = os.environ.get('CWN_TN_SIDE', '0') == '2'
_tn_side = {}
_tn_side_used = False


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _tn_side:
        _tn_side[key] = torch.cuda.Stream(device=key)
    return _tn_side[key]


def _side_flush(device) -> None:
    """The queued descriptors as launches on the side stream, forked off the current one here (TN_SIDE_CHUNKS)."""
    global _tn_queue, _tn_side_used
    pend = [q for q in _tn_queue if not q[3:]]
    descs = [d for q in pend for d in q[0]]
    main, side = torch.cuda.current_stream(device), _side_stream(device)
    side.wait_stream(main)
    _tn_side_used = True
    with torch.cuda.stream(side):
        _flush_descs(descs, device)
    # (the operands stay referenced until the join; the entries are marked as issued)
    _tn_queue = [q if q[3:] else ([], q[1], q[2], True) for q in _tn_queue]


def flush_tn(device=None) -> None:
    """Launch every queued weight gradient (queue order), MAX_TN_DESCS per launch, on the device (and its current stream)
    the operands live on -- recorded when they were queued (not the process' current device); join the side stream."""
    global _tn_queue, _tn_side_used
    if _tn_side_used:
        dev_ = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        torch.cuda.current_stream(dev_).wait_stream(_side_stream(dev_))
        _tn_side_used = False
    if not _tn_queue:
        return
    q, _tn_queue = _tn_queue, []
    by_dev = {}
    for ent in q:
        ds, dv = ent[0], ent[2]
        by_dev.setdefault(dv, []).extend(ds)
    for dv, descs in by_dev.items():
        _flush_descs(descs, dv if dv is not None else device)


def _flush_descs(descs, device) -> None:
    # one launch runs ONE form of the kernel: descriptors that need the element-wise loads (a width that is not a
    # multiple of 4: the head's lin2) go into launches of their own instead of slowing the 16-byte form of the rest
    def vec(d):
        ok = lambda p: p is None or p % 16 == 0
        return (ok(d.dZ) and ok(d.X) and ok(d.X2) and d.lddz % 4 == 0 and d.ldx % 4 == 0 and (d.K2 == 0 or d.ldx2 % 4 == 0)
                and d.N % 4 == 0 and d.K % 4 == 0 and d.K2 % 4 == 0)
    fast, slow = [d for d in descs if vec(d)], [d for d in descs if not vec(d)]
    # (which TILE a launch runs on -- 64 x 64 fp32 MFMA at hidden 64, 128 x 128 bf16-split above -- cwn_gemm_tn_f32 decides by
    #  the majority of its descriptors)
    for part in (fast, slow):
        if part:
            gemm_tn(part, device)               # (the operands stay referenced by `q` until here)


def gemm_tn(descs: Sequence[GemmTnDesc], device, keep=None, deferrable: bool = False) -> None:
    """dW += dZ^T [X | X2] for every descriptor.  `deferrable`: the targets are buffers the caller owns until
    flush_tn (never tensors handed back to autograd); `keep`: every tensor a descriptor points at."""
    if deferrable and _tn_queue is not None:
        if not TN_SIDE_STREAM:
            _tn_queue.append((list(descs), keep, torch.device(device)))
            if TN_SIDE_CHUNKS and sum(len(q[0]) for q in _tn_queue if not q[3:]) >= MAX_TN_DESCS:
                _side_flush(torch.device(device))
            return
        global _tn_side_used
        main, side = torch.cuda.current_stream(device), _side_stream(device)
        side.wait_stream(main)                  # everything the descriptors read has been issued on `main`
        _tn_queue.append(([], keep, torch.device(device)))            # the operands stay referenced until the join
        _tn_side_used = True
        with torch.cuda.stream(side):
            gemm_tn(descs, device)
        return
    L = lib()
    s = stream_ptr(device)
    _set_dyn(descs)
    for i in range(0, len(descs), MAX_TN_DESCS):
        chunk = descs[i:i + MAX_TN_DESCS]
        arr = (GemmTnDesc * len(chunk))(*chunk)
        ws, nbytes = None, 0
        if DETERMINISTIC_TN:
            nbytes = L.cwn_gemm_tn_workspace_bytes(arr, len(chunk))
            ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        check(L.cwn_gemm_tn_f32(arr, len(chunk), None if ws is None else ws.data_ptr(), nbytes, s),
              'cwn_gemm_tn_f32')


import itertools as _itertools
_no_version = _itertools.count(1)


def tver(t) -> int:
    """Version counter of a tensor, for the caches keyed on (tensor, version).  Inference tensors (created under
    torch.inference_mode()) keep none -- torch raises on `_version` -- and may still be written in place inside that mode:
    they get a value no earlier call returned, so every such cache MISSES and the uncached path runs (slower, never stale)."""
    try:
        return t._version
    except RuntimeError:
        return -next(_no_version)
