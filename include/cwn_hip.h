/*
 * cwn_hip.h -- C ABI of libcwn_hip.so, the MI355X (gfx950) cellular message-passing engine.
 *
 * The reference (twitter-research/cwn) has no FFI: its hot path is the Python method
 * CochainMessagePassing.propagate (mp/cell_mp.py:357-392) whose arithmetic is three PyTorch /
 * torch-scatter call sites.  Each entry point below replaces one of those call sites (cited per
 * function, paths relative to the reference root); cwn_amd/cell_mp.py binds them through ctypes
 * behind the reference's propagate(...) signature.  INTEGRATION.md shows the binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller owns every buffer; entry points never allocate, free or synchronise, so they
 *     are HIP-graph capturable; work is enqueued on `stream` (a hipStream_t, passed as void*);
 *   - return value: CWN_OK or an error code (cwn_error_string); device-side index errors are
 *     reported through a caller-provided int32 error word (see cwn_csr_build);
 *   - features are fp32 row-major [rows, F]; indices arrive as int64 (the reference asserts
 *     torch.long, mp/cell_mp.py:158) and are narrowed ONCE to int32 CSR by cwn_csr_build.
 *   - DEVICE-SIDE ROW COUNTS (ABI 17).  The reference's training loop draws a new shuffled batch every step
 *     (data/data_loading.py:84-111, exp/train_utils.py:35-75): every batch has its own cell and entry counts, and a
 *     captured hipGraph bakes kernel arguments in.  Descriptors whose row count is a host integer therefore carry an
 *     optional `m_dev` (device int64, or NULL): when given, the host count (M, n_dst, n_rows ...) is the CAPACITY of the
 *     buffers -- it sizes the grid and bounds every address -- and the kernel reads the ACTUAL count from *m_dev
 *     (0 <= *m_dev <= capacity; the caller vouches) for everything that is stored, counted or divided by.  Rows in
 *     [*m_dev, capacity) are never written and never enter a reduction.  One captured launch then serves batches of
 *     any shape up to the capacity; the item-table kernels (cwn_layer_fused_f32, cwn_layer_bwd_own_f32) have always
 *     worked this way (their sizes live in the item table).  cwn_collate_tables / cwn_layer_items_build_dev /
 *     cwn_layer_bwd_items_build_dev produce the per-batch tables on the device.
 */
#ifndef CWN_HIP_H
#define CWN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CWN_ABI_VERSION 24

typedef void* cwn_stream_t; /* hipStream_t */

enum {
    CWN_OK = 0,
    CWN_ERR_BAD_ARG = 1,    /* null pointer, negative size, too many descriptors, F <= 0 ...   */
    CWN_ERR_TOO_LARGE = 2,  /* E or n_dst does not fit int32                                   */
    CWN_ERR_WORKSPACE = 3,  /* workspace smaller than cwn_csr_workspace_bytes says             */
    CWN_ERR_LAUNCH = 4,     /* hipGetLastError() != hipSuccess after a launch                  */
    CWN_ERR_ALIGN = 5       /* a pointer misses its alignment: 4 bytes, or 16 where an entry says so */
};

#define CWN_MAX_DESCS 8 /* descriptors per batched call (one kernel launch covers all of them) */
#define CWN_CSR_MAX_DESCS 16 /* ... of cwn_csr_build / cwn_csr_workspace_bytes (ABI 23: the plans of two slots of a static batch,
                              * or of a training step's adjacencies AND their transposes, in one launch sequence) */

int cwn_abi_version(void);
const char* cwn_error_string(int code);
/* name of the gfx target the library was compiled for, e.g. "gfx950" */
const char* cwn_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * Index preprocessing: COO (as delivered by data/complex.py) -> destination-sorted CSR.
 *
 * Replaces nothing arithmetic in the reference; it is the one-time (per batch) conversion that
 * lets every later aggregation be an atomic-free segmented reduction.  The order inside a
 * segment is the ORIGINAL entry order (stable), i.e. the order torch's sequential index_add_
 * (the scatter under mp/cell_mp.py:439) visits them, so sums are reproducible and the integer
 * outputs are bit-exact against oracle/cwn_oracle.py::csr_from_coo.
 *
 *   key[e]  destination cell of entry e   (upper/lower_index[1], boundary_index[1])
 *   val[e]  source row of entry e         (index[0])              -> col[p]  = val[perm[p]]
 *   aux[e]  optional second index         (shared_coboundaries /
 *                                          shared_boundaries)     -> aux_out[p] = aux[perm[p]]
 *   rowptr[i]..rowptr[i+1]  the entries whose key is i;  perm[p] = original entry id.
 *
 * Passing (key=index[0], val=index[1]) yields the transposed structure used by the backward
 * pass.  Entries with key outside [0,n_dst) or val outside [0,n_val) set *err_flag (bit 0 / bit 1)
 * and are dropped; the Python layer turns that into the IndexError index_select would raise.
 * ------------------------------------------------------------------------------------------ */
typedef struct cwn_csr_desc {
    const int64_t* key;  /* [E] */
    const int64_t* val;  /* [E] */
    const int64_t* aux;  /* [E] or NULL */
    int64_t n_entries;   /* E >= 0 */
    int64_t n_dst;       /* rows of the CSR (>= 0) */
    int64_t n_val;       /* bound for val (rows of the gathered matrix) */
    int64_t n_aux;       /* bound for aux (ignored when aux == NULL) */
    int32_t* rowptr;     /* [n_dst + 1] out */
    int32_t* col;        /* [E] out */
    int32_t* perm;       /* [E] out */
    int32_t* aux_out;    /* [E] out or NULL */
    int32_t* long_rows;  /* [CWN_LONG_PARTS][E / CWN_LONG_ROW + 1] out or NULL: long-row lists */
    int32_t* n_long;     /* [CWN_LONG_PARTS] out or NULL: length of each list (all eight written) */
    const int64_t* e_dev; /* or NULL: device int64, the ACTUAL number of entries (n_entries is then the CAPACITY of key / val / aux and
                           * of the outputs: it sizes the grid, the workspace and the long-row lists; entries in [*e_dev, n_entries) are
                           * not read).  With it a plan is (re)built inside a captured graph for whatever batch a static buffer holds
                           * (cwn_amd/static_batch.py, mode 'csr'): rows past the batch's own cells simply have no entries. */
} cwn_csr_desc;

/* Rows with more entries than CWN_LONG_ROW are "long" (REDDIT-like hubs): cwn_csr_build lists
 * them, and cwn_aggregate_f32 reduces each of them with a whole workgroup instead of one lane
 * group.  The list comes in CWN_LONG_PARTS independent sub-lists of capacity
 * long_cap = E / CWN_LONG_ROW + 1 each (the single-launch build fills one per workgroup, so the
 * counters need no zeroing and no cross-workgroup atomics); order inside a list is unspecified. */
#define CWN_LONG_ROW 64
#define CWN_LONG_PARTS 8

/* Bytes of scratch cwn_csr_build needs for these descriptors (host array of n descriptors). */
size_t cwn_csr_workspace_bytes(const cwn_csr_desc* descs_host, int n);

/* Build up to CWN_MAX_DESCS CSR structures with one fixed sequence of launches.
 * `descs_host` is a HOST array (copied into kernel arguments).  `err_flag` is a device int32 the
 * caller zeroed. */
int cwn_csr_build(const cwn_csr_desc* descs_host, int n, void* workspace, size_t workspace_bytes,
                  int32_t* err_flag, cwn_stream_t stream);

/* The long-row lists of CSR structures that cwn_csr_build did NOT build (ABI 23): a static batch takes the CSR of a boundary
 * adjacency and of its TRANSPOSE as the concatenation of the per-complex CSRs kept with the dataset (cwn_collate_slots) -- and
 * the transpose of a REDDIT-like boundary has hub rows (a vertex of degree 300 is the boundary of 300 edges) that
 * cwn_aggregate_f32 then walked with one lane group each: its backward launch 54 against 37 us.  One workgroup per structure:
 * rows r < *m_dev (or n_rows) with rowptr[r + 1] - rowptr[r] > CWN_LONG_ROW go to sub-list 0 of long_rows, n_long[0] = their
 * number, n_long[1 ..] = 0; more than long_cap of them: the first long_cap (the others are walked the slow way). */
typedef struct cwn_long_rows_desc {
    const int32_t* rowptr;   /* [n_rows + 1] */
    int64_t n_rows;          /* capacity */
    const int64_t* m_dev;    /* or NULL: rows that exist */
    int32_t* long_rows;      /* [CWN_LONG_PARTS][long_cap] out */
    int32_t* n_long;         /* [CWN_LONG_PARTS] out */
    int64_t long_cap;
} cwn_long_rows_desc;
int cwn_csr_long_rows(const cwn_long_rows_desc* descs_host, int n, cwn_stream_t stream);   /* n <= CWN_CSR_MAX_DESCS */

/* ------------------------------------------------------------------------------------------
 * K1: row gather.  out[e, :] = src[idx[e], :]
 * Replaces  src.index_select(node_dim, index[dim])   mp/cell_mp.py:198   (and the up_attr /
 * down_attr gathers of data/complex.py:579-580, 587-588 when a caller wants them materialised).
 * Used by the generic path (arbitrary Python message hooks).  idx is int64 as delivered.
 * ------------------------------------------------------------------------------------------ */
int cwn_gather_rows_f32(const float* src, int64_t n_src, int64_t F, const int64_t* idx,
                        int64_t n_idx, float* out, cwn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K1+message+K2 fused: segmented gather-reduce over a destination-sorted CSR.
 *
 * For every destination row i of every descriptor
 *     out[i,:] = reduce_{p in [rowptr[i], rowptr[i+1])} msg(p)  [+ (1 + *eps) * self[i,:]]
 * with msg(p) selected by `msg_op`:
 *     CWN_MSG_A            A[ia[p], :]                       base-class message (identity),
 *                                                            mp/cell_mp.py:394-421; with ia=perm
 *                                                            it is the segment-reduce of rows a
 *                                                            Python hook produced (K2 alone)
 *     CWN_MSG_A_PLUS_B     A[ia[p], :] + B[ib[p], :]         DummyCochainMessagePassing,
 *                                                            mp/layers.py:23-31
 *     CWN_MSG_A_TIMES_B    A[ia[p], :] * B[ib[p], :]         OrientedConv, mp/layers.py:462-470
 *                                                            (B has width F or width 1)
 *     CWN_MSG_RELU_A_PLUS_B relu(A[ia[p], :] + B[ib[p], :])  SparseCIN coboundary message
 *                                                            ReLU(Linear(cat(x_j, up_attr))),
 *                                                            mp/layers.py:290-293, restructured as
 *                                                            A = X_d W1^T + b, B = X_{d+1} W2^T
 *     CWN_MSG_A_MASK_RELU  A[ia[p], :] * step(self_pre[i,:] + B[ib[p], :] > 0)
 *     CWN_MSG_RELU_A_PLUS_B_SQ  relu(A[ia[p], :] + B[ib[p], :])^2   the second moment of the CIN messages: BatchNorm(train) over
 *                               the ENTRIES of an adjacency (mp/models.py:40-47 conv_up / conv_down in training mode) needs
 *                               sum_e r_e and sum_e r_e^2 per column -- the column sums of two aggregations
 *     CWN_MSG_A_TIMES_2RELU     2 A[ia[p], :] * relu(self_pre[i, :] + B[ib[p], :])   its transposed (backward) form
 *                                                            backward of the previous form
 * and `reduce` one of add / mean / max (mp/cell_mp.py:104-105 -> torch_scatter.scatter, :439):
 * rows with no entry are 0 for every reduce; mean divides by max(count, 1).
 * A descriptor with rowptr == NULL is an ABSENT adjacency: out = 0 (+ self term), which is
 * CochainMessagePassing.update's zero fill (mp/cell_mp.py:517-522) without a separate launch.
 * The optional self term is the GIN-style  out += (1 + eps) * x  of mp/layers.py:191-192 (eps is a
 * DEVICE scalar because it may be a trainable parameter; NULL means eps = 0).
 * All descriptors of one call run in ONE kernel launch.
 * ------------------------------------------------------------------------------------------ */
enum { CWN_MSG_A = 0, CWN_MSG_A_PLUS_B = 1, CWN_MSG_A_TIMES_B = 2, CWN_MSG_RELU_A_PLUS_B = 3,
       CWN_MSG_A_MASK_RELU = 4, CWN_MSG_RELU_A_PLUS_B_SQ = 5, CWN_MSG_A_TIMES_2RELU = 6 };
enum { CWN_REDUCE_ADD = 0, CWN_REDUCE_MEAN = 1, CWN_REDUCE_MAX = 2 };

/* cwn_agg_desc.flags.  SMALL_OPERANDS: the caller vouches that every element of A and of B that an
 * index can address lies within 4 GiB of its base pointer (rows_a * F * 4 and rows_b * b_width * 4
 * below 2^32); when all descriptors of a launch say so the kernel addresses rows with 32-bit byte
 * offsets (one address register per load in flight instead of two).  Never required. */
#define CWN_AGG_SMALL_OPERANDS 1

typedef struct cwn_agg_desc {
    const int32_t* rowptr; /* [n_dst+1] or NULL (absent adjacency) */
    const int32_t* ia;     /* [E] row of A per CSR position */
    const int32_t* ib;     /* [E] row of B per CSR position, or NULL */
    const float* A;        /* [rows_a, F] */
    const float* B;        /* [rows_b, b_width] or NULL */
    const float* self_x;   /* [n_dst, F] or NULL */
    const float* eps;      /* device scalar or NULL */
    const float* self_pre; /* [n_dst, F], CWN_MSG_A_MASK_RELU only */
    float* out;            /* [n_dst, F] */
    const int32_t* long_rows; /* from cwn_csr_build, or NULL (every row is reduced by one lane group) */
    const int32_t* n_long;    /* [CWN_LONG_PARTS] from cwn_csr_build, or NULL */
    int64_t n_dst;
    int32_t F;
    int32_t b_width;       /* F or 1 */
    int32_t msg_op;
    int32_t reduce;
    int32_t long_cap;      /* capacity of one long-row sub-list (E / CWN_LONG_ROW + 1) */
    int32_t flags;         /* CWN_AGG_* bits (0: none) */
    const float* self_x2;  /* [n_dst, F] or NULL: a second self term, out += (1 + *eps2) * self_x2 */
    const float* eps2;     /* device scalar or NULL (= 0) */
    const int64_t* m_dev;  /* or NULL: the ACTUAL number of destination rows (n_dst is then the capacity), see "Conventions" */
} cwn_agg_desc;

int cwn_aggregate_f32(const cwn_agg_desc* descs_host, int n, cwn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A whole SparseCIN propagate step of one layer in ONE launch (inference / no-grad):
 * everything SparseCINConv.forward does before the update networks (mp/layers.py:333-342 ->
 * :184-192 per dimension -> CochainMessagePassing.propagate, mp/cell_mp.py:357-392), for all
 * cochain dimensions, straight from the reference's int64 COO indices -- no CSR plan, no
 * intermediate in HBM:
 *
 *     out_up_d[i] = sum_{e: up_index_d[1,e] = i} relu( W_d [x_d[up_index_d[0,e]] | x_{d+1}[shared_d[e]]] + b_d )
 *                   + (1 + eps1_d) x_d[i]                  (zeros + self term when d has no upper adjacency)
 *     out_b_d[i]  = sum_{e: b_index_d[1,e] = i} x_{d-1}[b_index_d[0,e]]  + (1 + eps2_d) x_d[i]
 *
 * i.e. message_up = msg_up_nn = ReLU(Linear(cat(x_j, up_attr))) (mp/layers.py:290-295) with
 * up_attr = x_{d+1}[shared_coboundaries] (data/complex.py:579-580), message_boundary = identity
 * (:299), both aggregations 'add', absent adjacency = zeros (mp/cell_mp.py:517-522), plus the GIN
 * self terms of mp/layers.py:191-192.
 *
 * A batched complex is a disjoint union (data/complex.py:148-169: every index is offset per
 * complex), so adjacency is block-diagonal and the cells / entries of one complex are contiguous
 * in every tensor.  The launch is cut into ITEMS -- one workgroup each -- that own a contiguous
 * range of complexes for a set of dimensions: the workgroup stages the item's rows of x_d and
 * x_{d+1}, forms Y1 = x_d W[:, :F]^T + b and Y2 = x_{d+1} W[:, F:]^T on the matrix cores into LDS
 * (exact three-way bf16 split, fp32 accuracy: csrc/cwn_split.h), sorts the item's COO entries by
 * destination in LDS (stable: sums run in the original entry order, like a sequential
 * index_add_), reduces relu(Y1[j] + Y2[c]) out of LDS and writes only the two output streams.
 *
 * The item table (device, int32[n_items][CWN_LAYER_ITEM_INTS]) is a property of the BATCH, built
 * once from the per-complex sizes the reference's collate keeps (`ptr`, `__slices__`,
 * data/complex.py:344-441): cwn_layer_items_build below.  Record layout (all offsets are into the batched
 * tensors):
 *   [0] flags      bit 0: the item has a GEMM dimension g (an upper adjacency with coboundary features);
 *                  bits 8-9: its SET (informative; the kernel derives it from cwn_layer_plan.set_start)
 *                  -- sets are numbered over the dimensions in ascending order: a dimension with
 *                  e_up > 0 opens a set as its GEMM dimension (the top dimension rides as its second
 *                  task when it has no upper adjacency itself), any other dimension is a set of its own
 *   [1] g          [2] first cell of dim g   [3] number of cells of dim g
 *   [4] first cell of dim g+1                [5] number of cells of dim g+1
 *   [6] first entry of up_index_g            [7] number of entries
 *   [8] number of tasks (0 = empty record, the workgroup leaves at once; 1 or 2); task t at [9 + 7 t],
 *       all seven fields 0 for an absent task:
 *       +0 dim d  +1 first cell  +2 number of cells (outputs out_up_d / out_b_d of these rows)
 *       +3 first entry of b_index_d  +4 number of entries
 *       +5 first cell of dim d-1     +6 number of cells of dim d-1 (boundary sources; 0 when the task has
 *          no boundary entries: sources are staged only when read)
 *   [23] R1: first staged row of the cells of g+1 = 16*ceil(n_g/16) rounded up to a multiple of
 *        cwn_layer_round_rows(F) (16*ceil(n/16) when there are no cells of g+1)
 *   [24] staged rows = R1 + 16*ceil(n_{g+1}/16) (16*ceil(n/16) without cells of g+1)
 *   [25] b1 = entries of up_index padded to 4   [26] b2 = (b1 + boundary entries of task 0) padded to 4
 *   [27] total = (b2 + boundary entries of task 1) padded to 4          [28..31] 0
 *   The derived fields [23..27] are part of the record so that sixteen waves do not each re-derive them;
 *   cwn_layer_items_check validates a host copy of the table (derived fields, ranges against the plan's
 *   summary, caps).  Without bit 0 of [0], fields [4..7] are 0.
 *   A task whose dim is g reduces the upper adjacency out of LDS; any other task must belong to a
 *   dimension without upper adjacency (out_up = self term); the GEMM dimension is task 0, a second task is
 *   the block of g+1.  Limits per item: staged rows <= CWN_LAYER_GEMM_ROWS(F); boundary-source cells of its
 *   tasks together <= CWN_LAYER_SOURCE_ROWS(F); LDS of both within 160 KiB (cwn_layer_fused_lds_bytes);
 *   cells per task <= CWN_LAYER_TASK_ROWS; total <= CWN_LAYER_MAX_ENTRIES.
 * An index that leaves its item's ranges (the batch is not block-diagonal, or the table does not
 * belong to it) sets bit 3 of *err_flag (the sticky word of cwn_csr_build) and is clamped.
 * F must be 64 or 128; every pointer 16-B aligned; one launch, no workspace, no host sync.
 * ------------------------------------------------------------------------------------------ */
#define CWN_LAYER_MAX_DIMS 3
#define CWN_LAYER_ITEM_INTS 32
#define CWN_LAYER_GEMM_ROWS(F) ((F) == 64 ? 256 : 96)    /* staged rows of an item (whole rounds of cwn_layer_round_rows) */
#define CWN_LAYER_SOURCE_ROWS(F) ((F) == 64 ? 256 : 96)  /* cells of dim d-1 the boundary stream of an item's first task reads */
#define CWN_LAYER_TASK_ROWS 192
#define CWN_LAYER_MAX_ENTRIES 1024
#define CWN_ERR_BIT_BLOCK 8                    /* *err_flag bit: index outside its item */
/* VARIANT 1 of the launch, the two-per-CU form (cwn_layer_plan.variant = 1): workgroups of 8 waves within 128 VGPRs
 * and CWN_LAYER_W8_LDS_BYTES of LDS, so that two are resident on a CU and one item's load / sort phases run under
 * the other's matrix-core / reduce phases -- for launches with more items than the chip has CUs.  Same record
 * layout, same arithmetic, bit-identical outputs; smaller caps, and the rows of EACH product (cells of g, cells of
 * g+1, each padded to 16) are bounded by CWN_LAYER_W8_HALF_ROWS.  Rows per round: cwn_layer_variant_round_rows. */
/* BIG ITEMS (record flag bit 1; variant 0 only; cwn_layer_sizes.allow_big).  A complex whose rows or entries exceed
 * what a workgroup's LDS holds (at F = 128: the atoms and the bonds of a molecule, each padded to 16, beyond 96 staged rows -- 48 atoms + 48 bonds at most, i.e. molecules of more than ~44 atoms; round 3 padded the first block to whole load rounds of 32 and stopped at 32 atoms -- or of more than ~115 atoms at F = 64) used to send its WHOLE
 * batch to the streaming path (cwn_gemm_f32 + cwn_aggregate_f32).  With allow_big the table builder gives such a
 * complex one BIG record per set instead: its workgroup runs the streaming algorithm by itself inside the same launch --
 * Y1 / Y2 of the complex row tile by row tile on the matrix cores straight from the fp32 rows into the scratch matrices
 * cwn_layer_dim.big_y1 / big_y2 (global memory, L2-resident), then the segmented reductions over the caller's CSR of the
 * big complexes' entries (big_up_* / big_b_*: built once per batch) -- same split, same MFMA order per tile, same entry
 * order: bit-identical to the other paths.  A BIG record uses the fields [0..22] as above (all ranges of the complex);
 * the derived fields [23..27] are 0.  An upper / boundary entry of a big complex whose source, coface or boundary cell
 * lies outside the complex sets CWN_ERR_BIT_BLOCK. */
#define CWN_LAYER_ITEM_BIG 2
#define CWN_LAYER_W8_LDS_BYTES (80 * 1024)
#define CWN_LAYER_W8_GEMM_ROWS(F) ((F) == 64 ? 128 : 80)
#define CWN_LAYER_W8_SOURCE_ROWS(F) ((F) == 64 ? 128 : 48)
#define CWN_LAYER_W8_HALF_ROWS(F) ((F) == 64 ? 128 : 64)

typedef struct cwn_layer_dim {
    const float* x;            /* [n_cells, F] */
    const int64_t* up_index;   /* [2, e_up] upper_index (row 0 source, row 1 destination) or NULL */
    const int64_t* up_shared;  /* [e_up] shared_coboundaries or NULL */
    const int64_t* b_index;    /* [2, n_b] boundary_index (row 0 boundary cell, row 1 cell) or NULL */
    const void* msg_w_packed;  /* msg_up_nn's Linear weight [F, 2F] packed by cwn_layer_pack_weights_f32, or NULL */
    const float* msg_bias;     /* [F] or NULL */
    const float* eps1;         /* device scalar or NULL (= 0) */
    const float* eps2;
    float* out_up;             /* [n_cells, F] */
    float* out_b;              /* [n_cells, F] */
    int64_t n_cells, e_up, n_b;
    /* BIG items only (all NULL otherwise; see "Big items" below): the destination-sorted int32 CSR (cwn_csr_build, GLOBAL
     * cell numbers) of the big complexes' entries of up_index / b_index of this dimension, and two [n_cells, F] scratch
     * matrices the launch may write (Y1 of this dimension as a GEMM dimension, Y2 of it as the coface dimension) */
    const int32_t* big_up_rowptr;   /* [n_cells + 1] */
    const int32_t* big_up_col;      /* source cell (dim d) */
    const int32_t* big_up_aux;      /* shared coface (dim d + 1) */
    const int32_t* big_b_rowptr;    /* [n_cells + 1] */
    const int32_t* big_b_col;       /* boundary cell (dim d - 1) */
    float* big_y1;
    float* big_y2;
    /* A THIRD output (ABI 22; NULL = not wanted): out_down[i] = (1 + eps3) x_i -- what CINppCochainConv.forward
     * (mp/layers.py:243-260) feeds update_down_nn when the lower stream is off, which is how the reference's molecular CIN++
     * models run it (include_down_features=False: down_index is None, propagate() returns zeros, :253 adds the self term).
     * Written by the workgroup that owns the row, from the registers that hold it: no extra read. */
    float* out_down;           /* [n_cells, F] or NULL */
    const float* eps3;         /* device scalar or NULL (= 0) */
} cwn_layer_dim;

/* The weight of the message Linear in the form the kernel's matrix-core loop reads it: the exact
 * three-way bf16 split (csrc/cwn_split.h) of W [F, 2F] (torch layout, row stride ldw), the pieces
 * laid out in MFMA-fragment order so that every wave instruction fetches 1 KiB of contiguous memory
 * (a fragment-shaped read of the fp32 weight -- 16 rows x 32 B per quarter wave -- runs the address
 * unit at 1/8 rate and re-splits the same numbers in every workgroup).  One small launch per weight
 * VERSION (the caller re-packs after an optimizer step); out: cwn_layer_packed_weight_bytes(F) bytes,
 * 16-B aligned.  Weight preparation, like folding BatchNorm into an affine: not part of a step. */
size_t cwn_layer_packed_weight_bytes(int32_t F);
int cwn_layer_pack_weights_f32(const float* W, int64_t ldw, int32_t F, void* out, cwn_stream_t stream);
/* the same for n weights of one width in ONE launch (a training step packs the message weights of all its layers once,
 * after the optimizer has written them): host arrays of n device pointers / row strides */
#define CWN_LAYER_PACK_MAX 32
int cwn_layer_pack_weights_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out, int32_t n,
                                    cwn_stream_t stream);
/* ... and in the form the BACKWARD launch (cwn_layer_bwd_f32) multiplies with: the same planes and chunk order of the
 * TRANSPOSED halves, chunk (ks, plane, h, ct) holding W[ks * 32 + kq * 8 .., h * F + ct * 16 + n] for lane kq * 16 + n
 * (output column = input feature of the Linear, reduction over its outputs: dX = gY W). */
int cwn_layer_pack_weights_t_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out, int32_t n,
                                      cwn_stream_t stream);
/* ... and both in ONE launch: out[e] the forward form, out_t[e] the transposed one. */
int cwn_layer_pack_weights_both_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out,
                                         void* const* out_t, int32_t n, cwn_stream_t stream);

/* The item table and what the launcher needs to know about it (HOST struct; built by
 * cwn_layer_items_build).  Items are ordered by set.  The *_end fields summarise what the table
 * addresses; the launcher checks them against the tensors, so that the kernel can form addresses
 * from a record without re-validating it (the caller vouches that the summary describes the table).
 * What only the device can see -- the VALUES of the int64 indices -- is checked in the kernel.
 * csr_cache (optional, n_items * CWN_LAYER_CSR_SLOT_BYTES device bytes): with CWN_LAYER_CSR_STORE
 * every workgroup also stores its item's finished CSR there, with CWN_LAYER_CSR_LOAD it loads it
 * back instead of reading and sorting the COO entries again -- the layers of one forward share
 * their index tensors (mp/molec_models.py:110-116), so layer 0 stores and layers 1.. load; the
 * caller invalidates (stores again) whenever the index tensors change. */
#define CWN_LAYER_CSR_SLOT_BYTES 5264          /* 2 * 2 * MAX_ENTRIES + 3 * 2 * (TASK_ROWS + 2), 16-B multiple */
#define CWN_LAYER_CSR_STORE 1
#define CWN_LAYER_CSR_LOAD 2
/* STORE_Y (any form, with or without the CSR flags): every item also writes its rows of the message products to the
 * matrices cwn_layer_dim.big_y1 (Y1 = x_d W[:, :F]^T + b of this dimension's cells) and the NEXT dimension's big_y2
 * (Y2 = x_{d+1} W[:, F:]^T) -- each row exactly once, by the item that owns its complex.  The training step runs its
 * forward through this launch and keeps Y1 / Y2 for the backward pass (the ReLU mask of the coboundary message needs
 * them), where the inference path never lets them leave LDS.  Both matrices are required for every dimension with an
 * upper adjacency (CWN_ERR_BAD_ARG otherwise). */
#define CWN_LAYER_STORE_Y 4

typedef struct cwn_layer_plan {
    const int32_t* items;                       /* device int32 [n_items][CWN_LAYER_ITEM_INTS] */
    void* csr_cache;                            /* device or NULL */
    int64_t n_items;
    int32_t set_start[CWN_LAYER_MAX_DIMS + 1];  /* first item of set s (set_start[0] = 0) */
    int32_t max_gemm_rows;                      /* bound (multiple of 16) of the staged rows of any item */
    int32_t max_source_rows;                    /* bound of the boundary-source cells of any item */
    int32_t variant;                            /* 0: one 16-wave workgroup per CU; 1: the two-per-CU form (caps above) */
    int64_t cells_end[CWN_LAYER_MAX_DIMS];      /* max (first cell + count) the table names, per dimension */
    int64_t up_end[CWN_LAYER_MAX_DIMS];         /* max (first entry + count) of up_index_d */
    int64_t b_end[CWN_LAYER_MAX_DIMS];          /* max (first entry + count) of b_index_d */
    int64_t n_big;                              /* BIG records in the table (0: none; the launcher then ignores big_*) */
    int64_t lds_bytes;                          /* variant 1: dynamic LDS of the launch = the largest per-item need
                                                   (cwn_layer_variant_lds_bytes of its staged rows and task-0 sources);
                                                   variant 0: unused (one layout per launch from the two maxima) */
} cwn_layer_plan;

int cwn_layer_fused_f32(const cwn_layer_dim* dims_host, int n_dims, int32_t F, const cwn_layer_plan* plan_host,
                        int32_t flags, int32_t* err_flag, cwn_stream_t stream);

/* The BACKWARD of the same step in one launch over the SAME item table (variant 0, no BIG records): see
 * csrc/cwn_layer_bwd.hip for the formulas.  Per dimension: the gradients of the two outputs (NULL = zero), the products
 * the forward launch stored (CWN_LAYER_STORE_Y: y1 = Y1 of this dimension, y2 = Y2 of the dimension BELOW, both
 * [n_cells, F]), the index tensors as in cwn_layer_dim, the transposed packed message weight of this dimension
 * (cwn_layer_pack_weights_t_many_f32; NULL without upper adjacency), eps.  Outputs: dx [n_cells, F] -- ZEROED by the
 * caller, every piece is added with fp32 atomics -- and gy1 / gy2 [n_cells, F] (written; NULL = not wanted): the
 * gradients of Y1 of this dimension / of Y2 of the dimension below, what the weight-gradient GEMM multiplies.
 * An index that leaves its item sets bit 3 of *err_flag.  cwn_layer_bwd_lds_bytes: the launch's LDS for a table whose
 * largest item stages max_gemm_rows rows, 0 = beyond a CU (CWN_ERR_TOO_LARGE from the launch: the caller keeps the
 * streaming backward). */
/* The REDUCE half of a BatchNorm backward taken over by the launch that PRODUCES the stage's dy (cwn_dense_stage_bwd_f32's
 * out_bn / out_bn2 below; cwn_layer_bwd_own_f32's out_bn, round 6): the launch adds, from its epilogue, the column sums of
 * dyh = dx * [z * scale + shift > 0] and of dyh * xhat (xhat = (z - mean) * rstd) over its rows into slot row
 * (workgroup number mod CWN_BN_SLOTS) with fp32 atomics; the consumer sums the slots in its prologue (s_slots). */
typedef struct cwn_bn_bwd_live {
    const float* z;          /* [M, F] (row stride ldz) pre-normalisation values of the stage that receives dx as its dy */
    const float* aff;        /* [4][F]: scale, shift, mean, rstd of its BatchNorm (cwn_bn_live.aff) */
    float* slots;            /* [CWN_BN_SLOTS][2][F]; NULL: unused */
    int64_t ldz;
} cwn_bn_bwd_live;
typedef struct cwn_layer_bwd_dim {
    const float* g_up;
    const float* g_b;
    const float* y1;
    const float* y2;
    const int64_t* up_index;
    const int64_t* up_shared;
    const int64_t* b_index;
    const void* wt_packed;
    const float* eps1;
    const float* eps2;
    float* dx;
    float* gy1;
    float* gy2;
    int64_t n_cells, e_up, n_b;
    /* (ABI 23; cwn_layer_bwd_own_f32 only, all-zero = unused) x_d is the OUTPUT of a Linear -> BatchNorm -> ReLU stage -- the
     * combine network of the previous conv layer, mp/layers.py:322-325 -- whose backward begins with the column sums of
     * dx_d * mask and dx_d * mask * xhat: every workgroup adds the sums of the rows of dx_d it owns into out_bn.slots (z, aff
     * of that stage), and the previous layer's backward launches no reduce (cwn_norm_bwd_reduce_f32: 5.9 us, four per ZINC step). */
    cwn_bn_bwd_live out_bn;
} cwn_layer_bwd_dim;
size_t cwn_layer_bwd_lds_bytes(int32_t F, int32_t max_gemm_rows);
int cwn_layer_bwd_f32(const cwn_layer_bwd_dim* dims_host, int n_dims, int32_t F, const cwn_layer_plan* plan_host,
                      int32_t* err_flag, cwn_stream_t stream);
/* The OWNER form of the same backward (csrc/cwn_layer_bwd_own.hip; what autograd derives for mp/layers.py:184-192, 290-295,
 * 333-342 -> mp/cell_mp.py:357-392): every row of dx has ONE writer.  An item is a
 * contiguous range of complexes for one dimension d whose cells it OWNS; its workgroup gathers everything those rows
 * receive -- gY1_d over the entries of up_index_d that name the row as source, the gradient of Y2 at d over the entries of
 * up_index_{d-1} that name it as coface, the boundary transposes over the entries of b_index_{d+1} that name it as
 * boundary cell -- multiplies [gY1_d | gY2] by the transposed message weights of d and d-1 on the matrix cores and
 * stores dx_d = products + self terms + transposes ONCE: no atomics, no zeroed output, sums in entry order
 * (deterministic).  A top dimension without upper adjacency rides with the dimension below it (its rows are owned by
 * the same item: third product + self terms), exactly as in the forward's sets.  Record (int32 x 16):
 *   [0] flags   bit 0 (A): dimension d reduces an upper adjacency (entries [8], [9] of up_index_d; product gY1_d W_d[:, :F])
 *               bit 1 (B): dimension d-1 does (entries [10], [11] of up_index_{d-1}; product gY2 W_{d-1}[:, F:])
 *               bit 2 (TOP): the item also owns the cells of d+1 (product gY2_{d+1} W_d[:, F:], self terms); bits 8-9: set
 *   [1] d       [2] first owned cell  [3] owned cells
 *   [4] first cell of d+1  [5] cells of d+1 (coface rows of A, sources of the transposes, the TOP rows; 0: none needed)
 *   [6] first cell of d-1  [7] cells of d-1 (0 unless B)
 *   [8] first entry of up_index_d      [9] entries (0 unless A)
 *   [10] first entry of up_index_{d-1} [11] entries (0 unless B)
 *   [12] first entry of b_index_{d+1}  [13] entries (0: no transposes)
 *   [14] LDS bytes of the item (csrc/cwn_layer_bwd_own.h: the layout follows from the record)   [15] 0
 * Limits per item: owned cells <= CWN_LAYER_GEMM_ROWS(F), TOP cells <= 1024 / (F / 4), each entry list <=
 * CWN_LAYER_MAX_ENTRIES, LDS <= 160 KiB.  The table is built on the host from the same prefix sums as the forward's
 * (cwn_layer_sizes; allow_big / skip / unfit are ignored: a complex beyond the limits gives CWN_LAYER_ITEMS_TOO_LARGE
 * and the caller keeps the streaming backward).  dims: as for cwn_layer_bwd_f32, except that dx needs no zeroing. */
#define CWN_LAYER_BWD_ITEM_INTS 16
typedef struct cwn_layer_bwd_plan {
    const int32_t* items;                       /* device int32 [n_items][CWN_LAYER_BWD_ITEM_INTS] */
    int64_t n_items;
    int64_t lds_bytes;                          /* dynamic LDS of the launch = the largest item */
    int64_t cells_end[CWN_LAYER_MAX_DIMS];      /* as in cwn_layer_plan: what the table addresses */
    int64_t up_end[CWN_LAYER_MAX_DIMS];
    int64_t b_end[CWN_LAYER_MAX_DIMS];
} cwn_layer_bwd_plan;
struct cwn_layer_sizes;
int64_t cwn_layer_bwd_items_build(const struct cwn_layer_sizes* sizes_host, int32_t F, int32_t* items_host, int64_t cap_items,
                                  cwn_layer_bwd_plan* plan_host);
int cwn_layer_bwd_own_f32(const cwn_layer_bwd_dim* dims_host, int n_dims, int32_t F, const cwn_layer_bwd_plan* plan_host,
                          int32_t* err_flag, cwn_stream_t stream);
/* The item table, built on the HOST from the per-complex prefix sums the reference's collate keeps (`ptr`:
 * data/complex.py:344, 432; `__slices__`: :349-394): contiguous ranges of complexes per set, greedily under the
 * limits above, with ONE launch's LDS split between staged rows and boundary sources so that the items are as
 * few as possible; heavy items first within a set.  Fills items[0 .. n) (host memory, cap_items records) and every
 * field of *plan but `items` / `csr_cache`.  Returns n >= 1, 0 for an empty batch, CWN_LAYER_ITEMS_TOO_LARGE when
 * a single complex exceeds what a workgroup holds (the caller uses cwn_csr_build + cwn_gemm_f32 +
 * cwn_aggregate_f32), CWN_LAYER_ITEMS_BAD_ARG otherwise.  Tens of microseconds for a batch of 128. */
typedef struct cwn_layer_sizes {
    int64_t n_complexes;
    int32_t n_dims;
    int32_t has_up[CWN_LAYER_MAX_DIMS];             /* dimension d reduces an upper adjacency with coboundary features */
    int32_t allow_big;                              /* 1: a complex beyond the caps becomes BIG records (variant 0) instead of
                                                       CWN_LAYER_ITEMS_TOO_LARGE */
    int32_t pad_;
    const int64_t* cell_ptr[CWN_LAYER_MAX_DIMS];    /* [n_complexes + 1] prefix sums of the cells per complex */
    const int64_t* up_ptr[CWN_LAYER_MAX_DIMS];      /* the same for the entries of upper_index_d, or NULL */
    const int64_t* b_ptr[CWN_LAYER_MAX_DIMS];       /* the same for the entries of boundary_index_d, or NULL */
    /* A batch served by TWO launches -- the two-per-CU form for the complexes that fit its (smaller) caps, the 16-wave
     * form for the rest -- is cut in two calls over complementary subsets of the complexes:
     * skip (in, [n_complexes] or NULL): complexes with a non-zero byte are left out of this table;
     * unfit (out, [n_complexes] or NULL): when given, a complex that does not fit the caps in some set is marked 1 and
     * left out of that set's items instead of failing the table (the caller then builds again with skip = unfit, so
     * that a complex is in every set of a table or in none). */
    const uint8_t* skip;
    uint8_t* unfit;
} cwn_layer_sizes;
#define CWN_LAYER_ITEMS_TOO_LARGE (-1)
#define CWN_LAYER_ITEMS_BAD_ARG (-2)
int64_t cwn_layer_items_build(const cwn_layer_sizes* sizes_host, int32_t F, int32_t* items_host, int64_t cap_items,
                              cwn_layer_plan* plan_host);     /* plan_host->variant (in): which form's caps to cut under */

/* The same tables built ON THE DEVICE (one workgroup per set, a few microseconds), for launches captured once over
 * capacity-sized buffers: the prefix sums are device arrays (the `seg` / `dst` rows cwn_collate_tables writes) and so is
 * the number of complexes.  Set s owns the FIXED region [set_start[s], set_start[s + 1]) of the table (set_start[n_sets] =
 * n_items = the capacity of the table; a region of `cap_complexes` records always suffices); records past a set's own
 * are zeroed (empty: their workgroups leave at once).  The cut is simpler than the host's: complexes are taken in
 * groups of `group` consecutive ones; a group that fits the caps is one item, one that does not is cut into its single
 * complexes, and a complex that does not fit alone sets bit 4 (16) of *err_flag and gets no record (the caller checks
 * that before it trusts a batch to this path -- the per-complex sizes are host metadata too).  No BIG records, no
 * heavy-first order.  Forward: plan_host gives `items` (device, written), n_items, set_start, variant, and the LDS
 * split of the captured launch (max_gemm_rows / max_source_rows; variant 1: lds_bytes).  Backward: plan_host gives
 * `items`, n_items, lds_bytes (the launch's dynamic LDS); set s owns records [s * cap_complexes, (s + 1) * cap_complexes). */
typedef struct cwn_layer_sizes_dev {
    const int64_t* n_complexes;                     /* DEVICE int64: complexes in this batch, <= cap_complexes */
    int64_t cap_complexes;
    int32_t n_dims;
    int32_t has_up[CWN_LAYER_MAX_DIMS];
    const int64_t* cell_ptr[CWN_LAYER_MAX_DIMS];    /* DEVICE [cap_complexes + 1]; entries past the batch's own = the total */
    const int64_t* up_ptr[CWN_LAYER_MAX_DIMS];      /* or NULL */
    const int64_t* b_ptr[CWN_LAYER_MAX_DIMS];       /* or NULL */
    int32_t n_slots;                                /* >= 1: the batches of a multi-slot static batch, cut by one launch: slot j reads every */
    int32_t pad_;                                   /* pointer above at + j * table_slot_stride (int64 elements) and writes the item table */
    int64_t table_slot_stride;                      /* at items + j * n_items records */
} cwn_layer_sizes_dev;
#define CWN_ERR_BIT_UNFIT 16                   /* *err_flag bit: a complex beyond what one workgroup holds (device table build) */
int cwn_layer_items_build_dev(const cwn_layer_sizes_dev* sizes_host, int32_t F, const cwn_layer_plan* plan_host, int32_t group,
                              int32_t* err_flag, cwn_stream_t stream);
int cwn_layer_bwd_items_build_dev(const cwn_layer_sizes_dev* sizes_host, int32_t F, const cwn_layer_bwd_plan* plan_host,
                                  int32_t group, int32_t* err_flag, cwn_stream_t stream);

/* HOST check of a host copy of the item table against its plan (record layout above): CWN_OK or
 * CWN_ERR_BAD_ARG.  The kernel re-checks only what keeps a workgroup inside its LDS. */
int cwn_layer_items_check(const int32_t* items_host, int64_t n_items, int32_t F, const cwn_layer_plan* plan_host);
/* rows a workgroup stages per round (threads / (F / 4)): the coface block of an item starts at a multiple of it */
int32_t cwn_layer_round_rows(int32_t F);                               /* variant 0 */
int32_t cwn_layer_variant_round_rows(int32_t F, int32_t variant);
/* dynamic LDS bytes such a launch uses, 0 for unsupported arguments or more than the variant's budget */
size_t cwn_layer_fused_lds_bytes(int32_t F, int32_t max_gemm_rows, int32_t max_source_rows);     /* variant 0: 160 KiB */
size_t cwn_layer_variant_lds_bytes(int32_t F, int32_t variant, int32_t max_gemm_rows, int32_t max_source_rows);

/* ------------------------------------------------------------------------------------------
 * The update / combine networks of a SparseCIN layer, all dimensions, in ONE launch (inference):
 *
 *     h_up = relu(bn(W2u relu(bn(W1u x_up + b1u)) + b2u))          update_up_nn          (mp/layers.py:303-311)
 *     h_b  = relu(bn(W2b relu(bn(W1b x_b  + b1b)) + b2b))          update_boundaries_nn  (:312-321)
 *     y    = relu(bn(Wc [h_up | h_b] + bc))                        combine_nn            (:322-325, :193-199)
 *
 * with every Linear F wide (Wc: 2F -> F; F = 64 or 128), BatchNorm in eval mode folded into a per-column
 * (scale, shift) or absent (NULL pair), ReLU after every stage.  x_up / x_b are the two outputs of the
 * propagate step (cwn_layer_fused_f32: out_up, out_b).  A workgroup takes 4096 / F rows through all five
 * Linear layers without leaving the CU: the intermediate activations never reach HBM.  Products on the
 * bf16 matrix pipe through the exact three-way operand split (fp32 accuracy, csrc/cwn_split.h).
 * w_packed: cwn_update_mlp_pack_weights_f32 of W1u, W2u, W1b, W2b, Wc[:, :F], Wc[:, F:] (the last two: the
 * column halves of the combine weight, ldw = 2F; at F = 128 cwn_gemm_pack_weights_f32 gives the same
 * layout); bias / scale / shift per stage in the order
 * (1u, 2u, 1b, 2b, c), bias may be NULL.  At most cwn_update_mlp_max_rows() rows per dimension
 * (CWN_ERR_TOO_LARGE beyond; the same networks as grouped launches: cwn_gemm_f32).
 * Every pointer 16-B aligned, row strides multiples of 4; no workspace, no host sync.
 * ------------------------------------------------------------------------------------------ */
typedef struct cwn_mlp_dim {
    const float* x_up;          /* [M, F], row stride ldx_up */
    const float* x_b;           /* [M, F], row stride ldx_b */
    const void* w_packed[6];
    const float* bias[5];       /* [F] or NULL */
    const float* scale[5];      /* [F] or NULL (then shift NULL too) */
    const float* shift[5];
    float* y;                   /* [M, F], row stride ldy */
    int64_t M, ldx_up, ldx_b, ldy;
    const int64_t* m_dev;       /* or NULL: actual rows (M = capacity) */
    int32_t in_width;           /* 0 (= F) or 1 .. F - 1: x_up / x_b have only this many columns (round 5: the first layer of a
                                   model over raw features -- REDDIT-BINARY's one constant feature, mp/models.py:112-260); the
                                   first weight of each branch is then the zero-padded [F, F] form of Linear(in_width -> F), the
                                   row strides may be any value >= in_width and the two pointers need 4-byte alignment only */
    int32_t pad_;
} cwn_mlp_dim;

int cwn_update_mlp_f32(const cwn_mlp_dim* dims_host, int n_dims, int32_t F, cwn_stream_t stream);
int64_t cwn_update_mlp_max_rows(void);
/* an [F, F] weight (row stride ldw) in the form the kernel streams it: cwn_update_mlp_packed_weight_bytes(F)
 * bytes, 16-B aligned; one small launch per weight VERSION */
size_t cwn_update_mlp_packed_weight_bytes(int32_t F);
int cwn_update_mlp_pack_weights_f32(const float* W, int64_t ldw, int32_t F, void* out, cwn_stream_t stream);
/* the same for n blocks in ONE launch (a training step packs the update / combine weights of all its layers once, after
 * the optimizer has written them): host arrays of n device pointers (the first element of each F x F block: a column
 * offset selects half of a combine weight) / row strides / outputs */
#define CWN_STAGE_PACK_MAX 160   /* blocks per launch (the table is a kernel argument: 160 x 25 B) */
int cwn_update_mlp_pack_weights_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out, int32_t n,
                                         cwn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same for a CIN++ layer (ABI 22): THREE update networks and a 3F-wide combine, all dimensions, ONE launch (inference):
 *
 *     h_k = relu(bn(W2_k relu(bn(W1_k x_k + b1_k)) + b2_k))     k = up, down, boundaries     (mp/layers.py:383-410)
 *     y   = relu(bn(Wc [h_up | h_down | h_b] + bc))                                          (:411-414, :255-260)
 *
 * x[k]: the three outputs of the propagate step (cwn_layer_fused_f32 with cwn_layer_dim.out_down: out_up, out_down, out_b),
 * [M, F] with row stride ldx[k].  w_packed (cwn_update_mlp_pack_weights_f32), in the order the launch multiplies: per branch
 * k its W1_k, W2_k and the k-th F-column block of the combine weight (ldw = 3F): [W1u, W2u, Wc[:, 0:F], W1d, W2d, Wc[:, F:2F],
 * W1b, W2b, Wc[:, 2F:3F]].  bias / scale / shift per stage in the order (1u, 2u, 1d, 2d, 1b, 2b, c); NULL as in cwn_mlp_dim.
 * The combine is accumulated branch by branch in the order of the cat, so three plane buffers serve three branches (two
 * workgroups per CU).  Same limits, alignment and arithmetic as cwn_update_mlp_f32; no narrow inputs (in_width).
 * ------------------------------------------------------------------------------------------ */
typedef struct cwn_mlp3_dim {
    const float* x[3];          /* [M, F] each */
    int64_t ldx[3];
    const void* w_packed[9];
    const float* bias[7];       /* [F] or NULL */
    const float* scale[7];      /* [F] or NULL (then shift NULL too) */
    const float* shift[7];
    float* y;                   /* [M, F], row stride ldy */
    int64_t M, ldy;
    const int64_t* m_dev;       /* or NULL: actual rows (M = capacity) */
} cwn_mlp3_dim;

int cwn_update_mlp3_f32(const cwn_mlp3_dim* dims_host, int n_dims, int32_t F, cwn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A BatchNorm1d(train) WITHOUT a launch of its own (round 4; torch.nn.BatchNorm1d inside update_up_nn / update_boundaries_nn /
 * combine_nn, mp/layers.py:303-325, exp/train_utils.py:57-75 in training mode): the launch that PRODUCES the pre-normalisation
 * values adds its workgroups' column sums into `slots` (fp64 atomics, one coalesced instruction per 64 columns and workgroup,
 * workgroup b into slot b % CWN_BN_SLOTS so that an address sees M / (32 * CWN_BN_SLOTS) adds); every workgroup of the launch
 * that CONSUMES them sums the slots in slot order and derives mean / rstd / scale / shift for its columns in its prologue
 * (cwn_bn_finalize_f32's arithmetic), and the FIRST workgroup of the consuming descriptor also writes `aff` (what the backward
 * reads) and updates the running statistics and the batch counter.  `slots` must be ZERO before the producing launch (the
 * caller's per-step fill -- cwn_amd/ops.py: step arena); slots == NULL: the record is unused.  The order of the adds inside a
 * slot is the order of arrival: the sums are fp64, so two runs differ by ~1e-16 relative before they are rounded to fp32 --
 * reproducible in practice, not by construction; cwn_bn_finalize_f32 over per-band partials stays the deterministic form.
 * ------------------------------------------------------------------------------------------ */
#define CWN_BN_SLOTS 4

/* ------------------------------------------------------------------------------------------
 * DROPOUT without a mask tensor (F.dropout of the callers: mp/molec_models.py:104-106 input features, :298-300 after every
 * conv layer of OGBEmbedSparseCIN, :129-146 / :338-346 before lin1 / the final readout / lin2 -- exp/scripts/cwn-molhiv.sh
 * trains with --drop_rate 0.5).  The keep decision of element e of one application is a pure function of (seed, step, site, e):
 *     r = Philox4x32-10(counter = (e / 4, site, step_lo, step_hi), key = (seed_lo, seed_hi))[e % 4]
 *     keep <=> r >= floor(p * 2^32);   y = keep ? x / (1 - p) : 0
 * so the forward applies it in the epilogue of the launch that produces x and the backward re-derives it in the prologue of
 * the launch that consumes the gradient -- no mask is written or read.  `state` = device int64 [2] {seed, step}: READ by the
 * kernels; cwn_step_begin advances step, so every replay of a captured training step draws fresh masks.  `site` tells the
 * applications of one step apart (a host-side counter baked into the launch).  e = row * N + column of the [M, N] matrix the
 * application covers (row stride irrelevant).  state == NULL or p == 0: no dropout.  The stream is Philox as torch uses it but
 * NOT torch's sequence (nothing in the reference pins a mask); cwn_dropout_f32 on a matrix of ones exports the mask for a
 * checker.
 * ------------------------------------------------------------------------------------------ */
typedef struct cwn_dropout {
    const int64_t* state;          /* device [2]: seed, step; NULL = off */
    float p;                       /* drop probability, 0 <= p < 1 */
    uint32_t site;
} cwn_dropout;

/* out[m, n] = x[m, n] * multiplier(m * N + n)  over an [M, N] matrix (the backward is the same call on the gradient).  x == out
 * is allowed.  m_dev (or NULL): the rows that exist (M = capacity). */
int cwn_dropout_f32(const float* x, float* out, int64_t M, int32_t N, int64_t ldx, int64_t ldout, const cwn_dropout* drop,
                    const int64_t* m_dev, cwn_stream_t stream);

typedef struct cwn_bn_live {
    double* slots;                 /* [CWN_BN_SLOTS][2][N]: column sums, column sums of squares */
    const float* gamma;            /* [N] or NULL (= 1) */
    const float* beta;             /* [N] or NULL (= 0) */
    float* running_mean;           /* [N] or NULL */
    float* running_var;
    int64_t* num_batches_tracked;  /* or NULL */
    float* aff;                    /* [4][N] out: scale, shift, mean, rstd (16-B aligned) */
    float eps;
    float momentum;
} cwn_bn_live;

/* ------------------------------------------------------------------------------------------
 * One STAGE of the same networks in TRAINING mode (csrc/cwn_stage.hip), up to CWN_MAX_DESCS products per launch:
 *
 *     Y = prologue([X | X2]) W^T + bias      col_sum / col_sumsq [CWN_STAT_ROWS(M), F]: per-32-row-band sums of Y, Y^2 (fp64)
 *
 * i.e. a Linear(F -> F) or, with X2, Linear(2F -> F) on the K-concatenation (combine_nn's torch.cat, mp/layers.py:199) whose
 * BatchNorm(train) statistics are accumulated in the epilogue exactly as cwn_gemm_f32 does (plain stores per band:
 * deterministic; cwn_bn_finalize_f32 reduces them), with the producing stage's BatchNorm apply + ReLU as the prologue
 * (in_scale / in_shift per input column or NULL, in_relu bit 0: X, bit 1: X2).  Arithmetic: the exact three-way bf16
 * split of csrc/cwn_split.h (fp32 in, fp32 accumulate, fp32 out; the inference kernels' path), weights pre-packed by
 * cwn_update_mlp_pack_weights_f32 or its _many form: w_packed = the block that multiplies X, w2_packed the block that multiplies X2.
 * F = 64 or 128 = the width of X, X2 and Y; pointers 16-B aligned, strides multiples of 4.
 * ------------------------------------------------------------------------------------------ */
typedef struct cwn_stage_desc {
    const float* X;          /* [M, F] */
    const float* X2;         /* [M, F] or NULL */
    const void* w_packed;
    const void* w2_packed;   /* NULL without X2 */
    const float* bias;       /* [F] or NULL */
    const float* in_scale;   /* [F] or NULL */
    const float* in_shift;
    const float* in_scale2;
    const float* in_shift2;
    float* Y;                /* [M, F] */
    double* col_sum;         /* [CWN_STAT_ROWS(M), F] or NULL */
    double* col_sumsq;
    int64_t M, ldx, ldx2, ldy;
    int32_t in_relu;
    int32_t pad_;
    const int64_t* m_dev;    /* or NULL: actual rows (M = capacity; col_sum / col_sumsq hold CWN_STAT_ROWS(M) bands, the first
                                CWN_STAT_ROWS(*m_dev) are written) */
    double* stat_slots;      /* or NULL: the statistics of Y go HERE ([CWN_BN_SLOTS][2][F], see cwn_bn_live) instead of col_sum */
    cwn_bn_live in_bn;       /* .slots != NULL: the prologue of X is this BatchNorm, derived in the kernel (in_scale / in_shift
                                must then be NULL) */
    cwn_bn_live in_bn2;      /* the same for X2 */
} cwn_stage_desc;
int cwn_dense_stage_f32(const cwn_stage_desc* descs_host, int n, int32_t F, cwn_stream_t stream);
/* ... with a THIRD and FOURTH K-block per product (ABI 19): Z = prologue([X | X2 | X3 | X4]) W^T + b -- the combine network of a
 * CIN++ layer, Linear(3F -> F) over cat(up, down, boundaries) (mp/layers.py:260, 408-410), 4F with the co-boundary stream.
 * extras_host[2 i], extras_host[2 i + 1] belong to descs_host[i] (X NULL: block absent; the fourth needs the third, both
 * need X2); a block's prologue is ReLU (`relu`) behind either nothing or a live BatchNorm (`bn.slots` != NULL, as
 * cwn_stage_desc.in_bn).  w_packed: cwn_update_mlp_pack_weights_many_f32 of W[:, 2F:3F] / W[:, 3F:4F].  extras_host NULL =
 * cwn_dense_stage_f32.  (A kernel of its own: the two-block launches keep their argument block and their code.) */
typedef struct cwn_stage_extra {
    const float* X;          /* [M, F] or NULL */
    const void* w_packed;
    int64_t ldx;
    int32_t relu;
    int32_t pad_;
    cwn_bn_live bn;
} cwn_stage_extra;
int cwn_dense_stage_ex_f32(const cwn_stage_desc* descs_host, const cwn_stage_extra* extras_host, int n, int32_t F,
                           cwn_stream_t stream);
/* ... and BACKWARD (autograd of the Linear / BatchNorm1d(train) / ReLU modules of mp/layers.py:303-325): dX = dz W (two halves
 * dX, dX2 for a Linear(2F -> F)) with the BatchNorm(train) + ReLU backward as the
 * prologue -- dz = scale * (dyh - s1 / M - xhat * s2 / M), dyh = dy * [z * scale + shift > 0]; scale NULL: dz = dy * [z > 0]
 * when relu, dy otherwise -- given the column sums s1, s2 of cwn_norm_bwd_reduce_f32.  dz is also written (the
 * weight-gradient GEMM reads it; NULL: not wanted); acc1 / acc2 (or NULL): beta.grad += s1, gamma.grad += s2, once.
 * wt_packed / wt2_packed: cwn_update_mlp_pack_weights_t_many_f32 of the block W[:, :F] / W[:, F:] (torch Linear layout).
 * Every acc1 / acc2 / s1 / s2 / constant pointer 16-B aligned. */
int cwn_update_mlp_pack_weights_t_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out, int32_t n,
                                           cwn_stream_t stream);
/* Both forms of the same n blocks in ONE launch (a training step re-packs every weight after the optimizer: one launch
 * instead of two): out[e] as cwn_update_mlp_pack_weights_many_f32, out_t[e] as cwn_update_mlp_pack_weights_t_many_f32. */
int cwn_update_mlp_pack_weights_both_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out,
                                              void* const* out_t, int32_t n, cwn_stream_t stream);
typedef struct cwn_stage_bwd_desc {
    const float* dy;         /* [M, F] */
    const float* z;          /* [M, F] */
    float* dz;               /* [M, F] or NULL */
    const float* scale;      /* [F] each, or all NULL (no norm) */
    const float* shift;
    const float* mean;
    const float* rstd;
    float* s1;               /* (read; written when s_slots != NULL) */
    float* s2;
    float* acc1;
    float* acc2;
    const void* wt_packed;
    const void* wt2_packed;  /* NULL: one product */
    float* dx;               /* [M, F] */
    float* dx2;              /* [M, F] or NULL */
    int64_t M, lddy, ldz, lddz, lddx, lddx2;
    int32_t relu;
    int32_t pad_;
    const int64_t* m_dev;    /* or NULL: actual rows (M = capacity) */
    /* The REDUCE half of the BatchNorm backward without a launch of its own (round 4; the backward twin of cwn_bn_live).
     * Producer side: dx (dx2) of this launch is the dy of an earlier Linear -> BatchNorm -> ReLU stage whose pre-normalisation
     * values are out_bn.z (out_bn2.z): the epilogue forms dyh = dx * [z * scale + shift > 0] and adds the workgroup's column
     * sums of dyh and dyh * xhat into out_bn.slots ([CWN_BN_SLOTS][2][F] fp32, ZERO on entry; workgroup b into slot
     * b % CWN_BN_SLOTS, one coalesced atomic instruction per 64 columns) -- what cwn_norm_bwd_reduce_f32 would have summed.
     * Consumer side: s_slots != NULL: s1 / s2 of THIS stage are the sums over those slots (slot order), taken in the prologue;
     * the first workgroup of the descriptor also stores them to s1 / s2 (then outputs: d beta, d gamma for the caller). */
    const float* s_slots;
    cwn_bn_bwd_live out_bn;
    cwn_bn_bwd_live out_bn2;
} cwn_stage_bwd_desc;
int cwn_dense_stage_bwd_f32(const cwn_stage_bwd_desc* descs_host, int n, int32_t F, cwn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense parts of the path on the matrix cores (fp32 MFMA, exact fp32):
 *
 *     Y = epilogue( prologue([X | X2]) . W^T + bias )         up to CWN_MAX_DESCS GEMMs per launch
 *
 * Replaces the torch.nn.Linear (+ BatchNorm1d + ReLU) calls of the message / update / combine
 * networks: msg_up_nn's Linear(2F->F) (mp/layers.py:290-293, evaluated as Y1 = X_d W[:, :F]^T + b
 * and Y2 = X_{d+1} W[:, F:]^T), update_up_nn / update_boundaries_nn (:303-321), combine_nn
 * (:322-325, whose torch.cat (:199) becomes the K-concatenation [X | X2]).
 *   X [M, K] (row stride ldx), X2 [M, K2] optional, W [N, K + K2] (row stride ldw, torch Linear
 *   layout), Y [M, N] (row stride ldy).
 *   With w_trans != 0, W is [K + K2, N] (row stride ldw) and the product is [X | X2] . W: the
 *   input-gradient GEMM dX = dY . W of a Linear layer reads the layer's own weight, no transposed
 *   copy.
 *   prologue (optional): x <- x * in_scale[k] + in_shift[k] for X (in_scale2 / in_shift2 for X2),
 *                        then ReLU on X if in_relu & 1, on X2 if in_relu & 2 (normalisation +
 *                        activation of the producing layer applied on the fly);
 *   epilogue: + bias[n]; per-column sum / sum of squares of that value over every 32-row band
 *             of Y, written (fp64, plain stores: no atomics, nothing to zero, deterministic) to
 *             col_sum / col_sumsq [CWN_STAT_ROWS(M), N] -- BatchNorm batch statistics, summed over
 *             the bands by cwn_bn_finalize_f32;
 *             then * out_scale[n] + out_shift[n] (BatchNorm eval), then ReLU if relu.
 * ------------------------------------------------------------------------------------------ */
/* bands of 32 rows the statistics epilogue writes one partial sum for */
#define CWN_STAT_ROWS(M) (((M) + 31) / 32)

/* Optional extension of a TRANSPOSED-weight descriptor (the input-gradient GEMM of the training step): the GEMM's input
 * is not X itself but the BatchNorm + ReLU backward of it,
 *     dyh = X * [act'(z * scale + shift)],   dz = scale * (dyh - s1 / M - xhat * s2 / M),   xhat = (z - mean) * rstd
 * (dz = dyh when scale == NULL), formed while the tile is staged -- what cwn_norm_bwd_apply_f32 computes, without its
 * launch and without reading dz back.  dz is also written to `dz` (the weight-gradient GEMM needs it), and the sums are
 * handed on to acc1 / acc2 (beta.grad / gamma.grad) by the descriptor's first workgroup.  s1 / s2 are complete when the
 * launch starts (cwn_norm_bwd_reduce_f32, or the epilogue of the previous launch). */
typedef struct cwn_gemm_bnb {
    const float* z;      /* [M, K] pre-normalisation values of the stage, row stride ldz (16-byte aligned) */
    float* dz;           /* [M, K] out, row stride lddz, or NULL (a second descriptor over the same input) */
    const float* scale;  /* [K] or NULL: identity normalisation */
    const float* shift;
    const float* mean;
    const float* rstd;
    const float* s1;     /* [K] sum_m dyh */
    const float* s2;     /* [K] sum_m dyh * xhat */
    float* acc1;         /* or NULL: acc1[k] += s1[k] */
    float* acc2;         /* or NULL: acc2[k] += s2[k] */
    int64_t ldz, lddz;
    int32_t relu;        /* activation of the stage: ReLU (1) or identity (0) */
    int32_t pad_;
} cwn_gemm_bnb;

typedef struct cwn_gemm_desc {
    const float* X;
    const float* X2;        /* or NULL */
    const float* W;
    const float* bias;      /* [N] or NULL */
    const float* in_scale;  /* [K] or NULL */
    const float* in_shift;  /* [K] or NULL */
    const float* in_scale2; /* [K2] or NULL */
    const float* in_shift2; /* [K2] or NULL */
    const float* out_scale; /* [N] or NULL */
    const float* out_shift; /* [N] or NULL */
    double* col_sum;        /* [CWN_STAT_ROWS(M), N] or NULL (fp64: var = E[y^2] - mean^2 is safe) */
    double* col_sumsq;      /* [CWN_STAT_ROWS(M), N] or NULL */
    float* Y;
    int64_t M;
    int64_t ldx, ldx2, ldw, ldy;
    int32_t N, K, K2;
    int32_t relu, in_relu;
    int32_t w_trans;
    int32_t flags;          /* CWN_GEMM_* bits, 0: none */
    int32_t pad_;
    const cwn_gemm_bnb* bnb;  /* HOST pointer or NULL, see the struct above: w_trans launches with 16-byte aligned operands and
                               K <= 128 (the launch's tile shapes for K <= 128: any N); CWN_ERR_BAD_ARG otherwise */
    const int64_t* m_dev;     /* (ABI 23) or NULL: the rows that exist (M = capacity, see "Conventions"): the workgroups walk the
                               * row tiles below *m_dev only -- a static batch's message products used to be taken over the
                               * CAPACITY of its buffers (REDDIT-32: 1.46 x the batch).  Not with col_sum / col_sumsq or bnb
                               * (CWN_ERR_BAD_ARG): those launches have cwn_dense_stage_f32's own count */
} cwn_gemm_desc;

/* cwn_gemm_desc.flags.  EXACT: keep this launch on the exact fp32-MFMA kernel (bitwise an fmaf chain
 * per output element; inf / NaN inputs propagate as in fp32) even when it is eligible for the
 * bf16-split path below.  Per call: the library keeps no precision state. */
#define CWN_GEMM_EXACT 1
/* W_PACKED: `W` of this descriptor is not the fp32 weight but the buffer cwn_gemm_pack_weights_f32 made of
 * it ([128, 128] weights only; ldw is ignored).  Valid only in launches that run on the bf16-split path
 * (cwn_gemm_would_split; otherwise CWN_ERR_BAD_ARG); results are bit-identical to passing the fp32 weight.
 * Weight preparation for inference: one small launch per weight VERSION instead of a split of the same
 * numbers in every workgroup of every launch. */
#define CWN_GEMM_W_PACKED 2
/* ADD_OUT: Y += the product (Y initialised by the caller; THIS descriptor is the only writer of Y while the launch
 * runs) instead of Y = the product -- the backward of a propagate step adds dX = dY W onto the gradient its aggregation
 * kernel has already written for the same cells, where the framework launched an add kernel per matrix.  A plain
 * read-modify-write: fp32 atomics here cost 11 us per launch at the ZINC batch.  Exact-kernel only. */
#define CWN_GEMM_ADD_OUT 4
size_t cwn_gemm_packed_weight_bytes(void);
int cwn_gemm_pack_weights_f32(const float* W, int64_t ldw, void* out, cwn_stream_t stream);

int cwn_gemm_f32(const cwn_gemm_desc* descs_host, int n, cwn_stream_t stream);

/* Launches whose every descriptor has N == 128, K == 128, K2 == 0, no prologue, no statistics, the
 * natural weight layout, 16-B aligned operands and no CWN_GEMM_EXACT flag run on the BF16 matrix pipe
 * (16x the fp32-MFMA rate) through an exact three-way split of both operands, x = hi + mid + lo
 * (round-to-nearest pieces, csrc/cwn_split.h), keeping six of the nine partial products (dropped:
 * <= 2^-26 |x||w|): fp32 accuracy (measured max error 1-4e-7 of |x|.|w|, the same as the fp32-MFMA
 * kernel) at 2.1x the speed at scale, but not bit-identical to an fmaf chain, and non-finite inputs
 * give NaN.  1 when cwn_gemm_f32 would run these descriptors on that path, else 0 (a pure function
 * of its arguments). */
int cwn_gemm_would_split(const cwn_gemm_desc* descs_host, int n);

/* ------------------------------------------------------------------------------------------
 * Training-mode pieces of the dense networks (torch.nn.BatchNorm1d in train mode + ReLU between
 * the Linear layers of update_up_nn / update_boundaries_nn / combine_nn, mp/layers.py:303-325,
 * and their backward pass).  The forward never materialises a normalised activation between two
 * Linear layers: cwn_gemm_f32 accumulates the batch statistics of its output in the epilogue,
 * cwn_bn_finalize_f32 turns them into a per-column affine, and the NEXT cwn_gemm_f32 applies
 * affine + ReLU in its prologue.
 * ------------------------------------------------------------------------------------------ */
#define CWN_MAX_NORM_DESCS 16

typedef struct cwn_bn_desc {
    const double* col_sum;    /* [CWN_STAT_ROWS(M), N] band sums of y        (cwn_gemm_f32 epilogue) */
    const double* col_sumsq;  /* [CWN_STAT_ROWS(M), N] band sums of y^2 */
    const float* gamma;       /* [N] or NULL (= 1) */
    const float* beta;        /* [N] or NULL (= 0) */
    float* running_mean;      /* [N] or NULL: <- (1 - momentum) * running + momentum * mean */
    float* running_var;       /* [N] or NULL: same with the UNBIASED batch variance (torch semantics) */
    float* scale;             /* [N] out: gamma * rstd */
    float* shift;             /* [N] out: beta - mean * scale */
    float* mean;              /* [N] out */
    float* rstd;              /* [N] out: 1 / sqrt(biased var + eps) */
    int64_t M;                /* rows the statistics were taken over */
    int32_t N;
    float eps;
    float momentum;
    int32_t pad_;
    int64_t* num_batches_tracked; /* or NULL: += 1 (BatchNorm1d's counter; one launch less per layer than a framework add) */
    float* bwd_sums;          /* or NULL: [2, N] set to 0 -- the s1 / s2 scratch of this stage's backward reduce, cleared
                                 here so that the backward pass launches no fill */
    const int64_t* m_dev;     /* or NULL: actual rows the statistics were taken over (M = capacity: the band buffers hold
                                 CWN_STAT_ROWS(M) bands, the first CWN_STAT_ROWS(*m_dev) are summed).  *m_dev < 1: the affine
                                 is the identity-statistics one (mean 0, var 0) and the running statistics are left alone */
} cwn_bn_desc;

/* One launch for up to CWN_MAX_NORM_DESCS normalisations. */
int cwn_bn_finalize_f32(const cwn_bn_desc* descs_host, int n, cwn_stream_t stream);

typedef struct cwn_norm_desc {
    const float* dy;     /* [M, N] gradient w.r.t. the activation output (backward), row stride lddy */
    const float* z;      /* [M, N] pre-normalisation values, row stride ldz */
    const float* scale;  /* [N] or NULL: identity normalisation (activation only) */
    const float* shift;  /* [N] or NULL */
    const float* mean;   /* [N]  (backward, when scale != NULL) */
    const float* rstd;   /* [N] */
    float* s1;           /* [N] sum_m dyh            reduce: accumulated (caller zeroes); apply: read */
    float* s2;           /* [N] sum_m dyh * xhat */
    float* out;          /* [M, N] row stride ldout: activation (forward) or dz (backward apply) */
    int64_t M;
    int64_t lddy, ldz, ldout;
    int32_t N;
    int32_t relu;        /* activation: ReLU (1) or identity (0) */
    float* acc1;         /* backward apply only, or NULL: acc1[n] += s1[n]  (beta.grad: d beta = s1) */
    float* acc2;         /* backward apply only, or NULL: acc2[n] += s2[n]  (gamma.grad: d gamma = s2); one writer per column */
    const int64_t* m_dev; /* or NULL: actual rows (M = capacity) */
    cwn_bn_live bn;       /* cwn_norm_act_f32 only; .slots != NULL: scale / shift are derived in the kernel (both NULL here) */
    cwn_dropout drop;     /* cwn_norm_act_f32: out = dropout(act(..)) (the conv layer's output dropout, mp/molec_models.py:298-300);
                           * cwn_norm_bwd_reduce_f32: dy is the gradient w.r.t. the DROPPED activation -- it is multiplied by the
                           * same multipliers on the way in, and ...  (.state == NULL: unused) */
    float* dy_out;        /* ... cwn_norm_bwd_reduce_f32 only, or NULL: the multiplied dy written here, row stride lddy_out (what
                           * the launches behind the reduce read instead of dy) */
    int64_t lddy_out;
} cwn_norm_desc;

/* out = act(z * scale + shift)                                    (the last stage's output) */
int cwn_norm_act_f32(const cwn_norm_desc* descs_host, int n, cwn_stream_t stream);
/* dyh = dy * [act'(z * scale + shift)];  s1 += sum dyh;  s2 += sum dyh * (z - mean) * rstd */
int cwn_norm_bwd_reduce_f32(const cwn_norm_desc* descs_host, int n, cwn_stream_t stream);
/* out = scale * (dyh - s1 / M - xhat * s2 / M)      (out = dyh when scale == NULL) */
int cwn_norm_bwd_apply_f32(const cwn_norm_desc* descs_host, int n, cwn_stream_t stream);
/* Both in ONE launch for matrices of at most CWN_NORM_BWD_FUSED_MAX_ROWS rows (the ZINC / MOLHIV batches): a workgroup
 * owns four columns of one matrix over ALL its rows, so the column sums never leave it -- no atomics (the sums are
 * bit-reproducible), no zeroed s1 / s2, nothing between the reduction and the apply but a workgroup barrier.
 * s1 / s2 are WRITTEN (accumulate == 0) or ADDED TO (accumulate != 0: they may be the .grad of beta / gamma);
 * when scale == NULL they are not touched.  Needs 16-byte aligned operands and strides (CWN_ERR_ALIGN otherwise).
 * SLOWER than reduce + apply at the ZINC batch (28 us against 12: every workgroup touches 16 B of every line); the
 * form for bit-reproducible BatchNorm gradients, not the default. */
#define CWN_NORM_BWD_FUSED_MAX_ROWS 4096
int cwn_norm_bwd_f32(const cwn_norm_desc* descs_host, int n, int accumulate, cwn_stream_t stream);

/* Weight gradient of a Linear layer on the matrix cores, accumulated:
 *     dW[n, k] += sum_m dZ[m, n] * prologue([X | X2])[m, k]          db[n] += sum_m dZ[m, n]
 * dW is [N, K + K2] (row stride lddw, torch Linear layout) and is ADDED to (the M rows are split
 * over workgroups whose partial tiles are then summed), so it can be the .grad buffer itself.
 * The prologue is the one of cwn_gemm_f32 (normalisation + ReLU of the producing layer). */
#define CWN_GEMM_TN_MAX_DESCS 24   /* weight gradients per launch (the other batched calls take CWN_MAX_DESCS) */
typedef struct cwn_gemm_tn_desc {
    const float* dZ;         /* [M, N] row stride lddz */
    const float* X;          /* [M, K] row stride ldx */
    const float* X2;         /* [M, K2] or NULL */
    const float* in_scale;   /* [K] or NULL */
    const float* in_shift;
    const float* in_scale2;  /* [K2] or NULL */
    const float* in_shift2;
    float* dW;               /* [N, K + K2] accumulated */
    float* db;               /* [N] accumulated, or NULL */
    int64_t M;
    int64_t lddz, ldx, ldx2, lddw;
    int32_t N, K, K2;
    int32_t in_relu;         /* bit 0: X, bit 1: X2 */
    const int64_t* m_dev;    /* or NULL: actual rows of the reduction (M = capacity: grid, workspace layout) */
} cwn_gemm_tn_desc;

/* With a workspace of cwn_gemm_tn_workspace_bytes() the row bands are combined in a fixed order by
 * a second small launch (deterministic); with workspace == NULL they are added with fp32 atomics. */
size_t cwn_gemm_tn_workspace_bytes(const cwn_gemm_tn_desc* descs_host, int n);
int cwn_gemm_tn_f32(const cwn_gemm_tn_desc* descs_host, int n, void* workspace, size_t workspace_bytes,
                    cwn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Device-side batching (collate): build the arrays of a ComplexBatch from a dataset that is
 * resident in HBM in packed form, with ONE launch.
 *
 * Replaces the CPU collate of the reference: CochainBatch.from_cochain_list /
 * ComplexBatch.from_complex_list (data/complex.py:323-458, 690-728) -- per key, concatenate the
 * selected complexes' tensors and add the running cell offsets of data/complex.py:148-169.
 * Every output array is a concatenation of `n_seg` segments (one per selected complex):
 *     out[r][dst_start[s] + q] = op( src[r][src_start[s] + q], add[r][s], s )   0 <= q < len(s)
 * with len(s) = dst_start[s+1] - dst_start[s], r < n_rows (2 for [2,E] index tensors).
 *   CWN_COLLATE_COPY32 / COPY64  plain copy of 4- / 8-byte elements (features, labels)
 *   CWN_COLLATE_ADD64            int64 element + add[r][s]            (index tensors)
 *   CWN_COLLATE_SEGID64          int64 segment number s               (the `batch` vector)
 *   CWN_COLLATE_ADD32            int32 element + (int32) add[r][s]    (CSR arrays, below)
 * Tables (dst_start [n_seg+1], src_start [n_seg], add [n_rows][n_seg]) are int64 device arrays.
 *
 * The CSR of a batch is the concatenation of its complexes' CSRs (the adjacency is block-diagonal and the cells of a
 * complex are contiguous, data/complex.py:148-169): `col` arrays collate like an index row (ADD32 with the source
 * dimension's cell offset), and a complex's row pointers WITHOUT their leading zero (n_c numbers) collate to
 * rowptr + 1 with the running entry count as `add` -- rowptr[0] = 0 is written once.  A packed dataset that keeps the
 * per-complex CSRs of its boundary adjacencies needs no cwn_csr_build per batch (cwn_amd/packed.py).
 * ------------------------------------------------------------------------------------------ */
enum { CWN_COLLATE_COPY32 = 0, CWN_COLLATE_COPY64 = 1, CWN_COLLATE_ADD64 = 2, CWN_COLLATE_SEGID64 = 3,
       CWN_COLLATE_ADD32 = 4   /* int32 element + (int32) add[r][s]: the CSR arrays of a block-diagonal adjacency */ };
#define CWN_MAX_COLLATE_DESCS 32

typedef struct cwn_collate_desc {
    const void* src;          /* row r starts at src + r * src_row_stride elements */
    void* dst;                /* row r starts at dst + r * dst_row_stride elements */
    const int64_t* dst_start; /* [n_seg + 1] */
    const int64_t* src_start; /* [n_seg] */
    const int64_t* add;       /* [n_rows * n_seg] or NULL */
    int64_t src_row_stride;
    int64_t dst_row_stride;
    int32_t n_rows;
    int32_t op;
} cwn_collate_desc;

int cwn_collate(const cwn_collate_desc* descs_host, int n, int64_t n_seg, cwn_stream_t stream);
/* The same launch for n_slots batches at once (blockIdx.z = slot): slot j writes dst + j * dst_slot_bytes[i] (host array [n]) of
 * every descriptor i and reads its tables at dst_start / src_start / add + j * table_slot_stride (int64 elements) -- the output
 * arrays and the tables of a multi-slot static batch are [n_slots, ...] tensors (cwn_amd/static_batch.py: the fill of several
 * steps of a captured graph in one launch).  cursor (device int64 or NULL): += n_slots, by one thread of the launch (the
 * tables it used were cut by an earlier launch, cwn_collate_tables, which only reads the cursor). */
int cwn_collate_slots(const cwn_collate_desc* descs_host, int n, int64_t n_seg, int32_t n_slots, int64_t table_slot_stride,
                      const int64_t* dst_slot_bytes_host, int64_t* cursor, cwn_stream_t stream);

/* The segment tables of a batch, built ON THE DEVICE from the per-complex metadata of a packed dataset (what
 * cwn_amd/packed.py's host path computes with numpy and uploads: the reference does it in CochainBatch.from_cochain_list,
 * data/complex.py:323-458).  With it a step needs nothing from the host but the batch's complex numbers -- or, with
 * `cursor`, nothing at all: an epoch's permutation is uploaded once and every replay of a captured step takes the next
 * batch (exp/train_utils.py:35: `for batch in train_loader`).
 *   meta   device int64 [num][W], W = 3 D + 3 K, one row per complex of the dataset:
 *            [3 d + 0 / 1 / 2]  cells of dimension d / cells below / cells above (the three running offsets of
 *                               data/complex.py:148-169)
 *            [3 D + k] length, [3 D + K + k] start, [3 D + 2 K + k] has      of key k (one array of the packed dataset)
 *   idx    device int64 [n_batches][B]: the complexes of the batches in order; a negative entry = no complex (a short last
 *          batch); an entry >= num sets bit 1 of *err_flag and counts as absent
 *   cursor device int64 or NULL (= 0), READ only: workgroup j of the launch cuts the tables of batch *cursor + j into
 *          tables + j * slot_stride, j < n_slots (cwn_collate_slots advances the cursor); a batch past n_batches is an empty
 *          batch and sets bit 1 of *err_flag
 *   tables (out) device int64 [cwn_collate_tables_len(D, K, B)]:
 *            dst   [K][B + 1]  at 0               dst[k][s] = sum_{s' < s} length_k(idx[s'])   (`__slices__`, :349-394)
 *            src   [K][B]      at K (B + 1)       start_k(idx[s])
 *            off   [D][5][B]   then               exclusive sums of (cells, cells, below, cells, above) of dimension d
 *            seg   [D][B + 1]  then               exclusive sums of the cells of dimension d, the total last (`ptr`, :344, 432)
 *            sizes [8 + K]     then               [d] cells of dimension d (0 beyond D; d < 3), [3] complexes in the batch,
 *                                                 [4..7] 0, [8 + k] total length of key k  -- what `m_dev` fields point at
 * One workgroup per slot; a few microseconds. */
size_t cwn_collate_tables_len(int32_t D, int32_t K, int64_t B);      /* int64 elements */
int cwn_collate_tables(const int64_t* meta, int64_t num, int32_t D, int32_t K, const int64_t* idx, int64_t B, int64_t n_batches,
                       const int64_t* cursor, int32_t n_slots, int64_t slot_stride, int64_t* tables, int32_t* err_flag,
                       cwn_stream_t stream);

/* The capacity guard of a static batch, between cwn_collate_tables and cwn_collate_slots: caps = device int64 [D + K], the
 * cells the feature / batch arrays of dimension d hold, then the elements the array of key k holds.  A slot whose scanned
 * totals exceed any of them has its tables zeroed -- it becomes a batch WITHOUT complexes: the collate launch writes nothing
 * for it, every m_dev row count reads 0, its training step changes nothing -- and CWN_ERR_BIT_CAPACITY is set in the sticky
 * word.  (The reference's collate allocates per batch, data/complex.py:323-458: there is no such failure there; a fixed-capacity
 * buffer must refuse instead of overrunning.) */
#define CWN_ERR_BIT_CAPACITY 32               /* *err_flag bit: a batch beyond the capacity of its static buffers was dropped */
int cwn_collate_guard(int64_t* tables, int32_t D, int32_t K, int64_t B, int32_t n_slots, int64_t slot_stride, const int64_t* caps,
                      int32_t* err_flag, cwn_stream_t stream);

/* Embedding lookup with a sum over index columns (torch.nn.Embedding for cols = 1; the OGB
 * Atom/BondEncoder sum over one table per integer feature column, mp/molec_models.py:44-52, 237-245):
 *     out[r, :] = sum_c W[col_off[c] + src[r, c], :]
 * W is the row-wise concatenation of the tables ([V, H], H % 4 == 0); col_off / col_size are device
 * arrays [cols] (both NULL for one table of V rows).  An index outside its OWN table sets bit 1 of
 * *err_flag (the sticky word of cwn_csr_build; raised as IndexError by the host) and contributes
 * nothing.  Columns are added in order: bit-identical to summing the per-column lookups. */
int cwn_embedding_fwd_f32(const float* W, const int64_t* src, const int64_t* col_off,
                          const int64_t* col_size, float* out, int64_t n_rows, int32_t cols, int32_t H,
                          int64_t V, int32_t* err_flag, cwn_stream_t stream);

/* Its backward: dW[col_off[c] + src[r, c], :] += g[r, :].  dW is [V, H], accumulated (the caller zeroes it).  H = 64 / 128 /
 * 256: a band of 64 cells is staged in LDS and the distinct values of every column are walked with ballots -- one partial per
 * (table row, feature) and band goes to dW with an fp32 atomic, no float atomics in LDS.  Other widths: every workgroup
 * accumulates its band into a private copy of the whole table in LDS, which must fit (V * H * 4 <= 60 KiB; CWN_ERR_TOO_LARGE
 * otherwise: callers then use the transposed aggregation). */
/* src_f32 != 0: `src` holds the integer features as float32 (as the containers deliver them; truncated like the front's);
 * n_dev (or NULL): device int64 = the actual number of rows (n_rows = capacity). */
int cwn_embedding_bwd_f32(const float* g, const void* src, const int64_t* col_off,
                          const int64_t* col_size, float* dW, int64_t n_rows, int32_t cols, int32_t H,
                          int64_t V, int32_t src_f32, const int64_t* n_dev, cwn_stream_t stream);

/* The backward of cwn_embed_front_f32 (EmbedVEWithReduce, mp/layers.py:490-593: vertex / edge embeddings + InitReduceConv
 * twice) in ONE launch, for one vertex table and at most one edge table with one integer feature each (the ZINC models):
 *     dv[v]    = g0[v] + sum_{edges e of v} t1[e],   t1[e] = (g1[e] when the edges have no table: x1 = red1)
 *                                                            + (halve ? 1/2 : 1) sum_{rings r of e} g2[r]
 *     dWv[type(v)] += dv[v]         dWe[type(e)] += g1[e]
 * A workgroup owns a band of 32 vertices (it gathers their dv into LDS: every incident edge, every ring of that edge) or of
 * 32 edges, then adds the band to the table rows by ballot, as cwn_embedding_bwd_f32's one-table form.  Replaces the
 * halving multiply, two transposed aggregations and two table-gradient launches of a training step.
 * rowptr1 / col1: CSR of the TRANSPOSED boundary adjacency of dimension 1 (per vertex its edges), rowptr2 / col2 of
 * dimension 2 (per edge its rings) -- int32, from cwn_csr_build or the collate; either may be NULL (no such cells).
 * H = 64 / 128 / 256; Vv, Ve <= 64; pointers 16-B aligned; dWv / dWe are ADDED to (fp32 atomics, one per band and
 * touched element). */
typedef struct cwn_front_bwd {
    const float* g0;         /* [n0, H] or NULL (= 0) */
    const float* g1;         /* [n1, H] or NULL */
    const float* g2;         /* [n2, H] or NULL */
    const int32_t* rowptr1;  /* [n0 + 1] */
    const int32_t* col1;
    const int32_t* rowptr2;  /* [n1 + 1] */
    const int32_t* col2;
    const void* v_src;       /* [n0] int64 or float32 */
    const void* e_src;       /* [n1], or NULL: no edge table (g1 then flows into the vertices) */
    float* dWv;              /* [Vv, H] */
    float* dWe;              /* [Ve, H] or NULL */
    int64_t n0, n1;          /* capacities */
    const int64_t* n0_dev;   /* or NULL: actual rows */
    const int64_t* n1_dev;
    int32_t H, Vv, Ve;
    int32_t src_f32;         /* bit 0: v_src is float32, bit 1: e_src is */
    int32_t halve;
    int32_t pad_;
} cwn_front_bwd;
int cwn_embed_front_bwd_f32(const cwn_front_bwd* args_host, cwn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The two ends of a model forward, one launch each (inference; csrc/cwn_ends.hip).
 *
 * FRONT -- EmbedVEWithReduce.forward / OGBEmbedVEWithReduce.forward (mp/layers.py:490-593), i.e.
 * v_embed_init / e_embed_init (mp/molec_models.py:44-52, 237-245) + InitReduceConv('sum') twice
 * (mp/layers.py:473-487, :526, :538-540):
 *     x0[v] = sum_c Tv_c[ids0[v, c]]
 *     x1[e] = sum_c Te_c[ids1[e, c]]            (e_tab == NULL: x1[e] = red1[e])
 *     x2[r] = (halve ? 1/2 : 1) * sum_{e in row r of CSR2} red1[e],   red1[e] = sum_{v in row e of CSR1} x0[v]
 * A table set is cwn_embedding_fwd_f32's (concatenated tables [V, H], per-column offsets / sizes, both NULL for
 * one table); `src` holds the integer features as int64 or -- as the reference's containers deliver them before
 * `.to(torch.long)` (mp/layers.py:556, :566) -- as float32 (src_is_f32: converted by truncation).  CSR1 / CSR2
 * are the destination-sorted plans cwn_csr_build makes of boundary_index_1 / boundary_index_2 (rowptr, col:
 * int32; nb1 / nb2 = their entry counts; a NULL pair or no entries = no reduction: zeros); sums run in CSR (= entry)
 * order.  H % 4 == 0; outputs [n_d, H]
 * contiguous, 16-B aligned.  An index outside its own table sets bit 1 of *err_flag and contributes nothing.
 * One launch when every table set is one table (torch.nn.Embedding: ZINC); table sets of several columns (the OGB
 * encoders, 9 + 3 at molhiv) take two on the same stream -- the embeddings of both cell types, then the rows that reduce,
 * from the x0 rows just written -- with the same results (x0 must not alias x1 / x2).
 * ------------------------------------------------------------------------------------------ */
typedef struct cwn_embed_table {
    const float* W;            /* [V, H]: the row-wise concatenation of the tables */
    const void* src;           /* [n_rows, cols] integer features: int64, or float32 when src_is_f32 */
    const int64_t* col_off;    /* device [cols] or NULL */
    const int64_t* col_size;   /* device [cols] or NULL (both or neither) */
    int64_t V;
    int32_t cols;
    int32_t src_is_f32;
} cwn_embed_table;

/* n_dev (or NULL): device int64 [3] = the ACTUAL n0, n1, n2 (the arguments are then capacities, see "Conventions"). */
int cwn_embed_front_f32(const cwn_embed_table* v_tab, int64_t n0, float* x0, const cwn_embed_table* e_tab,
                        int64_t n1, float* x1, const int32_t* rowptr1, const int32_t* col1, int64_t nb1, int64_t n2,
                        float* x2, const int32_t* rowptr2, const int32_t* col2, int64_t nb2, int32_t H, int32_t halve,
                        int32_t* err_flag, const int64_t* n_dev, cwn_stream_t stream);

/* HEAD -- pool_complex (mp/nn.py:50-60) + lin1s + final readout + lin2 (mp/molec_models.py:129-156,
 * mp/models.py:222-253), one workgroup per complex:
 *     pooled_d[c] = sum (mean_readout: mean, count clamped to 1) of rows cell_ptr_d[c] .. cell_ptr_d[c+1] of x_d
 *     out[c] = W2 (sum_d relu(W1_d pooled_d[c] + b1_d)  [/ n_dims when mean_final]) + b2
 * The cells of a complex are contiguous in a batched cochain (data/complex.py:148-169); cell_ptr_d is the
 * collate's `ptr` (data/complex.py:344, 432) as a device int64 [C + 1] array.  x == NULL: the dimension is
 * absent from the batch, pooled = 0 (mp/nn.py:55-56) and lin1 still contributes relu(b1).  w1t = the TRANSPOSE
 * of lin1s[d].weight, [K, H2] row-major (prepared once per weight version: coalesced over the outputs); w2
 * [O, H2] torch layout.  K % 4 == 0, K <= 2048, H2 % 4 == 0, H2 <= 512.  pooled_out (optional): [C, K].  fp32 FMA chains in a
 * fixed order; a complex's result does not depend on the rest of the batch.
 * ------------------------------------------------------------------------------------------ */
#define CWN_HEAD_MAX_DIMS 3
#define CWN_HEAD_MAX_PARTS 8
typedef struct cwn_head_dim {
    const float* x;             /* [n_cells, K], row stride ldx (multiple of 4), or NULL */
    const int64_t* cell_ptr;    /* device [C + 1] */
    const float* w1t;           /* [K, H2] */
    const float* b1;            /* [H2] or NULL */
    float* pooled_out;          /* [C, K] or NULL */
    float* h_out;               /* [C, H2] or NULL: W1_d pooled_d + b1_d BEFORE the ReLU (training: the backward's mask) */
    int64_t n_cells, ldx;
    /* Jumping knowledge, jump_mode 'cat' (mp/models.py:222-232: the readout of torch.cat(layer outputs, -1)) WITHOUT the
     * concatenation: n_parts > 1 -- the K columns are n_parts blocks of K / n_parts, block 0 read from x, block q from
     * x_more[q - 1], each [n_cells, K / n_parts] with row stride ldx (a layer's own output matrix).  0 / 1: x is [n_cells, K]. */
    const float* x_more[CWN_HEAD_MAX_PARTS - 1];
    int32_t n_parts;
    int32_t pad_;
} cwn_head_dim;

/* s_out (optional, training): [C, H2] the hidden vector lin2 multiplies (sum / mean over the dimensions).
 * drop (or NULL) + drop_pos: the head's dropout (`apply_dropout_before`, mp/molec_models.py:129-146): CWN_HEAD_DROP_LIN1 on
 * pooled_d (element (d C + c) K + k; pooled_out then holds the dropped vector -- what lin1 multiplies), CWN_HEAD_DROP_FINAL on
 * relu(h_d) before the sum over the dimensions (element (d C + c) H2 + j), CWN_HEAD_DROP_LIN2 on the summed hidden vector
 * (element c H2 + j; s_out holds the dropped vector). */
enum { CWN_HEAD_DROP_NONE = 0, CWN_HEAD_DROP_LIN1 = 1, CWN_HEAD_DROP_FINAL = 2, CWN_HEAD_DROP_LIN2 = 3 };
/* THE ORDER of the row sums: chunks of CWN_HEAD_CHUNK consecutive rows of a complex; inside a chunk 512 / (K / 4) row groups add
 * every (512 / (K / 4))-th row one after the other and are added in group order; chunk sums are added in chunk order -- the same
 * bits whichever launch forms them.  pool_partials / pool_split (round 5): LARGE complexes (REDDIT-like: thousands of cells per
 * complex, 32 complexes per batch -- one workgroup per complex pulled 4 MB through one CU): with pool_partials (device fp32,
 * at least cwn_head_pool_floats(...) floats) and pool_split > 1 a first launch writes the chunk sums -- one workgroup per
 * chunk of the whole batch (round 6; before: C x pool_split workgroups, a launch as long as the largest complex), plain
 * stores -- and the head launch adds them in chunk order instead of reading the rows.
 * NULL: one launch, which sums a large complex chunk by chunk itself (the same result). */
#define CWN_HEAD_CHUNK 128
int64_t cwn_head_pool_floats(const cwn_head_dim* dims_host, int n_dims, int64_t C, int32_t K);
int cwn_head_f32(const cwn_head_dim* dims_host, int n_dims, int64_t C, int32_t K, int32_t H2, int32_t mean_readout,
                 int32_t mean_final, const float* w2, const float* b2, int32_t O, float* out, float* s_out,
                 const cwn_dropout* drop, int32_t drop_pos, float* pool_partials, int64_t pool_partials_floats, int32_t pool_split,
                 cwn_stream_t stream);

/* Backward of the same head for the training step (exp/train_utils.py:62-73), one workgroup per complex, given
 * g_out = dL/dout [C, O] and what the forward left (h_out per dimension):
 *     ds = W2^T g_out [/ n_dims];  dh_d = ds . [h_d > 0]  (written to dh_out [C, H2]);
 *     dx_d[r] = W1_d^T dh_d  [/ cells of the complex]   for every row r of the complex   (dx == NULL: not wanted)
 * The weight gradients are sums over the complexes and go through cwn_gemm_tn_f32 on the [C, .] matrices:
 * dW1_d = dh_d^T pooled_d, db1_d = column sums of dh_d, dW2 = g_out^T s, db2 = column sums of g_out.
 * w1 is lin1s[d].weight in its own [H2, K] layout (16-B aligned); same size limits as the forward. */
typedef struct cwn_head_bwd_dim {
    const float* h;             /* [C, H2] pre-activation saved by the forward */
    const float* w1;            /* [H2, K] */
    const int64_t* cell_ptr;    /* device [C + 1] */
    float* dx;                  /* [n_cells, K], row stride lddx, or NULL */
    float* dh_out;              /* [C, H2] or NULL */
    int64_t n_cells, lddx;
    float* dx_more[CWN_HEAD_MAX_PARTS - 1];   /* n_parts > 1: block q of dpooled goes to the rows of dx (q = 0) / dx_more[q - 1], each */
    int32_t n_parts;                          /* [n_cells, K / n_parts] with row stride lddx (cwn_head_dim.x_more) */
    int32_t pad_;
} cwn_head_bwd_dim;

/* drop / drop_pos: the forward's (the same multipliers are re-derived: ds, dh_d or dpooled_d is multiplied where the forward
 * multiplied the value). */
/* row_split = P >= 1: C x P workgroups, every one of a complex derives ds / dh / dpooled (a few thousand FMAs) and writes ITS
 * chunk of the complex's rows (the broadcast of dpooled over thousands of cells is what a large complex costs). */
int cwn_head_bwd_f32(const cwn_head_bwd_dim* dims_host, int n_dims, int64_t C, int32_t K, int32_t H2, int32_t mean_readout,
                     int32_t mean_final, const float* w2, int32_t O, const float* g_out, const cwn_dropout* drop,
                     int32_t drop_pos, int32_t row_split, cwn_stream_t stream);

/* The loss of a training step and its gradient in one launch (exp/train_utils.py:62-73 with the elementwise-mean
 * criteria of :20-31): loss[0] = mean_i l(pred_i, y_i), grad_i = dl/dpred_i / n over n contiguous fp32 elements.
 * kinds: L1Loss, MSELoss, BCEWithLogitsLoss (torch's definitions, incl. sign(0) = 0 and the stable BCE form).
 * One workgroup, fixed reduction tree (deterministic); meant for the few hundred predictions of a batch. */
enum { CWN_LOSS_L1 = 0, CWN_LOSS_MSE = 1, CWN_LOSS_BCE_LOGITS = 2, CWN_LOSS_CE = 3 };
/* CWN_LOSS_CE (cwn_loss_cols_f32 only): torch.nn.CrossEntropyLoss() of exp/train_utils.py:21-22 -- pred = [rows, cols] logits and
 * `y` points at the rows' classes as int64 (not float); a class outside [0, cols) is ignored like torch's ignore_index. */
/* n_dev (or NULL): device int64 = the ACTUAL number of elements (n is then the capacity: grad[i] = 0 for i >= *n_dev).
 * A target that is NaN is a NULL label (exp/train_utils.py:64-66: `mask = ~torch.isnan(targets)`): it contributes no loss and
 * no gradient and does not count in the mean. */
int cwn_loss_f32(int32_t kind, const float* pred, const float* y, int64_t n, float* loss, float* grad,
                 const int64_t* n_dev, cwn_stream_t stream);
/* ... with `cols` predictions per complex (the multi-task ogbg-mol* heads: [complexes, tasks] row-major, n = capacity x cols):
 * *n_dev counts COMPLEXES, so the first *n_dev * cols elements are real. */
int cwn_loss_cols_f32(int32_t kind, const float* pred, const float* y, int64_t n, int64_t cols, float* loss, float* grad,
                      const int64_t* n_dev, cwn_stream_t stream);

/* The start of a training step in ONE launch (optimizer.zero_grad() of exp/train_utils.py:61 + what the step's own kernels
 * need zero on entry + the optimizer's step counter): a[0 .. a_bytes) = 0 (the flat gradient buffer), b[0 .. b_bytes) = 0 (the
 * step arena of cwn_amd/ops.py: slot sums of the live BatchNorms), and *step += 1 -- only when active == NULL or *active > 0
 * (cwn_adam_f32's convention: an empty batch of a static epoch is no step).  Pointers 16-B aligned, byte counts multiples of
 * 16; any of a / b / step may be NULL.  Replaces two fills and an add of the framework.  dropout_state (or NULL): the
 * {seed, step} record of cwn_dropout -- its step is advanced by one, unconditionally (a replayed step never repeats a mask). */
int cwn_step_begin(void* a, int64_t a_bytes, void* b, int64_t b_bytes, int32_t* step, const int64_t* active,
                   int64_t* dropout_state, cwn_stream_t stream);

/* y_i += (1 + *eps_i) * x_i for up to CWN_AXPY_MAX_DESCS vectors in ONE launch (ABI 24) -- the piece of a CIN++ layer's input
 * gradient that comes through its third output, out_down = zeros + (1 + eps2) x (mp/layers.py:253, lower stream off), added onto
 * the dx the blocked backward launch wrote: per dimension and layer the framework ran `1 + eps` (a launch over one element) and
 * an addcmul (24 launches of a ZINC CIN++ training step, 113 us of 1.15 ms).  eps: device float or NULL (= 0), read by the
 * launch (graph-capturable); n floats, n % 4 == 0, y / x 16-B aligned, y and x of one descriptor must not overlap.  The product
 * is rounded before the add (as the framework's addcmul). */
#define CWN_AXPY_MAX_DESCS 8
typedef struct cwn_axpy_desc {
    float* y;
    const float* x;
    const float* eps;
    int64_t n;
} cwn_axpy_desc;
int cwn_axpy_eps_f32(const cwn_axpy_desc* descs_host, int n, cwn_stream_t stream);

/* torch.optim.Adam's update (no amsgrad; weight_decay is the L2 form) for a whole model in one
 * launch: parameters p, gradients g and the moments m, v are each ONE contiguous fp32 buffer of n
 * elements (16-B aligned).  `step` is a device int32 holding the 1-based step number (the caller
 * increments it before the call), so the launch is graph-capturable.  Replaces optimizer.step() of
 * exp/train_utils.py:75 for the data-parallel training path. */
/* active (device int64 or NULL): when given and *active <= 0 the launch changes nothing -- the step of an EMPTY batch of a
 * static epoch (cwn_amd/static_batch.py: a replay holds a fixed number of steps, an epoch need not be a multiple of it); the
 * caller then also leaves `step` alone. */
int cwn_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, const int32_t* step, const int64_t* active, cwn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Graph -> 2-complex lifting on the HOST (integer preprocessing that produces the path's inputs;
 * the reference does it through graph-tool and gudhi):
 *   CWN_LIFT_RING    data/utils.py:400-498 compute_ring_2complex: 2-cells = chordless cycles with
 *                    3..max_k vertices (max_k is ignored by the clique lift);
 *   CWN_LIFT_CLIQUE  data/utils.py:224-272 compute_clique_complex_with_gudhi, expansion_dim 2.
 * `edges` is [n_edges, 2] (any orientation; sorted and de-duplicated internally).  Cells and
 * adjacency entries come in the reference's order (build_adj, data/utils.py:103-138).  Read the
 * results back with cwn_lift_size / cwn_lift_copy; index arrays are [2, L] row-major int64.
 * ------------------------------------------------------------------------------------------ */
typedef struct cwn_lift_s cwn_lift_t;
enum { CWN_LIFT_RING = 0, CWN_LIFT_CLIQUE = 1 };
enum {
    CWN_LIFT_EDGES = 0,      /* [E, 2]   vertices of edge e                                     */
    CWN_LIFT_CELLS2_PTR,     /* [C + 1]  offsets into CELLS2_VERTS                              */
    CWN_LIFT_CELLS2_VERTS,   /*          vertices of every 2-cell (cyclic order for rings)      */
    CWN_LIFT_UP0,            /* [2, L]   upper_index of the vertices                            */
    CWN_LIFT_COB0,           /* [L]      shared_coboundaries (edge ids)                         */
    CWN_LIFT_UP1,            /* [2, L]   upper_index of the edges                               */
    CWN_LIFT_COB1,           /* [L]      shared_coboundaries (2-cell ids)                       */
    CWN_LIFT_DOWN1,          /* [2, L]   lower_index of the edges        (include_down only)    */
    CWN_LIFT_BND1,           /* [L]      shared_boundaries (vertex ids)                         */
    CWN_LIFT_DOWN2,          /* [2, L]   lower_index of the 2-cells      (include_down only)    */
    CWN_LIFT_BND2,           /* [L]      shared_boundaries (edge ids)                           */
    CWN_LIFT_BINDEX1,        /* [2, 2E]  boundary_index of the edges                            */
    CWN_LIFT_BINDEX2,        /* [2, L]   boundary_index of the 2-cells                          */
    CWN_LIFT_N_ARRAYS
};
/* NULL on invalid input (vertex out of range, self loop, unknown kind). */
cwn_lift_t* cwn_lift_create(int kind, int64_t n_vertices, const int64_t* edges, int64_t n_edges,
                            int max_k, int include_down);
int64_t cwn_lift_size(const cwn_lift_t* lift, int which);          /* int64 elements, -1 on error */
int cwn_lift_copy(const cwn_lift_t* lift, int which, int64_t* out);
void cwn_lift_destroy(cwn_lift_t* lift);

/* A whole dataset at once (data/utils.py:501-560 convert_graph_dataset_with_rings / :275-297
 * convert_graph_dataset_with_gudhi lift graph by graph under joblib): graph g has n_vertices[g] vertices and the
 * edges edges[2 * edge_ptr[g] .. 2 * edge_ptr[g + 1]) (pairs of vertex ids LOCAL to the graph; edge_ptr[0] = 0), lifted
 * by up to n_threads host threads (<= 0: all hardware threads).  NULL on bad arguments or when any graph is invalid.
 * Results come back concatenated over the graphs, in graph order, ids still local to their graph:
 *   cwn_lift_many_lengths  per graph: the columns L_g of a [2, L_g] array, the elements of any other array;
 *   cwn_lift_many_copy     a [2, L_g] array as ONE [2, sum L_g] row-major array (row 0 of every graph, then row 1
 *                          of every graph), any other array end to end -- the layout of the HBM-resident packed
 *                          dataset (cwn_amd/packed.py), with no per-complex object in between. */
typedef struct cwn_lift_set_s cwn_lift_set_t;
cwn_lift_set_t* cwn_lift_many(int kind, int64_t n_graphs, const int64_t* n_vertices, const int64_t* edge_ptr,
                              const int64_t* edges, int max_k, int include_down, int n_threads);
int64_t cwn_lift_many_count(const cwn_lift_set_t* set);                       /* graphs, -1 on NULL */
int cwn_lift_many_lengths(const cwn_lift_set_t* set, int which, int64_t* out /* [n_graphs] */);
int cwn_lift_many_copy(const cwn_lift_set_t* set, int which, int64_t* out);
void cwn_lift_many_destroy(cwn_lift_set_t* set);

#ifdef __cplusplus
}
#endif
#endif /* CWN_HIP_H */
