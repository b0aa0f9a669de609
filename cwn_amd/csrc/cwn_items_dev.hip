// cwn_items_dev.hip -- the item tables of the complex-blocked launches (cwn_layer_fused_f32, cwn_layer_bwd_own_f32), built ON
// THE DEVICE from the device-resident prefix sums of a batch (the `ptr` / `__slices__` tables of data/complex.py:344-441, as
// cwn_collate_tables writes them).
//
// Why: the reference's training loop draws a new shuffled batch every step (data/data_loading.py:84-111,
// exp/train_utils.py:35-75).  With the table cut on the host (csrc/cwn_blockplan.cpp) every fresh batch costs a host pass
// over its per-complex sizes plus an upload -- 0.084 ms + H2D per batch measured in round 3, against a propagate step of
// 0.039 ms -- and a captured hipGraph cannot contain that upload at all.  Here the cut is two small launches inside the
// captured step: one workgroup per set, a thread per GROUP of consecutive complexes.
//
// The cut is simpler than the host's greedy one (which walks the complexes sequentially and tries several splits of the
// LDS): complexes are taken in groups of `group` consecutive ones; a group that fits the caps of a workgroup is one item,
// a group that does not is cut into its single complexes, a complex that does not fit alone sets CWN_ERR_BIT_UNFIT and gets
// no record.  Every record is what the host builder would write for the same range (same fields, same derived numbers:
// tests compare through cwn_layer_items_check and through the launches' outputs, which do not depend on the cut --
// tests/test_gpu_static.py).  Set s owns a FIXED region of the table; records past a set's own are zeroed (empty).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cwn_hip.h"
#include "cwn_layer_bwd_own.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kInts = CWN_LAYER_ITEM_INTS;
constexpr int kBInts = CWN_LAYER_BWD_ITEM_INTS;

__host__ __device__ inline int64_t pad16(int64_t n) { return (n + 15) / 16 * 16; }
__host__ __device__ inline int64_t pad4(int64_t n) { return (n + 3) / 4 * 4; }

struct DevSet { int32_t g, n_tasks, tasks[2]; };

struct ItemsArgs {
    const int64_t* n_complexes;
    int64_t cap_c;
    const int64_t* cell_ptr[CWN_LAYER_MAX_DIMS];
    const int64_t* up_ptr[CWN_LAYER_MAX_DIMS];
    const int64_t* b_ptr[CWN_LAYER_MAX_DIMS];
    int32_t* items;
    int32_t* err;
    DevSet sets[CWN_LAYER_MAX_DIMS];
    int32_t set_start[CWN_LAYER_MAX_DIMS + 1];
    int32_t n_sets, n_dims, F, variant, round_rows, group;
    int32_t has_up[CWN_LAYER_MAX_DIMS];
    int64_t row_cap, src_cap, lds_budget, half_cap, idx_bytes;
    int64_t bwd_lds;                 // backward table: the launch's dynamic LDS
    int64_t table_slot_stride;       // slots (blockIdx.y): int64 elements between the tables of consecutive slots ...
    int64_t items_slot_ints;         // ... and int32 elements between their item tables
};

// the arguments as slot `slot` sees them
__device__ __forceinline__ void to_slot(ItemsArgs& A, int slot) {
    const int64_t o = (int64_t)slot * A.table_slot_stride;
    A.n_complexes += o;
#pragma unroll
    for (int d = 0; d < CWN_LAYER_MAX_DIMS; ++d) {
        A.cell_ptr[d] += o;
        if (A.up_ptr[d] != nullptr) A.up_ptr[d] += o;
        if (A.b_ptr[d] != nullptr) A.b_ptr[d] += o;
    }
    A.items += (int64_t)slot * A.items_slot_ints;
}

// workgroup-wide exclusive scan of one small integer per thread; returns the thread's offset, *total = the sum
__device__ __forceinline__ int block_scan(int v, int* total, int* lds /* [2 * 16] */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wid] = x;
    __syncthreads();
    if (wid == 0) {
        int s = lane < kThreads / 64 ? lds[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int y = __shfl_up(s, o, 64);
            if (lane >= o) s += y;
        }
        if (lane < kThreads / 64) lds[16 + lane] = s;
    }
    __syncthreads();
    const int off = (wid == 0 ? 0 : lds[16 + wid - 1]) + x - v;
    *total = lds[16 + kThreads / 64 - 1];
    __syncthreads();
    return off;
}

__device__ __forceinline__ void store_record(int32_t* dst, const int32_t* r, int n_ints) {
    for (int k = 0; k < n_ints; k += 4) *reinterpret_cast<int4*>(dst + k) = make_int4(r[k], r[k + 1], r[k + 2], r[k + 3]);
}

// ---- forward table (record layout: include/cwn_hip.h; the arithmetic of cwn_blockplan.cpp: build_with) --------------------
struct FwdGeom {
    const ItemsArgs& A;
    const DevSet& S;
    __device__ int64_t cells(int d, int64_t a, int64_t b) const { return A.cell_ptr[d][b] - A.cell_ptr[d][a]; }
    __device__ static int64_t span(const int64_t* p, int64_t a, int64_t b) { return p != nullptr ? p[b] - p[a] : 0; }
    __device__ int64_t first_coface_row(int64_t n_g, int64_t n_c) const {
        const int64_t r1 = pad16(n_g);
        return n_c > 0 ? (r1 + A.round_rows - 1) / A.round_rows * A.round_rows : r1;
    }
    __device__ int64_t staged(int64_t n_g, int64_t n_c) const { return n_c > 0 ? first_coface_row(n_g, n_c) + pad16(n_c) : pad16(n_g); }
    __device__ int64_t lds(int64_t rows, int64_t src) const { return 3 * rows * (A.F + 8) * 2 + (src + 1) * A.F * 4 + A.idx_bytes; }
    __device__ const int64_t* bp(int t) const { return S.tasks[t] > 0 ? A.b_ptr[S.tasks[t]] : nullptr; }

    // do the complexes [a, b) fit one workgroup?  (*bad: cells of a higher dimension without cells of the set's first one)
    __device__ bool fits(int64_t a, int64_t b, bool* bad) const {
        const int g = S.g, d0 = S.tasks[0];
        const int64_t* up = g >= 0 ? A.up_ptr[g] : nullptr;
        const int64_t n0 = cells(d0, a, b);
        const int64_t rows = staged(n0, g >= 0 ? cells(g + 1, a, b) : 0);
        int64_t src = 0, ents = pad4(span(up, a, b));
        bool ok = true;
        for (int t = 0; t < S.n_tasks; ++t) {
            const int d = S.tasks[t];
            const int64_t be = span(bp(t), a, b);
            if (d > 0 && be > 0 && (A.variant == 0 || t == 0)) src += cells(d - 1, a, b);
            ents += pad4(be);
            ok = ok && cells(d, a, b) <= CWN_LAYER_TASK_ROWS;
            if (g >= 0 && n0 == 0 && t > 0 && cells(d, a, b) > 0) *bad = true;
        }
        ok = ok && rows <= A.row_cap && src <= A.src_cap && lds(rows, src) <= A.lds_budget && ents <= CWN_LAYER_MAX_ENTRIES;
        if (A.half_cap > 0 && g >= 0) ok = ok && pad16(n0) <= A.half_cap && pad16(cells(g + 1, a, b)) <= A.half_cap;
        return ok;
    }

    __device__ void record(int set, int64_t a, int64_t b, int32_t (&r)[kInts]) const {
#pragma unroll
        for (int k = 0; k < kInts; ++k) r[k] = 0;
        const int g = S.g, d0 = S.tasks[0];
        const int64_t* up = g >= 0 ? A.up_ptr[g] : nullptr;
        r[0] = set << 8;
        const int64_t n0 = cells(d0, a, b);
        int64_t nc = 0, une = 0;
        int live = S.n_tasks;
        if (g >= 0) {
            r[1] = g;
            if (n0 > 0) {
                nc = cells(g + 1, a, b);
                une = span(up, a, b);
                r[0] |= 1;
                r[2] = (int32_t)A.cell_ptr[g][a];
                r[3] = (int32_t)n0;
                r[4] = (int32_t)A.cell_ptr[g + 1][a];
                r[5] = (int32_t)nc;
                r[6] = (int32_t)(up != nullptr ? up[a] : 0);
                r[7] = (int32_t)une;
            } else {
                live = 1;
            }
        }
        r[8] = live;
        int64_t bne[2] = {0, 0};
        for (int t = 0; t < live; ++t) {
            const int d = S.tasks[t], o = 9 + 7 * t;
            const int64_t* p = bp(t);
            bne[t] = span(p, a, b);
            r[o] = d;
            r[o + 1] = (int32_t)A.cell_ptr[d][a];
            r[o + 2] = (int32_t)cells(d, a, b);
            r[o + 3] = (int32_t)(p != nullptr ? p[a] : 0);
            r[o + 4] = (int32_t)bne[t];
            if (d > 0 && bne[t] > 0) {          // boundary sources are staged only when read
                r[o + 5] = (int32_t)A.cell_ptr[d - 1][a];
                r[o + 6] = (int32_t)cells(d - 1, a, b);
            }
        }
        const int64_t b1 = pad4(une), b2 = pad4(b1 + bne[0]);
        r[23] = (int32_t)first_coface_row(n0, nc);
        r[24] = (int32_t)staged(n0, nc);
        r[25] = (int32_t)b1;
        r[26] = (int32_t)b2;
        r[27] = (int32_t)pad4(b2 + bne[1]);
    }
};

__global__ __launch_bounds__(kThreads) void items_fwd_kernel(ItemsArgs A_) {
    __shared__ int scan_lds[32];
    ItemsArgs A = A_;
    to_slot(A, blockIdx.y);
    const int set = blockIdx.x;
    const DevSet S = A.sets[set];
    const FwdGeom G{A, S};
    const int64_t nc_dev = *A.n_complexes;
    const int64_t C = nc_dev < A.cap_c ? (nc_dev < 0 ? 0 : nc_dev) : A.cap_c;
    const int region0 = A.set_start[set], region1 = A.set_start[set + 1];
    const int64_t groups = (A.cap_c + A.group - 1) / A.group;
    int base = 0;
    bool unfit = false;
    for (int64_t q0 = 0; q0 < groups; q0 += kThreads) {          // (uniform trip count)
        const int64_t a = (q0 + threadIdx.x) * A.group;
        const int64_t b = a + A.group < C ? a + A.group : C;
        int cnt = 0;
        bool whole = false, bad = false;
        if (a < C) {
            whole = G.fits(a, b, &bad);
            if (whole) {
                cnt = 1;
            } else {
                for (int64_t c = a; c < b; ++c) {
                    const bool one = G.fits(c, c + 1, &bad);
                    cnt += one ? 1 : 0;
                    unfit = unfit || !one;
                }
            }
        }
        unfit = unfit || bad;
        int total;
        const int off = block_scan(cnt, &total, scan_lds);
        if (cnt > 0) {
            int32_t r[kInts];
            int slot = region0 + base + off;
            if (whole) {
                if (slot < region1) {
                    G.record(set, a, b, r);
                    store_record(A.items + (size_t)slot * kInts, r, kInts);
                } else {
                    unfit = true;
                }
            } else {
                for (int64_t c = a; c < b; ++c) {
                    bool dummy = false;
                    if (!G.fits(c, c + 1, &dummy)) continue;
                    if (slot < region1) {
                        G.record(set, c, c + 1, r);
                        store_record(A.items + (size_t)slot * kInts, r, kInts);
                    } else {
                        unfit = true;
                    }
                    ++slot;
                }
            }
        }
        base += total;
    }
    // records past the set's own: empty
    for (int i = region0 + base + (int)threadIdx.x; i < region1; i += kThreads) {
        int4* dst = reinterpret_cast<int4*>(A.items + (size_t)i * kInts);
#pragma unroll
        for (int k = 0; k < kInts / 4; ++k) dst[k] = make_int4(0, 0, 0, 0);
    }
    if (unfit) atomicOr(A.err, CWN_ERR_BIT_UNFIT);
}

// ---- backward table, owner form (record layout: include/cwn_hip.h; cwn_blockplan.cpp: cwn_layer_bwd_items_build) ----------
struct BwdGeom {
    const ItemsArgs& A;
    int d;
    bool top, pa, pb, above;
    int flags;
    const int64_t* upa;
    const int64_t* upb;
    const int64_t* bnd;
    __device__ int64_t cells(int dd, int64_t a, int64_t b) const { return A.cell_ptr[dd][b] - A.cell_ptr[dd][a]; }
    __device__ static int64_t span(const int64_t* p, int64_t a, int64_t b) { return p != nullptr ? p[b] - p[a] : 0; }

    // 1: an item; 0: nothing to write (no owned cells); -1: beyond a limit; -2: entries without cells (not a cell complex)
    __device__ int classify(int64_t a, int64_t b, int32_t (&r)[kBInts]) const {
        namespace bo = cwn_bwd_own;
        const int F = A.F;
        const int64_t n_o = cells(d, a, b), n_a = above ? cells(d + 1, a, b) : 0, n_b = pb ? cells(d - 1, a, b) : 0;
        const int64_t ea = span(upa, a, b), eb = span(upb, a, b), bd = span(bnd, a, b);
        if (n_o > bo::own_rows_cap(F) || (top && n_a > bo::top_rows_cap(F))) return -1;
        if (ea > CWN_LAYER_MAX_ENTRIES || eb > CWN_LAYER_MAX_ENTRIES || bd > CWN_LAYER_MAX_ENTRIES) return -1;
        if (n_a > 4096 || n_b > 4096) return -1;
        const bool need_a = pa || top || bd > 0;
        const bo::Layout L = bo::layout(F, flags, (int)n_o, need_a ? (int)n_a : 0, (int)n_b, (int)ea, (int)eb, (int)bd);
        if (L.total > A.bwd_lds) return -1;
        if (n_o == 0) return ((top && n_a > 0) || ea > 0 || eb > 0 || bd > 0) ? -2 : 0;
#pragma unroll
        for (int k = 0; k < kBInts; ++k) r[k] = 0;
        r[bo::R_FLAGS] = flags | ((int)blockIdx.x << 8);
        r[bo::R_DIM] = d;
        r[bo::R_OWN_R0] = (int32_t)A.cell_ptr[d][a];
        r[bo::R_OWN_N] = (int32_t)n_o;
        if (need_a) {
            r[bo::R_ABOVE_R0] = (int32_t)A.cell_ptr[d + 1][a];
            r[bo::R_ABOVE_N] = (int32_t)n_a;
        }
        if (pb) {
            r[bo::R_BELOW_R0] = (int32_t)A.cell_ptr[d - 1][a];
            r[bo::R_BELOW_N] = (int32_t)n_b;
            r[bo::R_UPB_E0] = (int32_t)upb[a];
            r[bo::R_UPB_NE] = (int32_t)eb;
        }
        if (pa) {
            r[bo::R_UPA_E0] = (int32_t)upa[a];
            r[bo::R_UPA_NE] = (int32_t)ea;
        }
        if (bd > 0) {
            r[bo::R_BND_E0] = (int32_t)bnd[a];
            r[bo::R_BND_NE] = (int32_t)bd;
        }
        r[bo::R_LDS_BYTES] = L.total;
        return 1;
    }
};

__global__ __launch_bounds__(kThreads) void items_bwd_kernel(ItemsArgs A_) {
    __shared__ int scan_lds[32];
    ItemsArgs A = A_;
    to_slot(A, blockIdx.y);
    const int set = blockIdx.x;
    const DevSet S = A.sets[set];
    const int d = S.tasks[0];
    BwdGeom G{A, d, S.n_tasks == 2, A.has_up[d] != 0, d > 0 && A.has_up[d - 1] != 0, d + 1 < A.n_dims, 0, nullptr, nullptr, nullptr};
    G.flags = (G.pa ? cwn_bwd_own::F_PA : 0) | (G.pb ? cwn_bwd_own::F_PB : 0) | (G.top ? cwn_bwd_own::F_TOP : 0);
    G.upa = G.pa ? A.up_ptr[d] : nullptr;
    G.upb = G.pb ? A.up_ptr[d - 1] : nullptr;
    G.bnd = G.above ? A.b_ptr[d + 1] : nullptr;
    const int64_t nc_dev = *A.n_complexes;
    const int64_t C = nc_dev < A.cap_c ? (nc_dev < 0 ? 0 : nc_dev) : A.cap_c;
    const int region0 = A.set_start[set], region1 = A.set_start[set + 1];
    const int64_t groups = (A.cap_c + A.group - 1) / A.group;
    int base = 0;
    bool unfit = false;
    for (int64_t q0 = 0; q0 < groups; q0 += kThreads) {
        const int64_t a = (q0 + threadIdx.x) * A.group;
        const int64_t b = a + A.group < C ? a + A.group : C;
        int32_t r[kBInts];
        int cnt = 0, whole = 0;
        if (a < C) {
            whole = G.classify(a, b, r);
            if (whole == 1) {
                cnt = 1;
            } else if (whole == -1) {
                for (int64_t c = a; c < b; ++c) {
                    int32_t r1[kBInts];
                    const int k = G.classify(c, c + 1, r1);
                    cnt += k == 1 ? 1 : 0;
                    unfit = unfit || k < 0;
                }
            } else if (whole == -2) {
                unfit = true;
            }
        }
        int total;
        const int off = block_scan(cnt, &total, scan_lds);
        if (cnt > 0) {
            int slot = region0 + base + off;
            if (whole == 1) {
                if (slot < region1) store_record(A.items + (size_t)slot * kBInts, r, kBInts);
                else unfit = true;
            } else {
                for (int64_t c = a; c < b; ++c) {
                    if (G.classify(c, c + 1, r) != 1) continue;
                    if (slot < region1) store_record(A.items + (size_t)slot * kBInts, r, kBInts);
                    else unfit = true;
                    ++slot;
                }
            }
        }
        base += total;
    }
    for (int i = region0 + base + (int)threadIdx.x; i < region1; i += kThreads) {
        int4* dst = reinterpret_cast<int4*>(A.items + (size_t)i * kBInts);
#pragma unroll
        for (int k = 0; k < kBInts / 4; ++k) dst[k] = make_int4(0, 0, 0, 0);
    }
    if (unfit) atomicOr(A.err, CWN_ERR_BIT_UNFIT);
}

// sets in ascending order of dimension (cwn_blockplan.cpp: make_sets)
int make_sets(const cwn_layer_sizes_dev& in, DevSet (&sets)[CWN_LAYER_MAX_DIMS]) {
    int n = 0;
    for (int d = 0; d < in.n_dims;) {
        DevSet& s = sets[n++];
        s.tasks[0] = d;
        s.tasks[1] = 0;
        s.n_tasks = 1;
        if (in.has_up[d]) {
            s.g = d;
            if (d + 1 < in.n_dims && !in.has_up[d + 1] && d + 2 >= in.n_dims) s.tasks[s.n_tasks++] = d + 1;
        } else {
            s.g = -1;
        }
        d += s.n_tasks;
    }
    return n;
}

int fill_common(const cwn_layer_sizes_dev* in, int32_t F, int32_t group, int32_t* err_flag, ItemsArgs& A) {
    if (in == nullptr || (F != 64 && F != 128) || in->n_dims < 1 || in->n_dims > CWN_LAYER_MAX_DIMS || in->cap_complexes < 1 ||
        in->n_complexes == nullptr || group < 1 || err_flag == nullptr)
        return CWN_ERR_BAD_ARG;
    if (in->cap_complexes >= INT32_MAX / 4) return CWN_ERR_TOO_LARGE;
    for (int d = 0; d < in->n_dims; ++d) {
        if (in->cell_ptr[d] == nullptr) return CWN_ERR_BAD_ARG;
        if (in->has_up[d] && (d + 1 >= in->n_dims || in->up_ptr[d] == nullptr)) return CWN_ERR_BAD_ARG;
        A.cell_ptr[d] = in->cell_ptr[d];
        A.up_ptr[d] = in->up_ptr[d];
        A.b_ptr[d] = in->b_ptr[d];
        A.has_up[d] = in->has_up[d] ? 1 : 0;
    }
    A.n_complexes = in->n_complexes;
    A.cap_c = in->cap_complexes;
    A.n_dims = in->n_dims;
    A.F = F;
    A.group = group;
    A.err = err_flag;
    A.n_sets = make_sets(*in, A.sets);
    if (in->n_slots < 1 || in->n_slots > 1024 || (in->n_slots > 1 && in->table_slot_stride <= 0)) return CWN_ERR_BAD_ARG;
    A.table_slot_stride = in->n_slots > 1 ? in->table_slot_stride : 0;
    return CWN_OK;
}

}  // namespace

extern "C" int cwn_layer_items_build_dev(const cwn_layer_sizes_dev* in, int32_t F, const cwn_layer_plan* plan, int32_t group,
                                         int32_t* err_flag, cwn_stream_t stream_) {
    ItemsArgs A{};
    const int rc = fill_common(in, F, group, err_flag, A);
    if (rc != CWN_OK) return rc;
    if (plan == nullptr || plan->items == nullptr || plan->n_items < 1 || ((uintptr_t)plan->items & 15u)) return CWN_ERR_BAD_ARG;
    const int variant = plan->variant;
    if (variant != 0 && variant != 1) return CWN_ERR_BAD_ARG;
    A.items = const_cast<int32_t*>(plan->items);
    A.variant = variant;
    A.round_rows = cwn_layer_variant_round_rows(F, variant);
    if (A.round_rows <= 0) return CWN_ERR_BAD_ARG;
    // the LDS split of the captured launch: the caller's (variant 0: one layout per launch; variant 1: per item inside
    // the launch's dynamic LDS)
    A.row_cap = plan->max_gemm_rows;
    A.src_cap = plan->max_source_rows;
    A.lds_budget = variant == 1 ? plan->lds_bytes : (int64_t)160 * 1024;
    A.half_cap = variant == 1 ? (int64_t)CWN_LAYER_W8_HALF_ROWS(F) : 0;
    A.idx_bytes = (int64_t)cwn_layer_variant_lds_bytes(128, variant, 16, 0) - 3 * 16 * (128 + 8) * 2 - 128 * 4;
    const int64_t cap_rows = variant == 1 ? CWN_LAYER_W8_GEMM_ROWS(F) : CWN_LAYER_GEMM_ROWS(F);
    const int64_t cap_src = variant == 1 ? CWN_LAYER_W8_SOURCE_ROWS(F) : CWN_LAYER_SOURCE_ROWS(F);
    if (A.row_cap < 16 || A.row_cap > cap_rows || A.src_cap < 0 || A.src_cap > cap_src || A.lds_budget <= 0) return CWN_ERR_BAD_ARG;
    if (variant == 0 && cwn_layer_fused_lds_bytes(F, (int32_t)A.row_cap, (int32_t)A.src_cap) == 0) return CWN_ERR_BAD_ARG;
    // fixed regions: set_start[0] = 0 <= set_start[1] <= ... ; the last set ends at n_items
    for (int s = 0; s <= A.n_sets; ++s) {
        const int64_t v = s == A.n_sets ? plan->n_items : plan->set_start[s];
        const int64_t lo = s == 0 ? 0 : A.set_start[s - 1];
        if ((s == 0 && v != 0) || v < lo || v > plan->n_items || v >= INT32_MAX) return CWN_ERR_BAD_ARG;
        A.set_start[s] = (int32_t)v;
    }
    A.items_slot_ints = plan->n_items * kInts;
    items_fwd_kernel<<<dim3(A.n_sets, in->n_slots), dim3(kThreads), 0, (hipStream_t)stream_>>>(A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_layer_bwd_items_build_dev(const cwn_layer_sizes_dev* in, int32_t F, const cwn_layer_bwd_plan* plan, int32_t group,
                                             int32_t* err_flag, cwn_stream_t stream_) {
    ItemsArgs A{};
    const int rc = fill_common(in, F, group, err_flag, A);
    if (rc != CWN_OK) return rc;
    if (plan == nullptr || plan->items == nullptr || ((uintptr_t)plan->items & 15u)) return CWN_ERR_BAD_ARG;
    if (plan->lds_bytes <= 0 || plan->lds_bytes > 160 * 1024) return CWN_ERR_BAD_ARG;
    if (plan->n_items != (int64_t)A.n_sets * in->cap_complexes) return CWN_ERR_BAD_ARG;     // one region of cap_complexes records per set
    A.items = const_cast<int32_t*>(plan->items);
    A.bwd_lds = plan->lds_bytes;
    for (int s = 0; s <= A.n_sets; ++s) A.set_start[s] = (int32_t)(s * in->cap_complexes);
    A.items_slot_ints = plan->n_items * kBInts;
    items_bwd_kernel<<<dim3(A.n_sets, in->n_slots), dim3(kThreads), 0, (hipStream_t)stream_>>>(A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
