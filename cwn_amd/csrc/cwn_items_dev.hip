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

// Everything below reads the kernel arguments with COMPILE-TIME indices (sel3, unrolled loops): a run-time index into a
// by-value argument array makes the compiler copy the whole structure to scratch (the first form of these kernels: 23 us for
// eight slots, all of it scratch traffic).
template <typename T>
__device__ __forceinline__ T sel3(int i, T v0, T v1, T v2) { return i == 0 ? v0 : (i == 1 ? v1 : v2); }

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

// the prefix sums of a range of complexes [a, b): first cell / entry and count, per dimension (absent tables: zeros)
struct Range {
    int64_t c0[CWN_LAYER_MAX_DIMS], cn[CWN_LAYER_MAX_DIMS];     // cells
    int64_t u0[CWN_LAYER_MAX_DIMS], un[CWN_LAYER_MAX_DIMS];     // entries of upper_index_d
    int64_t b0[CWN_LAYER_MAX_DIMS], bn[CWN_LAYER_MAX_DIMS];     // entries of boundary_index_d
    __device__ __forceinline__ int64_t cells(int d) const { return sel3(d, cn[0], cn[1], cn[2]); }
    __device__ __forceinline__ int64_t cell0(int d) const { return sel3(d, c0[0], c0[1], c0[2]); }
    __device__ __forceinline__ int64_t ups(int d) const { return sel3(d, un[0], un[1], un[2]); }
    __device__ __forceinline__ int64_t up0(int d) const { return sel3(d, u0[0], u0[1], u0[2]); }
    __device__ __forceinline__ int64_t bnds(int d) const { return sel3(d, bn[0], bn[1], bn[2]); }
    __device__ __forceinline__ int64_t bnd0(int d) const { return sel3(d, b0[0], b0[1], b0[2]); }
};

__device__ __forceinline__ Range load_range(const ItemsArgs& A, int64_t o, int64_t a, int64_t b) {
    Range R;
#pragma unroll
    for (int d = 0; d < CWN_LAYER_MAX_DIMS; ++d) {
        R.c0[d] = R.cn[d] = R.u0[d] = R.un[d] = R.b0[d] = R.bn[d] = 0;
        if (d < A.n_dims) {
            const int64_t* cp = A.cell_ptr[d] + o;
            R.c0[d] = cp[a];
            R.cn[d] = cp[b] - R.c0[d];
            if (A.up_ptr[d] != nullptr) {
                const int64_t* up = A.up_ptr[d] + o;
                R.u0[d] = up[a];
                R.un[d] = up[b] - R.u0[d];
            }
            if (A.b_ptr[d] != nullptr) {
                const int64_t* bp = A.b_ptr[d] + o;
                R.b0[d] = bp[a];
                R.bn[d] = bp[b] - R.b0[d];
            }
        }
    }
    return R;
}

__device__ __forceinline__ void store_record(int32_t* dst, const int32_t* r, int n_ints) {
    for (int k = 0; k < n_ints; k += 4) *reinterpret_cast<int4*>(dst + k) = make_int4(r[k], r[k + 1], r[k + 2], r[k + 3]);
}

// ---- forward table (record layout: include/cwn_hip.h; the arithmetic of cwn_blockplan.cpp: build_with) --------------------
struct FwdGeom {
    int F, variant, round_rows, g, n_tasks, t0, t1;
    int64_t row_cap, src_cap, lds_budget, half_cap, idx_bytes;
    __device__ __forceinline__ int task(int t) const { return t == 0 ? t0 : t1; }
    __device__ int64_t first_coface_row(int64_t n_g, int64_t /*n_c*/) const { return pad16(n_g); }   // (= cwn_blockplan.cpp)
    __device__ int64_t staged(int64_t n_g, int64_t n_c) const { return n_c > 0 ? first_coface_row(n_g, n_c) + pad16(n_c) : pad16(n_g); }
    __device__ int64_t lds(int64_t rows, int64_t src) const { return 3 * rows * (F + 8) * 2 + (src + 1) * F * 4 + idx_bytes; }

    // does the range fit one workgroup?  (*bad: cells of a higher dimension without cells of the set's first one)
    __device__ bool fits(const Range& R, bool* bad) const {
        const int64_t n0 = R.cells(t0);
        const int64_t ncf = g >= 0 ? R.cells(g + 1) : 0;
        const int64_t rows = staged(n0, ncf);
        int64_t src = 0, ents = pad4(g >= 0 ? R.ups(g) : 0);
        bool ok = true;
        for (int t = 0; t < n_tasks; ++t) {
            const int d = task(t);
            const int64_t be = d > 0 ? R.bnds(d) : 0;
            if (d > 0 && be > 0 && (variant == 0 || t == 0)) src += R.cells(d - 1);
            ents += pad4(be);
            ok = ok && R.cells(d) <= CWN_LAYER_TASK_ROWS;
            if (g >= 0 && n0 == 0 && t > 0 && R.cells(d) > 0) *bad = true;
        }
        ok = ok && rows <= row_cap && src <= src_cap && lds(rows, src) <= lds_budget && ents <= CWN_LAYER_MAX_ENTRIES;
        if (half_cap > 0 && g >= 0) ok = ok && pad16(n0) <= half_cap && pad16(ncf) <= half_cap;
        return ok;
    }

    __device__ void record(int set, const Range& R, int32_t (&r)[kInts]) const {
#pragma unroll
        for (int k = 0; k < kInts; ++k) r[k] = 0;
        r[0] = set << 8;
        const int64_t n0 = R.cells(t0);
        int64_t nc = 0, une = 0;
        int live = n_tasks;
        if (g >= 0) {
            r[1] = g;
            if (n0 > 0) {
                nc = R.cells(g + 1);
                une = R.ups(g);
                r[0] |= 1;
                r[2] = (int32_t)R.cell0(g);
                r[3] = (int32_t)n0;
                r[4] = (int32_t)R.cell0(g + 1);
                r[5] = (int32_t)nc;
                r[6] = (int32_t)R.up0(g);
                r[7] = (int32_t)une;
            } else {
                live = 1;
            }
        }
        r[8] = live;
        int64_t bne[2] = {0, 0};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < live) {
                const int d = task(t), o = 9 + 7 * t;
                bne[t] = d > 0 ? R.bnds(d) : 0;
                r[o] = d;
                r[o + 1] = (int32_t)R.cell0(d);
                r[o + 2] = (int32_t)R.cells(d);
                r[o + 3] = (int32_t)(d > 0 ? R.bnd0(d) : 0);
                r[o + 4] = (int32_t)bne[t];
                if (d > 0 && bne[t] > 0) {          // boundary sources are staged only when read
                    r[o + 5] = (int32_t)R.cell0(d - 1);
                    r[o + 6] = (int32_t)R.cells(d - 1);
                }
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

__global__ __launch_bounds__(kThreads) void items_fwd_kernel(ItemsArgs A) {
    __shared__ int scan_lds[32];
    const int set = blockIdx.x;
    const int64_t o = (int64_t)blockIdx.y * A.table_slot_stride;
    int32_t* const items = A.items + (int64_t)blockIdx.y * A.items_slot_ints;
    const DevSet S0 = A.sets[0], S1 = A.sets[1], S2 = A.sets[2];
    const DevSet S = set == 0 ? S0 : (set == 1 ? S1 : S2);
    const FwdGeom G{A.F, A.variant, A.round_rows, S.g, S.n_tasks, S.tasks[0], S.tasks[1], A.row_cap, A.src_cap, A.lds_budget,
                    A.half_cap, A.idx_bytes};
    const int64_t nc_dev = A.n_complexes[o];
    const int64_t C = nc_dev < A.cap_c ? (nc_dev < 0 ? 0 : nc_dev) : A.cap_c;
    const int region0 = sel3(set, A.set_start[0], A.set_start[1], A.set_start[2]);
    const int region1 = sel3(set, A.set_start[1], A.set_start[2], A.set_start[3]);
    const int64_t groups = (A.cap_c + A.group - 1) / A.group;
    int base = 0;
    bool unfit = false;
    for (int64_t q0 = 0; q0 < groups; q0 += kThreads) {          // (uniform trip count)
        const int64_t a = (q0 + threadIdx.x) * A.group;
        const int64_t b = a + A.group < C ? a + A.group : C;
        int cnt = 0;
        bool whole = false, bad = false;
        Range R;
        if (a < C) {
            R = load_range(A, o, a, b);
            whole = G.fits(R, &bad);
            if (whole) {
                cnt = 1;
            } else {
                for (int64_t c = a; c < b; ++c) {
                    const bool one = G.fits(load_range(A, o, c, c + 1), &bad);
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
                    G.record(set, R, r);
                    store_record(items + (size_t)slot * kInts, r, kInts);
                } else {
                    unfit = true;
                }
            } else {
                for (int64_t c = a; c < b; ++c) {
                    bool dummy = false;
                    const Range R1 = load_range(A, o, c, c + 1);
                    if (!G.fits(R1, &dummy)) continue;
                    if (slot < region1) {
                        G.record(set, R1, r);
                        store_record(items + (size_t)slot * kInts, r, kInts);
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
        int4* dst = reinterpret_cast<int4*>(items + (size_t)i * kInts);
#pragma unroll
        for (int k = 0; k < kInts / 4; ++k) dst[k] = make_int4(0, 0, 0, 0);
    }
    if (unfit) atomicOr(A.err, CWN_ERR_BIT_UNFIT);
}

// ---- backward table, owner form (record layout: include/cwn_hip.h; cwn_blockplan.cpp: cwn_layer_bwd_items_build) ----------
struct BwdGeom {
    int F, d, set;
    bool top, pa, pb, above;
    int flags;
    int64_t bwd_lds;

    // 1: an item; 0: nothing to write (no owned cells); -1: beyond a limit; -2: entries without cells (not a cell complex)
    __device__ int classify(const Range& R, int32_t (&r)[kBInts]) const {
        namespace bo = cwn_bwd_own;
        const int64_t n_o = R.cells(d), n_a = above ? R.cells(d + 1) : 0, n_b = pb ? R.cells(d - 1) : 0;
        const int64_t ea = pa ? R.ups(d) : 0, eb = pb ? R.ups(d - 1) : 0, bd = above ? R.bnds(d + 1) : 0;
        if (n_o > bo::own_rows_cap(F) || (top && n_a > bo::top_rows_cap(F))) return -1;
        if (ea > CWN_LAYER_MAX_ENTRIES || eb > CWN_LAYER_MAX_ENTRIES || bd > CWN_LAYER_MAX_ENTRIES) return -1;
        if (n_a > 4096 || n_b > 4096) return -1;
        const bool need_a = pa || top || bd > 0;
        const bo::Layout L = bo::layout(F, flags, (int)n_o, need_a ? (int)n_a : 0, (int)n_b, (int)ea, (int)eb, (int)bd);
        if (L.total > bwd_lds) return -1;
        if (n_o == 0) return ((top && n_a > 0) || ea > 0 || eb > 0 || bd > 0) ? -2 : 0;
#pragma unroll
        for (int k = 0; k < kBInts; ++k) r[k] = 0;
        r[bo::R_FLAGS] = flags | (set << 8);
        r[bo::R_DIM] = d;
        r[bo::R_OWN_R0] = (int32_t)R.cell0(d);
        r[bo::R_OWN_N] = (int32_t)n_o;
        if (need_a) {
            r[bo::R_ABOVE_R0] = (int32_t)R.cell0(d + 1);
            r[bo::R_ABOVE_N] = (int32_t)n_a;
        }
        if (pb) {
            r[bo::R_BELOW_R0] = (int32_t)R.cell0(d - 1);
            r[bo::R_BELOW_N] = (int32_t)n_b;
            r[bo::R_UPB_E0] = (int32_t)R.up0(d - 1);
            r[bo::R_UPB_NE] = (int32_t)eb;
        }
        if (pa) {
            r[bo::R_UPA_E0] = (int32_t)R.up0(d);
            r[bo::R_UPA_NE] = (int32_t)ea;
        }
        if (bd > 0) {
            r[bo::R_BND_E0] = (int32_t)R.bnd0(d + 1);
            r[bo::R_BND_NE] = (int32_t)bd;
        }
        r[bo::R_LDS_BYTES] = L.total;
        return 1;
    }
};

__global__ __launch_bounds__(kThreads) void items_bwd_kernel(ItemsArgs A) {
    __shared__ int scan_lds[32];
    const int set = blockIdx.x;
    const int64_t o = (int64_t)blockIdx.y * A.table_slot_stride;
    int32_t* const items = A.items + (int64_t)blockIdx.y * A.items_slot_ints;
    const DevSet S0 = A.sets[0], S1 = A.sets[1], S2 = A.sets[2];
    const DevSet S = set == 0 ? S0 : (set == 1 ? S1 : S2);
    const int d = S.tasks[0];
    const bool pa = sel3(d, A.has_up[0], A.has_up[1], A.has_up[2]) != 0;
    const bool pb = d > 0 && sel3(d - 1, A.has_up[0], A.has_up[1], A.has_up[2]) != 0;
    BwdGeom G{A.F, d, set, S.n_tasks == 2, pa, pb, d + 1 < A.n_dims, 0, A.bwd_lds};
    G.flags = (G.pa ? cwn_bwd_own::F_PA : 0) | (G.pb ? cwn_bwd_own::F_PB : 0) | (G.top ? cwn_bwd_own::F_TOP : 0);
    const int64_t nc_dev = A.n_complexes[o];
    const int64_t C = nc_dev < A.cap_c ? (nc_dev < 0 ? 0 : nc_dev) : A.cap_c;
    const int region0 = sel3(set, A.set_start[0], A.set_start[1], A.set_start[2]);
    const int region1 = sel3(set, A.set_start[1], A.set_start[2], A.set_start[3]);
    const int64_t groups = (A.cap_c + A.group - 1) / A.group;
    int base = 0;
    bool unfit = false;
    for (int64_t q0 = 0; q0 < groups; q0 += kThreads) {
        const int64_t a = (q0 + threadIdx.x) * A.group;
        const int64_t b = a + A.group < C ? a + A.group : C;
        int32_t r[kBInts];
        int cnt = 0, whole = 0;
        if (a < C) {
            whole = G.classify(load_range(A, o, a, b), r);
            if (whole == 1) {
                cnt = 1;
            } else if (whole == -1) {
                for (int64_t c = a; c < b; ++c) {
                    int32_t r1[kBInts];
                    const int k = G.classify(load_range(A, o, c, c + 1), r1);
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
                if (slot < region1) store_record(items + (size_t)slot * kBInts, r, kBInts);
                else unfit = true;
            } else {
                for (int64_t c = a; c < b; ++c) {
                    if (G.classify(load_range(A, o, c, c + 1), r) != 1) continue;
                    if (slot < region1) store_record(items + (size_t)slot * kBInts, r, kBInts);
                    else unfit = true;
                    ++slot;
                }
            }
        }
        base += total;
    }
    for (int i = region0 + base + (int)threadIdx.x; i < region1; i += kThreads) {
        int4* dst = reinterpret_cast<int4*>(items + (size_t)i * kBInts);
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
    if (plan->lds_bytes <= 0 || plan->lds_bytes > cwn_bwd_own::kLdsCap) return CWN_ERR_BAD_ARG;
    if (plan->n_items != (int64_t)A.n_sets * in->cap_complexes) return CWN_ERR_BAD_ARG;     // one region of cap_complexes records per set
    A.items = const_cast<int32_t*>(plan->items);
    A.bwd_lds = plan->lds_bytes;
    for (int s = 0; s <= A.n_sets; ++s) A.set_start[s] = (int32_t)(s * in->cap_complexes);
    A.items_slot_ints = plan->n_items * kBInts;
    items_bwd_kernel<<<dim3(A.n_sets, in->n_slots), dim3(kThreads), 0, (hipStream_t)stream_>>>(A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
