// cwn_ends.hip -- the two ends of a model forward around the message-passing layers, one launch each (inference).
//
// FRONT  cwn_embed_front_f32: the input features of all three cochain dimensions
//     x0[v] = sum_c Tv_c[ids0[v, c]]                                 v_embed_init / OGB AtomEncoder
//     x1[e] = sum_c Te_c[ids1[e, c]]      (or red1[e] without an edge table)   e_embed_init / BondEncoder
//     x2[r] = 1/2 sum_{e in boundary(r)} red1[e],   red1[e] = sum_{v in boundary(e)} x0[v]
//   = EmbedVEWithReduce.forward / OGBEmbedVEWithReduce.forward (mp/layers.py:490-593: embed, InitReduceConv of the
//   vertex embeddings onto the edges :526, InitReduceConv of THAT onto the rings, halved :538-540; InitReduceConv
//   itself :473-487).  The library ran this as 8 launches (two embedding gathers, two dtype conversions, two
//   segmented reductions, a copy, a scale): 29 us of a 167 us forward at the ZINC batch of 128, every one of them a
//   few microseconds of latency for a few hundred kilobytes.  Here a ring row walks its boundary edges and their
//   boundary vertices itself (a ring has ~6 edges x 2 vertices: twelve table rows out of L2), in CSR order -- the
//   order the segmented reduction adds them in, so the result is bit-identical to the launches it replaces.
//
// HEAD   cwn_head_f32: readout + lin1s + final readout + lin2 (mp/nn.py:50-60, mp/molec_models.py:129-156 /
//   mp/models.py:222-253), one workgroup per complex:
//     pooled_d[c] = sum (or mean) of the rows of x_d that belong to complex c        pool_complex
//     h_d = relu(W1_d pooled_d + b1_d);  s = sum_d h_d (or mean);  out[c] = W2 s + b2
//   The library ran this as one segmented-reduce launch, one grouped GEMM, two adds and a 256 -> 1 GEMM that alone
//   took 19.9 us (two workgroups walking K = 256): 43 us of the 167.  The cells of a complex are contiguous in every
//   x_d (data/complex.py:148-169; `ptr`, :344, 432), so a workgroup sums its complex's rows with coalesced loads and
//   keeps everything after that in LDS; the weights (lin1s transposed once per weight version: [K, H2], coalesced
//   over the outputs) come out of L2.  fp32 FMA chains in a fixed order: deterministic, and a complex's result does
//   not depend on what else is in the batch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cwn_hip.h"
#include "cwn_mem.h"
#include "cwn_dropout.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// front
// ---------------------------------------------------------------------------------------------------------------
struct FrontArgs {
    cwn_embed_table tv, te;
    float* x0; float* x1; float* x2;
    const int32_t* rowptr1; const int32_t* col1;
    const int32_t* rowptr2; const int32_t* col2;
    int64_t n0, n1, n2;
    int32_t H, G, has_te, halve;
    int32_t* err;
    const int64_t* n_dev;        // [3] the actual n0, n1, n2 (the fields above are then capacities: grid, output blocks), or NULL
};

__device__ __forceinline__ int64_t clampi(int64_t v, int64_t n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// ---- ONE table per cell type (torch.nn.Embedding: the ZINC models): the whole front in one launch -------------------------
// (Several tables per cell type -- the OGB encoders -- take the two launches further down: walked in this kernel a ring row
// was 108 table rows behind a chain of seven dependent loads.)
constexpr int kMaxCols = 16;          // columns of a table set (cwn_embed_table.cols)

// one row of a single table, 4 features at h.  An index outside the table sets bit 1 of the sticky word and contributes
// nothing (as cwn_embedding_fwd_f32)
__device__ __forceinline__ float4 emb_row(const cwn_embed_table& T, int64_t r, int H, int h, int32_t* err) {
    const int64_t id = T.src_is_f32 ? (int64_t)reinterpret_cast<const float*>(T.src)[r]     // .to(torch.long): truncation
                                    : reinterpret_cast<const int64_t*>(T.src)[r];
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool ok = id >= 0 && id < T.V;
    if (!ok && h == 0) atomicOr(err, 2);
    if (ok) w = *reinterpret_cast<const float4*>(T.W + id * H + h);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
    return acc;
}

// the same for TWO rows at once (the two boundary vertices of an edge): features of both, then table rows of both
__device__ __forceinline__ void emb_row_pair(const cwn_embed_table& T, int64_t r0, int64_t r1, int H, int h, int32_t* err, float4& a0,
                                             float4& a1) {
    int64_t id[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t r = q == 0 ? r0 : r1;
        id[q] = T.src_is_f32 ? (int64_t)reinterpret_cast<const float*>(T.src)[r] : reinterpret_cast<const int64_t*>(T.src)[r];
    }
    float4 w[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        w[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool ok = id[q] >= 0 && id[q] < T.V;
        if (!ok && h == 0) atomicOr(err, 2);
        if (ok) w[q] = *reinterpret_cast<const float4*>(T.W + id[q] * H + h);
    }
    a0 = a1 = make_float4(0.f, 0.f, 0.f, 0.f);
    a0.x += w[0].x; a0.y += w[0].y; a0.z += w[0].z; a0.w += w[0].w;
    a1.x += w[1].x; a1.y += w[1].y; a1.z += w[1].z; a1.w += w[1].w;
}

// red1[e] = sum of the embedded boundary vertices of edge e, in CSR order; two boundary vertices at a time (an edge has
// exactly two)
__device__ __forceinline__ float4 reduce_edge(const FrontArgs& A, int64_t e, int h) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (A.rowptr1 == nullptr) return acc;
    const int s = A.rowptr1[e], t = A.rowptr1[e + 1];
    for (int p = s; p < t; p += 2) {
        const int64_t v0 = clampi(A.col1[p], A.n0), v1 = clampi(A.col1[p + 1 < t ? p + 1 : p], A.n0);   // the plan build reported bad ones; never fault
        float4 w0, w1;
        emb_row_pair(A.tv, v0, v1, A.H, h, A.err, w0, w1);
        acc.x += w0.x; acc.y += w0.y; acc.z += w0.z; acc.w += w0.w;
        if (p + 1 < t) { acc.x += w1.x; acc.y += w1.y; acc.z += w1.z; acc.w += w1.w; }
    }
    return acc;
}

// The same sums for up to kChunk boundary edges of a ring AT ONCE.  A ring row is a chain of six dependent loads
// (rowptr2 -> col2 -> rowptr1 -> col1 -> integer feature -> table row); walked edge by edge and vertex by vertex a
// hexagon was 42 round trips to L2 (12 us for a launch that moves 3 MB).  Here every level is issued for all edges
// of the chunk before the next level needs it, and only the ADDS run in CSR order (the result is bit-identical to the
// sequential walk).  An edge has two boundary vertices; further ones (any other CSR1) take the sequential tail.
constexpr int kChunk = 8;

__device__ __forceinline__ float4 ring_chunk(const FrontArgs& A, int p0, int n, int h, float4 acc) {
    int64_t e[kChunk];
    int s1[kChunk], t1[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) e[u] = clampi(A.col2[p0 + (u < n ? u : 0)], A.n1);
#pragma unroll
    for (int u = 0; u < kChunk; ++u) { s1[u] = A.rowptr1[e[u]]; t1[u] = A.rowptr1[e[u] + 1]; }
    // vertex numbers, integer features and table rows of both endpoints of every edge, level by level
    int64_t v[kChunk][2], id[kChunk][2];
    float4 w[kChunk][2];
#pragma unroll
    for (int u = 0; u < kChunk; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q) v[u][q] = clampi(A.col1[s1[u] + q < t1[u] ? s1[u] + q : (t1[u] > s1[u] ? t1[u] - 1 : 0)], A.n0);
#pragma unroll
    for (int u = 0; u < kChunk; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q)
            id[u][q] = A.tv.src_is_f32 ? (int64_t)reinterpret_cast<const float*>(A.tv.src)[v[u][q]]
                                       : reinterpret_cast<const int64_t*>(A.tv.src)[v[u][q]];
#pragma unroll
    for (int u = 0; u < kChunk; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool ok = id[u][q] >= 0 && id[u][q] < A.tv.V;
            if (!ok && h == 0 && u < n && s1[u] + q < t1[u]) atomicOr(A.err, 2);
            w[u][q] = *reinterpret_cast<const float4*>(A.tv.W + (ok ? id[u][q] : 0) * A.H + h);
            if (!ok) w[u][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
        if (u >= n) break;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (s1[u] + q < t1[u]) { r.x += w[u][q].x; r.y += w[u][q].y; r.z += w[u][q].z; r.w += w[u][q].w; }
        for (int p = s1[u] + 2; p < t1[u]; ++p) {                   // not a 1-cell's boundary: the plain walk
            const float4 x = emb_row(A.tv, clampi(A.col1[p], A.n0), A.H, h, A.err);
            r.x += x.x; r.y += x.y; r.z += x.z; r.w += x.w;
        }
        acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
    }
    return acc;
}

__global__ __launch_bounds__(256) void embed_front_kernel(FrontArgs A) {
    // device-side row counts: the ROW -> (dimension, cell) mapping below follows the capacities (the grid was sized with
    // them); a cell past its dimension's actual count leaves
    // (the index clamps of the reductions keep the capacities: they are there so that no address leaves the buffers)
    const int64_t cap0 = A.n0, cap1 = A.n1;
    int64_t live0 = A.n0, live1 = A.n1, live2 = A.n2;
    if (A.n_dev != nullptr) {
        const int64_t d0 = A.n_dev[0], d1 = A.n_dev[1], d2 = A.n_dev[2];
        live0 = d0 < live0 ? d0 : live0;
        live1 = d1 < live1 ? d1 : live1;
        live2 = d2 < live2 ? d2 : live2;
    }
    const int G = A.G, gl = threadIdx.x & (G - 1);
    const int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (row >= cap0 + cap1 + A.n2) return;
    if (row < cap0 ? row >= live0 : (row < cap0 + cap1 ? row - cap0 >= live1 : row - cap0 - cap1 >= live2)) return;
    for (int h = 4 * gl; h < A.H; h += 4 * G) {
        float4 v;
        float* dst;
        if (row < cap0) {
            v = emb_row(A.tv, row, A.H, h, A.err);
            dst = A.x0 + row * A.H + h;
        } else if (row < cap0 + cap1) {
            const int64_t e = row - cap0;
            v = A.has_te ? emb_row(A.te, e, A.H, h, A.err) : reduce_edge(A, e, h);
            dst = A.x1 + e * A.H + h;
        } else {
            const int64_t r = row - cap0 - cap1;
            v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (A.rowptr2 != nullptr && A.rowptr1 != nullptr) {
                const int s = A.rowptr2[r], t = A.rowptr2[r + 1];
                for (int p = s; p < t; p += kChunk) v = ring_chunk(A, p, min(kChunk, t - p), h, v);
            }
            if (A.halve) { v.x *= 0.5f; v.y *= 0.5f; v.z *= 0.5f; v.w *= 0.5f; }
            dst = A.x2 + r * A.H + h;
        }
        cwn::store_result4(dst, v.x, v.y, v.z, v.w);
    }
}

// ---- several tables per cell type (the OGB encoders: 9 atom + 3 bond columns): the reductions read x0 ----------------------
// In one launch a ring row walks 12 vertices of 9 columns each (108 table rows behind a chain of seven dependent loads): 44 - 60
// us at the molhiv batch of 512, against ~25 for the separate launches it was meant to replace.  Two launches instead: the
// embeddings of both cell types by embed_pair_kernel (rows of dimensions 0 and 1 only), then THIS kernel for the rows
// that reduce -- edges without a table of their own, rings -- from the x0 rows the first launch wrote (in L2):
//     x1[e] = sum_{v in row e of CSR1} x0[v]                        (no edge table)
//     x2[r] = (halve ? 1/2 : 1) sum_{e in row r of CSR2} (sum_{v in row e of CSR1} x0[v])
// the adds in CSR order, every level of a chunk of kChunk edges requested before the next needs it -- bit-identical to the
// separate launches (an x0 row IS the column sum they make).
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// the embeddings of both cell types in one launch, the columns of a row in a plain loop (cwn_embedding_fwd_f32's walk: the
// 16-column form of embed_front_kernel that round 3 unrolled held 116 registers and took 22 us for the rows this takes 10 for)
__device__ __forceinline__ float4 embed_cols(const float* __restrict__ W, const void* __restrict__ src, const int64_t* __restrict__ col_off,
                                             const int64_t* __restrict__ col_size, int64_t V, int cols, bool f32, int64_t r, int H, int h,
                                             int32_t* err) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < cols; ++c) {
        int64_t v = f32 ? (int64_t)reinterpret_cast<const float*>(src)[r * cols + c] : reinterpret_cast<const int64_t*>(src)[r * cols + c];
        const int64_t lim = col_size != nullptr ? col_size[c] : V;   // per TABLE, not per concatenation
        if (v < 0 || v >= lim) {
            if (h == 0) atomicOr(err, 2);
            continue;
        }
        if (col_off != nullptr) v += col_off[c];
        const float4 w = ld4(W + v * H + h);
        acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
    }
    return acc;
}

__global__ __launch_bounds__(256) void embed_pair_kernel(FrontArgs A) {
    const int64_t cap0 = A.n0, cap1 = A.has_te ? A.n1 : 0;
    int64_t live0 = cap0, live1 = cap1;
    if (A.n_dev != nullptr) {
        const int64_t d0 = A.n_dev[0], d1 = A.n_dev[1];
        live0 = d0 < live0 ? d0 : live0;
        live1 = d1 < live1 ? d1 : live1;
    }
    const int G = A.G, gl = threadIdx.x & (G - 1);
    const int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (row >= cap0 + cap1) return;
    if (row < cap0 ? row >= live0 : row - cap0 >= live1) return;
    for (int h = 4 * gl; h < A.H; h += 4 * G) {
        if (row < cap0) {
            const float4 v = embed_cols(A.tv.W, A.tv.src, A.tv.col_off, A.tv.col_size, A.tv.V, A.tv.cols, A.tv.src_is_f32 != 0, row, A.H, h, A.err);
            cwn::store_result4(A.x0 + row * A.H + h, v.x, v.y, v.z, v.w);
        } else {
            const int64_t e = row - cap0;
            const float4 v = embed_cols(A.te.W, A.te.src, A.te.col_off, A.te.col_size, A.te.V, A.te.cols, A.te.src_is_f32 != 0, e, A.H, h, A.err);
            cwn::store_result4(A.x1 + e * A.H + h, v.x, v.y, v.z, v.w);
        }
    }
}

__global__ __launch_bounds__(256) void front_reduce_kernel(FrontArgs A) {
    const int64_t cap1 = A.has_te ? 0 : A.n1;          // edge rows of this launch
    int64_t live0 = A.n0, live1 = A.n1, live2 = A.n2;
    if (A.n_dev != nullptr) {
        const int64_t d0 = A.n_dev[0], d1 = A.n_dev[1], d2 = A.n_dev[2];
        live0 = d0 < live0 ? d0 : live0;
        live1 = d1 < live1 ? d1 : live1;
        live2 = d2 < live2 ? d2 : live2;
    }
    const int G = A.G, gl = threadIdx.x & (G - 1);
    const int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (row >= cap1 + A.n2) return;
    if (row < cap1 ? row >= live1 : row - cap1 >= live2) return;
    for (int h = 4 * gl; h < A.H; h += 4 * G) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float* dst;
        if (row < cap1) {
            dst = A.x1 + row * A.H + h;
            if (A.rowptr1 != nullptr) {
                const int s = A.rowptr1[row], t = A.rowptr1[row + 1];
                for (int p = s; p < t; p += 2) {
                    const int64_t v0 = clampi(A.col1[p], A.n0), v1 = clampi(A.col1[p + 1 < t ? p + 1 : p], A.n0);
                    const float4 w0 = ld4(A.x0 + v0 * A.H + h), w1 = ld4(A.x0 + v1 * A.H + h);
                    acc.x += w0.x; acc.y += w0.y; acc.z += w0.z; acc.w += w0.w;
                    if (p + 1 < t) { acc.x += w1.x; acc.y += w1.y; acc.z += w1.z; acc.w += w1.w; }
                }
            }
        } else {
            const int64_t r = row - cap1;
            dst = A.x2 + r * A.H + h;
            if (A.rowptr2 != nullptr && A.rowptr1 != nullptr) {
                const int s2 = A.rowptr2[r], t2 = A.rowptr2[r + 1];
                for (int p0 = s2; p0 < t2; p0 += kChunk) {
                    const int n = min(kChunk, t2 - p0);
                    int64_t e[kChunk], v[kChunk][2];
                    int s1[kChunk], t1[kChunk];
                    float4 w[kChunk][2];
#pragma unroll
                    for (int u = 0; u < kChunk; ++u) e[u] = clampi(A.col2[p0 + (u < n ? u : 0)], A.n1);
#pragma unroll
                    for (int u = 0; u < kChunk; ++u) { s1[u] = A.rowptr1[e[u]]; t1[u] = A.rowptr1[e[u] + 1]; }
#pragma unroll
                    for (int u = 0; u < kChunk; ++u)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            v[u][q] = clampi(A.col1[s1[u] + q < t1[u] ? s1[u] + q : (t1[u] > s1[u] ? t1[u] - 1 : 0)], A.n0);
#pragma unroll
                    for (int u = 0; u < kChunk; ++u)
#pragma unroll
                        for (int q = 0; q < 2; ++q) w[u][q] = ld4(A.x0 + v[u][q] * A.H + h);
#pragma unroll
                    for (int u = 0; u < kChunk; ++u) {
                        if (u >= n) break;
                        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            if (s1[u] + q < t1[u]) { x.x += w[u][q].x; x.y += w[u][q].y; x.z += w[u][q].z; x.w += w[u][q].w; }
                        for (int p = s1[u] + 2; p < t1[u]; ++p) {                   // not a 1-cell's boundary: the plain walk
                            const float4 y = ld4(A.x0 + clampi(A.col1[p], A.n0) * A.H + h);
                            x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
                        }
                        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
                    }
                }
            }
            if (A.halve) { acc.x *= 0.5f; acc.y *= 0.5f; acc.z *= 0.5f; acc.w *= 0.5f; }
        }
        cwn::store_result4(dst, acc.x, acc.y, acc.z, acc.w);
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

inline bool table_ok(const cwn_embed_table& T) {
    return T.W != nullptr && T.src != nullptr && T.cols > 0 && T.cols <= kMaxCols && T.V > 0 &&
           (T.col_off == nullptr) == (T.col_size == nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// head
// ---------------------------------------------------------------------------------------------------------------
constexpr int kHeadThreads = 512;
constexpr int kPartFloats = 4 * kHeadThreads;          // partial row sums of one dimension: [row groups][K] <= 512 float4

struct HeadArgs {
    cwn_head_dim d[CWN_HEAD_MAX_DIMS];
    const float* w2; const float* b2; float* out;
    float* s_out;                                  // [C, H2] the summed hidden vector (training), or NULL
    int64_t C;
    int32_t n_dims, K, H2, O, mean_readout, mean_final;
    cwn_dropout drop;                              // the head's dropout (cwn_dropout.h) at position drop_pos (CWN_HEAD_DROP_*)
    int32_t drop_pos;
    float* partials;                               // chunk sums [slots][K] (head_pool_kernel wrote them), or NULL
    int64_t slot_base[CWN_HEAD_MAX_DIMS];          // first slot of dimension d: chunk j of complex c sits at slot_base[d] + r0 / kHeadChunk + c + j
    int32_t pool_split;                            // P: workgroups per complex of head_pool_kernel
};

// the source of columns 4 l .. 4 l + 3 of row r of dimension D: the matrix itself, or -- a jumping-knowledge concatenation
// that is never materialised -- the block's own matrix (cwn_head_dim.x_more)
__device__ __forceinline__ const float* head_row(const cwn_head_dim& D, int64_t r, int col, int Kp) {
    if (D.n_parts <= 1) return D.x + r * D.ldx + col;
    const int q = col / Kp;
    const float* base = q == 0 ? D.x : D.x_more[q - 1];
    return base + r * D.ldx + (col - q * Kp);
}

// THE ORDER in which the rows of a complex are summed, whichever launch does it (so that a complex's pooled vector is the same
// bits in a batch of molecules, in a static batch of another capacity, with its rows summed by one workgroup or by many):
// chunks of kHeadChunk (= CWN_HEAD_CHUNK) consecutive rows; inside a chunk row group g adds rows a + g, a + g + NG, ... one after the other, the
// NG group partials are added in group order; the chunk sums are added in chunk order.  (A complex of at most kHeadChunk
// cells per dimension -- every molecule -- is one chunk: the order this kernel has always had.)
constexpr int kHeadChunk = CWN_HEAD_CHUNK;

__device__ __forceinline__ float4 head_group_sum(const cwn_head_dim& D, int64_t a, int64_t b, int g, int NG, int l, int Kp) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = a + g; r < b; r += 8 * (int64_t)NG) {          // eight rows in flight, added one after the other
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t ru = r + (int64_t)u * NG;
            w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ru < b) w[u] = *reinterpret_cast<const float4*>(head_row(D, ru, 4 * l, Kp));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += w[u].x; acc.y += w[u].y; acc.z += w[u].z; acc.w += w[u].w; }
    }
    return acc;
}


__host__ __device__ constexpr size_t head_lds_floats(int K, int H2) {
    return (size_t)CWN_HEAD_MAX_DIMS * kPartFloats + (size_t)CWN_HEAD_MAX_DIMS * K + (size_t)CWN_HEAD_MAX_DIMS * 4 * kHeadThreads + H2;
}

__global__ __launch_bounds__(kHeadThreads) void head_kernel(HeadArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = A.K, H2 = A.H2, nd = A.n_dims;
    float* const part = sm;                                                    // [3][NG][K]
    float* const pooled = part + CWN_HEAD_MAX_DIMS * kPartFloats;              // [3][K]
    float* const hpart = pooled + CWN_HEAD_MAX_DIMS * K;                       // [3][S][H2], S * H2 <= 4 * 512
    float* const sbuf = hpart + CWN_HEAD_MAX_DIMS * 4 * kHeadThreads;          // [H2]
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x;
    cwn::Dropout drop;
    drop.init(A.drop);
    const int dpos = drop.on ? A.drop_pos : CWN_HEAD_DROP_NONE;

    // ---- 1. pooled_d = sum of this complex's rows of x_d: K / 4 lanes a row, row groups side by side ------------
    const int G = K / 4, NG = kHeadThreads / G;
    const int g = tid / G, l = tid - g * G;
    int64_t r0[CWN_HEAD_MAX_DIMS], r1[CWN_HEAD_MAX_DIMS];
    int Kp[CWN_HEAD_MAX_DIMS];
#pragma unroll
    for (int d = 0; d < CWN_HEAD_MAX_DIMS; ++d) {
        Kp[d] = (d < nd && A.d[d].n_parts > 1) ? K / A.d[d].n_parts : K;
        r0[d] = r1[d] = 0;
        if (d < nd && A.d[d].x != nullptr && A.d[d].n_cells > 0) {
            const int64_t a = A.d[d].cell_ptr[c], b = A.d[d].cell_ptr[c + 1];
            r0[d] = a < 0 ? 0 : (a > A.d[d].n_cells ? A.d[d].n_cells : a);      // a table that is not this batch's cannot fault
            r1[d] = b < r0[d] ? r0[d] : (b > A.d[d].n_cells ? A.d[d].n_cells : b);
        }
    }
    const bool from_partials = A.partials != nullptr;      // (uniform) head_pool_kernel summed the rows, chunk by chunk
    bool big = false;                                      // (uniform) a complex of more than one chunk, summed here
#pragma unroll
    for (int d = 0; d < CWN_HEAD_MAX_DIMS; ++d) big = big || (r1[d] - r0[d] > kHeadChunk);
    if (from_partials || big) {
        float* const tmp = hpart;                          // [NG][K] group partials of one chunk (phase 2 has not begun)
        for (int d = 0; d < CWN_HEAD_MAX_DIMS; ++d) {
            const int64_t nch = (r1[d] - r0[d] + kHeadChunk - 1) / kHeadChunk;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};           // columns tid, tid + 512, ... (K <= 2048)
            if (from_partials) {
                const float* mine = A.partials + (size_t)(A.slot_base[d] + r0[d] / kHeadChunk + c) * K;
                for (int64_t j = 0; j < nch; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = tid + u * kHeadThreads;
                        if (k < K) acc[u] += mine[(size_t)j * K + k];
                    }
            } else {
                for (int64_t j = 0; j < nch; ++j) {
                    const int64_t a = r0[d] + j * kHeadChunk, b = a + kHeadChunk < r1[d] ? a + kHeadChunk : r1[d];
                    if (g < NG) *reinterpret_cast<float4*>(tmp + (size_t)g * K + 4 * l) = head_group_sum(A.d[d], a, b, g, NG, l, Kp[d]);
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = tid + u * kHeadThreads;
                        if (k < K) {
                            float t = 0.f;
                            for (int q = 0; q < NG; ++q) t += tmp[(size_t)q * K + k];
                            acc[u] += t;
                        }
                    }
                    __syncthreads();
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = tid + u * kHeadThreads;
                if (k < K) part[(size_t)d * kPartFloats + k] = acc[u];                 // as row group 0's partial
            }
        }
    } else if (g < NG) {
        // the first two rows of every dimension are requested before any is added: a molecule gives a row group one
        // or two rows per dimension, and three dimensions one after the other would be three memory round trips
        float4 v[CWN_HEAD_MAX_DIMS][2];
#pragma unroll
        for (int d = 0; d < CWN_HEAD_MAX_DIMS; ++d)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                v[d][u] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int64_t r = r0[d] + g + (int64_t)u * NG;
                if (r < r1[d]) v[d][u] = *reinterpret_cast<const float4*>(head_row(A.d[d], r, 4 * l, Kp[d]));
            }
#pragma unroll
        for (int d = 0; d < CWN_HEAD_MAX_DIMS; ++d) {
            float4 acc = v[d][0];
            acc.x += v[d][1].x; acc.y += v[d][1].y; acc.z += v[d][1].z; acc.w += v[d][1].w;
            for (int64_t r = r0[d] + g + 2 * (int64_t)NG; r < r1[d]; r += 4 * (int64_t)NG) {       // large complexes
                float4 w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t ru = r + (int64_t)u * NG;
                    w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ru < r1[d]) w[u] = *reinterpret_cast<const float4*>(head_row(A.d[d], ru, 4 * l, Kp[d]));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc.x += w[u].x; acc.y += w[u].y; acc.z += w[u].z; acc.w += w[u].w; }
            }
            *reinterpret_cast<float4*>(part + (size_t)d * kPartFloats + (size_t)g * K + 4 * l) = acc;
        }
    }
    __syncthreads();
    for (int i = tid; i < CWN_HEAD_MAX_DIMS * K; i += kHeadThreads) {
        const int d = i / K, k = i - d * K;
        float s = 0.f;
        for (int q = 0; q < ((from_partials || big) ? 1 : NG); ++q) s += part[(size_t)d * kPartFloats + (size_t)q * K + k];     // fixed order
        if (A.mean_readout) {
            const int64_t n = r1[d] - r0[d];
            s = s / (float)(n > 0 ? n : 1);
        }
        if (dpos == CWN_HEAD_DROP_LIN1) s *= drop.mul1(((uint64_t)d * (uint64_t)A.C + (uint64_t)c) * (uint64_t)K + (uint64_t)k);
        pooled[i] = s;
        if (d < nd && A.d[d].pooled_out != nullptr) A.d[d].pooled_out[c * K + k] = s;
    }
    __syncthreads();

    // ---- 2. h_d = W1_d pooled_d.  A thread owns FOUR consecutive outputs (one 16-byte load of the transposed weight
    // per k: the form the memory pipeline takes at full rate -- with 4-byte loads this phase was 8 of the launch's
    // 15 us) and one of the S = 512 / (H2 / 4) slices of the K range; every load of a slice is in flight before the
    // first FMA.  The slices' partial sums meet in LDS, in slice order.
    const int JQ = H2 / 4, S = kHeadThreads / JQ;
    const int jq = tid % JQ, kh = tid / JQ;
    if (kh < S) {
        const int kb = (K + S - 1) / S, k0 = min(K, kh * kb), k1 = min(K, k0 + kb);
#pragma unroll
        for (int d = 0; d < CWN_HEAD_MAX_DIMS; ++d) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (d < nd) {
                const float* w = A.d[d].w1t + (size_t)k0 * H2 + 4 * jq;
                const float* p = pooled + d * K + k0;
                int k = 0;
                for (; k + 8 <= k1 - k0; k += 8) {
                    float4 wv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const float4*>(w + (size_t)(k + u) * H2);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float pk = p[k + u];
                        a.x = __builtin_fmaf(wv[u].x, pk, a.x); a.y = __builtin_fmaf(wv[u].y, pk, a.y);
                        a.z = __builtin_fmaf(wv[u].z, pk, a.z); a.w = __builtin_fmaf(wv[u].w, pk, a.w);
                    }
                }
                for (; k < k1 - k0; ++k) {
                    const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)k * H2);
                    const float pk = p[k];
                    a.x = __builtin_fmaf(wv.x, pk, a.x); a.y = __builtin_fmaf(wv.y, pk, a.y);
                    a.z = __builtin_fmaf(wv.z, pk, a.z); a.w = __builtin_fmaf(wv.w, pk, a.w);
                }
            }
            *reinterpret_cast<float4*>(hpart + (size_t)d * 4 * kHeadThreads + (size_t)kh * H2 + 4 * jq) = a;
        }
    }
    __syncthreads();
    if (tid < H2) {
        float s = 0.f;
        for (int d = 0; d < nd; ++d) {
            float h = 0.f;
            for (int q = 0; q < S; ++q) h += hpart[(size_t)d * 4 * kHeadThreads + q * H2 + tid];
            if (A.d[d].b1 != nullptr) h += A.d[d].b1[tid];
            if (A.d[d].h_out != nullptr) A.d[d].h_out[c * H2 + tid] = h;      // pre-activation: the backward's ReLU mask
            float a = fmaxf(h, 0.f);
            if (dpos == CWN_HEAD_DROP_FINAL) a *= drop.mul1(((uint64_t)d * (uint64_t)A.C + (uint64_t)c) * (uint64_t)H2 + (uint64_t)tid);
            s += a;
        }
        if (A.mean_final) s = s / (float)nd;
        if (dpos == CWN_HEAD_DROP_LIN2) s *= drop.mul1((uint64_t)c * (uint64_t)H2 + (uint64_t)tid);
        sbuf[tid] = s;
        if (A.s_out != nullptr) A.s_out[c * H2 + tid] = s;
    }
    __syncthreads();

    // ---- 3. out[c] = W2 s + b2: one wave per output, lanes stride the H2 terms, fixed xor tree ------------------
    const int lane = tid & 63, wave = tid >> 6;
    for (int o = wave; o < A.O; o += kHeadThreads / 64) {
        float v = 0.f;
        for (int q = lane; q < H2; q += 64) v = __builtin_fmaf(A.w2[(size_t)o * H2 + q], sbuf[q], v);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) A.out[c * A.O + o] = v + (A.b2 != nullptr ? A.b2[o] : 0.f);
    }
}

// The row sums of LARGE complexes ahead of the head launch (cwn_head_f32: pool_split = P > 1): workgroup (c, p) sums chunk p
// of the rows of complex c in every dimension -- K / 4 lanes a row, row groups side by side, four rows in flight per group, the
// groups' partials summed in group order -- and stores [CWN_HEAD_MAX_DIMS][K] floats.  A REDDIT-like complex has ~4 000 cells
// of 1 KiB (4 layers x 64 under jumping knowledge): one workgroup pulled 4 MB through its CU in 68 us while 224 CUs idled.
// One workgroup per SLOT of the partials (slot_base[d] + lo / chunk + c + j = chunk j of complex c in dimension d; the gaps of
// that numbering -- at most C + 1 per dimension -- leave at once): the slot's complex by a binary search over the `ptr` table
// (in LDS while it fits), then the chunk's 128 rows.  (Until round 6 the grid was (complex, P): a launch lasted as long as the
// LARGEST complex's chunks over P workgroups -- REDDIT-32, sizes 1 : 3 across a batch: 40 us, most workgroups idle.)
constexpr int kPoolPtrLds = 2048;                 // `ptr` entries staged in LDS (int64); beyond: searched in global memory

__global__ __launch_bounds__(kHeadThreads) void head_pool_kernel(HeadArgs A, int64_t n_slots) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = A.K, nd = A.n_dims;
    float* const part = sm;                                  // [NG][K]
    int64_t* const ptr_lds = reinterpret_cast<int64_t*>(sm + kPartFloats);
    const int tid = threadIdx.x;
    const int64_t s = blockIdx.x;
    if (s >= n_slots) return;
    const int d = (nd > 2 && s >= A.slot_base[2]) ? 2 : ((nd > 1 && s >= A.slot_base[1]) ? 1 : 0);
    if (A.d[d].x == nullptr || A.d[d].n_cells <= 0) return;                                  // (uniform)
    const int64_t local = s - A.slot_base[d], n_cells = A.d[d].n_cells, C = A.C;
    const int64_t* cp = A.d[d].cell_ptr;
    if (C + 1 <= kPoolPtrLds) {
        for (int64_t i = tid; i <= C; i += kHeadThreads) ptr_lds[i] = cp[i];
        __syncthreads();
        cp = ptr_lds;
    }
    auto clamp_lo = [&](int64_t v) { return v < 0 ? 0 : (v > n_cells ? n_cells : v); };
    // the last complex whose first slot (lo / chunk + c, increasing in c) is not beyond this one
    int64_t c0 = 0, c1 = C - 1;
    while (c0 < c1) {
        const int64_t m = (c0 + c1 + 1) >> 1;
        if (clamp_lo(cp[m]) / kHeadChunk + m <= local) c0 = m; else c1 = m - 1;
    }
    const int64_t c = c0;
    const int64_t lo = clamp_lo(cp[c]), h1 = cp[c + 1];
    const int64_t hi = h1 < lo ? lo : (h1 > n_cells ? n_cells : h1);
    const int64_t j = local - (lo / kHeadChunk + c);
    if (j < 0 || j >= (hi - lo + kHeadChunk - 1) / kHeadChunk) return;                       // a gap of the numbering (uniform)
    const int G = K / 4, NG = kHeadThreads / G;
    const int g = tid / G, l = tid - g * G;
    const int Kp = A.d[d].n_parts > 1 ? K / A.d[d].n_parts : K;
    const int64_t a = lo + j * kHeadChunk, b = a + kHeadChunk < hi ? a + kHeadChunk : hi;
    if (g < NG) *reinterpret_cast<float4*>(part + (size_t)g * K + 4 * l) = head_group_sum(A.d[d], a, b, g, NG, l, Kp);
    __syncthreads();
    float* const out = A.partials + (size_t)s * K;
    for (int k = tid; k < K; k += kHeadThreads) {
        float t = 0.f;
        for (int q = 0; q < NG; ++q) t += part[(size_t)q * K + k];                            // group order
        out[k] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// head, backward (training): per complex  ds = W2^T g_out  ->  dh_d = ds . [h_d > 0]  ->  dpooled_d = W1_d^T dh_d  ->  every
// row of the complex in x_d gets dpooled_d (/ count for a mean readout).  The weight gradients are sums over the
// complexes -- dW1_d = dh_d^T pooled_d, dW2 = g_out^T s -- and are left to cwn_gemm_tn_f32 on the [C, .] matrices this
// kernel writes (dh_out); here nothing is reduced across workgroups.  16 framework / GEMM launches of a training step
// before (readout backward, two GEMMs and their weight gradients, six mask / multiply kernels).
// ---------------------------------------------------------------------------------------------------------------
struct HeadBwdArgs {
    cwn_head_bwd_dim d[CWN_HEAD_MAX_DIMS];
    const float* w2; const float* g_out;
    int64_t C;
    int32_t n_dims, K, H2, O, mean_readout, mean_final;
    cwn_dropout drop;
    int32_t drop_pos;
    int32_t row_split;                            // P: workgroup (c, p) writes chunk p of the complex's rows
};

__global__ __launch_bounds__(kHeadThreads) void head_bwd_kernel(HeadBwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = A.K, H2 = A.H2, nd = A.n_dims;
    float* const dh = sm;                                          // [3][H2]
    float* const part = dh + CWN_HEAD_MAX_DIMS * H2;               // [slices][K] partial dpooled of ONE dimension (<= 4 * 512)
    float* const dp = part + 4 * kHeadThreads;                     // [K]
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x;
    cwn::Dropout drop;
    drop.init(A.drop);
    const int dpos = drop.on ? A.drop_pos : CWN_HEAD_DROP_NONE;
    // 1. + 2.  ds and the masked dh of every dimension
    if (tid < H2) {
        float ds = 0.f;
        for (int o = 0; o < A.O; ++o) ds = __builtin_fmaf(A.w2[(size_t)o * H2 + tid], A.g_out[c * A.O + o], ds);
        if (dpos == CWN_HEAD_DROP_LIN2) ds *= drop.mul1((uint64_t)c * (uint64_t)H2 + (uint64_t)tid);
        if (A.mean_final) ds = ds / (float)nd;
        for (int d = 0; d < nd; ++d) {
            float v = A.d[d].h[c * H2 + tid] > 0.f ? ds : 0.f;
            if (dpos == CWN_HEAD_DROP_FINAL) v *= drop.mul1(((uint64_t)d * (uint64_t)A.C + (uint64_t)c) * (uint64_t)H2 + (uint64_t)tid);
            dh[d * H2 + tid] = v;
            if (A.d[d].dh_out != nullptr && blockIdx.y == 0) A.d[d].dh_out[c * H2 + tid] = v;
        }
    }
    __syncthreads();
    const int KQ = K / 4, S = kHeadThreads / KQ;                    // a thread: four consecutive k, one slice of the j range
    const int kq = tid % KQ, sl = tid / KQ;
    const int G = KQ, NG = kHeadThreads / G, g = tid / G, l = tid - g * G;
    for (int d = 0; d < nd; ++d) {
        if (A.d[d].dx == nullptr || A.d[d].n_cells == 0) continue;   // uniform
        // 3. dpooled_d[k] = sum_j W1_d[j][k] dh_d[j]   (W1 in its own [H2, K] layout: coalesced over k)
        if (sl < S) {
            const int jb = (H2 + S - 1) / S, j0 = min(H2, sl * jb), j1 = min(H2, j0 + jb);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* w = A.d[d].w1 + (size_t)j0 * K + 4 * kq;
            int j = 0;
            for (; j + 8 <= j1 - j0; j += 8) {
                float4 wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const float4*>(w + (size_t)(j + u) * K);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float t = dh[d * H2 + j0 + j + u];
                    a.x = __builtin_fmaf(wv[u].x, t, a.x); a.y = __builtin_fmaf(wv[u].y, t, a.y);
                    a.z = __builtin_fmaf(wv[u].z, t, a.z); a.w = __builtin_fmaf(wv[u].w, t, a.w);
                }
            }
            for (; j < j1 - j0; ++j) {
                const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)j * K);
                const float t = dh[d * H2 + j0 + j];
                a.x = __builtin_fmaf(wv.x, t, a.x); a.y = __builtin_fmaf(wv.y, t, a.y);
                a.z = __builtin_fmaf(wv.z, t, a.z); a.w = __builtin_fmaf(wv.w, t, a.w);
            }
            *reinterpret_cast<float4*>(part + (size_t)sl * K + 4 * kq) = a;
        }
        __syncthreads();
        int64_t r0 = A.d[d].cell_ptr[c], r1 = A.d[d].cell_ptr[c + 1];
        r0 = r0 < 0 ? 0 : (r0 > A.d[d].n_cells ? A.d[d].n_cells : r0);
        r1 = r1 < r0 ? r0 : (r1 > A.d[d].n_cells ? A.d[d].n_cells : r1);
        if (tid < K) {
            float v = 0.f;
            for (int q = 0; q < S; ++q) v += part[(size_t)q * K + tid];       // fixed order
            if (dpos == CWN_HEAD_DROP_LIN1) v *= drop.mul1(((uint64_t)d * (uint64_t)A.C + (uint64_t)c) * (uint64_t)K + (uint64_t)tid);
            if (A.mean_readout) v = v / (float)(r1 - r0 > 0 ? r1 - r0 : 1);
            dp[tid] = v;
        }
        __syncthreads();
        // 4. every row of the complex gets it (this workgroup: its chunk of the rows; a concatenation: every block to its matrix)
        if (g < NG) {
            const float4 v = *reinterpret_cast<const float4*>(dp + 4 * l);
            const int64_t chunk = (r1 - r0 + A.row_split - 1) / A.row_split;
            const int64_t a = r0 + (int64_t)blockIdx.y * chunk, b = a + chunk < r1 ? a + chunk : r1;
            float* base = A.d[d].dx;
            int col = 4 * l;
            if (A.d[d].n_parts > 1) {
                const int Kp = K / A.d[d].n_parts, q = col / Kp;
                base = q == 0 ? A.d[d].dx : A.d[d].dx_more[q - 1];
                col -= q * Kp;
            }
            for (int64_t r = a + g; r < b; r += NG)
                cwn::store_result4(base + r * A.d[d].lddx + col, v.x, v.y, v.z, v.w);
        }
        __syncthreads();                       // `part` / `dp` are reused by the next dimension
    }
}

}  // namespace

extern "C" int cwn_head_bwd_f32(const cwn_head_bwd_dim* dims, int n_dims, int64_t C, int32_t K, int32_t H2,
                                int32_t mean_readout, int32_t mean_final, const float* w2, int32_t O, const float* g_out,
                                const cwn_dropout* drop, int32_t drop_pos, int32_t row_split, cwn_stream_t stream_) {
    if (row_split < 1) row_split = 1;
    if (row_split > 64) return CWN_ERR_BAD_ARG;
    if (dims == nullptr || n_dims < 1 || n_dims > CWN_HEAD_MAX_DIMS || C < 0 || O < 1) return CWN_ERR_BAD_ARG;
    if (K < 4 || (K & 3) != 0 || K > 4 * kHeadThreads || H2 < 4 || (H2 & 3) != 0 || H2 > kHeadThreads) return CWN_ERR_BAD_ARG;
    if (C == 0) return CWN_OK;
    if (C >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    if (w2 == nullptr || g_out == nullptr) return CWN_ERR_BAD_ARG;
    HeadBwdArgs A{};
    for (int d = 0; d < n_dims; ++d) {
        const cwn_head_bwd_dim& D = dims[d];
        if (D.h == nullptr || D.n_cells < 0) return CWN_ERR_BAD_ARG;
        const int np = D.n_parts > 1 ? D.n_parts : 1;
        if (np > CWN_HEAD_MAX_PARTS || K % np != 0 || ((K / np) & 3) != 0) return CWN_ERR_BAD_ARG;
        if (D.dx != nullptr && D.n_cells > 0 && (D.cell_ptr == nullptr || D.w1 == nullptr || D.lddx < K / np || (D.lddx & 3) != 0))
            return CWN_ERR_BAD_ARG;
        if (!al16(D.dx) || !al16(D.w1)) return CWN_ERR_ALIGN;
        for (int q = 1; q < np; ++q)
            if (D.dx != nullptr && (D.dx_more[q - 1] == nullptr || !al16(D.dx_more[q - 1]))) return CWN_ERR_BAD_ARG;
        A.d[d] = D;
    }
    A.w2 = w2; A.g_out = g_out;
    if (drop != nullptr && drop->state != nullptr && drop->p > 0.f) {
        if (!(drop->p < 1.f) || drop_pos < CWN_HEAD_DROP_LIN1 || drop_pos > CWN_HEAD_DROP_LIN2) return CWN_ERR_BAD_ARG;
        A.drop = *drop;
        A.drop_pos = drop_pos;
    }
    A.C = C;
    A.n_dims = n_dims; A.K = K; A.H2 = H2; A.O = O;
    A.mean_readout = mean_readout ? 1 : 0;
    A.mean_final = mean_final ? 1 : 0;
    A.row_split = row_split;
    const size_t lds = ((size_t)CWN_HEAD_MAX_DIMS * H2 + 4 * kHeadThreads + K) * sizeof(float);
    head_bwd_kernel<<<dim3((unsigned)C, (unsigned)row_split), dim3(kHeadThreads), lds, (hipStream_t)stream_>>>(A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_embed_front_f32(const cwn_embed_table* v_tab, int64_t n0, float* x0, const cwn_embed_table* e_tab,
                                   int64_t n1, float* x1, const int32_t* rowptr1, const int32_t* col1, int64_t nb1, int64_t n2,
                                   float* x2, const int32_t* rowptr2, const int32_t* col2, int64_t nb2, int32_t H,
                                   int32_t halve, int32_t* err_flag, const int64_t* n_dev, cwn_stream_t stream_) {
    if (v_tab == nullptr || n0 < 0 || n1 < 0 || n2 < 0 || H <= 0 || (H & 3) != 0 || err_flag == nullptr) return CWN_ERR_BAD_ARG;
    if (n0 + n1 + n2 == 0) return CWN_OK;
    if (nb1 < 0 || nb2 < 0) return CWN_ERR_BAD_ARG;
    if (nb1 == 0) rowptr1 = nullptr, col1 = nullptr;          // a plan without entries reduces nothing: zeros
    if (nb2 == 0) rowptr2 = nullptr, col2 = nullptr;
    if (!table_ok(*v_tab) || (e_tab != nullptr && !table_ok(*e_tab))) return CWN_ERR_BAD_ARG;
    if ((n0 > 0 && x0 == nullptr) || (n1 > 0 && x1 == nullptr) || (n2 > 0 && x2 == nullptr)) return CWN_ERR_BAD_ARG;
    if ((rowptr1 == nullptr) != (col1 == nullptr) || (rowptr2 == nullptr) != (col2 == nullptr)) return CWN_ERR_BAD_ARG;
    // a reduction onto the edges / rings walks into the vertex table: it needs vertices (and the rings need the edges' CSR)
    if ((rowptr1 != nullptr || rowptr2 != nullptr) && n0 == 0) return CWN_ERR_BAD_ARG;
    if (rowptr2 != nullptr && n1 == 0) return CWN_ERR_BAD_ARG;
    if (!(al16(v_tab->W) && al16(x0) && al16(x1) && al16(x2) && (e_tab == nullptr || al16(e_tab->W)))) return CWN_ERR_ALIGN;
    FrontArgs A{};
    A.tv = *v_tab;
    if (e_tab != nullptr) A.te = *e_tab;
    A.has_te = e_tab != nullptr ? 1 : 0;
    A.x0 = x0; A.x1 = x1; A.x2 = x2;
    A.rowptr1 = rowptr1; A.col1 = col1; A.rowptr2 = rowptr2; A.col2 = col2;
    A.n0 = n0; A.n1 = n1; A.n2 = n2;
    A.H = H;
    A.halve = halve ? 1 : 0;
    A.err = err_flag;
    A.n_dev = n_dev;
    int G = 1;
    while (G < H / 4 && G < 64) G <<= 1;
    A.G = G;
    const int64_t rows = n0 + n1 + n2, per = 256 / G, blocks = (rows + per - 1) / per;
    if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    const bool single = A.tv.cols == 1 && (!A.has_te || A.te.cols == 1);
    if (single) {
        embed_front_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_>>>(A);
        return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
    }
    // several tables per cell type: the embeddings, then the rows that reduce from x0 (front_reduce_kernel)
    const int64_t e_rows = n0 + (A.has_te ? n1 : 0), e_blocks = (e_rows + per - 1) / per;
    if (e_rows > 0) embed_pair_kernel<<<dim3((unsigned)e_blocks), dim3(256), 0, (hipStream_t)stream_>>>(A);
    const int64_t r_rows = (A.has_te ? 0 : n1) + n2, r_blocks = (r_rows + per - 1) / per;
    if (r_rows > 0) front_reduce_kernel<<<dim3((unsigned)r_blocks), dim3(256), 0, (hipStream_t)stream_>>>(A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int64_t cwn_head_pool_floats(const cwn_head_dim* dims, int n_dims, int64_t C, int32_t K) {
    if (dims == nullptr || n_dims < 1 || n_dims > CWN_HEAD_MAX_DIMS || C < 0 || K <= 0) return 0;
    int64_t slots = 0;
    for (int d = 0; d < n_dims; ++d) slots += (dims[d].n_cells > 0 ? dims[d].n_cells : 0) / kHeadChunk + C + 1;
    return slots * K;
}

extern "C" int cwn_head_f32(const cwn_head_dim* dims, int n_dims, int64_t C, int32_t K, int32_t H2, int32_t mean_readout,
                            int32_t mean_final, const float* w2, const float* b2, int32_t O, float* out, float* s_out,
                            const cwn_dropout* drop, int32_t drop_pos, float* pool_partials, int64_t pool_partials_floats,
                            int32_t pool_split, cwn_stream_t stream_) {
    if (pool_partials == nullptr || pool_split < 1) { pool_partials = nullptr; pool_split = 1; }
    if (pool_split > 64 || !al16(pool_partials)) return CWN_ERR_BAD_ARG;
    if (dims == nullptr || n_dims < 1 || n_dims > CWN_HEAD_MAX_DIMS || C < 0 || O < 1) return CWN_ERR_BAD_ARG;
    // K / 4 lanes a row inside 512 threads; output j by thread j
    if (K < 4 || (K & 3) != 0 || K > 4 * kHeadThreads || H2 < 4 || (H2 & 3) != 0 || H2 > kHeadThreads) return CWN_ERR_BAD_ARG;
    if (C == 0) return CWN_OK;
    if (C >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    if (w2 == nullptr || out == nullptr) return CWN_ERR_BAD_ARG;
    HeadArgs A{};
    for (int d = 0; d < n_dims; ++d) {
        const cwn_head_dim& D = dims[d];
        if (D.w1t == nullptr || D.n_cells < 0) return CWN_ERR_BAD_ARG;
        const int np = D.n_parts > 1 ? D.n_parts : 1;
        if (np > CWN_HEAD_MAX_PARTS || K % np != 0 || ((K / np) & 3) != 0) return CWN_ERR_BAD_ARG;
        if (D.x != nullptr && D.n_cells > 0 && (D.cell_ptr == nullptr || D.ldx < K / np || (D.ldx & 3) != 0)) return CWN_ERR_BAD_ARG;
        if (!al16(D.x) || !al16(D.w1t)) return CWN_ERR_ALIGN;
        for (int q = 1; q < np; ++q)
            if (D.x != nullptr && (D.x_more[q - 1] == nullptr || !al16(D.x_more[q - 1]))) return CWN_ERR_BAD_ARG;
        A.d[d] = D;
    }
    A.w2 = w2; A.b2 = b2; A.out = out;
    A.s_out = s_out;
    if (drop != nullptr && drop->state != nullptr && drop->p > 0.f) {
        if (!(drop->p < 1.f) || drop_pos < CWN_HEAD_DROP_LIN1 || drop_pos > CWN_HEAD_DROP_LIN2) return CWN_ERR_BAD_ARG;
        A.drop = *drop;
        A.drop_pos = drop_pos;
    }
    A.C = C;
    A.n_dims = n_dims; A.K = K; A.H2 = H2; A.O = O;
    A.mean_readout = mean_readout ? 1 : 0;
    A.mean_final = mean_final ? 1 : 0;
    A.partials = pool_partials;
    A.pool_split = pool_split;
    if (pool_partials != nullptr) {
        int64_t slots = 0;
        for (int d = 0; d < n_dims; ++d) {
            A.slot_base[d] = slots;
            slots += dims[d].n_cells / kHeadChunk + C + 1;
        }
        if (pool_partials_floats < slots * K) return CWN_ERR_WORKSPACE;
    }
    const size_t lds = head_lds_floats(K, H2) * sizeof(float);
    if (lds > 64 * 1024) return CWN_ERR_BAD_ARG;
    if (pool_partials != nullptr) {
        int64_t n_slots = 0;
        for (int d = 0; d < n_dims; ++d) n_slots += dims[d].n_cells / kHeadChunk + C + 1;
        if (n_slots >= INT32_MAX) return CWN_ERR_TOO_LARGE;
        head_pool_kernel<<<dim3((unsigned)n_slots), dim3(kHeadThreads), (size_t)kPartFloats * sizeof(float) + kPoolPtrLds * sizeof(int64_t),
                           (hipStream_t)stream_>>>(A, n_slots);
    }
    head_kernel<<<dim3((unsigned)C), dim3(kHeadThreads), lds, (hipStream_t)stream_>>>(A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
