// cwn_layer_bwd.hip -- the BACKWARD of one SparseCIN propagate step of a layer in one launch, complex-blocked.
//
// Forward (cwn_layer.hip; mp/layers.py:184-192, 290-295, 333-342), per dimension d:
//     out_up_d[i] = sum_{p: dst_p = i} relu(Y1_d[src_p] + Y2_d[cof_p]) + (1 + eps1_d) x_d[i]
//     out_b_d[i]  = sum_{b in boundary(i)} x_{d-1}[b]               + (1 + eps2_d) x_d[i]
//     Y1_d = x_d W_d[:, :F]^T + bias_d,    Y2_d = x_{d+1} W_d[:, F:]^T
// Backward, given gU_d = dL/d out_up_d and gB_d = dL/d out_b_d (autograd of the reference's modules):
//     m_p     = gU_d[dst_p] * [Y1_d[src_p] + Y2_d[cof_p] > 0]                         per entry
//     gY1_d[j] = sum_{p: src_p = j} m_p          gY2_d[c] = sum_{p: cof_p = c} m_p     (-> the weight gradients)
//     dx_d    = (1 + eps1_d) gU_d + (1 + eps2_d) gB_d + gY1_d W_d[:, :F] + gY2_{d-1} W_{d-1}[:, F:]
//               + sum_{i in dim d+1: b in boundary(i)} gB_{d+1}[i]
// The training step ran this as a transposed CSR aggregation (gY1, gY2, boundary transposes, self terms: one launch
// over plans that had to be built per batch), a transposed-weight GEMM that added gY W onto it, and framework adds.
// Here the workgroup that owns a range of complexes for a GEMM dimension g (the SAME item table as the forward launch)
// gathers, for every row of gY1 | gY2 it owns, the masked gradients of the upper entries that name the row as source or
// coface (out of LDS, in entry order: no float atomics, no sort, no transposed plan), writes them out for the weight-gradient GEMM, multiplies them by the transposed message weight on the matrix
// cores (the forward's exact three-way bf16 split, cwn_split.h) and adds every piece of dx -- products, self terms,
// boundary transposes -- to the caller's ZEROED dx matrices with fp32 atomics: a cell receives pieces from up to three
// workgroups (its own set's, the set below's second product, the set above's boundary entries).  Sums therefore run in
// arrival order: results agree with the streaming path to rounding, not bit for bit (as the weight gradients already do).
//
// STATE (round 3): correct -- dx, gY1, gY2 against float64 autograd, and whole training steps through it -- and NOT the
// default: 25 us per launch (+ the fill of dx) at the ZINC batch of 128 (256 items) against ~34 us for the three launches it
// replaces; the training step reads the same with and without it.  What was measured on the way (tools/ubench_layer_bwd.py
// with CWN_LBWD_DBG): the first form scattered the masked gradients with fp32 LDS atomics -- ds_add_f32 runs at ~100 cycles
// per wave instruction: 31 us of a 56-us launch; written with a branch per entry the compiler waited for each entry's
// indices before requesting the next one's; atomics for ABSENT entries (zeros onto row 0) serialised on that one address
// (95 us); walking the entries four keys per LDS read and a branch per key cost 15 us where one key per lane and a ballot
// costs 6.  Left: ~10 us of staging and launch, 6 of walk, 6 of global fp32 atomics into dx, 2 of MFMA.  Next: one item per
// complex over all dimensions gives every dx row a single owner (plain stores, no fill).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include <stdlib.h>
#include "../../include/cwn_hip.h"
#include "cwn_split.h"

namespace {

using cwn::frag_cd;

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;

enum { I_FLAGS = 0, I_G, I_GR0, I_GN, I_CR0, I_CN, I_UE0, I_UNE, I_NT, I_TASK0, I_R1 = 23, I_ROWS };
enum { T_DIM = 0, T_R0, T_N, T_BE0, T_BNE, T_SR0, T_SN, T_INTS };

struct BwdArgs {
    cwn_layer_bwd_dim d[CWN_LAYER_MAX_DIMS];
    const int32_t* items;
    int32_t* err;
    int32_t rows_cap;
    int32_t n_dims;
    int32_t dbg;          // timing experiments (CWN_LBWD_DBG): 1 no entry scatter, 2 no MFMA / product atomics, 4 no self terms / boundary transposes, 8 no gY store
};

template <int F> struct Geo {
    static constexpr int kPlaneStride = F + 8;          // bf16 elements per plane row (fragment reads conflict-free)
    static constexpr int kYStride = F + 4;              // floats per gY row
    static constexpr int kKS = F / 32;
    static constexpr int kNCT = F / 16;
    static constexpr int kWPC = kWaves / kNCT / 2;      // waves sharing a column tile of ONE product (row-tile parity)
    static constexpr int kG = F / 4;                    // lanes per row
    static constexpr int kNG = kThreads / kG;           // rows / entries per round
    __host__ __device__ static constexpr size_t gy_bytes(int rows) { return (size_t)rows * kYStride * 4; }
    __host__ __device__ static constexpr size_t planes_bytes(int rows) { return (size_t)3 * rows * kPlaneStride * 2; }
    // + the item's upper entries as three int arrays (source, destination, coface; padded to 4)
    __host__ __device__ static constexpr size_t lds_bytes(int rows) { return gy_bytes(rows) + planes_bytes(rows) + (size_t)3 * CWN_LAYER_MAX_ENTRIES * 4; }
};

__device__ __forceinline__ void atomic_add4(float* p, const float4& v) {
    atomicAdd(p, v.x);
    atomicAdd(p + 1, v.y);
    atomicAdd(p + 2, v.z);
    atomicAdd(p + 3, v.w);
}

template <int F>
__global__ __launch_bounds__(kThreads) void layer_bwd_kernel(BwdArgs A) {
    using G = Geo<F>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int32_t* const it = A.items + (size_t)blockIdx.x * CWN_LAYER_ITEM_INTS;
    const int flags = it[I_FLAGS], nt = it[I_NT];
    if (nt == 0) return;
    if ((flags & CWN_LAYER_ITEM_BIG) != 0) {            // streamed complexes: the caller keeps those batches on the CSR path
        if (tid == 0) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
        return;
    }
    const bool has_gemm = (flags & 1) != 0;
    const int g = it[I_G], g_r0 = it[I_GR0], g_n = it[I_GN], c_r0 = it[I_CR0], c_n = it[I_CN];
    const int ue0 = it[I_UE0], une = it[I_UNE];
    const int R1 = it[I_R1], rows_pad = it[I_ROWS];
    const int rows_cap = A.rows_cap;
    float* const gY = reinterpret_cast<float*>(smem);                                          // [rows_cap][F + 4]
    uint16_t* const planes = reinterpret_cast<uint16_t*>(smem + G::gy_bytes(rows_cap));       // [3][rows_cap][F + 8]
    const size_t plane = (size_t)rows_cap * G::kPlaneStride;
    const int gl = tid % G::kG, gq = tid / G::kG, f = gl * 4;

    if (has_gemm) {
        if (rows_pad > rows_cap || R1 > rows_pad || g + 1 >= A.n_dims) {       // table / launch mismatch
            if (tid == 0) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
            return;
        }
        const cwn_layer_bwd_dim& Dg = A.d[g];
        const cwn_layer_bwd_dim& Dc = A.d[g + 1];
        // this wave's slice of the TRANSPOSED packed weight (cwn_layer_pack_weights_many_f32 with transposed = 1; the
        // forward's chunk order): product h = 0 is gY1 W[:, :F], h = 1 is gY2 W[:, F:]; requested first, used last
        const int ct = wave % G::kNCT, w2 = wave / G::kNCT;
        const int my_h = w2 & 1, rt_par = w2 >> 1;
        uint4 wsp[G::kKS][3];
        {
            const unsigned char* wp = reinterpret_cast<const unsigned char*>(Dg.wt_packed) + (size_t)ct * 1024 + lane * 16;
#pragma unroll
            for (int ks = 0; ks < G::kKS; ++ks)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    wsp[ks][pl] = *reinterpret_cast<const uint4*>(wp + (size_t)(((ks * 3 + pl) * 2 + my_h) * G::kNCT) * 1024);
        }
        // ---- 1. stage: Y1 | Y2 rows (fp32) into S, the rows of gU into the plane region (free until phase 3), the item's
        //         upper entries as LOCAL row numbers (structure of arrays: source, destination, coface) --------------------------
        float* const S = gY;                                                      // [rows_cap][F + 4]: Y1 rows, from R1 the Y2 rows
        float* const T = reinterpret_cast<float*>(planes);                        // [g_n][F + 4]: gU rows
        int* const ej = reinterpret_cast<int*>(smem + G::gy_bytes(rows_cap) + G::planes_bytes(rows_cap));
        const int une4 = (une + 3) & ~3;
        int* const ei = ej + une4;
        int* const ec = ei + une4;
        const bool have_gu = Dg.g_up != nullptr && !(A.dbg & 1);
        {
            const int64_t E = Dg.e_up;
            int bad = 0;
            for (int p = tid; p < une4; p += kThreads) {
                const int q = min(p, une - 1);
                const int64_t jj = Dg.up_index[ue0 + q] - g_r0, ii = Dg.up_index[E + ue0 + q] - g_r0, cc = Dg.up_shared[ue0 + q] - c_r0;
                const int ok = (int)((uint64_t)jj < (uint64_t)g_n) & (int)((uint64_t)ii < (uint64_t)g_n) & (int)((uint64_t)cc < (uint64_t)c_n);
                const int live = (int)(p < une);
                bad |= live & (ok ^ 1);
                ej[p] = (live & ok) ? (int)jj : -1;                               // -1: matches no row
                ei[p] = (live & ok) ? (int)ii : 0;
                ec[p] = (live & ok) ? (int)cc : -1;
            }
            if (bad) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
            for (int r = gq; r < g_n; r += G::kNG) {
                *reinterpret_cast<float4*>(S + (size_t)r * G::kYStride + f) = *reinterpret_cast<const float4*>(Dg.y1 + (size_t)(g_r0 + r) * F + f);
                if (have_gu)
                    *reinterpret_cast<float4*>(T + (size_t)r * G::kYStride + f) = *reinterpret_cast<const float4*>(Dg.g_up + (size_t)(g_r0 + r) * F + f);
            }
            for (int r = gq; r < c_n; r += G::kNG)
                *reinterpret_cast<float4*>(S + (size_t)(R1 + r) * G::kYStride + f) = *reinterpret_cast<const float4*>(Dc.y2 + (size_t)(c_r0 + r) * F + f);
        }
        __syncthreads();
        // ---- 2. every lane group OWNS the rows gq, gq + kNG, ... of gY1 | gY2 and walks the item's entries for each: those
        //         whose source (gY1) or coface (gY2) is the row add their masked gradient, out of LDS, in entry order.  No
        //         float atomics (the first form scattered with ds_add_f32: ~100 cycles per wave instruction, 31 us of a
        //         56-us launch), no sort, deterministic; the walk is rows x entries compares -- a few hundred per lane group.
        constexpr int kMaxOwn = (CWN_LAYER_GEMM_ROWS(F) + G::kNG - 1) / G::kNG;
        float4 own[kMaxOwn];
#pragma unroll
        for (int k = 0; k < kMaxOwn; ++k) {
            own[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int r = gq + k * G::kNG;
            const bool is1 = r < g_n, is2 = r >= R1 && r - R1 < c_n;
            if (have_gu && (is1 || is2) && !(A.dbg & 16)) {
                const int* const key = is1 ? ej : ec;
                const int want = is1 ? r : r - R1;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                // the lanes of the group test kG entries AT ONCE (one key each, a ballot gives the group its matches); only
                // the matches -- a cell's degree, two or three -- are then walked, in entry order.  (Walked entry by entry
                // the group paid an LDS round trip per four keys and a branch per key: 15 us of the launch.)
                const int gsh = (lane / G::kG) * G::kG;                      // this group's bits of the wave's ballot
                for (int e0 = 0; e0 < une4; e0 += G::kG) {
                    const int e = e0 + gl;
                    const int kv = e < une4 ? key[e] : -2;
                    unsigned long long bal = __ballot(kv == want);
                    unsigned m = (unsigned)((bal >> gsh) & (G::kG == 32 ? 0xffffffffull : 0xffffull));
                    while (m != 0u) {
                        const int q = e0 + __builtin_ctz(m);
                        m &= m - 1u;
                        const int j = ej[q], i = ei[q], c = ec[q];
                        const float4 a = *reinterpret_cast<const float4*>(S + (size_t)j * G::kYStride + f);
                        const float4 b = *reinterpret_cast<const float4*>(S + (size_t)(R1 + c) * G::kYStride + f);
                        const float4 mm = *reinterpret_cast<const float4*>(T + (size_t)i * G::kYStride + f);
                        acc.x += a.x + b.x > 0.f ? mm.x : 0.f;
                        acc.y += a.y + b.y > 0.f ? mm.y : 0.f;
                        acc.z += a.z + b.z > 0.f ? mm.z : 0.f;
                        acc.w += a.w + b.w > 0.f ? mm.w : 0.f;
                    }
                }
                own[k] = acc;
            }
        }
        __syncthreads();                     // every group is done with S and T: the planes may overwrite T
        // ---- 3. gY out (the weight-gradient GEMM reads it) and into the bf16 planes ------------------------------------
#pragma unroll
        for (int k = 0; k < kMaxOwn; ++k) {
            const int r = gq + k * G::kNG;
            if (r < rows_pad) {
                const float4 v = own[k];
                if (r < g_n) {
                    if (Dg.gy1 != nullptr && !(A.dbg & 8)) *reinterpret_cast<float4*>(Dg.gy1 + (size_t)(g_r0 + r) * F + f) = v;
                } else if (r >= R1 && r - R1 < c_n) {
                    if (Dc.gy2 != nullptr && !(A.dbg & 8)) *reinterpret_cast<float4*>(Dc.gy2 + (size_t)(c_r0 + r - R1) * F + f) = v;
                }
                uint2 ph, pm, pl;
                cwn::split4(v, ph, pm, pl);
                uint16_t* dst = planes + (size_t)r * G::kPlaneStride + f;
                *reinterpret_cast<uint2*>(dst) = ph;
                *reinterpret_cast<uint2*>(dst + plane) = pm;
                *reinterpret_cast<uint2*>(dst + 2 * plane) = pl;
            }
        }
        __syncthreads();
        // ---- 4. dx_g += gY1 W[:, :F],  dx_{g+1} += gY2 W[:, F:] on the matrix cores ---------------------------------------
        // Operand roles NOT swapped here (A = rows of gY, B = rows of the transposed weight = output columns): a lane then
        // holds rows 4 kq + reg of output column l15, so one atomic instruction adds 16 CONSECUTIVE floats of four rows
        // (whole 64-byte segments) -- with the forward's roles it would touch 16 rows x 4 bytes x 4.  Same six terms.
        {
            using cwn::as_frag;
            const int T1 = (g_n + 15) >> 4, T2 = (c_n + 15) >> 4;
            const int rt0 = my_h == 0 ? 0 : R1 / 16, rt1 = rt0 + (my_h == 0 ? T1 : T2);
            const int first = rt0 + ((rt_par - rt0) % G::kWPC + G::kWPC) % G::kWPC;
            float* const dx = (my_h == 0 ? Dg.dx + (size_t)g_r0 * F : Dc.dx + (size_t)c_r0 * F) + ct * 16 + l15;
            const int n_rows = my_h == 0 ? g_n : c_n;
            for (int rt = first; rt < rt1 && !(A.dbg & 2); rt += G::kWPC) {
                frag_cd acc = {0.f, 0.f, 0.f, 0.f};
                const uint16_t* p0 = planes + (size_t)(rt * 16 + l15) * G::kPlaneStride + kq * 8;
#pragma unroll
                for (int ks = 0; ks < G::kKS; ++ks) {
                    const uint4 xh = *reinterpret_cast<const uint4*>(p0 + ks * 32);
                    const uint4 xm = *reinterpret_cast<const uint4*>(p0 + plane + ks * 32);
                    const uint4 xl = *reinterpret_cast<const uint4*>(p0 + 2 * plane + ks * 32);
                    const uint4 &wh = wsp[ks][0], &wm = wsp[ks][1], &wl = wsp[ks][2];
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xh), as_frag(wl), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xl), as_frag(wh), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xm), as_frag(wm), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xh), as_frag(wm), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xm), as_frag(wh), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xh), as_frag(wh), acc, 0, 0, 0);
                }
                // D[i][j]: i = gY row 4 kq + reg, j = output column l15
                const int row = (rt - rt0) * 16 + 4 * kq;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row + r < n_rows) atomicAdd(dx + (size_t)(row + r) * F, acc[r]);
            }
        }
    }
    // ---- 5. self terms and boundary transposes of the item's tasks -----------------------------------------------------
    // lane gl of a lane group takes columns gl, gl + kG, gl + 2 kG, gl + 3 kG: an atomic instruction then adds kG
    // CONSECUTIVE floats of a row (float4 lanes would each touch 4 bytes of every 16).  Loads level by level, branch-free.
    for (int t = 0; t < nt && t < 2 && !(A.dbg & 4); ++t) {
        const int o = I_TASK0 + t * T_INTS;
        const int dt = it[o + T_DIM], r0 = it[o + T_R0], n = it[o + T_N], be0 = it[o + T_BE0], bne = it[o + T_BNE];
        const int sr0 = it[o + T_SR0], sn = it[o + T_SN];
        if (dt < 0 || dt >= A.n_dims) continue;
        const cwn_layer_bwd_dim& D = A.d[dt];
        const float e1 = D.eps1 != nullptr ? *D.eps1 : 0.f, e2 = D.eps2 != nullptr ? *D.eps2 : 0.f;
        const float s1 = D.g_up != nullptr ? 1.0f + e1 : 0.f, s2 = D.g_b != nullptr ? 1.0f + e2 : 0.f;
        const float* const gup = (D.g_up != nullptr ? D.g_up : D.dx) + (size_t)r0 * F + gl;      // (never NULL: a masked read)
        const float* const gbp = (D.g_b != nullptr ? D.g_b : D.dx) + (size_t)r0 * F + gl;
        float* const dxt = D.dx + (size_t)r0 * F + gl;
        for (int r = gq; r < n; r += G::kNG) {
            float u[4], w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u[q] = gup[(size_t)r * F + q * G::kG];
                w[q] = gbp[(size_t)r * F + q * G::kG];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(dxt + (size_t)r * F + q * G::kG, s1 * u[q] + s2 * w[q]);
        }
        if (bne > 0 && dt > 0 && D.g_b != nullptr && D.b_index != nullptr) {
            float* const dxs = A.d[dt - 1].dx + gl;
            const float* const gb = D.g_b + gl;
            const int64_t* const bs = D.b_index + be0;
            const int64_t* const bd = D.b_index + D.n_b + be0;
            int bad = 0;
            for (int p0 = gq; p0 < bne; p0 += 4 * G::kNG) {
                int64_t b[4], i[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = min(p0 + u * G::kNG, bne - 1);
                    b[u] = bs[q];
                    i[u] = bd[q];
                }
                float on[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ok = (int)((uint64_t)(i[u] - r0) < (uint64_t)n) & (int)((uint64_t)(b[u] - sr0) < (uint64_t)sn);
                    const int live = (int)(p0 + u * G::kNG < bne);
                    bad |= live & (ok ^ 1);
                    on[u] = (live & ok) ? 1.0f : 0.0f;
                    b[u] = ok ? b[u] : sr0;
                    i[u] = ok ? i[u] : r0;
                }
                float m[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) m[u][q] = gb[(size_t)i[u] * F + q * G::kG];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (on[u] != 0.f) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) atomicAdd(dxs + (size_t)b[u] * F + q * G::kG, m[u][q]);
                    }
            }
            if (bad) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
        }
    }
}

inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; }

template <int F>
int launch(const BwdArgs& A, int64_t n_items, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr = hipSuccess;
    std::call_once(once, [] {
        attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_bwd_kernel<F>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr != hipSuccess) return CWN_ERR_LAUNCH;
    layer_bwd_kernel<F><<<dim3((unsigned)n_items), dim3(kThreads), Geo<F>::lds_bytes(A.rows_cap), stream>>>(A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

}  // namespace

extern "C" size_t cwn_layer_bwd_lds_bytes(int32_t F, int32_t max_gemm_rows) {
    if ((F != 64 && F != 128) || max_gemm_rows < 16 || max_gemm_rows % 16 != 0) return 0;
    const size_t b = F == 128 ? Geo<128>::lds_bytes(max_gemm_rows) : Geo<64>::lds_bytes(max_gemm_rows);
    return b <= 160 * 1024 ? b : 0;
}

extern "C" int cwn_layer_bwd_f32(const cwn_layer_bwd_dim* dims, int n_dims, int32_t F, const cwn_layer_plan* plan,
                                 int32_t* err_flag, cwn_stream_t stream_) {
    if (dims == nullptr || plan == nullptr || n_dims < 1 || n_dims > CWN_LAYER_MAX_DIMS || plan->n_items < 0)
        return CWN_ERR_BAD_ARG;
    if (F != 64 && F != 128) return CWN_ERR_BAD_ARG;
    if (plan->variant != 0 || plan->n_big != 0) return CWN_ERR_BAD_ARG;           // the 16-wave form's tables without BIG records
    const int64_t n_items = plan->n_items;
    if (n_items == 0) return CWN_OK;
    if (plan->items == nullptr || err_flag == nullptr) return CWN_ERR_BAD_ARG;
    if (n_items >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    if (cwn_layer_bwd_lds_bytes(F, plan->max_gemm_rows) == 0) return CWN_ERR_TOO_LARGE;
    if (!al16(plan->items)) return CWN_ERR_ALIGN;
    BwdArgs A{};
    for (int d = 0; d < n_dims; ++d) {
        const cwn_layer_bwd_dim& D = dims[d];
        if (D.n_cells < 0 || D.e_up < 0 || D.n_b < 0) return CWN_ERR_BAD_ARG;
        if (D.n_cells > 0 && D.dx == nullptr) return CWN_ERR_BAD_ARG;
        if (D.out_bn.slots != nullptr) return CWN_ERR_BAD_ARG;       // (the BatchNorm sums are the owner form's: a dx row has one writer there)
        if (D.e_up > 0 && (D.up_index == nullptr || D.up_shared == nullptr || D.wt_packed == nullptr || D.y1 == nullptr ||
                           d + 1 >= n_dims || dims[d + 1].y2 == nullptr))
            return CWN_ERR_BAD_ARG;
        if (D.n_b > 0 && (D.b_index == nullptr || d == 0)) return CWN_ERR_BAD_ARG;
        if (!(al16(D.g_up) && al16(D.g_b) && al16(D.y1) && al16(D.y2) && al16(D.dx) && al16(D.gy1) && al16(D.gy2) &&
              al16(D.wt_packed)))
            return CWN_ERR_ALIGN;
        if (plan->cells_end[d] < 0 || plan->cells_end[d] > D.n_cells || plan->up_end[d] < 0 || plan->up_end[d] > D.e_up ||
            plan->b_end[d] < 0 || plan->b_end[d] > D.n_b)
            return CWN_ERR_BAD_ARG;
        A.d[d] = D;
    }
    A.items = plan->items;
    A.err = err_flag;
    A.rows_cap = plan->max_gemm_rows;
    A.n_dims = n_dims;
    static const int dbg = getenv("CWN_LBWD_DBG") ? atoi(getenv("CWN_LBWD_DBG")) : 0;
    A.dbg = dbg;
    return F == 128 ? launch<128>(A, n_items, (hipStream_t)stream_) : launch<64>(A, n_items, (hipStream_t)stream_);
}
