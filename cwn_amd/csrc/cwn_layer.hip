// cwn_layer.hip -- one SparseCIN propagate step of a layer in ONE launch, complex-blocked.
//
// What it replaces (reference): the three propagate calls of SparseCINConv.forward
// (mp/layers.py:333-342 -> :184-192 -> mp/cell_mp.py:357-392) with the coboundary message
// ReLU(Linear(cat(x_j, up_attr))) (mp/layers.py:290-295), the up_attr gather of
// data/complex.py:579-580, the zero fills of mp/cell_mp.py:517-522 and the self terms of
// mp/layers.py:191-192.  Round 1 ran this as CSR build + grouped GEMM (Y1, Y2 to HBM) + aggregate
// (Y1, Y2 read back): 2 launches per layer + 1 per batch, 15 us per layer at ZINC-128, the GEMM on
// 161 of 256 CUs.
//
// A batched complex is block-diagonal and per-complex contiguous (data/complex.py:148-169), a
// ZINC-like complex is ~55 cells x 512 B.  So a workgroup (512 threads, 8 waves) OWNS a range of
// complexes for one "GEMM dimension" g (plus, as a second task, the top dimension that has no
// upper adjacency) and never leaves the CU's LDS between the dense and the sparse half:
//
//   1. item record (96 B, scalar loads)                                  1st dependent round trip
//   2. all global loads of the item at once: its COO entries (int64, as  2nd (and last) round trip
//      delivered), its rows of x_g and x_{g+1}, this wave's slice of W    before the epilogue gathers
//   3. COO -> LDS, stable counting rank by destination (P lanes per entry scan the keys, combined
//      by shuffles) -> per-item CSR in LDS (rowptr, col, aux as 16-bit LOCAL row numbers)
//   4. x rows -> exact 3-way bf16 split (cwn_split.h) -> three bf16 planes in LDS
//   5. boundary stream + self terms of every task (global gathers of rows this CU just touched)
//   6. Y1 = x_g W[:, :F]^T + b,  Y2 = x_{g+1} W[:, F:]^T : v_mfma_f32_16x16x32_bf16, six per 32
//      k-values, wave w owns output columns 16w..16w+15 (F = 128) for every row tile -> fp32 in LDS
//   7. out_up[i] = sum_p relu(Y1[col[p]] + Y2[aux[p]]) + (1 + eps1) x_i out of LDS, in entry order
//
// HBM traffic = the item's x rows once, its COO entries once, the two output streams once.  No
// atomics, no zero-fill pass; sums are sequential in the original entry order (deterministic, the
// order a sequential index_add_ visits them).  Results are bit-identical to the two-kernel path
// (cwn_gemm_split.hip + cwn_aggregate.hip): same split, same MFMA order, same epilogue arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/cwn_hip.h"
#include "cwn_split.h"

namespace {

using cwn::frag_cd;

constexpr int kThreads = 512;
constexpr int kEcap = CWN_LAYER_MAX_ENTRIES;
constexpr int kTaskRows = CWN_LAYER_TASK_ROWS;
constexpr int kNX = 6;                      // float4 of x per thread at the row cap (12288 / F rows)

// item record fields (include/cwn_hip.h)
enum { I_FLAGS = 0, I_G, I_GR0, I_GN, I_CR0, I_CN, I_UE0, I_UNE, I_NT, I_TASK0 };
enum { T_DIM = 0, T_R0, T_N, T_BE0, T_BNE, T_SR0, T_SN, T_INTS };

struct LayerArgs {
    cwn_layer_dim d[CWN_LAYER_MAX_DIMS];
    const int32_t* items;
    int32_t* err;
    int32_t rows_cap;                        // padded GEMM rows the LDS of this launch holds
};

// Field-wise select instead of d[dim]: a dynamically indexed by-value struct lands in scratch
// (cwn_aggregate.hip has the measurement).
#define CWN_PICK(field, dim) ((dim) == 0 ? A.d[0].field : (dim) == 1 ? A.d[1].field : A.d[2].field)

template <int F> struct Geo {
    static constexpr int kPlaneStride = F + 8;              // bf16 elements per plane row
    static constexpr int kYStride = F + 4;                  // floats per Y row
    static constexpr int kKS = F / 32;                      // k-steps of 32
    static constexpr int kNCT = F / 16;                     // column tiles
    static constexpr int kWPC = 8 / kNCT;                   // waves sharing a column tile (row-tile parity)
    static constexpr int kG = F / 4;                        // lanes per row in the reduce phases
    static constexpr int kNG = kThreads / kG;               // rows in flight
    static constexpr int kF4 = F / 4;                       // float4 per row
    __host__ __device__ static constexpr size_t planes_bytes(int rows) { return (size_t)3 * rows * kPlaneStride * 2; }
    __host__ __device__ static constexpr size_t y_bytes(int rows) { return (size_t)rows * kYStride * 4; }
};

// index scratch behind the planes and Y: six u16 arrays of kEcap + three row-pointer arrays
constexpr size_t kIdxBytes = (size_t)6 * kEcap * 2 + (size_t)3 * (kTaskRows + 2) * 2;

template <int F>
__host__ __device__ constexpr size_t lds_bytes(int rows) {
    return Geo<F>::planes_bytes(rows) + Geo<F>::y_bytes(rows) + ((kIdxBytes + 15) & ~(size_t)15);
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int F>
__global__ __launch_bounds__(kThreads) void layer_kernel(LayerArgs A) {
    using G = Geo<F>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int rows_cap = A.rows_cap;

    uint16_t* const planes = reinterpret_cast<uint16_t*>(smem);                    // [3][rows_cap][F + 8]
    float* const Y = reinterpret_cast<float*>(smem + G::planes_bytes(rows_cap));   // [rows_cap][F + 4]
    uint16_t* const idx = reinterpret_cast<uint16_t*>(smem + G::planes_bytes(rows_cap) + G::y_bytes(rows_cap));
    uint16_t* const ukey = idx;                 // unsorted local destination / source / shared row
    uint16_t* const uval = idx + kEcap;
    uint16_t* const uaux = idx + 2 * kEcap;
    uint16_t* const skey = idx + 3 * kEcap;     // sorted by destination, stable
    uint16_t* const scol = idx + 4 * kEcap;
    uint16_t* const saux = idx + 5 * kEcap;
    uint16_t* const rowptr = idx + 6 * kEcap;   // [3][kTaskRows + 2]: upper, boundary of task 0, of task 1

    // ---- 1. item record (uniform address: scalar loads) -------------------------------------------
    const int32_t* it = A.items + (size_t)blockIdx.x * CWN_LAYER_ITEM_INTS;
    const int flags = it[I_FLAGS];
    const bool has_gemm = (flags & 1) != 0;
    const int g = it[I_G], g_r0 = it[I_GR0], g_n = has_gemm ? it[I_GN] : 0;
    const int c_r0 = it[I_CR0], c_n = has_gemm ? it[I_CN] : 0;
    const int u_e0 = it[I_UE0], u_ne = has_gemm ? it[I_UNE] : 0;
    const int n_tasks = it[I_NT];
    int t_dim[2], t_r0[2], t_n[2], t_be0[2], t_bne[2], t_sr0[2], t_sn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int32_t* tk = it + I_TASK0 + t * T_INTS;
        const bool on = t < n_tasks;
        t_dim[t] = on ? tk[T_DIM] : 0;
        t_r0[t] = tk[T_R0];
        t_n[t] = on ? tk[T_N] : 0;
        t_be0[t] = tk[T_BE0];
        t_bne[t] = on ? tk[T_BNE] : 0;
        t_sr0[t] = tk[T_SR0];
        t_sn[t] = tk[T_SN];
    }
    const int T1 = (g_n + 15) >> 4, T2 = (c_n + 15) >> 4;      // row tiles of Y1, Y2
    const int rows_pad = (T1 + T2) << 4;
    // segments of the item's combined entry list: [0, s1) upper, [s1, s2) boundary 0, [s2, s3) boundary 1
    const int s1 = u_ne, s2 = s1 + t_bne[0], s3 = s2 + t_bne[1];
    // a record that does not fit the launch's LDS: report and leave (uniform over the workgroup)
    if (rows_pad > rows_cap || s3 > kEcap || s1 < 0 || s2 < s1 || s3 < s2 || t_n[0] > kTaskRows ||
        t_n[1] > kTaskRows || g_n > kTaskRows || (has_gemm && (g < 0 || g > 1))) {
        if (tid == 0) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
        return;
    }
    bool bad = false;

    // ---- 2. every global load of the item, entries first (vector loads return in order) ------------
    const float* xg = CWN_PICK(x, g);
    const float* xc = g == 0 ? A.d[1].x : A.d[2].x;           // x_{g+1}
    const int64_t* up_index = CWN_PICK(up_index, g);
    const int64_t* up_shared = CWN_PICK(up_shared, g);
    const int64_t up_E = CWN_PICK(e_up, g);
    const int64_t* b_index0 = CWN_PICK(b_index, t_dim[0]);
    const int64_t* b_index1 = CWN_PICK(b_index, t_dim[1]);
    const int64_t b_E0 = CWN_PICK(n_b, t_dim[0]), b_E1 = CWN_PICK(n_b, t_dim[1]);

    int64_t ek[2], ev[2], ea[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int w = tid + i * kThreads;
        ek[i] = ev[i] = ea[i] = 0;
        if (w < s1) {
            const int64_t e = (int64_t)u_e0 + w;
            ev[i] = up_index[e];
            ek[i] = up_index[up_E + e];
            ea[i] = up_shared[e];
        } else if (w < s2) {
            const int64_t e = (int64_t)t_be0[0] + (w - s1);
            ev[i] = b_index0[e];
            ek[i] = b_index0[b_E0 + e];
        } else if (w < s3) {
            const int64_t e = (int64_t)t_be0[1] + (w - s2);
            ev[i] = b_index1[e];
            ek[i] = b_index1[b_E1 + e];
        }
    }
    // x rows of the GEMM operands: rows [0, 16 T1) from x_g, [16 T1, rows_pad) from x_{g+1}; rows past
    // the real ones are clamped to the last real row (never stored; guarded loads serialise)
    float4 xv[kNX];
    const int nx = (rows_pad * G::kF4 + kThreads - 1) / kThreads;
#pragma unroll
    for (int i = 0; i < kNX; ++i) {
        xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nx) {
            const int q = tid + i * kThreads, row = q / G::kF4, c4 = q % G::kF4;
            const bool first = row < (T1 << 4);
            const int r = first ? min(row, g_n - 1) : min(row - (T1 << 4), c_n - 1);
            const float* base = first ? xg + (int64_t)(g_r0 + r) * F : xc + (int64_t)(c_r0 + max(r, 0)) * F;
            if (row < rows_pad) xv[i] = ldg4(base + c4 * 4);
        }
    }
    // this wave's slice of W: output columns ct*16 .. +15, both halves of the [F, 2F] weight
    const int ct = wave % G::kNCT, rt_par = wave / G::kNCT;
    float4 wraw[2][G::kKS][2];
    if (has_gemm) {
        const float* msg_w = CWN_PICK(msg_w, g);
        const float* wrow = msg_w + (int64_t)(ct * 16 + l15) * (2 * F) + kq * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < G::kKS; ++ks) {
                wraw[h][ks][0] = ldg4(wrow + h * F + ks * 32);
                wraw[h][ks][1] = ldg4(wrow + h * F + ks * 32 + 4);
            }
    }

    // ---- 3a. entries -> LDS as local row numbers, range-checked --------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int w = tid + i * kThreads;
        if (w < s3 && w < kEcap) {
            int64_t k, v, a = 0;
            int64_t nk, nv, na = 1;
            if (w < s1) {
                k = ek[i] - g_r0; v = ev[i] - g_r0; a = ea[i] - c_r0;
                nk = g_n; nv = g_n; na = c_n;
            } else if (w < s2) {
                k = ek[i] - t_r0[0]; v = ev[i] - t_sr0[0];
                nk = t_n[0]; nv = t_sn[0];
            } else {
                k = ek[i] - t_r0[1]; v = ev[i] - t_sr0[1];
                nk = t_n[1]; nv = t_sn[1];
            }
            if (k < 0 || k >= nk || v < 0 || v >= nv || v > 65535 || a < 0 || a >= na) {
                bad = true;
                k = 0; v = 0; a = 0;
            }
            ukey[w] = (uint16_t)k;
            uval[w] = (uint16_t)v;
            uaux[w] = (uint16_t)a;
        }
    }
    if (bad) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
    __syncthreads();

    // ---- 3b. stable rank by destination: P lanes per entry, each scans 1/P of the entry's segment ----
    const int total = min(s3, kEcap);
    {
        int P = 1;
        while (P < 8 && total * (P * 2) <= kThreads) P *= 2;
        const int per_pass = kThreads / P;
        for (int base = 0; base < total; base += per_pass) {
            const int w = base + tid / P, sub = tid % P;
            int cnt = 0, k = 0, seg0 = 0;
            const bool live = w < total;
            if (live) {
                seg0 = w < s1 ? 0 : (w < s2 ? s1 : s2);
                const int seg1 = w < s1 ? s1 : (w < s2 ? s2 : s3);
                k = ukey[w];
                const int len = seg1 - seg0, chunk = (len + P - 1) / P;
                const int lo = seg0 + sub * chunk, hi = min(seg1, lo + chunk);
                for (int e = lo; e < hi; ++e) {
                    const int ke = ukey[e];
                    cnt += (ke < k || (ke == k && e < w)) ? 1 : 0;
                }
            }
            for (int off = 1; off < P; off <<= 1) cnt += __shfl_xor(cnt, off, 64);
            if (live && sub == 0) {
                const int pos = seg0 + cnt;
                skey[pos] = (uint16_t)k;
                scol[pos] = uval[w];
                saux[pos] = uaux[w];
            }
        }
    }
    __syncthreads();

    // ---- 3c. row pointers from the sorted keys (run boundaries), empty rows included -----------------
    for (int p = tid; p < total; p += kThreads) {
        const int which = p < s1 ? 0 : (p < s2 ? 1 : 2);
        const int seg0 = which == 0 ? 0 : (which == 1 ? s1 : s2);
        const int seg1 = which == 0 ? s1 : (which == 1 ? s2 : s3);
        const int n_rows = which == 0 ? g_n : t_n[which - 1];
        uint16_t* rp = rowptr + which * (kTaskRows + 2);
        const int k = skey[p];
        const int kprev = p > seg0 ? (int)skey[p - 1] : -1;
        for (int r = kprev + 1; r <= k; ++r) rp[r] = (uint16_t)(p - seg0);
        if (p == seg1 - 1)
            for (int r = k + 1; r <= n_rows; ++r) rp[r] = (uint16_t)(seg1 - seg0);
    }
    if (s1 == 0)
        for (int r = tid; r <= g_n; r += kThreads) rowptr[r] = 0;
    if (s2 == s1)
        for (int r = tid; r <= t_n[0]; r += kThreads) rowptr[(kTaskRows + 2) + r] = 0;
    if (s3 == s2)
        for (int r = tid; r <= t_n[1]; r += kThreads) rowptr[2 * (kTaskRows + 2) + r] = 0;

    // ---- 4. x rows -> three bf16 planes ---------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < kNX; ++i) {
        if (i < nx) {
            const int q = tid + i * kThreads, row = q / G::kF4, c4 = q % G::kF4;
            if (row < rows_pad) {
                uint2 ph, pm, pl;
                cwn::split4(xv[i], ph, pm, pl);
                const size_t plane = (size_t)rows_cap * G::kPlaneStride;
                uint16_t* dst = planes + (size_t)row * G::kPlaneStride + c4 * 4;
                *reinterpret_cast<uint2*>(dst) = ph;
                *reinterpret_cast<uint2*>(dst + plane) = pm;
                *reinterpret_cast<uint2*>(dst + 2 * plane) = pl;
            }
        }
    }
    __syncthreads();

    // ---- 5. boundary stream and self terms of every task ---------------------------------------------
    const int gl = tid % G::kG, gq = tid / G::kG, f = gl * 4;
    int z = 0;
    asm volatile("" : "+v"(z));              // keeps the eps loads in VMEM (see cwn_aggregate.hip)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t < n_tasks) {
            const int d = t_dim[t];
            const float* x = CWN_PICK(x, d);
            const float* xs = d == 1 ? A.d[0].x : A.d[1].x;          // x_{d-1} (unused when d == 0)
            float* out_up = CWN_PICK(out_up, d);
            float* out_b = CWN_PICK(out_b, d);
            const float* e1p = CWN_PICK(eps1, d);
            const float* e2p = CWN_PICK(eps2, d);
            const float scale1 = 1.0f + (e1p != nullptr ? e1p[z] : 0.0f);
            const float scale2 = 1.0f + (e2p != nullptr ? e2p[z] : 0.0f);
            const uint16_t* rp = rowptr + (t + 1) * (kTaskRows + 2);
            const uint16_t* col = scol + (t == 0 ? s1 : s2);
            const bool up_here = has_gemm && d == g;
            for (int r = gq; r < t_n[t]; r += G::kNG) {
                const int start = rp[r], end = rp[r + 1];
                const int64_t row = (int64_t)(t_r0[t] + r) * F + f;
                const float4 xi = ldg4(x + row);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                int p = start;
                for (; p + 4 <= end; p += 4) {
                    float4 a[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = ldg4(xs + (int64_t)(t_sr0[t] + col[p + u]) * F + f);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc.x += a[u].x; acc.y += a[u].y; acc.z += a[u].z; acc.w += a[u].w;
                    }
                }
                for (; p < end; ++p) {
                    const float4 a = ldg4(xs + (int64_t)(t_sr0[t] + col[p]) * F + f);
                    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
                }
                *reinterpret_cast<float4*>(out_b + row) =
                    make_float4(acc.x + scale2 * xi.x, acc.y + scale2 * xi.y, acc.z + scale2 * xi.z,
                                acc.w + scale2 * xi.w);
                if (!up_here)        // no upper adjacency in this dimension: zeros + self term
                    *reinterpret_cast<float4*>(out_up + row) =
                        make_float4(0.0f + scale1 * xi.x, 0.0f + scale1 * xi.y, 0.0f + scale1 * xi.z,
                                    0.0f + scale1 * xi.w);
            }
        }
    }
    if (!has_gemm) return;

    // ---- 6. Y1 | Y2 on the matrix cores ---------------------------------------------------------------
    {
        const size_t plane = (size_t)rows_cap * G::kPlaneStride;
        const float* bias = CWN_PICK(msg_bias, g);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias != nullptr) b4 = ldg4(bias + ct * 16 + kq * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 wf[G::kKS][3];
#pragma unroll
            for (int ks = 0; ks < G::kKS; ++ks)
                cwn::split8(wraw[h][ks][0], wraw[h][ks][1], wf[ks][0], wf[ks][1], wf[ks][2]);
            const int rt0 = h == 0 ? 0 : T1, rt1 = h == 0 ? T1 : T1 + T2;
            // row tiles of this half that are this wave's (F = 64: two waves share a column tile)
            int rt = rt0 + ((rt_par - rt0) % G::kWPC + G::kWPC) % G::kWPC;
            for (; rt < rt1; rt += 2 * G::kWPC) {
                const bool two = rt + G::kWPC < rt1;
                frag_cd c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
                const uint16_t* p0 = planes + (size_t)(rt * 16 + l15) * G::kPlaneStride + kq * 8;
                const uint16_t* p1 = p0 + (size_t)(two ? G::kWPC * 16 : 0) * G::kPlaneStride;
#pragma unroll
                for (int ks = 0; ks < G::kKS; ++ks) {
                    const uint4 xh0 = *reinterpret_cast<const uint4*>(p0 + ks * 32);
                    const uint4 xm0 = *reinterpret_cast<const uint4*>(p0 + plane + ks * 32);
                    const uint4 xl0 = *reinterpret_cast<const uint4*>(p0 + 2 * plane + ks * 32);
                    const uint4 xh1 = *reinterpret_cast<const uint4*>(p1 + ks * 32);
                    const uint4 xm1 = *reinterpret_cast<const uint4*>(p1 + plane + ks * 32);
                    const uint4 xl1 = *reinterpret_cast<const uint4*>(p1 + 2 * plane + ks * 32);
                    c0 = cwn::mfma_split6(wf[ks][0], wf[ks][1], wf[ks][2], xh0, xm0, xl0, c0);
                    c1 = cwn::mfma_split6(wf[ks][0], wf[ks][1], wf[ks][2], xh1, xm1, xl1, c1);
                }
                // D[i][j]: i = output column (lane >> 4) * 4 + reg, j = x row (lane & 15)
                if (h == 0 && bias != nullptr) {       // Y1 carries the Linear's bias
                    c0[0] += b4.x; c0[1] += b4.y; c0[2] += b4.z; c0[3] += b4.w;
                    c1[0] += b4.x; c1[1] += b4.y; c1[2] += b4.z; c1[3] += b4.w;
                }
                float* y0 = Y + (size_t)(rt * 16 + l15) * G::kYStride + ct * 16 + kq * 4;
                *reinterpret_cast<float4*>(y0) = make_float4(c0[0], c0[1], c0[2], c0[3]);
                if (two)
                    *reinterpret_cast<float4*>(y0 + (size_t)(G::kWPC * 16) * G::kYStride) =
                        make_float4(c1[0], c1[1], c1[2], c1[3]);
            }
        }
    }
    __syncthreads();

    // ---- 7. upper stream out of LDS: out_up[i] = sum_p relu(Y1[col[p]] + Y2[aux[p]]) + (1 + eps1) x_i -
    {
        float* out_up = CWN_PICK(out_up, g);
        const float* e1p = CWN_PICK(eps1, g);
        const float scale1 = 1.0f + (e1p != nullptr ? e1p[z] : 0.0f);
        const float* Y2 = Y + (size_t)(T1 << 4) * G::kYStride;
        for (int r = gq; r < g_n; r += G::kNG) {
            const int start = rowptr[r], end = rowptr[r + 1];
            const int64_t row = (int64_t)(g_r0 + r) * F + f;
            const float4 xi = ldg4(xg + row);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = start; p < end; ++p) {
                const float4 a = *reinterpret_cast<const float4*>(Y + (size_t)scol[p] * G::kYStride + f);
                const float4 b = *reinterpret_cast<const float4*>(Y2 + (size_t)saux[p] * G::kYStride + f);
                acc.x += fmaxf(a.x + b.x, 0.0f);
                acc.y += fmaxf(a.y + b.y, 0.0f);
                acc.z += fmaxf(a.z + b.z, 0.0f);
                acc.w += fmaxf(a.w + b.w, 0.0f);
            }
            *reinterpret_cast<float4*>(out_up + row) =
                make_float4(acc.x + scale1 * xi.x, acc.y + scale1 * xi.y, acc.z + scale1 * xi.z,
                            acc.w + scale1 * xi.w);
        }
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <int F>
int launch(const LayerArgs& A, int64_t n_items, hipStream_t stream) {
    static std::once_flag once;            // raise the dynamic-LDS limit of this instantiation, once
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel<F>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) return CWN_ERR_LAUNCH;
    const size_t lds = lds_bytes<F>(A.rows_cap);
    layer_kernel<F><<<dim3((unsigned)n_items), dim3(kThreads), lds, stream>>>(A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

}  // namespace

extern "C" size_t cwn_layer_fused_lds_bytes(int32_t F, int32_t max_gemm_rows) {
    if (max_gemm_rows < 0 || max_gemm_rows % 16 != 0) return 0;
    if (F == 128 && max_gemm_rows <= CWN_LAYER_GEMM_ROWS(128)) return lds_bytes<128>(max_gemm_rows);
    if (F == 64 && max_gemm_rows <= CWN_LAYER_GEMM_ROWS(64)) return lds_bytes<64>(max_gemm_rows);
    return 0;
}

extern "C" int cwn_layer_fused_f32(const cwn_layer_dim* dims, int n_dims, int32_t F, const int32_t* items,
                                   int64_t n_items, int32_t max_gemm_rows, int32_t flags, int32_t* err_flag,
                                   cwn_stream_t stream_) {
    if (dims == nullptr || n_dims < 1 || n_dims > CWN_LAYER_MAX_DIMS || n_items < 0 || flags != 0)
        return CWN_ERR_BAD_ARG;
    if (F != 64 && F != 128) return CWN_ERR_BAD_ARG;
    if (n_items == 0) return CWN_OK;
    if (items == nullptr || err_flag == nullptr) return CWN_ERR_BAD_ARG;
    if (n_items >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    if (cwn_layer_fused_lds_bytes(F, max_gemm_rows) == 0) return CWN_ERR_BAD_ARG;
    LayerArgs A{};
    for (int d = 0; d < n_dims; ++d) {
        const cwn_layer_dim& D = dims[d];
        if (D.n_cells < 0 || D.e_up < 0 || D.n_b < 0) return CWN_ERR_BAD_ARG;
        if (D.n_cells >= INT32_MAX || D.e_up >= INT32_MAX || D.n_b >= INT32_MAX) return CWN_ERR_TOO_LARGE;
        if (D.n_cells > 0 && (D.x == nullptr || D.out_up == nullptr || D.out_b == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.e_up > 0 && (D.up_index == nullptr || D.up_shared == nullptr || D.msg_w == nullptr ||
                           d + 1 >= n_dims))
            return CWN_ERR_BAD_ARG;
        if (D.n_b > 0 && (D.b_index == nullptr || d == 0)) return CWN_ERR_BAD_ARG;
        if (!(al16(D.x) && al16(D.out_up) && al16(D.out_b) && al16(D.msg_w) && al16(D.msg_bias)))
            return CWN_ERR_ALIGN;
        A.d[d] = D;
    }
    A.items = items;
    A.err = err_flag;
    A.rows_cap = max_gemm_rows;
    hipStream_t stream = (hipStream_t)stream_;
    return F == 128 ? launch<128>(A, n_items, stream) : launch<64>(A, n_items, stream);
}
