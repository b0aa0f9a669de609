// cwn_norm.hip -- training-mode normalisation + activation around the dense GEMMs, and its
// backward pass (torch.nn.BatchNorm1d(train) + ReLU of update_up_nn / update_boundaries_nn /
// combine_nn, mp/layers.py:303-325).
//
//   cwn_bn_finalize_f32       batch statistics (fp64 band sums from the GEMM epilogue) -> per-column
//                             affine (scale, shift) + mean / rstd for backward + running statistics
//   cwn_norm_act_f32          out = act(z * scale + shift)             (materialised last stage)
//   cwn_norm_bwd_reduce_f32   s1 = sum dyh, s2 = sum dyh * xhat        (= d beta, d gamma)
//   cwn_norm_bwd_apply_f32    dz = scale * (dyh - s1/M - xhat * s2/M)
//
// All of it is HBM/L2-bound elementwise or column-reduction work over [M, N] matrices with
// N = hidden width (64..256): a workgroup takes a band of rows, 32 lanes x 16 B span 128 columns
// (one full 512-B row segment per half wave), column reductions go registers -> LDS -> one atomic
// per column per workgroup.  Up to CWN_MAX_NORM_DESCS matrices (all dimensions and both branches
// of a layer) per launch.
#include <hip/hip_runtime.h>
#include "../../include/cwn_hip.h"
#include "cwn_mem.h"
#include "cwn_bn_live.h"
#include "cwn_dropout.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTPR = 32;                    // threads per row
constexpr int kRowsPerPass = kThreads / kTPR;   // 8
constexpr int kBand = 64;                   // rows per workgroup

struct NormBatch {
    cwn_norm_desc d[CWN_MAX_NORM_DESCS];
    int32_t blk_start[CWN_MAX_NORM_DESCS + 1];
    int32_t n;
};

struct BnBatch {
    cwn_bn_desc d[CWN_MAX_NORM_DESCS];
    int32_t n;
};

__device__ __forceinline__ int find_desc(const int32_t* start, int n, int b) {
    int d = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_NORM_DESCS; ++i)
        if (i < n && b >= start[i]) d = i;
    return d;
}

// grid (column chunk of 64, descriptor).  The GEMM epilogue left one fp64 partial per 32-row band
// and column; lanes read consecutive columns (coalesced), the SIXTEEN waves take every sixteenth band
// with eight loads in flight each: the 105 bands of a ZINC-128 dimension are one round trip (a dependent
// round trip is ~1 us; four waves walking them in four rounds made this a 6.6-us kernel -- 9 us start to
// start, twelve times per training step -- and one thread walking them a 12-us one), and the
// sixteen partials meet in LDS.
constexpr int kFinThreads = 1024, kFinSlices = kFinThreads / 64;

__global__ __launch_bounds__(kFinThreads) void bn_finalize_kernel(BnBatch B) {
    __shared__ double part[2][kFinSlices][64];
    const cwn_bn_desc& D = B.d[blockIdx.y];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    const int64_t Mv = D.m_dev != nullptr ? *D.m_dev : D.M;       // rows the statistics were taken over (D.M: capacity)
    const int64_t bands = CWN_STAT_ROWS(Mv);
    const bool ok = n < D.N;
    const int nc = ok ? n : D.N - 1;
    double s = 0.0, sq = 0.0;
    for (int64_t b0 = slice; b0 < bands; b0 += kFinSlices * 8) {
        double t[8], u[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t b = b0 + kFinSlices * q;
            const int64_t bc = b < bands ? b : (bands > 0 ? bands - 1 : 0);
            t[q] = D.col_sum[bc * D.N + nc];
            u[q] = D.col_sumsq[bc * D.N + nc];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (b0 + kFinSlices * q < bands) {
                s += t[q];
                sq += u[q];
            }
        }
    }
    part[0][slice][threadIdx.x & 63] = s;
    part[1][slice][threadIdx.x & 63] = sq;
    __syncthreads();
    if (slice != 0 || !ok) return;
    s = sq = 0.0;
#pragma unroll
    for (int q = 0; q < kFinSlices; ++q) {
        s += part[0][q][threadIdx.x];
        sq += part[1][q][threadIdx.x];
    }
    const double invM = 1.0 / (double)(Mv > 0 ? Mv : 1);
    const double mean = s * invM;
    double var = sq * invM - mean * mean;     // biased, as BatchNorm normalises
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)D.eps));
    const float g = D.gamma != nullptr ? D.gamma[n] : 1.0f;
    const float beta = D.beta != nullptr ? D.beta[n] : 0.0f;
    const float scale = g * rstd;
    D.scale[n] = scale;
    D.shift[n] = beta - (float)mean * scale;
    D.mean[n] = (float)mean;
    D.rstd[n] = rstd;
    if (D.bwd_sums != nullptr) {
        D.bwd_sums[n] = 0.f;
        D.bwd_sums[D.N + n] = 0.f;
    }
    if (Mv < 1) return;        // (device-side count: a batch without cells of this dimension leaves the module's state alone)
    if (D.num_batches_tracked != nullptr && n == 0) *D.num_batches_tracked += 1;
    if (D.running_mean != nullptr) {
        const float mom = D.momentum;
        const double unbiased = Mv > 1 ? var * ((double)Mv / (double)(Mv - 1)) : var;
        D.running_mean[n] = (1.0f - mom) * D.running_mean[n] + mom * (float)mean;
        D.running_var[n] = (1.0f - mom) * D.running_var[n] + mom * (float)unbiased;
    }
}

// per-column constants of a thread's VEC columns
template <int VEC>
struct Cols {
    float scale[VEC], shift[VEC], mean[VEC], rstd[VEC];
};

template <int VEC>
__device__ __forceinline__ void ld_vec(float (&v)[VEC], const float* p) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}

template <int VEC>
__device__ __forceinline__ void st_vec(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) cwn::store_result4(p, v[0], v[1], v[2], v[3]);
    else *p = v[0];
}

// MODE 0: activation forward; 1: backward reduce; 2: backward apply
template <int VEC, int MODE>
__global__ __launch_bounds__(kThreads) void norm_kernel(NormBatch B) {
    __shared__ float red[2][kRowsPerPass][kTPR * 4 + 4];
    const int di = find_desc(B.blk_start, B.n, blockIdx.x);
    const cwn_norm_desc& D = B.d[di];
    const int64_t row0 = (int64_t)(blockIdx.x - B.blk_start[di]) * kBand;
    const int64_t Mv = D.m_dev != nullptr ? *D.m_dev : D.M;      // rows that exist (D.M: the capacity, bounds the addresses)
    const int tc = threadIdx.x % kTPR, tr = threadIdx.x / kTPR;
    const int N = D.N;
    // (activation only) a live BatchNorm (cwn_bn_live.h): the affine is derived here from the producing launch's slot sums
    const bool live = MODE == 0 && D.bn.slots != nullptr;
    const bool has_norm = D.scale != nullptr || live;
    const bool relu = D.relu != 0;
    const float invM = 1.0f / (float)(Mv > 0 ? Mv : 1);
    // dropout of the activation (MODE 0: the epilogue) / of the incoming gradient (MODE 1: the prologue; cwn_dropout.h)
    cwn::Dropout drop;
    drop.on = false;
    if constexpr (MODE != 2) drop.init(D.drop);
    // (a band past the batch's own rows has nothing to do -- except the apply form's first band, which hands the sums on)
    if (row0 >= Mv && !(MODE == 2 && row0 == 0)) return;
    for (int c0 = 0; c0 < N; c0 += kTPR * VEC) {     // column chunks of 128 (VEC = 4) or 32
        const int c = c0 + tc * VEC;
        const bool cok = c < N;                       // N % VEC == 0 (host-checked)
        float scale[VEC], shift[VEC], mean[VEC], rstd[VEC], k1[VEC], k2[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            scale[v] = 1.f; shift[v] = 0.f; mean[v] = 0.f; rstd[v] = 0.f; k1[v] = 0.f; k2[v] = 0.f;
        }
        if (cok && has_norm && !live) {
            ld_vec<VEC>(scale, D.scale + c);
            ld_vec<VEC>(shift, D.shift + c);
            if constexpr (MODE != 0) {
                ld_vec<VEC>(mean, D.mean + c);
                ld_vec<VEC>(rstd, D.rstd + c);
            }
            if constexpr (MODE == 2) {
                ld_vec<VEC>(k1, D.s1 + c);
                ld_vec<VEC>(k2, D.s2 + c);
                if (row0 == 0 && tr == 0) {          // the first band's first row group hands the sums on: one writer per column
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        if (D.acc1 != nullptr) D.acc1[c + v] += k1[v];
                        if (D.acc2 != nullptr) D.acc2[c + v] += k2[v];
                    }
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    k1[v] *= invM;
                    k2[v] *= invM;
                }
            }
        }
        float a1[VEC], a2[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) a1[v] = a2[v] = 0.f;
        // all of a thread's row loads are issued before the first use
        float z[kBand / kRowsPerPass][VEC], g[kBand / kRowsPerPass][VEC];
#pragma unroll
        for (int i = 0; i < kBand / kRowsPerPass; ++i) {
            const int64_t r = row0 + tr + i * kRowsPerPass;
            const int64_t rc = r < D.M ? r : D.M - 1;
#pragma unroll
            for (int v = 0; v < VEC; ++v) z[i][v] = g[i][v] = 0.f;
            if (cok) {
                ld_vec<VEC>(z[i], D.z + rc * D.ldz + c);
                if constexpr (MODE != 0) ld_vec<VEC>(g[i], D.dy + rc * D.lddy + c);
            }
        }
        if constexpr (MODE == 1) {
            if (drop.on) {                            // (uniform) dy arrives w.r.t. the dropped activation
#pragma unroll
                for (int i = 0; i < kBand / kRowsPerPass; ++i) {
                    const int64_t r = row0 + tr + i * kRowsPerPass;
                    if (!(cok && r < Mv)) continue;
                    if constexpr (VEC == 4) {
                        const float4 m = drop.mul4((uint32_t)(((uint64_t)r * (uint64_t)N + (uint64_t)c) >> 2));
                        g[i][0] *= m.x; g[i][1] *= m.y; g[i][2] *= m.z; g[i][3] *= m.w;
                    } else {
                        g[i][0] *= drop.mul1((uint64_t)r * (uint64_t)N + (uint64_t)c);
                    }
                    if (D.dy_out != nullptr) st_vec<VEC>(D.dy_out + r * D.lddy_out + c, g[i]);
                }
            }
        }
        if constexpr (MODE == 0) {
            if (live) {                               // (uniform) thread i < chunk width: column c0 + i; the first band writes.  BEHIND the row
                                                      // requests above: the derive runs while they travel
                if ((int)threadIdx.x < kTPR * VEC && c0 + (int)threadIdx.x < N) {
                    float sc, sh;
                    cwn::bn_live_column(D.bn, N, Mv, c0 + threadIdx.x, row0 == 0, sc, sh);
                    red[0][0][threadIdx.x] = sc;
                    red[1][0][threadIdx.x] = sh;
                }
                __syncthreads();
                if (cok) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        scale[v] = red[0][0][tc * VEC + v];
                        shift[v] = red[1][0][tc * VEC + v];
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < kBand / kRowsPerPass; ++i) {
            const int64_t r = row0 + tr + i * kRowsPerPass;
            const bool ok = cok && r < Mv;
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float y = z[i][v] * scale[v] + shift[v];
                if constexpr (MODE == 0) {
                    o[v] = relu ? fmaxf(y, 0.f) : y;
                } else {
                    const float dyh = (!relu || y > 0.f) ? g[i][v] : 0.f;
                    const float xhat = (z[i][v] - mean[v]) * rstd[v];
                    if constexpr (MODE == 1) {
                        if (ok) {
                            a1[v] += dyh;
                            a2[v] += dyh * xhat;
                        }
                    } else {
                        o[v] = has_norm ? scale[v] * (dyh - k1[v] - xhat * k2[v]) : dyh;
                    }
                }
            }
            if constexpr (MODE == 0) {
                if (drop.on && ok) {
                    if constexpr (VEC == 4) {
                        const float4 m = drop.mul4((uint32_t)(((uint64_t)r * (uint64_t)N + (uint64_t)c) >> 2));
                        o[0] *= m.x; o[1] *= m.y; o[2] *= m.z; o[3] *= m.w;
                    } else {
                        o[0] *= drop.mul1((uint64_t)r * (uint64_t)N + (uint64_t)c);
                    }
                }
            }
            if constexpr (MODE != 1) {
                if (ok) st_vec<VEC>(D.out + r * D.ldout + c, o);
            }
        }
        if constexpr (MODE == 1) {
            // 8 row groups -> one partial per column -> one atomic per column per workgroup
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                red[0][tr][tc * VEC + v] = a1[v];
                red[1][tr][tc * VEC + v] = a2[v];
            }
            __syncthreads();
            if (threadIdx.x < kTPR * VEC) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int q = 0; q < kRowsPerPass; ++q) {
                    t1 += red[0][q][threadIdx.x];
                    t2 += red[1][q][threadIdx.x];
                }
                const int cc = c0 + threadIdx.x;
                if (cc < N) {
                    atomicAdd(D.s1 + cc, t1);
                    atomicAdd(D.s2 + cc, t2);
                }
            }
            __syncthreads();
        }
    }
}

// Backward reduce + apply in one launch (cwn_norm_bwd_f32): workgroup = (matrix, 4 columns), thread t owns rows
// t, t + 512, ... (at most kFR of them, all loads issued before the first use); the column sums are a wave xor tree and
// eight LDS partials summed by every thread in the same order.  A wave's load touches 64 rows x 16 B: the L2 -> L1
// traffic is 4 - 8x the useful bytes and the stores are partial lines shared by 32 workgroups: measured 28 us per launch
// at the ZINC batch against 6.3 + 5.9 us for the two coalesced launches.  The deterministic form, not the fast one.
constexpr int kFT = 512;
constexpr int kFR = CWN_NORM_BWD_FUSED_MAX_ROWS / kFT;     // 8 rows per thread

__global__ __launch_bounds__(kFT) void norm_bwd_fused_kernel(NormBatch B, int accumulate) {
    __shared__ float red[2][kFT / 64][4];
    const int di = find_desc(B.blk_start, B.n, blockIdx.x);
    const cwn_norm_desc& D = B.d[di];
    const int c = ((int)blockIdx.x - B.blk_start[di]) * 4;
    const bool has_norm = D.scale != nullptr, relu = D.relu != 0;
    const int64_t Mcap = D.M;
    const int64_t M = D.m_dev != nullptr ? *D.m_dev : D.M;
    float scale[4] = {1.f, 1.f, 1.f, 1.f}, shift[4] = {0.f, 0.f, 0.f, 0.f}, mean[4] = {0.f, 0.f, 0.f, 0.f},
          rstd[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_norm) {
        ld_vec<4>(scale, D.scale + c);
        ld_vec<4>(shift, D.shift + c);
        ld_vec<4>(mean, D.mean + c);
        ld_vec<4>(rstd, D.rstd + c);
    }
    float z[kFR][4], g[kFR][4];
#pragma unroll
    for (int i = 0; i < kFR; ++i) {
        const int64_t r = threadIdx.x + (int64_t)i * kFT;
        const int64_t rc = r < Mcap ? r : Mcap - 1;
        ld_vec<4>(z[i], D.z + rc * D.ldz + c);
        ld_vec<4>(g[i], D.dy + rc * D.lddy + c);
    }
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kFR; ++i) {
        const bool ok = threadIdx.x + (int64_t)i * kFT < M;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float y = z[i][v] * scale[v] + shift[v];
            const float dyh = (!relu || y > 0.f) ? g[i][v] : 0.f;
            const float xhat = (z[i][v] - mean[v]) * rstd[v];
            g[i][v] = dyh;
            z[i][v] = xhat;
            if (ok) {
                a1[v] += dyh;
                a2[v] += dyh * xhat;
            }
        }
    }
    float k1[4] = {0.f, 0.f, 0.f, 0.f}, k2[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_norm) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                a1[v] += __shfl_xor(a1[v], m, 64);
                a2[v] += __shfl_xor(a2[v], m, 64);
            }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                red[0][threadIdx.x >> 6][v] = a1[v];
                red[1][threadIdx.x >> 6][v] = a2[v];
            }
        }
        __syncthreads();
        const float invM = 1.0f / (float)(M > 0 ? M : 1);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int q = 0; q < kFT / 64; ++q) {
                t1 += red[0][q][v];
                t2 += red[1][q][v];
            }
            if (threadIdx.x == 0) {
                D.s1[c + v] = accumulate ? D.s1[c + v] + t1 : t1;
                D.s2[c + v] = accumulate ? D.s2[c + v] + t2 : t2;
            }
            k1[v] = t1 * invM;
            k2[v] = t2 * invM;
        }
    }
#pragma unroll
    for (int i = 0; i < kFR; ++i) {
        const int64_t r = threadIdx.x + (int64_t)i * kFT;
        if (r < M) {
            float o[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) o[v] = has_norm ? scale[v] * (g[i][v] - k1[v] - z[i][v] * k2[v]) : g[i][v];
            st_vec<4>(D.out + r * D.ldout + c, o);
        }
    }
}

inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; }

template <int MODE>
int launch_norm(const cwn_norm_desc* descs, int n, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_NORM_DESCS) return CWN_ERR_BAD_ARG;
    NormBatch B{};
    B.n = n;
    int64_t blocks = 0;
    bool vec = true;
    for (int i = 0; i < n; ++i) {
        const cwn_norm_desc& D = descs[i];
        if (D.M < 0 || D.N <= 0) return CWN_ERR_BAD_ARG;
        if ((D.scale == nullptr) != (D.shift == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.bn.slots != nullptr && (MODE != 0 || D.scale != nullptr || D.bn.aff == nullptr || ((uintptr_t)D.bn.slots & 7u) ||
                                      (D.bn.running_mean == nullptr) != (D.bn.running_var == nullptr)))
            return CWN_ERR_BAD_ARG;
        if (D.M > 0) {
            if (D.z == nullptr || D.ldz < D.N) return CWN_ERR_BAD_ARG;
            if (MODE != 1 && (D.out == nullptr || D.ldout < D.N)) return CWN_ERR_BAD_ARG;
            if (MODE != 0 && (D.dy == nullptr || D.lddy < D.N)) return CWN_ERR_BAD_ARG;
            if (MODE != 0 && D.scale != nullptr &&
                (D.mean == nullptr || D.rstd == nullptr || D.s1 == nullptr || D.s2 == nullptr))
                return CWN_ERR_BAD_ARG;
            if (MODE == 1 && (D.s1 == nullptr || D.s2 == nullptr)) return CWN_ERR_BAD_ARG;
            if (D.drop.state != nullptr && (MODE == 2 || !(D.drop.p >= 0.f && D.drop.p < 1.f))) return CWN_ERR_BAD_ARG;
            if (D.dy_out != nullptr && (MODE != 1 || D.lddy_out < D.N)) return CWN_ERR_BAD_ARG;
        }
        const void* ptrs[] = {D.dy, D.z, D.scale, D.shift, D.mean, D.rstd, D.s1, D.s2, D.out, D.dy_out};
        for (const void* p : ptrs) {
            if (p != nullptr && ((uintptr_t)p & 3u)) return CWN_ERR_ALIGN;
            vec = vec && al16(p);
        }
        vec = vec && D.N % 4 == 0 && D.ldz % 4 == 0 && (MODE == 0 || D.lddy % 4 == 0) &&
              (MODE == 1 || D.ldout % 4 == 0) && (D.dy_out == nullptr || D.lddy_out % 4 == 0);
        B.d[i] = D;
        B.blk_start[i] = (int32_t)blocks;
        blocks += (D.M + kBand - 1) / kBand;
        if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    }
    for (int i = n; i <= CWN_MAX_NORM_DESCS; ++i) B.blk_start[i] = (int32_t)blocks;
    if (blocks == 0) return CWN_OK;
    hipStream_t stream = (hipStream_t)stream_;
    if (vec) norm_kernel<4, MODE><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
    else norm_kernel<1, MODE><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

}  // namespace

extern "C" int cwn_bn_finalize_f32(const cwn_bn_desc* descs, int n, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_NORM_DESCS) return CWN_ERR_BAD_ARG;
    BnBatch B{};
    B.n = n;
    for (int i = 0; i < n; ++i) {
        const cwn_bn_desc& D = descs[i];
        if (D.N <= 0 || D.M <= 0) return CWN_ERR_BAD_ARG;
        if (D.col_sum == nullptr || D.col_sumsq == nullptr || D.scale == nullptr || D.shift == nullptr ||
            D.mean == nullptr || D.rstd == nullptr)
            return CWN_ERR_BAD_ARG;
        if ((D.running_mean == nullptr) != (D.running_var == nullptr)) return CWN_ERR_BAD_ARG;
        B.d[i] = D;
    }
    int nmax = 0;
    for (int i = 0; i < n; ++i) nmax = descs[i].N > nmax ? descs[i].N : nmax;
    bn_finalize_kernel<<<dim3((nmax + 63) / 64, n), dim3(kFinThreads), 0, (hipStream_t)stream_>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_norm_act_f32(const cwn_norm_desc* descs, int n, cwn_stream_t stream) {
    return launch_norm<0>(descs, n, stream);
}

extern "C" int cwn_norm_bwd_reduce_f32(const cwn_norm_desc* descs, int n, cwn_stream_t stream) {
    return launch_norm<1>(descs, n, stream);
}

extern "C" int cwn_norm_bwd_apply_f32(const cwn_norm_desc* descs, int n, cwn_stream_t stream) {
    return launch_norm<2>(descs, n, stream);
}

extern "C" int cwn_norm_bwd_f32(const cwn_norm_desc* descs, int n, int accumulate, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_NORM_DESCS) return CWN_ERR_BAD_ARG;
    NormBatch B{};
    B.n = n;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        const cwn_norm_desc& D = descs[i];
        if (D.M < 0 || D.N <= 0 || D.M > CWN_NORM_BWD_FUSED_MAX_ROWS) return CWN_ERR_BAD_ARG;
        if ((D.scale == nullptr) != (D.shift == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.M > 0) {
            if (D.z == nullptr || D.ldz < D.N || D.out == nullptr || D.ldout < D.N || D.dy == nullptr || D.lddy < D.N)
                return CWN_ERR_BAD_ARG;
            if (D.scale != nullptr && (D.mean == nullptr || D.rstd == nullptr || D.s1 == nullptr || D.s2 == nullptr))
                return CWN_ERR_BAD_ARG;
        }
        const void* ptrs[] = {D.dy, D.z, D.scale, D.shift, D.mean, D.rstd, D.s1, D.s2, D.out};
        for (const void* p : ptrs)
            if (!al16(p)) return CWN_ERR_ALIGN;
        if (D.N % 4 != 0 || D.ldz % 4 != 0 || D.lddy % 4 != 0 || D.ldout % 4 != 0) return CWN_ERR_ALIGN;
        B.d[i] = D;
        B.blk_start[i] = (int32_t)blocks;
        blocks += D.M > 0 ? D.N / 4 : 0;
    }
    for (int i = n; i <= CWN_MAX_NORM_DESCS; ++i) B.blk_start[i] = (int32_t)blocks;
    if (blocks == 0) return CWN_OK;
    norm_bwd_fused_kernel<<<dim3((unsigned)blocks), dim3(kFT), 0, (hipStream_t)stream_>>>(B, accumulate);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

// ---- Adam on one flat parameter buffer -----------------------------------------------------------
// torch.optim.Adam's update (exp/train_utils.py's optimizer.step(), no amsgrad) for ALL parameters
// of a model in one launch: parameters, gradients and both moments each live in one contiguous
// fp32 buffer (cwn_amd/dist.py::FlatGradBucket, cwn_amd/train.py::FlatAdam).  The multi-tensor
// fused Adam of the framework takes 8 launches x 22 us for the 265 tensors of the ZINC model; this
// is 48 MB of traffic in one pass.  `step` is a device counter (already incremented by the
// caller), so the launch is graph-capturable.
namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr, float b1, float b2, float eps, float wd,
                                                   const int32_t* __restrict__ step, const int64_t* __restrict__ active) {
    if (active != nullptr && *active <= 0) return;       // (uniform) an empty batch of a static epoch: the step changes nothing
    const float t = (float)*step;
    const float bc1 = 1.0f - powf(b1, t), bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    const float step_size = lr / bc1;
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    float pp[4], gg[4], mm[4], vv[4];
    const bool full = i0 + 3 < n;
    if (full) {
        const float4 a = *reinterpret_cast<const float4*>(p + i0), b = *reinterpret_cast<const float4*>(g + i0);
        const float4 c = *reinterpret_cast<const float4*>(m + i0), d = *reinterpret_cast<const float4*>(v + i0);
        pp[0] = a.x; pp[1] = a.y; pp[2] = a.z; pp[3] = a.w;
        gg[0] = b.x; gg[1] = b.y; gg[2] = b.z; gg[3] = b.w;
        mm[0] = c.x; mm[1] = c.y; mm[2] = c.z; mm[3] = c.w;
        vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
        for (int k = 0; k < 4; ++k) {
            const bool ok = i0 + k < n;
            pp[k] = ok ? p[i0 + k] : 0.f; gg[k] = ok ? g[i0 + k] : 0.f;
            mm[k] = ok ? m[i0 + k] : 0.f; vv[k] = ok ? v[i0 + k] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gr = gg[k] + wd * pp[k];
        mm[k] = b1 * mm[k] + (1.0f - b1) * gr;
        vv[k] = b2 * vv[k] + (1.0f - b2) * gr * gr;
        const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
        pp[k] = pp[k] - step_size * (mm[k] / denom);
    }
    if (full) {
        *reinterpret_cast<float4*>(p + i0) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        *reinterpret_cast<float4*>(m + i0) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *reinterpret_cast<float4*>(v + i0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
        for (int k = 0; k < 4 && i0 + k < n; ++k) {
            p[i0 + k] = pp[k]; m[i0 + k] = mm[k]; v[i0 + k] = vv[k];
        }
    }
}

}  // namespace

extern "C" int cwn_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, const int32_t* step, const int64_t* active,
                            cwn_stream_t stream_) {
    if (n < 0) return CWN_ERR_BAD_ARG;
    if (n == 0) return CWN_OK;
    if (p == nullptr || g == nullptr || m == nullptr || v == nullptr || step == nullptr) return CWN_ERR_BAD_ARG;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15u) return CWN_ERR_ALIGN;
    const int64_t threads = (n + 3) / 4, blocks = (threads + 255) / 256;
    if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    adam_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_>>>(p, g, m, v, n, lr, beta1, beta2,
                                                                                 eps, weight_decay, step, active);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

// ---- the start of a training step: zero the gradients, zero the step arena, count the step -----------------------------
namespace {
__global__ __launch_bounds__(256) void step_begin_kernel(uint4* __restrict__ a, int64_t na, uint4* __restrict__ b, int64_t nb,
                                                         int32_t* __restrict__ step, const int64_t* __restrict__ active,
                                                         int64_t* __restrict__ dropout_state) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += stride) {
        if (i < na) a[i] = z;
        else b[i - na] = z;
    }
    if (step != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && (active == nullptr || *active > 0)) *step += 1;
    if (dropout_state != nullptr && blockIdx.x == 0 && threadIdx.x == 1) dropout_state[1] += 1;      // (cwn_dropout.h: fresh masks per step)
}
}  // namespace

// ---- y += (1 + eps) x, several vectors per launch (include/cwn_hip.h: cwn_axpy_eps_f32) --------------------------------------
namespace {
constexpr int kAxpyThreads = 256, kAxpyPer = 4;            // float4 per thread
struct AxpyBatch {
    cwn_axpy_desc d[CWN_AXPY_MAX_DESCS];
    int32_t blk_start[CWN_AXPY_MAX_DESCS + 1];
    int32_t n;
};
__global__ __launch_bounds__(kAxpyThreads) void axpy_eps_kernel(AxpyBatch B) {
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_AXPY_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_axpy_desc& D = B.d[di];
    const float s = 1.0f + (D.eps != nullptr ? *D.eps : 0.f);
    const int64_t n4 = D.n / 4;
    const int64_t base = ((int64_t)blockIdx.x - B.blk_start[di]) * (kAxpyThreads * kAxpyPer) + threadIdx.x;
    float4 xv[kAxpyPer], yv[kAxpyPer];
#pragma unroll
    for (int u = 0; u < kAxpyPer; ++u) {
        const int64_t i = base + (int64_t)u * kAxpyThreads;
        if (i < n4) {
            xv[u] = reinterpret_cast<const float4*>(D.x)[i];
            yv[u] = reinterpret_cast<const float4*>(D.y)[i];
        }
    }
#pragma unroll
    for (int u = 0; u < kAxpyPer; ++u) {
        const int64_t i = base + (int64_t)u * kAxpyThreads;
        if (i < n4) {
            float4 r;
            r.x = yv[u].x + xv[u].x * s; r.y = yv[u].y + xv[u].y * s; r.z = yv[u].z + xv[u].z * s; r.w = yv[u].w + xv[u].w * s;
            reinterpret_cast<float4*>(D.y)[i] = r;
        }
    }
}
}  // namespace

extern "C" int cwn_axpy_eps_f32(const cwn_axpy_desc* descs, int n, cwn_stream_t stream_) {
    if (descs == nullptr || n < 1 || n > CWN_AXPY_MAX_DESCS) return CWN_ERR_BAD_ARG;
    AxpyBatch B{};
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        const cwn_axpy_desc& D = descs[i];
        if (D.n < 0 || (D.n & 3) != 0 || (D.n > 0 && (D.y == nullptr || D.x == nullptr))) return CWN_ERR_BAD_ARG;
        if ((((uintptr_t)D.y) | ((uintptr_t)D.x)) & 15u) return CWN_ERR_ALIGN;
        B.d[i] = D;
        B.blk_start[i] = (int32_t)blocks;
        blocks += (D.n / 4 + kAxpyThreads * kAxpyPer - 1) / (kAxpyThreads * kAxpyPer);
        if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    }
    for (int i = n; i <= CWN_AXPY_MAX_DESCS; ++i) B.blk_start[i] = (int32_t)blocks;
    B.n = n;
    if (blocks == 0) return CWN_OK;
    axpy_eps_kernel<<<dim3((unsigned)blocks), dim3(kAxpyThreads), 0, (hipStream_t)stream_>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_step_begin(void* a, int64_t a_bytes, void* b, int64_t b_bytes, int32_t* step, const int64_t* active,
                              int64_t* dropout_state, cwn_stream_t stream_) {
    if (a_bytes < 0 || b_bytes < 0 || (a_bytes & 15) || (b_bytes & 15)) return CWN_ERR_BAD_ARG;
    if ((a_bytes > 0 && a == nullptr) || (b_bytes > 0 && b == nullptr)) return CWN_ERR_BAD_ARG;
    if ((((uintptr_t)a) | ((uintptr_t)b)) & 15u) return CWN_ERR_ALIGN;
    const int64_t n = (a_bytes + b_bytes) / 16;
    if (n == 0 && step == nullptr && dropout_state == nullptr) return CWN_OK;
    int64_t blocks = (n + 4 * 256 - 1) / (4 * 256);        // four 16-B stores per thread
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    step_begin_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_>>>((uint4*)a, a_bytes / 16, (uint4*)b,
                                                                                       b_bytes / 16, step, active, dropout_state);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

// ---- dropout as a launch of its own: the applications no producing / consuming kernel takes (cwn_dropout.h) ------------------
namespace {
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t M, int N, int64_t ldx,
                                                      int64_t ldout, cwn_dropout d, const int64_t* __restrict__ m_dev, int vec) {
    cwn::Dropout drop;
    drop.init(d);
    const int64_t Mv = m_dev != nullptr ? (*m_dev < M ? *m_dev : M) : M;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t nq = Mv * (N / 4);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
            const int64_t r = i / (N / 4), c = (i - r * (N / 4)) * 4;
            float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
            const float4 m = drop.mul4((uint32_t)(((uint64_t)r * (uint64_t)N + (uint64_t)c) >> 2));
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            *reinterpret_cast<float4*>(out + r * ldout + c) = v;
        }
    } else {
        const int64_t n = Mv * N;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const int64_t r = i / N, c = i - r * N;
            out[r * ldout + c] = x[r * ldx + c] * drop.mul1((uint64_t)i);
        }
    }
}
}  // namespace

extern "C" int cwn_dropout_f32(const float* x, float* out, int64_t M, int32_t N, int64_t ldx, int64_t ldout, const cwn_dropout* drop,
                               const int64_t* m_dev, cwn_stream_t stream_) {
    if (M < 0 || N <= 0 || drop == nullptr || !(drop->p >= 0.f && drop->p < 1.f)) return CWN_ERR_BAD_ARG;
    if (M == 0) return CWN_OK;
    if (x == nullptr || out == nullptr || ldx < N || ldout < N) return CWN_ERR_BAD_ARG;
    if ((((uintptr_t)x) | ((uintptr_t)out)) & 3u) return CWN_ERR_ALIGN;
    const int vec = (al16(x) && al16(out) && N % 4 == 0 && ldx % 4 == 0 && ldout % 4 == 0) ? 1 : 0;
    const int64_t work = vec ? M * (N / 4) : M * N;
    int64_t blocks = (work + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
    dropout_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_>>>(x, out, M, N, ldx, ldout, *drop, m_dev, vec);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

// ---- the loss of a training step and its gradient in ONE launch ------------------------------------------
// exp/train_utils.py:62-73: loss = criterion(pred, targets); loss.backward().  For the elementwise-mean criteria
// (L1Loss 'regression', MSELoss 'mse_regression', BCEWithLogitsLoss 'bin_classification': exp/train_utils.py:20-31)
// the framework runs ~9 launches of a few hundred elements each (sub, abs, mean, fill, sign, div, mul ...: 45 us of a
// step); here one workgroup computes  loss = mean_i l(pred_i, y_i)  and  grad_i = dl/dpred_i / n  together.
namespace {

__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ pred, const float* __restrict__ y, int64_t n_cap,
                                                   int kind, float* __restrict__ loss, float* __restrict__ grad,
                                                   const int64_t* __restrict__ n_dev, int64_t cols) {
    __shared__ float part[256];
    __shared__ int cnt[256];
    // (a static batch: *n_dev complexes with `cols` predictions each are real, the rest of the n_cap elements is capacity)
    const int64_t n = n_dev != nullptr ? (*n_dev * cols < n_cap ? *n_dev * cols : n_cap) : n_cap;
    if (kind == CWN_LOSS_CE) {
        // torch.nn.CrossEntropyLoss() (exp/train_utils.py:21-22, REDDIT-BINARY / the TU datasets): pred [rows, cols] logits, y the
        // class of every row as int64; loss = mean over the rows with a class >= 0 of logsumexp(row) - row[class] (a negative
        // class = torch's ignore_index; a class >= cols poisons the loss with NaN), grad = (softmax(row) - onehot) / rows counted.  A thread per row, fixed reduction tree.
        const int64_t* cls = reinterpret_cast<const int64_t*>(y);
        const int64_t rows = n / cols, rows_cap = n_cap / cols;
        int valid = 0;
        for (int64_t r = threadIdx.x; r < rows; r += 256) valid += cls[r] >= 0 ? 1 : 0;
        cnt[threadIdx.x] = valid;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) cnt[threadIdx.x] += cnt[threadIdx.x + off];
            __syncthreads();
        }
        const float inv = 1.0f / (float)cnt[0];
        float s = 0.f;
        for (int64_t i = rows * cols + threadIdx.x; i < rows_cap * cols; i += 256) grad[i] = 0.f;
        for (int64_t r = threadIdx.x; r < rows; r += 256) {
            const float* p = pred + r * cols;
            float* g = grad + r * cols;
            const int64_t c = cls[r];
            if (c < 0) {                                   // ignored row (torch's ignore_index is negative)
                for (int64_t j = 0; j < cols; ++j) g[j] = 0.f;
                continue;
            }
            if (c >= cols) {
                // a class the logits have no column for: torch.nn.CrossEntropyLoss asserts on the device.  Never a silent
                // smaller denominator: the loss and this row's gradient are NaN, the caller sees a non-finite step.
                s = __builtin_nanf("");
                for (int64_t j = 0; j < cols; ++j) g[j] = __builtin_nanf("");
                continue;
            }
            float m = p[0];
            for (int64_t j = 1; j < cols; ++j) m = fmaxf(m, p[j]);
            float z = 0.f;
            for (int64_t j = 0; j < cols; ++j) z += expf(p[j] - m);
            const float lse = m + logf(z);
            s += lse - p[c];
            for (int64_t j = 0; j < cols; ++j) g[j] = (expf(p[j] - lse) - (j == c ? 1.f : 0.f)) * inv;
        }
        part[threadIdx.x] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) *loss = cnt[0] > 0 ? part[0] * inv : __int_as_float(0x7fc00000);
        return;
    }
    // null labels (exp/train_utils.py:64-66: `mask = ~torch.isnan(targets)`; ogbg-mol* tasks): not in the mean, no gradient
    int valid = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) valid += (y[i] == y[i]) ? 1 : 0;
    cnt[threadIdx.x] = valid;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) cnt[threadIdx.x] += cnt[threadIdx.x + off];
        __syncthreads();
    }
    const float inv = 1.0f / (float)cnt[0];           // (no valid target: the mean of nothing is NaN, as torch's)
    float s = 0.f;
    for (int64_t i = n + threadIdx.x; i < n_cap; i += 256) grad[i] = 0.f;       // rows past the batch's own
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float p = pred[i], t = y[i], d = p - t;
        if (!(t == t)) { grad[i] = 0.f; continue; }
        float l, gr;
        if (kind == CWN_LOSS_L1) {
            l = fabsf(d);
            gr = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);             // torch: sign(0) = 0
        } else if (kind == CWN_LOSS_MSE) {
            l = d * d;
            gr = 2.f * d;
        } else {                                                      // BCE with logits, the stable form torch uses
            l = fmaxf(p, 0.f) - p * t + log1pf(expf(-fabsf(p)));
            gr = 1.f / (1.f + expf(-p)) - t;
        }
        s += l;
        grad[i] = gr * inv;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {                         // fixed tree: deterministic
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = cnt[0] > 0 ? part[0] * inv : __int_as_float(0x7fc00000);
}

}  // namespace

extern "C" int cwn_loss_cols_f32(int32_t kind, const float* pred, const float* y, int64_t n, int64_t cols, float* loss, float* grad,
                                 const int64_t* n_dev, cwn_stream_t stream_) {
    if (kind < 0 || kind > CWN_LOSS_CE || n <= 0 || cols <= 0 || n % cols != 0 || pred == nullptr || y == nullptr ||
        loss == nullptr || grad == nullptr)
        return CWN_ERR_BAD_ARG;
    loss_kernel<<<dim3(1), dim3(256), 0, (hipStream_t)stream_>>>(pred, y, n, kind, loss, grad, n_dev, cols);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_loss_f32(int32_t kind, const float* pred, const float* y, int64_t n, float* loss, float* grad,
                            const int64_t* n_dev, cwn_stream_t stream_) {
    if (kind == CWN_LOSS_CE) return CWN_ERR_BAD_ARG;          // (needs the number of classes: cwn_loss_cols_f32)
    return cwn_loss_cols_f32(kind, pred, y, n, 1, loss, grad, n_dev, stream_);
}

// ---- embedding backward: dW[v, :] += sum over the cells that looked row v up of g[cell, :] ----------
// (torch.nn.Embedding / OGB Atom-BondEncoder tables: v_embed_init, e_embed_init of
// mp/molec_models.py:44-52, 237-245.)  The tables are tiny (28 x 128, 173 x 64) and a few rows take
// almost every lookup (most atoms are carbon), so a transposed CSR would have rows of 10^4 entries
// and same-address global atomics would serialise: every workgroup accumulates its band of cells
// into a private copy of the WHOLE table in LDS (ds_add_f32), then adds the non-zero rows to dW
// once.  fp32 atomics: the order of the band partials varies run to run (like the dW of the Linear
// layers).
namespace {

// cells per workgroup.  Round 2 took 128 with one float per lane: 26 workgroups of 64 dependent passes (index -> gradient
// row -> LDS add), 33 us per table at the ZINC batch of 128 -- the two calls were 4.5 % of the training step.  Four
// features per lane put eight cells of a 128-wide table in flight per pass, and 64-cell bands fill a hundred CUs.
constexpr int kEmbBand = 64;

// the integer features as the containers deliver them: int64, or float32 (truncated, as `.to(torch.long)` does)
__device__ __forceinline__ int64_t emb_index(const void* src, int f32, int64_t i) {
    return f32 ? (int64_t)reinterpret_cast<const float*>(src)[i] : reinterpret_cast<const int64_t*>(src)[i];
}

__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ g,
                                                            const void* __restrict__ src,
                                                            const int64_t* __restrict__ col_off,
                                                            const int64_t* __restrict__ col_size,
                                                            float* __restrict__ dW, int64_t n_cap, int cols,
                                                            int H, int64_t V, int src_f32, const int64_t* __restrict__ n_dev) {
    extern __shared__ float table[];          // [V][H]
    const int64_t n_rows = n_dev != nullptr ? (*n_dev < n_cap ? *n_dev : n_cap) : n_cap;
    const int64_t r0 = (int64_t)blockIdx.x * kEmbBand;
    if (r0 >= n_rows) return;                 // (uniform) a band past the batch's own rows
    const int64_t total = V * H;
    for (int64_t i = threadIdx.x; i < total; i += 256) table[i] = 0.f;
    __syncthreads();
    const int64_t r1 = r0 + kEmbBand < n_rows ? r0 + kEmbBand : n_rows;
    if ((H & 3) == 0 && H <= 1024) {
        const int lanes = H / 4;                  // threads walking one cell's features, four each
        const int per = 256 / lanes;              // cells in flight
        const int h0 = 4 * (threadIdx.x % lanes), sub = threadIdx.x / lanes;
        if (sub < per) {
            for (int64_t r = r0 + sub; r < r1; r += per) {
                const float4 gv = *reinterpret_cast<const float4*>(g + r * H + h0);
                for (int c = 0; c < cols; ++c) {
                    int64_t v = emb_index(src, src_f32, r * cols + c);
                    if (v < 0 || v >= (col_size != nullptr ? col_size[c] : V)) continue;   // flagged by the forward
                    if (col_off != nullptr) v += col_off[c];
                    float* t = table + v * H + h0;
                    atomicAdd(t, gv.x); atomicAdd(t + 1, gv.y); atomicAdd(t + 2, gv.z); atomicAdd(t + 3, gv.w);
                }
            }
        }
    } else {
        const int lanes = H < 256 ? H : 256;      // threads walking one cell's features
        const int per = 256 / lanes;              // cells in flight
        const int h0 = threadIdx.x % lanes, sub = threadIdx.x / lanes;
        if (sub < per) {
            for (int64_t r = r0 + sub; r < r1; r += per) {
                for (int c = 0; c < cols; ++c) {
                    int64_t v = emb_index(src, src_f32, r * cols + c);
                    if (v < 0 || v >= (col_size != nullptr ? col_size[c] : V)) continue;   // flagged by the forward
                    if (col_off != nullptr) v += col_off[c];
                    for (int h = h0; h < H; h += lanes) atomicAdd(&table[v * H + h], g[r * H + h]);
                }
            }
        }
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const float t = table[i];
        if (t != 0.f) atomicAdd(dW + i, t);
    }
}

// SEVERAL tables summed per cell (the OGB Atom / BondEncoder: nine / three integer feature columns, mp/molec_models.py:237-245),
// H = 64 / 128 / 256, without float atomics in LDS: the table-in-LDS form above spent 97 us on the atom tables of a molhiv batch
// of 512 (twice per training step: 13 % of it) -- a column has 2 .. 12 distinct values, so the 16 cells a pass has in flight
// hit the same few LDS rows and every ds_add_f32 serialises.  Here the band's 64 gradient rows are staged once, lane = cell
// holds the cell's index of the current column, and the DISTINCT values present in the band are walked with ballots: for value
// v one thread per feature adds ITS feature of the cells that hold v out of LDS (conflict-free reads, ascending cell order)
// and hands one partial per (v, feature) to dW.  The 256 / H feature groups take the distinct values in turn.
template <int H>
__global__ __launch_bounds__(256) void embedding_bwd_cols_kernel(const float* __restrict__ g, const void* __restrict__ src,
                                                                 const int64_t* __restrict__ col_off, const int64_t* __restrict__ col_size,
                                                                 float* __restrict__ dW, int64_t n_cap, int cols, int64_t V, int src_f32,
                                                                 const int64_t* __restrict__ n_dev) {
    __shared__ __attribute__((aligned(16))) float rows[kEmbBand][H];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t n_rows = n_dev != nullptr ? (*n_dev < n_cap ? *n_dev : n_cap) : n_cap;
    const int64_t r0 = (int64_t)blockIdx.x * kEmbBand;
    if (r0 >= n_rows) return;                 // (uniform) a band past the batch's own rows
    const int n = (int)((n_rows - r0) < kEmbBand ? (n_rows - r0) : kEmbBand);
    for (int i = tid; i < kEmbBand * (H / 4); i += 256) {
        const int r = i / (H / 4), c4 = i % (H / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) v = *reinterpret_cast<const float4*>(g + (r0 + r) * H + 4 * c4);
        *reinterpret_cast<float4*>(&rows[r][4 * c4]) = v;
    }
    __syncthreads();
    constexpr int kG = 256 / H > 0 ? 256 / H : 1;                   // feature groups (H = 64: 4, 128: 2, 256: 1)
    const int h = tid % H, q = tid / H;
    // the indices of kColChunk columns are REQUESTED together (with their table bounds) before the first is walked: a column's
    // walk ends in atomics, which the compiler will not move a load across (molhiv-512: bond tables 18.5 -> 13.9 us, atom tables
    // 31 -> 29: what is left there is the ~650 k global atomics of the launch -- splitting a value's cells over the feature
    // groups instead of the values, four times the atomics, took 56 us)
    constexpr int kColChunk = 8;
    for (int c0 = 0; c0 < cols; c0 += kColChunk) {
        int64_t ids[kColChunk], lims[kColChunk], offs[kColChunk];
#pragma unroll
        for (int u = 0; u < kColChunk; ++u) {
            const int c = c0 + u < cols ? c0 + u : cols - 1;
            ids[u] = lane < n ? emb_index(src, src_f32, (r0 + lane) * cols + c) : -1;     // every wave holds all 64 cells
            lims[u] = col_size != nullptr ? col_size[c] : V;
            offs[u] = col_off != nullptr ? col_off[c] : 0;
        }
#pragma unroll
        for (int u = 0; u < kColChunk; ++u) {
            if (c0 + u >= cols) break;                                                         // (uniform)
            int64_t id = ids[u];
            const int64_t off = offs[u];
            if (id < 0 || id >= lims[u]) id = -1;                                              // (flagged by the forward)
            unsigned long long todo = __ballot(id >= 0);
            int it = 0;
            while (todo != 0ull) {                                                             // (uniform across the workgroup)
                const int leader = __builtin_ctzll(todo);
                const int64_t v = __shfl(id, leader, 64);
                const unsigned long long bal = __ballot(id == v);
                todo &= ~bal;
                if ((it++ % kG) != q) continue;
                unsigned long long m = bal;
                float acc = 0.f;
                while (m != 0ull) {
                    const int r = __builtin_ctzll(m);
                    m &= m - 1ull;
                    acc += rows[r][h];
                }
                if (acc != 0.f) atomicAdd(dW + (size_t)(off + v) * H + h, acc);
            }
        }
    }
}

// One table, few rows (ZINC: 28 atom types, 4 bond types), H = 64 / 128 / 256: the band's 64 gradient rows go to LDS in one
// coalesced pass, every wave knows each cell's table row (lane = cell), and for table row v a BALLOT gives the cells that
// hit it -- a thread then adds ITS feature of those cells out of LDS and hands one partial per (v, feature) to global
// memory.  No float atomics in LDS: ds_add_f32 runs at ~100 cycles per wave instruction (measured in cwn_layer_bwd.hip),
// which is what made the table-in-LDS form above 14 us per ZINC table.
template <int H>
__global__ __launch_bounds__(256) void embedding_bwd_one_table_kernel(const float* __restrict__ g, const void* __restrict__ src,
                                                                     float* __restrict__ dW, int64_t n_cap, int V, int src_f32,
                                                                     const int64_t* __restrict__ n_dev) {
    __shared__ __attribute__((aligned(16))) float rows[kEmbBand][H];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t n_rows = n_dev != nullptr ? (*n_dev < n_cap ? *n_dev : n_cap) : n_cap;
    const int64_t r0 = (int64_t)blockIdx.x * kEmbBand;
    if (r0 >= n_rows) return;                 // (uniform) a band past the batch's own rows
    const int n = (int)((n_rows - r0) < kEmbBand ? (n_rows - r0) : kEmbBand);
    // stage: kEmbBand x H floats, 16 bytes a thread and pass, row-contiguous
    for (int i = tid; i < kEmbBand * (H / 4); i += 256) {
        const int r = i / (H / 4), c4 = i % (H / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) v = *reinterpret_cast<const float4*>(g + (r0 + r) * H + 4 * c4);
        *reinterpret_cast<float4*>(&rows[r][4 * c4]) = v;
    }
    const int64_t id = lane < n ? emb_index(src, src_f32, r0 + lane) : -1;   // lane = cell of the band (every wave holds all 64)
    __syncthreads();
    constexpr int kSlices = 256 / H > 0 ? 256 / H : 1;               // threads per feature (H = 256: 1, 128: 2, 64: 4)
    constexpr int kPer = kEmbBand / kSlices;                          // cells of a slice
    const int h = tid % H, sl = tid / H;
    for (int v = 0; v < V; ++v) {
        const unsigned long long bal = __ballot(id == (int64_t)v);
        if (bal == 0ull) continue;                                     // uniform
        unsigned long long m = kPer == 64 ? bal : (bal >> (sl * kPer)) & ((1ull << kPer) - 1ull);
        float acc = 0.f;
        while (m != 0ull) {
            const int r = sl * kPer + __builtin_ctzll(m);
            m &= m - 1ull;
            acc += rows[r][h];
        }
        if (acc != 0.f) atomicAdd(dW + (size_t)v * H + h, acc);
    }
}

}  // namespace

namespace {

// forward of the same lookup: out[r, :] = sum_c W[src[r, c], :], columns in order (bit-identical to
// summing the per-column torch.nn.Embedding outputs); an index outside [0, V) sets bit 1 of the
// sticky error word (the plan builds' flag: raised as IndexError by the host) and is skipped
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const float* __restrict__ W,
                                                            const int64_t* __restrict__ src,
                                                            const int64_t* __restrict__ col_off,
                                                            const int64_t* __restrict__ col_size,
                                                            float* __restrict__ out, int64_t n_rows, int cols,
                                                            int H, int64_t V, int G, int32_t* err) {
    const int gl = threadIdx.x & (G - 1);
    const int64_t r = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (r >= n_rows) return;
    for (int h = 4 * gl; h < H; h += 4 * G) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c < cols; ++c) {
            int64_t v = src[r * cols + c];
            const int64_t lim = col_size != nullptr ? col_size[c] : V;   // per TABLE, not per concatenation
            if (v < 0 || v >= lim) {
                if (h == 0) atomicOr(err, 2);
                continue;
            }
            if (col_off != nullptr) v += col_off[c];
            const float4 w = *reinterpret_cast<const float4*>(W + v * H + h);
            acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
        }
        *reinterpret_cast<float4*>(out + r * H + h) = acc;
    }
}

}  // namespace

extern "C" int cwn_embedding_fwd_f32(const float* W, const int64_t* src, const int64_t* col_off,
                                     const int64_t* col_size, float* out, int64_t n_rows, int32_t cols,
                                     int32_t H, int64_t V, int32_t* err_flag, cwn_stream_t stream_) {
    if (n_rows < 0 || cols <= 0 || H <= 0 || V <= 0 || (H & 3) != 0) return CWN_ERR_BAD_ARG;
    if (n_rows == 0) return CWN_OK;
    if (W == nullptr || src == nullptr || out == nullptr || err_flag == nullptr) return CWN_ERR_BAD_ARG;
    if ((((uintptr_t)W) | ((uintptr_t)out)) & 15u) return CWN_ERR_ALIGN;
    int G = 1;
    while (G < H / 4 && G < 64) G <<= 1;
    const int64_t blocks = (n_rows + 256 / G - 1) / (256 / G);
    if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    if ((col_off == nullptr) != (col_size == nullptr)) return CWN_ERR_BAD_ARG;
    embedding_fwd_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_>>>(
        W, src, col_off, col_size, out, n_rows, cols, H, V, G, err_flag);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

// ---- the backward of the model's front in one launch (include/cwn_hip.h: cwn_front_bwd) ---------------------------------
namespace {
// (bands of 32 cells, not cwn_embedding_bwd_f32's 64: a thread walks its items' edges and rings one dependent load after
// another -- four items per thread instead of eight, twice the workgroups: 21 -> ~12 us at the ZINC batch)
constexpr int kFrontBand = 32;

template <int H>
__global__ __launch_bounds__(256) void front_bwd_kernel(cwn_front_bwd A, int nb0) {
    __shared__ __attribute__((aligned(16))) float rows[kFrontBand][H];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool vert = (int)blockIdx.x < nb0;
    const int64_t cap = vert ? A.n0 : A.n1;
    const int64_t* nd = vert ? A.n0_dev : A.n1_dev;
    const int64_t n_rows = nd != nullptr ? (*nd < cap ? *nd : cap) : cap;
    const int64_t r0 = (int64_t)((int)blockIdx.x - (vert ? 0 : nb0)) * kFrontBand;
    if (r0 >= n_rows) return;                 // (uniform) a band past the batch's own rows
    const int n = (int)((n_rows - r0) < kFrontBand ? (n_rows - r0) : kFrontBand);
    const float scale = A.halve ? 0.5f : 1.0f;
    const bool g1_to_v = A.e_src == nullptr && A.g1 != nullptr;
    for (int i = tid; i < kFrontBand * (H / 4); i += 256) {
        const int r = i / (H / 4), c4 = i % (H / 4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) {
            const int64_t row = r0 + r;
            if (!vert) {
                acc = *reinterpret_cast<const float4*>(A.g1 + row * H + 4 * c4);
            } else {
                if (A.g0 != nullptr) acc = *reinterpret_cast<const float4*>(A.g0 + row * H + 4 * c4);
                if (A.rowptr1 != nullptr) {
                    const int k1 = A.rowptr1[row + 1];
                    for (int k = A.rowptr1[row]; k < k1; ++k) {
                        const int64_t e = A.col1[k];
                        if (g1_to_v) {
                            const float4 t = *reinterpret_cast<const float4*>(A.g1 + e * H + 4 * c4);
                            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                        }
                        if (A.rowptr2 != nullptr && A.g2 != nullptr) {
                            const int q1 = A.rowptr2[e + 1];
                            for (int q = A.rowptr2[e]; q < q1; ++q) {
                                const float4 t = *reinterpret_cast<const float4*>(A.g2 + (int64_t)A.col2[q] * H + 4 * c4);
                                acc.x += scale * t.x; acc.y += scale * t.y; acc.z += scale * t.z; acc.w += scale * t.w;
                            }
                        }
                    }
                }
            }
        }
        *reinterpret_cast<float4*>(&rows[r][4 * c4]) = acc;
    }
    const void* src = vert ? A.v_src : A.e_src;
    const int f32 = vert ? (A.src_f32 & 1) : ((A.src_f32 >> 1) & 1);
    const int V = vert ? A.Vv : A.Ve;
    float* const dW = vert ? A.dWv : A.dWe;
    const int64_t id = lane < n ? emb_index(src, f32, r0 + lane) : -1;       // lane = cell of the band (every wave holds all 64)
    __syncthreads();
    constexpr int kSlices = 256 / H > 0 ? 256 / H : 1;
    constexpr int kPer = kFrontBand / kSlices;
    static_assert(kPer >= 1 && kFrontBand <= 64, "a wave's ballot covers the band");
    const int h = tid % H, sl = tid / H;
    for (int v = 0; v < V; ++v) {
        const unsigned long long bal = __ballot(id == (int64_t)v);
        if (bal == 0ull) continue;                                     // uniform
        unsigned long long m = (bal >> (sl * kPer)) & ((1ull << kPer) - 1ull);
        float acc = 0.f;
        while (m != 0ull) {
            const int r = sl * kPer + __builtin_ctzll(m);
            m &= m - 1ull;
            acc += rows[r][h];
        }
        if (acc != 0.f) atomicAdd(dW + (size_t)v * H + h, acc);
    }
}
}  // namespace

extern "C" int cwn_embed_front_bwd_f32(const cwn_front_bwd* a, cwn_stream_t stream_) {
    if (a == nullptr || a->n0 < 0 || a->n1 < 0) return CWN_ERR_BAD_ARG;
    if ((a->H != 64 && a->H != 128 && a->H != 256) || a->Vv <= 0 || a->Vv > 64 || a->Ve < 0 || a->Ve > 64) return CWN_ERR_BAD_ARG;
    const bool edges = a->e_src != nullptr && a->n1 > 0;
    if (a->n0 > 0 && (a->v_src == nullptr || a->dWv == nullptr)) return CWN_ERR_BAD_ARG;
    if (edges && (a->dWe == nullptr || a->g1 == nullptr || a->Ve <= 0)) return CWN_ERR_BAD_ARG;
    if ((a->rowptr1 == nullptr) != (a->col1 == nullptr) || (a->rowptr2 == nullptr) != (a->col2 == nullptr)) return CWN_ERR_BAD_ARG;
    const void* ptrs[] = {a->g0, a->g1, a->g2};
    for (const void* p : ptrs)
        if ((uintptr_t)p & 15u) return CWN_ERR_ALIGN;
    const int64_t nb0 = (a->n0 + kFrontBand - 1) / kFrontBand, nb1 = edges ? (a->n1 + kFrontBand - 1) / kFrontBand : 0;
    if (nb0 + nb1 == 0) return CWN_OK;
    if (nb0 + nb1 >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream_;
    const dim3 grid((unsigned)(nb0 + nb1));
    if (a->H == 64) front_bwd_kernel<64><<<grid, dim3(256), 0, st>>>(*a, (int)nb0);
    else if (a->H == 128) front_bwd_kernel<128><<<grid, dim3(256), 0, st>>>(*a, (int)nb0);
    else front_bwd_kernel<256><<<grid, dim3(256), 0, st>>>(*a, (int)nb0);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_embedding_bwd_f32(const float* g, const void* src, const int64_t* col_off,
                                     const int64_t* col_size, float* dW, int64_t n_rows, int32_t cols,
                                     int32_t H, int64_t V, int32_t src_f32, const int64_t* n_dev, cwn_stream_t stream_) {
    if (n_rows < 0 || cols <= 0 || H <= 0 || V <= 0) return CWN_ERR_BAD_ARG;
    if (n_rows == 0) return CWN_OK;
    if (g == nullptr || src == nullptr || dW == nullptr) return CWN_ERR_BAD_ARG;
    if ((H & 3) == 0 && ((uintptr_t)g & 15u)) return CWN_ERR_ALIGN;       // gradient rows are read 16 bytes a lane
    const int64_t blocks = (n_rows + kEmbBand - 1) / kEmbBand;
    if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    if (cols == 1 && col_off == nullptr && V <= 64 && (H == 64 || H == 128 || H == 256)) {
        // one small table: the ballot form (no float atomics in LDS)
        hipStream_t st = (hipStream_t)stream_;
        const int sf = src_f32 ? 1 : 0;
        if (H == 64) embedding_bwd_one_table_kernel<64><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(g, src, dW, n_rows, (int)V, sf, n_dev);
        else if (H == 128) embedding_bwd_one_table_kernel<128><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(g, src, dW, n_rows, (int)V, sf, n_dev);
        else embedding_bwd_one_table_kernel<256><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(g, src, dW, n_rows, (int)V, sf, n_dev);
        return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
    }
    if ((H == 64 || H == 128 || H == 256) && !((uintptr_t)g & 15u)) {
        // several tables, or one table of many rows: the ballot form over the distinct values of every column
        hipStream_t st = (hipStream_t)stream_;
        const int sf = src_f32 ? 1 : 0;
        if (H == 64) embedding_bwd_cols_kernel<64><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(g, src, col_off, col_size, dW, n_rows, cols, V, sf, n_dev);
        else if (H == 128) embedding_bwd_cols_kernel<128><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(g, src, col_off, col_size, dW, n_rows, cols, V, sf, n_dev);
        else embedding_bwd_cols_kernel<256><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(g, src, col_off, col_size, dW, n_rows, cols, V, sf, n_dev);
        return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
    }
    const int64_t bytes = V * H * 4;
    if (bytes > 60 * 1024) return CWN_ERR_TOO_LARGE;       // (other widths: the table-in-LDS form; the table must fit one workgroup's LDS)
    embedding_bwd_kernel<<<dim3((unsigned)blocks), dim3(256), (size_t)bytes, (hipStream_t)stream_>>>(
        g, src, col_off, col_size, dW, n_rows, cols, H, V, src_f32 ? 1 : 0, n_dev);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
