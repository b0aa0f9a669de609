// cwn_gemm.hip -- grouped fp32 GEMM on the gfx950 matrix cores for the DENSE parts of the path:
// the coboundary-message products Y1 = X_d W1^T + b, Y2 = X_{d+1} W2^T (mp/layers.py:290-293,
// restructured) and the update / combine MLPs (mp/layers.py:193-199, 303-325).
//
//   Y[g] = epilogue( prologue(X[g] | X2[g]) . W[g]^T + bias[g] )        g = 0 .. n-1, ONE launch
//
// * v_mfma_f32_16x16x4_f32: exact fp32 (an fmaf chain), 157 TF peak; no TF32 on gfx950.
// * operand roles are swapped (A = W rows, B = X rows) so that a lane's four accumulator
//   registers are four CONSECUTIVE output columns of one output row: the epilogue is one 16-B
//   store per lane per tile, bias / scale / shift are 16-B loads.
// * K is walked in slabs of 16; inside a slab lane group g = lane>>4 owns k = 4g..4g+3, so both
//   MFMA fragments are one 16-B read per lane; MFMA step s multiplies the s-th component.
// * tiles go through LDS: a block stages a 32-row X tile and the 128-row W tile (K chunk of 128,
//   80 KiB) with row-contiguous, fully coalesced 16-B global reads (a fragment-shaped global load
//   touches 16 different rows per quarter-wave and is bound by the per-CU address unit: measured
//   14-17 us vs the 4 us of this form on the ZINC-128 shape), stores them XOR-swizzled
//   (chunk ^ (row & 15)) and reads fragments with conflict-free ds_read_b128.
// * blocks are persistent over their descriptor's M tiles, so the W tile is staged once per block
//   (N <= 128, K <= 128: the shape of every GEMM on the path); at M ~ 1e4 rows the grid is one
//   tile per block and fills the 256 CUs.
// * block = 4 waves = 32 rows x 128 columns; each wave 32 x 32 (2 x 2 MFMA tiles, 16 acc VGPRs).
// * optional fused pieces: K-concatenation of two inputs (combine_nn's cat), per-input-column
//   affine + ReLU prologue (BatchNorm apply of the producing layer), bias, per-output-column
//   affine (BatchNorm in eval mode), ReLU, and per-column sum / sum-of-squares accumulation
//   (BatchNorm batch statistics in training mode).
#include <hip/hip_runtime.h>
#include "../../include/cwn_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int BM = 32;    // rows per tile
constexpr int BN = 128;   // columns per tile (4 waves x 32)
constexpr int BK = 128;   // K chunk staged in LDS (one 512-B row per tile row)
constexpr int RT = 2;     // 16-row MFMA tiles per wave
constexpr int CT = 2;     // 16-col MFMA tiles per wave
constexpr int kLdsBytes = (BM + BN) * BK * 4;   // 80 KiB: X tile + W tile

struct GemmBatch {
    cwn_gemm_desc d[CWN_MAX_DESCS];
    int32_t blk_start[CWN_MAX_DESCS + 1];   // first block of each descriptor
    int32_t n_tiles_n[CWN_MAX_DESCS];
    int32_t n_tiles[CWN_MAX_DESCS];         // tiles_m * tiles_n
    int32_t vec[CWN_MAX_DESCS];             // 16-B global accesses allowed (host-checked)
    int32_t n;
};

// LDS image of a tile: row r (512 B) holds 32 chunks of 16 B; chunk c is stored at chunk position
// c ^ (r & 15).  Fragment reads (16 lanes = 16 different rows, same chunk) then hit 16 different
// 16-B slots of the 256-B bank row: ds_read_b128 is conflict-free (MI355X_MICROARCH.md, LDS).
__device__ __forceinline__ int lds_off(int row, int chunk) {   // in floats
    return row * BK + ((chunk ^ (row & 15)) << 2);
}

// Staging of a `ROWS` x BK tile, columns [k0, k0+BK) of the (possibly K-concatenated) matrix, in
// two phases so that EVERY global load of the block's tiles (16 per thread for W, 4 for X) is in
// flight before the first LDS write.  Global reads are row-contiguous: 32 consecutive lanes read
// one full 512-B row.
template <int ROWS>
struct Staged {
    static constexpr int U = ROWS * (BK / 4) / kThreads;
    f32x4 v[U];
};

template <bool FAST, bool PRO, int ROWS>
__device__ __forceinline__ void stage_load(Staged<ROWS>& st, int64_t row0, int64_t row_max,
                                           const float* __restrict__ P1, int64_t ld1, int K1,
                                           const float* __restrict__ P2, int64_t ld2, int K2, int k0) {
#pragma unroll
    for (int u = 0; u < Staged<ROWS>::U; ++u) {
        const int q = u * kThreads + threadIdx.x;
        const int r = q >> 5, c = q & 31;
        const int64_t grow = row0 + r < row_max ? row0 + r : row_max - 1;   // clamped, never faults
        const int k = k0 + 4 * c;
        const bool second = K2 > 0 && k >= K1;   // never touch P2 when there is no second input
        const float* base = second ? P2 : P1;
        const int64_t ld = second ? ld2 : ld1;
        const int kk = second ? k - K1 : k;
        const int kmax = second ? K2 : K1;
        if constexpr (FAST) {
            // kmax % 4 == 0; the column is clamped into range, the zeroing happens at store time
            st.v[u] = *reinterpret_cast<const f32x4*>(base + grow * ld + (kk < kmax ? kk : 0));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) st.v[u][t] = base[grow * ld + (kk + t < kmax ? kk + t : 0)];
        }
    }
}

template <bool FAST, bool PRO, int ROWS>
__device__ __forceinline__ void stage_store(float* lds, Staged<ROWS>& st, int K1, int K2, int k0,
                                            const float* __restrict__ in_scale,
                                            const float* __restrict__ in_shift, bool in_relu) {
#pragma unroll
    for (int u = 0; u < Staged<ROWS>::U; ++u) {
        const int q = u * kThreads + threadIdx.x;
        const int r = q >> 5, c = q & 31;
        const int k = k0 + 4 * c;
        const bool second = K2 > 0 && k >= K1;   // never touch P2 when there is no second input
        const int kk = second ? k - K1 : k;
        const int kmax = second ? K2 : K1;
        f32x4 v = st.v[u];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float x = v[t];
            if constexpr (PRO) {
                if (in_scale != nullptr && !second && kk + t < kmax) {
                    x = x * in_scale[kk + t] + in_shift[kk + t];
                    x = in_relu ? fmaxf(x, 0.f) : x;
                }
            }
            v[t] = kk + t < kmax ? x : 0.f;
        }
        *reinterpret_cast<f32x4*>(lds + lds_off(r, c)) = v;
    }
}

template <bool FAST, bool PRO>
__global__ __launch_bounds__(kThreads) void gemm_kernel(GemmBatch B) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ldsX = smem;               // [BM][BK]
    float* ldsW = smem + BM * BK;     // [BN][BK]
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_gemm_desc& D = B.d[di];
    const int nblk = B.blk_start[di + 1] - B.blk_start[di];
    const int tiles_n = B.n_tiles_n[di], tiles = B.n_tiles[di];
    const bool vec = FAST;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    // descriptor fields into registers once
    const float* const Xp = D.X;
    const float* const X2p = D.X2;
    const float* const Wp = D.W;
    const int64_t ldx = D.ldx, ldx2 = D.ldx2, ldw = D.ldw, ldy = D.ldy, M = D.M;
    const int N = D.N, K1 = D.K, K2 = D.K2, Ktot = D.K + D.K2;
    const float* const in_scale = D.in_scale;
    const float* const in_shift = D.in_shift;
    const bool in_relu = D.in_relu != 0;
    const int kchunks = (Ktot + BK - 1) / BK;

    // persistent over the descriptor's tiles, tile_n-major so that consecutive iterations of a
    // block reuse the W tile already in LDS (always, when N <= 128 and K <= 128)
    int cur_tn = -1;
    for (int tile = blockIdx.x - B.blk_start[di]; tile < tiles; tile += nblk) {
        const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
        const int64_t m_base = (int64_t)tile_m * BM;
        const int n_base = tile_n * BN + wave * (CT * 16);

        f32x4 acc[CT][RT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[ct][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kc = 0; kc < kchunks; ++kc) {
            const int k0 = kc * BK;
            const bool need_w = kchunks > 1 || cur_tn != tile_n;
            Staged<BN> sw;
            Staged<BM> sx;
            if (need_w)
                stage_load<FAST, false, BN>(sw, (int64_t)tile_n * BN, N, Wp, ldw, Ktot, nullptr, 0, 0, k0);
            stage_load<FAST, PRO, BM>(sx, m_base, M, Xp, ldx, K1, X2p, ldx2, K2, k0);
            __syncthreads();                     // previous readers of the LDS tiles are done
            if (need_w) stage_store<FAST, false, BN>(ldsW, sw, Ktot, 0, k0, nullptr, nullptr, false);
            stage_store<FAST, PRO, BM>(ldsX, sx, K1, K2, k0, in_scale, in_shift, in_relu);
            __syncthreads();
            const int kslabs = (min(BK, Ktot - k0) + 15) / 16;
            for (int sl = 0; sl < kslabs; ++sl) {
                f32x4 w[CT], x[RT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    w[ct] = *reinterpret_cast<const f32x4*>(
                        ldsW + lds_off(wave * (CT * 16) + ct * 16 + j, 4 * sl + g));
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    x[rt] = *reinterpret_cast<const f32x4*>(ldsX + lds_off(rt * 16 + j, 4 * sl + g));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][t], x[rt][t],
                                                                               acc[ct][rt], 0, 0, 0);
            }
        }
        cur_tn = tile_n;

        // epilogue: acc[ct][rt][r] = Y[m_base + rt*16 + j][n_base + ct*16 + 4g + r]
        bool xok[RT];
        int64_t xrow[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            xrow[rt] = m_base + rt * 16 + j;
            xok[rt] = xrow[rt] < M;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int n0 = n_base + ct * 16 + 4 * g;
            if (n0 >= N) continue;
            const bool full = n0 + 3 < N;
            float bias[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n0 + r < N) {
                    if (D.bias != nullptr) bias[r] = D.bias[n0 + r];
                    if (D.out_scale != nullptr) {
                        sc[r] = D.out_scale[n0 + r];
                        sh[r] = D.out_shift[n0 + r];
                    }
                }
            }
            float csum[4] = {0.f, 0.f, 0.f, 0.f}, csq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 v = acc[ct][rt];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y = v[r] + bias[r];
                    if (D.col_sum != nullptr && xok[rt]) {   // statistics of the pre-normalisation value
                        csum[r] += y;
                        csq[r] += y * y;
                    }
                    y = y * sc[r] + sh[r];
                    v[r] = D.relu ? fmaxf(y, 0.f) : y;
                }
                if (!xok[rt]) continue;
                float* yp = D.Y + xrow[rt] * ldy + n0;
                if (full && vec) {
                    *reinterpret_cast<f32x4*>(yp) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + r < N) yp[r] = v[r];
                }
            }
            if (D.col_sum != nullptr) {
                // reduce over the 16 rows held by lanes j = 0..15 of this lane group, one atomic each
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = csum[r], b = csq[r];
#pragma unroll
                    for (int o = 8; o >= 1; o >>= 1) {
                        a += __shfl_xor(a, o, 16);
                        b += __shfl_xor(b, o, 16);
                    }
                    if (j == 0 && n0 + r < N) {
                        atomicAdd(D.col_sum + n0 + r, a);
                        atomicAdd(D.col_sumsq + n0 + r, b);
                    }
                }
            }
        }
    }
}

inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; }

}  // namespace

extern "C" int cwn_gemm_f32(const cwn_gemm_desc* descs, int n, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_DESCS) return CWN_ERR_BAD_ARG;
    GemmBatch B{};
    B.n = n;
    int64_t total_tiles = 0;
    for (int i = 0; i < n; ++i) {
        const cwn_gemm_desc& D = descs[i];
        if (D.M < 0 || D.N <= 0 || D.K <= 0 || D.K2 < 0) return CWN_ERR_BAD_ARG;
        if (D.M > 0 && (D.X == nullptr || D.W == nullptr || D.Y == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.K2 > 0 && (D.X2 == nullptr || (D.K % 4) != 0)) return CWN_ERR_BAD_ARG;
        if ((D.in_scale == nullptr) != (D.in_shift == nullptr)) return CWN_ERR_BAD_ARG;
        if ((D.out_scale == nullptr) != (D.out_shift == nullptr)) return CWN_ERR_BAD_ARG;
        if ((D.col_sum == nullptr) != (D.col_sumsq == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.ldx < D.K || D.ldw < D.K + D.K2 || D.ldy < D.N || (D.K2 > 0 && D.ldx2 < D.K2))
            return CWN_ERR_BAD_ARG;
        const void* ptrs[] = {D.X, D.X2, D.W, D.Y};
        for (const void* p : ptrs)
            if (p != nullptr && ((uintptr_t)p & 3u)) return CWN_ERR_ALIGN;
        B.vec[i] = (al16(D.X) && al16(D.X2) && al16(D.W) && al16(D.Y) && D.ldx % 4 == 0 &&
                    D.ldw % 4 == 0 && D.ldy % 4 == 0 && (D.K2 == 0 || D.ldx2 % 4 == 0) &&
                    D.K % 4 == 0 && D.K2 % 4 == 0) ? 1 : 0;
        B.d[i] = D;
        const int64_t tm = (D.M + BM - 1) / BM;
        const int tn = (D.N + BN - 1) / BN;
        if (tm * tn >= INT32_MAX) return CWN_ERR_TOO_LARGE;
        B.n_tiles_n[i] = tn;
        B.n_tiles[i] = (int32_t)(tm * tn);
        total_tiles += tm * tn;
    }
    if (total_tiles == 0) return CWN_OK;
    // persistent blocks: at most ~2 per CU in total (80 KiB of LDS each), shared between the
    // descriptors in proportion to their tile counts; each block walks its descriptor's tiles
    const int64_t budget = 512;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        int64_t nb = B.n_tiles[i];
        if (total_tiles > budget) {
            nb = (B.n_tiles[i] * budget + total_tiles - 1) / total_tiles;
            if (nb < 1 && B.n_tiles[i] > 0) nb = 1;
            if (nb > B.n_tiles[i]) nb = B.n_tiles[i];
        }
        B.blk_start[i] = (int32_t)blocks;
        blocks += nb;
    }
    for (int i = n; i <= CWN_MAX_DESCS; ++i) B.blk_start[i] = (int32_t)blocks;
    bool fast = true, pro = false;
    for (int i = 0; i < n; ++i) {
        fast = fast && B.vec[i] != 0;
        pro = pro || B.d[i].in_scale != nullptr;
    }
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[] = {(const void*)gemm_kernel<true, false>, (const void*)gemm_kernel<true, true>,
                             (const void*)gemm_kernel<false, false>, (const void*)gemm_kernel<false, true>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess)
                return CWN_ERR_LAUNCH;
        attr_set = true;
    }
    const dim3 grid((unsigned)blocks), block(kThreads);
    hipStream_t st = (hipStream_t)stream_;
    if (fast && !pro) gemm_kernel<true, false><<<grid, block, kLdsBytes, st>>>(B);
    else if (fast) gemm_kernel<true, true><<<grid, block, kLdsBytes, st>>>(B);
    else if (!pro) gemm_kernel<false, false><<<grid, block, kLdsBytes, st>>>(B);
    else gemm_kernel<false, true><<<grid, block, kLdsBytes, st>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
