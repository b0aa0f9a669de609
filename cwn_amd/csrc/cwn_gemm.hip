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
// * WEIGHTS STATIONARY: each wave keeps the W fragments of its 32 output columns for the whole K in
//   registers (64 VGPRs at K <= 128) across all tiles of a persistent block; LDS only
//   double-buffers the 32-row X tile (2 x 16 KiB), so there is ONE barrier per tile and three
//   blocks fit a CU.  (With the 64-KiB W tile in LDS only two waves per SIMD fit and the MFMA
//   pipe measured 56 % busy on the 650 k-row shape.)
// * X tiles (and W, once per block, through the same LDS buffer) are read from global memory
//   row-contiguously -- a fragment-shaped global load touches 16 different rows per quarter-wave
//   and is bound by the per-CU address unit (measured 14-17 us vs 4 on the ZINC-128 shape) --
//   stored XOR-swizzled (chunk ^ (row & 15)) and read back with conflict-free ds_read_b128.
//   The next tile's loads are in flight during the current tile's MFMAs.
// * block = 4 waves, each wave 32 x 32 outputs (2 x 2 MFMA tiles, 16 acc VGPRs).  Wide layers
//   (N > 64) arrange the waves 1 x 4: a 32-row x 128-column tile.  Narrow layers (N <= 64: the
//   hidden-64 models of the MOLHIV / TU configurations) arrange them 2 x 2: 64 rows x 64 columns,
//   so no wave multiplies columns that do not exist; K <= 64 gets a 64-wide LDS image as well.
// * optional fused pieces: K-concatenation of two inputs (combine_nn's cat), per-input-column
//   affine + ReLU prologue (BatchNorm apply of the producing layer), bias, per-output-column
//   affine (BatchNorm in eval mode), ReLU, and per-column sum / sum-of-squares accumulation
//   (BatchNorm batch statistics in training mode).
#include <hip/hip_runtime.h>
#include <mutex>
#include <type_traits>
#include <stdlib.h>
#include "../../include/cwn_hip.h"
#include "cwn_mem.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int CT = 2;     // 16-col MFMA tiles per wave
constexpr int kCUs = 256;    // MI355X
constexpr int kMaxK = 256;  // K + K2 supported (the W tile of the whole K stays in LDS)

struct GemmBatch {
    cwn_gemm_desc d[CWN_MAX_DESCS];
    int32_t blk_start[CWN_MAX_DESCS + 1];   // first block of each descriptor
    int32_t n_tiles_n[CWN_MAX_DESCS];
    int32_t n_tiles[CWN_MAX_DESCS];         // tiles_m * tiles_n
    int32_t vec[CWN_MAX_DESCS];             // 16-B global accesses allowed (host-checked)
    int32_t n;
};

// LDS image of a tile: a row holds KP floats (KP = 128 or 256) = KP/4 chunks of 16 B; chunk c of
// row r is stored at chunk position c ^ (r & 15).  Fragment reads (16 lanes = 16 different rows,
// same chunk) then hit 16 different 16-B slots of the 256-B bank row: ds_read_b128 is
// conflict-free (MI355X_MICROARCH.md, LDS).
template <int KP>
__device__ __forceinline__ int lds_off(int row, int chunk) {   // in floats
    return row * KP + ((chunk ^ (row & 15)) << 2);
}

// A ROWS x KP tile in registers between its global loads and its LDS stores, so that every load
// is in flight before the first store (and, for X, during the previous tile's MFMAs).
template <int ROWS, int KP>
struct Staged {
    static constexpr int U = ROWS * (KP / 4) / kThreads;
    f32x4 v[U];
};

// Global reads are row-contiguous: KP/4 consecutive lanes read one full row of the tile.
template <bool FAST, int ROWS, int KP, int U0, int U1>
__device__ __forceinline__ void stage_load(Staged<ROWS, KP>& st, int64_t row0, int64_t row_max,
                                           const float* __restrict__ P1, int64_t ld1, int K1,
                                           const float* __restrict__ P2, int64_t ld2, int K2) {
    constexpr int CPR = KP / 4;   // chunks per row
#pragma unroll
    for (int u = U0; u < U1; ++u) {
        const int q = u * kThreads + threadIdx.x;
        const int r = q / CPR, c = q % CPR;
        const int64_t grow = row0 + r < row_max ? row0 + r : row_max - 1;   // clamped, never faults
        const int k = 4 * c;
        const bool second = K2 > 0 && k >= K1;   // never touch P2 when there is no second input
        const float* base = second ? P2 : P1;
        const int64_t ld = second ? ld2 : ld1;
        const int kk = second ? k - K1 : k;
        const int kmax = second ? K2 : K1;
        if constexpr (FAST) {
            // kmax % 4 == 0; the column is clamped into range, zeroing happens at store time
            st.v[u] = *reinterpret_cast<const f32x4*>(base + grow * ld + (kk < kmax ? kk : 0));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) st.v[u][t] = base[grow * ld + (kk + t < kmax ? kk + t : 0)];
        }
    }
}

struct Prologue {
    const float* scale;    // X:  [K] or NULL
    const float* shift;
    const float* scale2;   // X2: [K2] or NULL
    const float* shift2;
    int relu;              // bit 0: ReLU on X, bit 1: ReLU on X2
};

// A thread's part of the prologue: q = u * kThreads + tid walks ROWS only (kThreads is a multiple of KP / 4), so its four
// input columns -- and their scale / shift -- are the same for every row of every tile: loaded once per workgroup.  (Round 2
// read them from global memory in front of every LDS store: 3 of the 16.5 us of a ZINC-128 stage-2 launch,
// tools/ubench_gemm_train.py.)
// v of the lane CTRL's rotation away within its 16-lane row (DPP row_ror:n = 0x120 + n), for a double
template <int CTRL>
__device__ __forceinline__ double row_ror_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u & 0xffffffffull), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}

struct ProConst {
    f32x4 sc, sh;
    bool affine, relu;
    int kk, kmax;
};

template <int KP>
__device__ __forceinline__ ProConst make_pro(int K1, int K2, const Prologue& P) {
    constexpr int CPR = KP / 4;
    static_assert(kThreads % CPR == 0, "a thread keeps its columns");
    const int k = 4 * (threadIdx.x % CPR);
    const bool second = K2 > 0 && k >= K1;
    ProConst C;
    C.kk = second ? k - K1 : k;
    C.kmax = second ? K2 : K1;
    const float* sc = second ? P.scale2 : P.scale;
    const float* sh = second ? P.shift2 : P.shift;
    C.affine = sc != nullptr;
    C.relu = (P.relu & (second ? 2 : 1)) != 0;
    C.sc = (f32x4){1.f, 1.f, 1.f, 1.f};
    C.sh = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (C.affine) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (C.kk + t < C.kmax) {
                C.sc[t] = sc[C.kk + t];
                C.sh[t] = sh[C.kk + t];
            }
    }
    return C;
}

template <bool PRO, int ROWS, int KP, int U0, int U1>
__device__ __forceinline__ void stage_store(float* lds, Staged<ROWS, KP>& st, const ProConst& C) {
    constexpr int CPR = KP / 4;
#pragma unroll
    for (int u = U0; u < U1; ++u) {
        const int q = u * kThreads + threadIdx.x;
        const int r = q / CPR, c = q % CPR;
        f32x4 v = st.v[u];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float x = v[t];
            if constexpr (PRO) {
                if (C.affine) x = x * C.sc[t] + C.sh[t];
                if (C.relu) x = fmaxf(x, 0.f);
            }
            v[t] = C.kk + t < C.kmax ? x : 0.f;
        }
        *reinterpret_cast<f32x4*>(lds + lds_off<KP>(r, c)) = v;
    }
}

// ---- BatchNorm + ReLU backward as the prologue of the input-gradient GEMM (cwn_gemm_bnb) -------------------------------
// The tile staged for the MFMAs is dz = scale * (dyh - s1 / M - xhat * s2 / M), formed from the dy tile and the z tile
// on their way into LDS; it also goes to global memory for the weight-gradient GEMM.  Round 2 ran this as a launch of
// its own (cwn_norm_bwd_apply_f32: 6.5 us start to start, twelve times per ZINC training step) that wrote dz and the
// GEMM read it back.
struct BnbBatch {
    cwn_gemm_bnb x[CWN_MAX_DESCS];
};

struct NoBnb {
    cwn_gemm_bnb x[1];       // never read
};
template <bool BNB>
using BnbArg = typename std::conditional<BNB, BnbBatch, NoBnb>::type;

// Per input column:  dz = scale * dyh + c0 + c1 * (z - mean),  dyh = dy * [z * scale + shift > 0].  Five constants a column;
// they live in LDS (written once per workgroup, read back 16 B at a time when a tile is staged): as 20 registers per thread
// next to the stationary weight fragments and two staged tiles they pushed 26 registers into scratch (21 us per launch
// against 12 + 6.5 for GEMM + apply).
struct BnbFlags {
    bool on, relu;
};

template <int KP>
__device__ __forceinline__ BnbFlags make_bnb(const cwn_gemm_bnb& E, int K, int64_t M, bool first_block, float* cst) {
    BnbFlags F;
    F.on = E.z != nullptr;
    F.relu = E.relu != 0;
    if (F.on && threadIdx.x < KP) {
        const int k = threadIdx.x;
        float sc = 1.f, sh = 0.f, mu = 0.f, c0 = 0.f, c1 = 0.f;
        if (E.scale != nullptr && k < K) {
            const float invM = 1.0f / (float)M;
            const float rs = E.rstd[k], s1 = E.s1[k], s2 = E.s2[k];
            sc = E.scale[k];
            sh = E.shift[k];
            mu = E.mean[k];
            c0 = -sc * (s1 * invM);
            c1 = -sc * (rs * (s2 * invM));
            if (first_block) {                  // the sums go on to beta.grad / gamma.grad: one writer per column
                if (E.acc1 != nullptr) E.acc1[k] += s1;
                if (E.acc2 != nullptr) E.acc2[k] += s2;
            }
        }
        cst[k] = sc;
        cst[KP + k] = sh;
        cst[2 * KP + k] = mu;
        cst[3 * KP + k] = c0;
        cst[4 * KP + k] = c1;
    }
    return F;                                   // (the caller's first __syncthreads() publishes cst)
}

template <int ROWS, int KP, int U0, int U1>
__device__ __forceinline__ void stage_store_bnb(float* lds, const Staged<ROWS, KP>& sy, const Staged<ROWS, KP>& sz,
                                                const BnbFlags& F, const float* cst, int K, int64_t row0, int64_t M,
                                                float* dz, int64_t lddz) {
    constexpr int CPR = KP / 4;
    const int c = threadIdx.x % CPR;             // a thread's columns are the same in every row
    const f32x4 scale = *reinterpret_cast<const f32x4*>(cst + 4 * c), shift = *reinterpret_cast<const f32x4*>(cst + KP + 4 * c);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(cst + 2 * KP + 4 * c), c0 = *reinterpret_cast<const f32x4*>(cst + 3 * KP + 4 * c);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(cst + 4 * KP + 4 * c);
#pragma unroll
    for (int u = U0; u < U1; ++u) {
        const int q = u * kThreads + threadIdx.x;
        const int r = q / CPR;
        const f32x4 dy = sy.v[u], z = sz.v[u];
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float y = z[t] * scale[t] + shift[t];
            const float dyh = (!F.relu || y > 0.f) ? dy[t] : 0.f;
            const float d = scale[t] * dyh + c0[t] + c1[t] * (z[t] - mean[t]);
            v[t] = 4 * c + t < K ? d : 0.f;
        }
        *reinterpret_cast<f32x4*>(lds + lds_off<KP>(r, c)) = v;
        if (dz != nullptr && row0 + r < M && 4 * c < K) cwn::store_result4(dz + (row0 + r) * lddz + 4 * c, v[0], v[1], v[2], v[3]);
    }
}

template <bool FAST, bool PRO, int KP, int WN, bool WT, int RT, bool BNB>
#ifndef CWN_GEMM_LB
#define CWN_GEMM_LB 2
#endif
#ifndef CWN_GEMM_DEEP
#define CWN_GEMM_DEEP 1
#endif
#ifndef CWN_GEMM_FRAGPF
#define CWN_GEMM_FRAGPF 0
#endif
// (Round 4, VERDICT r3 item 8: the instantiations that do not fit 256 registers -- the 64 x 64 tile with a prologue or a transposed
// weight at K <= 128, the BatchNorm-backward prologue: 6 - 51 VGPRs in scratch -- compiled for ONE workgroup per SIMD pair
// (-DCWN_GEMM_NOSPILL=1: no spill left in this file) are SLOWER where they run: training steps with CWN_STAGE_KERNEL=0, molhiv-512
// 1.075 vs 1.061 ms, ZINC-128 0.952 vs 0.900 ms (profiles/r4_gemm_spills.txt).  Two co-resident workgroups that spill a few
// registers beat one that does not: the bound stays.)
#ifndef CWN_GEMM_NOSPILL
#define CWN_GEMM_NOSPILL 0
#endif
__global__ __launch_bounds__(kThreads, (KP <= 128 && RT == 2 && !(CWN_GEMM_NOSPILL && KP == 128 && ((WN == 2 && (PRO || WT)) || BNB))
                                        ? CWN_GEMM_LB : 1)) void gemm_kernel(GemmBatch B, BnbArg<BNB> E) {
    // RT = 16-row MFMA tiles per wave (2; 3 for the one-round small-M case, see the host side)
    constexpr int BM = 16 * RT * (4 / WN);   // rows per tile: WN waves side by side along N, 4/WN along M
    constexpr int BN = 32 * WN;         // columns per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];   // two [BM][KP] X buffers
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_gemm_desc& D = B.d[di];
    const int nblk = B.blk_start[di + 1] - B.blk_start[di];
    const int tiles_n = B.n_tiles_n[di];
    int tiles = B.n_tiles[di];
    if (D.m_dev != nullptr) {        // (uniform) a static batch: the row tiles below the rows that exist -- a PREFIX of the tile numbers
        const int64_t mv = *D.m_dev;
        const int64_t live = (mv < 0 ? 0 : (mv < D.M ? mv : D.M));
        const int64_t t = (live + BM - 1) / BM * tiles_n;
        tiles = t < tiles ? (int)t : tiles;
    }
    const bool vec = FAST;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % WN, wm = wave / WN;     // the wave's 32 x 32 patch inside the tile
    const int j = lane & 15, g = lane >> 4;
    // descriptor fields into registers once
    const float* const Xp = D.X;
    const float* const X2p = D.X2;
    const float* const Wp = D.W;
    const int64_t ldx = D.ldx, ldx2 = D.ldx2, ldw = D.ldw, ldy = D.ldy, M = D.M;
    const int N = D.N, K1 = D.K, K2 = D.K2, Ktot = D.K + D.K2;
    const ProConst pro = make_pro<KP>(K1, K2, Prologue{D.in_scale, D.in_shift, D.in_scale2, D.in_shift2, D.in_relu});
    const bool add_out = (D.flags & CWN_GEMM_ADD_OUT) != 0;
    BnbFlags bnb{};
    __shared__ __attribute__((aligned(16))) float bnb_cst[BNB ? 5 * KP : 4];
    const float* Zp = nullptr;
    float* dzp = nullptr;
    int64_t ldz = 0, lddz = 0;
    if constexpr (BNB) {
        const cwn_gemm_bnb& X = E.x[di];
        bnb = make_bnb<KP>(X, D.K, D.M, (int)blockIdx.x == B.blk_start[di], bnb_cst);
        Zp = X.z;
        dzp = X.dz;
        ldz = X.ldz;
        lddz = X.lddz;
    }
    const int dbg = D.flags >> 8;   // timing experiments (tools/ubench_gemm.py): 1 no MFMA, 2 no W staging, 4 no store
    constexpr int SLABS = KP / 16;
    using SX = Staged<BM, KP>;                 // a 32-row tile in flight: KP/32 x 16 B per thread

    // WEIGHTS STATIONARY: the wave's W fragments for the whole K (its 32 output columns) live in
    // registers for all tiles of the persistent loop (KP/2 VGPRs), so LDS only double-buffers the
    // 32-row X tile: one barrier per tile, 32 KiB (K <= 128) per block, three blocks per CU.
    f32x4 wreg[SLABS][CT];

    // Persistent over the descriptor's tiles, tile_n-major: a block keeps its tile_n (always, when
    // N <= 128) and reloads W only when it changes.
    int tile = blockIdx.x - B.blk_start[di];
    int cur_tn = -1;
    int it = 0;
    // two X tiles in flight in registers (prefetch distance 2 when K <= 128; 1 for the K = 256 variant,
    // whose tiles are twice as large)
    constexpr bool DEEP = KP <= 128 && CWN_GEMM_DEEP && !WT;   // (the WT variants have no registers to spare)
    SX sx, sx2;
    SX sz;                                   // BNB: the z tile that goes with sx
    static_assert(!BNB || (WT && !DEEP), "the backward prologue belongs to the transposed-weight variants");
    if (tile < tiles) {
        stage_load<FAST, BM, KP, 0, SX::U>(sx, (int64_t)(tile / tiles_n) * BM, M, Xp, ldx, K1, X2p, ldx2, K2);
        if constexpr (BNB)
            if (bnb.on) stage_load<FAST, BM, KP, 0, SX::U>(sz, (int64_t)(tile / tiles_n) * BM, M, Zp, ldz, K1, nullptr, 0, 0);
    }
    if (DEEP && tile + nblk < tiles)
        stage_load<FAST, BM, KP, 0, SX::U>(sx2, (int64_t)((tile + nblk) / tiles_n) * BM, M, Xp, ldx, K1, X2p, ldx2, K2);
    for (; tile < tiles; tile += nblk, ++it) {
        const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
        const int64_t m_base = (int64_t)tile_m * BM;
        const int n_base = tile_n * BN + wn * (CT * 16);

        if (cur_tn != tile_n && !(dbg & 2)) {
            // W fragments into registers, WAVE-PRIVATELY: the wave reads only its own 32 W rows
            // (row-contiguous 256-B segments, all loads in flight at once), pushes them through an
            // 8-KiB slice of LDS that no other wave touches and lifts its MFMA fragments back --
            // no workgroup barrier between the passes (LDS operations of one wave execute in order).
            constexpr int HW = 64;                  // floats of K per pass
            constexpr int PASSES = KP / HW;
            constexpr int LPP = 32 * (HW / 4) / 64; // 16-B loads per lane per pass (= 8)
            float* priv = smem + wave * (32 * HW);
            __syncthreads();                 // nobody still reads the X buffers we are about to reuse
            if constexpr (WT) {
                // W is [Ktot, N] (the layer's own weight, read for dX = dY . W): the wave's 32
                // output columns are 32 CONSECUTIVE floats of every W row.  Stage 32 k-rows per pass
                // in their natural [k][n] layout (16-B loads and stores, rows padded to 36 floats)
                // and lift the fragments with 4-B reads -- lanes j walk consecutive words, the lane
                // groups g land 16 banks apart.  The fragments stay in registers for the whole
                // block, so the narrow reads are paid once.
                constexpr int HK = 32, RS = 36;
                constexpr int WPASSES = KP / HK;
                const int wrow0 = tile_n * BN + wn * (CT * 16);
#pragma unroll
                for (int ps = 0; ps < WPASSES; ++ps) {
                    f32x4 wl[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kr = (lane >> 3) + 8 * i, c4 = 4 * (lane & 7);
                        const int k = ps * HK + kr;
                        const int64_t row = (int64_t)(k < Ktot ? k : 0) * ldw;
                        if (FAST && wrow0 + c4 + 3 < N) {
                            wl[i] = *reinterpret_cast<const f32x4*>(Wp + row + wrow0 + c4);
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; ++t) wl[i][t] = Wp[row + (wrow0 + c4 + t < N ? wrow0 + c4 + t : N - 1)];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kr = (lane >> 3) + 8 * i, c4 = 4 * (lane & 7);
                        f32x4 v = wl[i];
                        if (ps * HK + kr >= Ktot) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4*>(priv + kr * RS + c4) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int sl = 0; sl < HK / 16; ++sl)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                wreg[ps * (HK / 16) + sl][ct][t] = priv[(16 * sl + 4 * g + t) * RS + ct * 16 + j];
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                f32x4 wl[LPP];
                const int wrow0 = tile_n * BN + wn * (CT * 16);
#pragma unroll
                for (int i = 0; i < LPP; ++i) {
                    const int r = (lane >> 4) + 4 * i;
                    const int k = ps * HW + 4 * (lane & 15);
                    const int grow = wrow0 + r < N ? wrow0 + r : N - 1;
                    if constexpr (FAST) {
                        wl[i] = *reinterpret_cast<const f32x4*>(Wp + (int64_t)grow * ldw + (k < Ktot ? k : 0));
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) wl[i][t] = Wp[(int64_t)grow * ldw + (k + t < Ktot ? k + t : 0)];
                    }
                }
#pragma unroll
                for (int i = 0; i < LPP; ++i) {
                    const int r = (lane >> 4) + 4 * i, c = lane & 15;
                    const int k = ps * HW + 4 * c;
                    f32x4 v = wl[i];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = k + t < Ktot ? v[t] : 0.f;
                    *reinterpret_cast<f32x4*>(priv + r * HW + ((c ^ (r & 15)) << 2)) = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int sl = 0; sl < HW / 16; ++sl)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int r = ct * 16 + j, c = 4 * sl + g;
                        wreg[ps * (HW / 16) + sl][ct] =
                            *reinterpret_cast<const f32x4*>(priv + r * HW + ((c ^ (r & 15)) << 2));
                    }
                __builtin_amdgcn_wave_barrier();
            }
            }
            __syncthreads();                 // private slices are free again before any X store
            cur_tn = tile_n;
            it = 0;
        }
        float* ldsX = smem + (it & 1) * (BM * KP);
        if (BNB && bnb.on) stage_store_bnb<BM, KP, 0, SX::U>(ldsX, sx, sz, bnb, bnb_cst, K1, m_base, M, tile_n == 0 ? dzp : nullptr, lddz);
        else stage_store<PRO, BM, KP, 0, SX::U>(ldsX, sx, pro);
        __syncthreads();                     // tile visible; everyone is done with the other buffer
        if constexpr (DEEP) {
            sx = sx2;                        // tile + nblk is already on its way
            const int next2 = tile + 2 * nblk;
            if (next2 < tiles)               // two tiles ahead, in flight during two MFMA phases
                stage_load<FAST, BM, KP, 0, SX::U>(sx2, (int64_t)(next2 / tiles_n) * BM, M, Xp, ldx, K1, X2p, ldx2, K2);
        } else {
            const int next = tile + nblk;
            if (next < tiles)                // in flight during the MFMAs below
                stage_load<FAST, BM, KP, 0, SX::U>(sx, (int64_t)(next / tiles_n) * BM, M, Xp, ldx, K1, X2p, ldx2, K2);
        }

        f32x4 acc[CT][RT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[ct][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if CWN_GEMM_FRAGPF
        {   // explicit software pipeline of the X fragments: slab s+1 is read while slab s multiplies
            f32x4 xa[RT], xb[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                xa[rt] = *reinterpret_cast<const f32x4*>(ldsX + lds_off<KP>(wm * (16 * RT) + rt * 16 + j, g));
#pragma unroll
            for (int sl = 0; sl < SLABS; ++sl) {
                if (sl * 16 < Ktot && !(dbg & 1)) {
                    if (sl + 1 < SLABS) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            xb[rt] = *reinterpret_cast<const f32x4*>(ldsX + lds_off<KP>(wm * (16 * RT) + rt * 16 + j, 4 * (sl + 1) + g));
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
                                acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    wreg[sl][ct][t], xa[rt][t], acc[ct][rt], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) xa[rt] = xb[rt];
                }
            }
        }
#else
#pragma unroll
        for (int sl = 0; sl < SLABS; ++sl) {
            if (sl * 16 < Ktot && !(dbg & 1)) {   // wave-uniform; slabs past K hold zeros anyway
                f32x4 x[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    x[rt] = *reinterpret_cast<const f32x4*>(ldsX + lds_off<KP>(wm * (16 * RT) + rt * 16 + j, 4 * sl + g));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                wreg[sl][ct][t], x[rt][t], acc[ct][rt], 0, 0, 0);
            }
        }

#endif

        // epilogue: acc[ct][rt][r] = Y[m_base + rt*16 + j][n_base + ct*16 + 4g + r]
        // bias / affine / ReLU / BatchNorm statistics first, in the MFMA layout
        bool xok[RT];
        int64_t xrow[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            xrow[rt] = m_base + wm * (16 * RT) + rt * 16 + j;
            xok[rt] = xrow[rt] < M;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int n0 = n_base + ct * 16 + 4 * g;
            float bias[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n0 + r < N ? n0 + r : N - 1;       // clamped: unconditional loads
                if (D.bias != nullptr) bias[r] = D.bias[nn];
                if (D.out_scale != nullptr) {
                    sc[r] = D.out_scale[nn];
                    sh[r] = D.out_shift[nn];
                }
            }
            double csum[4] = {0., 0., 0., 0.}, csq[4] = {0., 0., 0., 0.};   // fp64: see cwn_bn_finalize_f32
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y = acc[ct][rt][r] + bias[r];
                    if (D.col_sum != nullptr && xok[rt]) {   // statistics of the pre-normalisation value
                        csum[r] += (double)y;
                        csq[r] += (double)y * (double)y;
                    }
                    y = y * sc[r] + sh[r];
                    acc[ct][rt][r] = D.relu ? fmaxf(y, 0.f) : y;
                }
            }
            if (D.col_sum != nullptr && n0 < N) {
                // reduce over the 16 rows held by lanes j = 0..15 of this lane group (both row
                // tiles are already summed in registers), then ONE plain store per column into
                // the slot of this wave's 32-row band: no atomics, no zero fill, deterministic.
                // (fp64 atomics on 2 x N addresses from ~100 workgroups per descriptor serialised
                // at L2 and cost 20 us of a 30 us launch.)
                const int64_t slot = m_base / 32 + wm;   // RT == 2 only (host-checked): tiles align to bands
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (DPP rotations within the 16-lane row, two 32-bit halves per double: __shfl_xor on a double is two
                    // ds_bpermute round trips per step, 3 us of a 13.5-us stage launch, tools/ubench_gemm_train.py)
                    double a = csum[r], b = csq[r];
                    a += row_ror_f64<0x128>(a); b += row_ror_f64<0x128>(b);
                    a += row_ror_f64<0x124>(a); b += row_ror_f64<0x124>(b);
                    a += row_ror_f64<0x122>(a); b += row_ror_f64<0x122>(b);
                    a += row_ror_f64<0x121>(a); b += row_ror_f64<0x121>(b);
                    if (j == 0 && n0 + r < N && slot < CWN_STAT_ROWS(M)) {
                        D.col_sum[slot * N + n0 + r] = a;
                        D.col_sumsq[slot * N + n0 + r] = b;
                    }
                }
            }
        }
        if (dbg & 4) continue;
        // Stores.  In the MFMA layout one store instruction would write 16 rows x 64 B (half
        // lines).  A DPP row_ror:8 exchange between lanes j and j^8 of each 16-lane row (same g)
        // pairs the two column tiles of a row instead: lanes j < 8 keep (row j, ct 0) and receive
        // (row j+8, ct 0); lanes j >= 8 receive (row j-8, ct 1) and keep (row j, ct 1).  Each
        // store instruction then writes 8 rows x 128 B = 8 FULL cache lines.
        static_assert(CT == 2, "the line-pairing epilogue assumes two column tiles per wave");
        const bool lo = j < 8;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 send, recv;
#pragma unroll
            for (int r = 0; r < 4; ++r) send[r] = lo ? acc[1][rt][r] : acc[0][rt][r];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                recv[r] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send[r]), 0x128, 0xf, 0xf, false));
            // instruction A: rows 0..7 of the 16-row tile; instruction B: rows 8..15
            f32x4 va, vb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                va[r] = lo ? acc[0][rt][r] : recv[r];      // lo: (row j, ct0)    hi: (row j-8, ct1)
                vb[r] = lo ? recv[r] : acc[1][rt][r];      // lo: (row j+8, ct0)  hi: (row j, ct1)
            }
            const int64_t ra = m_base + wm * (16 * RT) + rt * 16 + (lo ? j : j - 8);
            const int64_t rb = m_base + wm * (16 * RT) + rt * 16 + (lo ? j + 8 : j);
            const int n0 = n_base + (lo ? 0 : 16) + 4 * g;
            const bool full = n0 + 3 < N;
            if (ra < M && n0 < N) {
                float* yp = D.Y + ra * ldy + n0;
                if (add_out) {      // Y += : this launch is the only writer of Y (host contract), a plain read-modify-write
if (full && vec) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(yp);
                        cwn::store_result4(yp, o[0] + va[0], o[1] + va[1], o[2] + va[2], o[3] + va[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n0 + r < N) yp[r] += va[r];
                    }
                } else if (full && vec) cwn::store_result4(yp, va[0], va[1], va[2], va[3]);
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + r < N) yp[r] = va[r];
            }
            if (rb < M && n0 < N) {
                float* yp = D.Y + rb * ldy + n0;
                if (add_out) {
if (full && vec) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(yp);
                        cwn::store_result4(yp, o[0] + vb[0], o[1] + vb[1], o[2] + vb[2], o[3] + vb[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n0 + r < N) yp[r] += vb[r];
                    }
                } else if (full && vec) cwn::store_result4(yp, vb[0], vb[1], vb[2], vb[3]);
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + r < N) yp[r] = vb[r];
            }
        }
        if constexpr (BNB) {
            // the next tile's z only now: across the MFMAs it would be 16 more live registers (-> scratch); its latency
            // hides behind the stores above
            const int next = tile + nblk;
            if (bnb.on && next < tiles)
                stage_load<FAST, BM, KP, 0, SX::U>(sz, (int64_t)(next / tiles_n) * BM, M, Zp, ldz, K1, nullptr, 0, 0);
        }
    }
}

inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; }

}  // namespace

// cwn_gemm_split.hip: the same product on the bf16 matrix pipe (three-way exact split, fp32 accuracy)
int cwn_gemm_split_eligible(const cwn_gemm_desc* descs, int n);
int cwn_gemm_split_launch(const cwn_gemm_desc* descs, int n, hipStream_t stream);

extern "C" int cwn_gemm_would_split(const cwn_gemm_desc* descs, int n) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_DESCS) return 0;
    for (int i = 0; i < n; ++i)
        if (descs[i].flags & (CWN_GEMM_EXACT | CWN_GEMM_ADD_OUT)) return 0;
    return cwn_gemm_split_eligible(descs, n) ? 1 : 0;
}

extern "C" int cwn_gemm_f32(const cwn_gemm_desc* descs, int n, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_DESCS) return CWN_ERR_BAD_ARG;
    GemmBatch B{};
    BnbBatch E{};
    bool any_bnb = false;
    B.n = n;
    int64_t total_tiles = 0;
    for (int i = 0; i < n; ++i) {
        const cwn_gemm_desc& D = descs[i];
        if (D.M < 0 || D.N <= 0 || D.K <= 0 || D.K2 < 0) return CWN_ERR_BAD_ARG;
        if (D.K + D.K2 > kMaxK) return CWN_ERR_TOO_LARGE;   // the whole-K W tile must fit in LDS
        if (D.M > 0 && (D.X == nullptr || D.W == nullptr || D.Y == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.K2 > 0 && (D.X2 == nullptr || (D.K % 4) != 0)) return CWN_ERR_BAD_ARG;
        if ((D.in_scale == nullptr) != (D.in_shift == nullptr)) return CWN_ERR_BAD_ARG;
        if ((D.in_scale2 == nullptr) != (D.in_shift2 == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.in_scale2 != nullptr && D.K2 == 0) return CWN_ERR_BAD_ARG;
        if ((D.out_scale == nullptr) != (D.out_shift == nullptr)) return CWN_ERR_BAD_ARG;
        if ((D.col_sum == nullptr) != (D.col_sumsq == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.m_dev != nullptr && (D.col_sum != nullptr || D.bnb != nullptr)) return CWN_ERR_BAD_ARG;   // (include/cwn_hip.h: m_dev)
        if (D.ldx < D.K || (D.ldw < (D.w_trans ? D.N : D.K + D.K2) && !(D.flags & CWN_GEMM_W_PACKED)) || D.ldy < D.N ||
            (D.K2 > 0 && D.ldx2 < D.K2))
            return CWN_ERR_BAD_ARG;
        const void* ptrs[] = {D.X, D.X2, D.W, D.Y};
        for (const void* p : ptrs)
            if (p != nullptr && ((uintptr_t)p & 3u)) return CWN_ERR_ALIGN;
        B.vec[i] = (al16(D.X) && al16(D.X2) && al16(D.W) && al16(D.Y) && D.ldx % 4 == 0 &&
                    D.ldw % 4 == 0 && D.ldy % 4 == 0 && (D.K2 == 0 || D.ldx2 % 4 == 0) &&
                    D.K % 4 == 0 && D.K2 % 4 == 0) ? 1 : 0;
        B.d[i] = D;
        B.d[i].bnb = nullptr;              // (a host pointer: the extension travels in its own argument)
        if (D.bnb != nullptr) {
            const cwn_gemm_bnb& X = *D.bnb;
            if (!D.w_trans || D.K2 != 0 || D.K > 128 || X.z == nullptr || X.ldz < D.K || (X.dz != nullptr && X.lddz < D.K))
                return CWN_ERR_BAD_ARG;
            if ((X.scale == nullptr) != (X.shift == nullptr)) return CWN_ERR_BAD_ARG;
            if (X.scale != nullptr && (X.mean == nullptr || X.rstd == nullptr || X.s1 == nullptr || X.s2 == nullptr))
                return CWN_ERR_BAD_ARG;
            if (!al16(X.z) || !al16(X.dz) || X.ldz % 4 != 0 || X.lddz % 4 != 0) return CWN_ERR_ALIGN;
            const void* ps[] = {X.scale, X.shift, X.mean, X.rstd, X.s1, X.s2, X.acc1, X.acc2};
            for (const void* q : ps)
                if (q != nullptr && ((uintptr_t)q & 3u)) return CWN_ERR_ALIGN;
            E.x[i] = X;
            any_bnb = true;
        }
    }
    // precision policy is PER CALL (no process-wide state): CWN_GEMM_EXACT on any descriptor keeps the
    // launch on the exact fp32-MFMA kernel below
    if (cwn_gemm_would_split(descs, n)) return cwn_gemm_split_launch(descs, n, (hipStream_t)stream_);
    for (int i = 0; i < n; ++i)              // a packed weight exists in the split kernel's form only
        if (descs[i].flags & CWN_GEMM_W_PACKED) return CWN_ERR_BAD_ARG;
    // tile shape of the launch: narrow (64 x 64) when no descriptor has more than 64 output columns
    int kmax = 0, nmax = 0;
    for (int i = 0; i < n; ++i) {
        kmax = descs[i].K + descs[i].K2 > kmax ? descs[i].K + descs[i].K2 : kmax;
        nmax = descs[i].N > nmax ? descs[i].N : nmax;
    }
    const bool narrow = nmax <= 64;
    const int KP = kmax <= 64 && narrow ? 64 : (kmax <= 128 ? 128 : 256);
    int BM = narrow ? 64 : 32;
    const int BN = narrow ? 64 : 128;
    // 48-row tiles for the one case where they pay: a small launch (the ZINC batch of 128 is 318
    // tiles of 32 rows on 256 CUs) whose 32-row tiling needs two rounds of workgroups on some CUs
    // while a 48-row tiling fits in one -- the MFMA phase of the critical CU drops from 2 x 2 to
    // 1 x 3 row tiles (3.9 -> ~2.6 us measured on that shape).  Not with the statistics epilogue,
    // whose partial sums are laid out in 32-row bands.
    bool rt3 = false;
    if (!narrow && KP == 128) {
        int64_t t32 = 0, t48 = 0;
        bool stats = false;
        for (int i = 0; i < n; ++i) {
            const int tn_ = (descs[i].N + BN - 1) / BN;
            t32 += ((descs[i].M + 31) / 32) * tn_;
            t48 += ((descs[i].M + 47) / 48) * tn_;
            stats = stats || descs[i].col_sum != nullptr;
        }
        rt3 = !stats && !any_bnb && t32 > kCUs && t48 <= kCUs;
        if (rt3) BM = 48;
    }
    for (int i = 0; i < n; ++i) {
        const cwn_gemm_desc& D = descs[i];
        const int64_t tm = (D.M + BM - 1) / BM;
        const int tn = (D.N + BN - 1) / BN;
        if (tm * tn >= INT32_MAX) return CWN_ERR_TOO_LARGE;
        B.n_tiles_n[i] = tn;
        B.n_tiles[i] = (int32_t)(tm * tn);
        total_tiles += tm * tn;
    }
    if (total_tiles == 0) return CWN_OK;
    // two X buffers; never less than the four 8-KiB wave-private W staging slices
    int lds_bytes = 2 * BM * KP * 4;
    if (lds_bytes < 4 * 32 * 64 * 4) lds_bytes = 4 * 32 * 64 * 4;
    // persistent blocks, shared between the descriptors in proportion to their tile counts; each
    // block walks its descriptor's tiles with its weights stationary
    const int64_t budget = lds_bytes <= 48 * 1024 ? 4096 : 512;
    // (K <= 128: measured on the 650 k-row shape -- 512 blocks 363 us, 768: 331, 1024: 318, 2048: 312,
    //  4096: 302, 8192: 324, one block per tile: 371; two are resident per CU, the queue behind them
    //  keeps every CU busy to the end while W is still re-staged only 16 times per CU)
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
    // w_trans is a property of the launch (all descriptors or none), and excludes the prologue
    const bool wt = descs[0].w_trans != 0;
    bool fast = true, pro = false;
    for (int i = 0; i < n; ++i) {
        if ((descs[i].w_trans != 0) != wt) return CWN_ERR_BAD_ARG;
        fast = fast && B.vec[i] != 0;
        pro = pro || B.d[i].in_scale != nullptr || B.d[i].in_scale2 != nullptr || B.d[i].in_relu != 0;
    }
    using Kern = void (*)(GemmBatch, NoBnb);
    // shape index: 0 = 32x128 tile, K <= 128;  1 = 32x128, K <= 256;  2 = 64x64, K <= 64;
    //              3 = 64x64, K <= 128;        4 = 64x64, K <= 256;   5 = 48x128, K <= 128
#define CWN_SHAPES(F, P, T)                                                                           \
    {gemm_kernel<F, P, 128, 4, T, 2, false>, gemm_kernel<F, P, 256, 4, T, 2, false>, gemm_kernel<F, P, 64, 2, T, 2, false>,   \
     gemm_kernel<F, P, 128, 2, T, 2, false>, gemm_kernel<F, P, 256, 2, T, 2, false>, gemm_kernel<F, P, 128, 4, T, 3, false>}
    static const Kern kerns[2][2][6] = {{CWN_SHAPES(false, false, false), CWN_SHAPES(false, true, false)},
                                        {CWN_SHAPES(true, false, false), CWN_SHAPES(true, true, false)}};
    // transposed-weight variants (the input-gradient GEMM): no prologue
    static const Kern kerns_wt[2][6] = {CWN_SHAPES(false, false, true), CWN_SHAPES(true, false, true)};
#undef CWN_SHAPES
    static const int kShapeLds[6] = {2 * 32 * 128 * 4, 2 * 32 * 256 * 4, 2 * 64 * 64 * 4, 2 * 64 * 128 * 4,
                                     2 * 64 * 256 * 4, 2 * 48 * 128 * 4};
    static std::once_flag attr_once;        // thread-safe, once per process
    static bool attr_ok = true;
    std::call_once(attr_once, [&] {
        for (int c = 0; c < 6; ++c)
            for (int a = 0; a < 2; ++a) {
                for (int b = 0; b < 2; ++b)
                    attr_ok = attr_ok && hipFuncSetAttribute((const void*)kerns[a][b][c],
                                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             kShapeLds[c]) == hipSuccess;
                attr_ok = attr_ok && hipFuncSetAttribute((const void*)kerns_wt[a][c],
                                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         kShapeLds[c]) == hipSuccess;
            }
    });
    if (!attr_ok) return CWN_ERR_LAUNCH;
    if (wt && pro) return CWN_ERR_BAD_ARG;
    const int shape = narrow ? (KP == 64 ? 2 : (KP == 128 ? 3 : 4)) : (rt3 ? 5 : (KP == 128 ? 0 : 1));
    if (any_bnb) {
        // the backward prologue: transposed weight, 16-byte operands, the two shapes of the hidden-128 / hidden-64 models
        if (!wt || !fast || (shape != 0 && shape != 2 && shape != 3)) return CWN_ERR_BAD_ARG;
        static std::once_flag bnb_once;
        static bool bnb_ok = true;
        std::call_once(bnb_once, [&] {
            bnb_ok = hipFuncSetAttribute((const void*)gemm_kernel<true, false, 128, 4, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         kShapeLds[0]) == hipSuccess &&
                     hipFuncSetAttribute((const void*)gemm_kernel<true, false, 64, 2, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         kShapeLds[2]) == hipSuccess &&
                     hipFuncSetAttribute((const void*)gemm_kernel<true, false, 128, 2, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         kShapeLds[3]) == hipSuccess;
        });
        if (!bnb_ok) return CWN_ERR_LAUNCH;
        if (shape == 0) hipLaunchKernelGGL((gemm_kernel<true, false, 128, 4, true, 2, true>), dim3((unsigned)blocks), dim3(kThreads), lds_bytes, (hipStream_t)stream_, B, E);
        else if (shape == 2) hipLaunchKernelGGL((gemm_kernel<true, false, 64, 2, true, 2, true>), dim3((unsigned)blocks), dim3(kThreads), lds_bytes, (hipStream_t)stream_, B, E);
        else hipLaunchKernelGGL((gemm_kernel<true, false, 128, 2, true, 2, true>), dim3((unsigned)blocks), dim3(kThreads), lds_bytes, (hipStream_t)stream_, B, E);
        return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
    }
    const Kern k = wt ? kerns_wt[fast ? 1 : 0][shape] : kerns[fast ? 1 : 0][pro ? 1 : 0][shape];
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), lds_bytes, (hipStream_t)stream_, B, NoBnb{});
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
