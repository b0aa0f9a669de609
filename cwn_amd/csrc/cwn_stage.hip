// cwn_stage.hip -- ONE stage of the update / combine networks of a SparseCIN layer in TRAINING mode, all dimensions and
// both branches in one launch:
//
//     Z = prologue([X | X2]) W^T + b,     per-column sum / sum of squares of Z per 32-row band (fp64)
//
// (mp/layers.py:193-199, 303-325: Linear -> BatchNorm(train) -> ReLU; the prologue is the BatchNorm apply + ReLU of the
// stage before, cwn_bn_finalize_f32 turns the band sums into the next prologue's affine.)  Training cannot chain the
// stages inside a workgroup the way cwn_update_mlp_f32 does -- batch statistics are a reduction over ALL rows between any
// two of them -- so a stage stays a launch; this is the launch with the inference kernel's arithmetic: 32 (F = 128) or
// 64 (F = 64) rows per workgroup, the tile split ONCE per element into bf16 planes in LDS (cwn_split.h: exact three-way
// split, six MFMAs per product term, fp32 accuracy), the weight pre-split and packed in fragment order once per optimizer
// step (cwn_update_mlp_pack_weights_many_f32) and streamed 1 KiB per load instruction into registers, eight waves = eight
// column tiles.  The grouped fp32-MFMA launch it replaces (cwn_gemm_f32: weights stationary per workgroup, staged through
// LDS; 157 TF peak) took 13 - 14.4 us per stage at the ZINC batch of 128 -- with 1.7 tiles per workgroup it never reaches
// a steady state: weight staging, tile load, 128 dependent MFMAs and the epilogue run back to back.
// (Measured and dropped: 64 rows per workgroup at F = 128 -- four row tiles a wave, half the weight bytes per row, at most
// one workgroup per CU at the ZINC batch (214 instead of 428): the training step read 0.829 ms against 0.822, MOLHIV-512
// 0.965 against 0.931.  Two small workgroups on a CU overlap each other's load and multiply phases; one large one does not.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include "../../include/cwn_hip.h"
#include "cwn_split.h"
#include "cwn_mem.h"
#include "cwn_bn_live.h"

namespace {

using cwn::frag_cd;

constexpr int kThreads = 512;
constexpr int kV = 2;                             // float4 of an input tile per thread
constexpr int kRT = 2;                            // 16-row tiles per wave

template <int F> struct Shape {
    static constexpr int kTM = 4096 / F;
    static constexpr int kNCT = F / 16;
    static constexpr int kKS = F / 32;
    static constexpr int kRowStride = F + 8;          // bf16 elements per LDS row (fragment reads conflict-free)
    static constexpr int kChunksPerTile = kKS * 3;    // packed weight: 1-KiB chunks per 16-column tile (k steps x planes)
    static constexpr size_t kPlaneElems = (size_t)kTM * kRowStride;
    static constexpr size_t kBufBytes = 3 * kPlaneElems * 2;   // three planes
    static constexpr size_t kLdsBytes = 2 * kBufBytes;         // the tile of X and of X2
    static_assert(kTM * (F / 4) == kV * kThreads && (kTM / 16) * kNCT == 8 * kRT, "tile shape");
    static_assert(kThreads % (F / 4) == 0, "a thread keeps its input columns");
};

struct StageBatch {
    cwn_stage_desc d[CWN_MAX_DESCS];
    int32_t blk_start[CWN_MAX_DESCS + 1];
    int32_t n;
    int32_t dbg;      // timing experiments (CWN_STAGE_DBG): 1 live prologue without its arithmetic, 2 without its loads, 4 no statistics atomics
};
// ... with a third / fourth K-block per product (cwn_dense_stage_ex_f32: CIN++'s 3F / 4F-wide combine): its own kernel, so
// that the two-input launches of a SparseCIN step keep their argument block and their code
struct StageBatchEx : StageBatch {
    cwn_stage_extra more[CWN_MAX_DESCS][2];
};
static_assert(sizeof(StageBatchEx) <= 4096 - 256, "kernel-argument segment (with the hidden arguments)");

// v of the lane CTRL's rotation away within its 16-lane row (DPP row_ror:n = 0x120 + n), for a double
template <int CTRL>
__device__ __forceinline__ double row_ror_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u & 0xffffffffull), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}

template <int CTRL>
__device__ __forceinline__ float row_ror_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <int F, bool EX, typename BatchT>
__device__ __forceinline__ void dense_stage_body(const BatchT& B) {
    using S = Shape<F>;
    constexpr int TM = S::kTM, kRowStride = S::kRowStride, kChunksPerTile = S::kChunksPerTile, kKS = S::kKS;
    constexpr size_t kPlaneElems = S::kPlaneElems, kBufBytes = S::kBufBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // 2 KiB next to the tiles: the affines a live BatchNorm prologue derives ([input][scale | shift][F] floats) and, behind the
    // products, the workgroup's column statistics on their way to one coalesced atomic per column ([half][sum | sq][F] doubles)
    // (EX: up to four inputs -> 4 KiB)
    __shared__ __attribute__((aligned(16))) double live_scratch[EX ? 512 : 256];
    uint16_t* const buf0 = reinterpret_cast<uint16_t*>(smem);
    uint16_t* const buf1 = reinterpret_cast<uint16_t*>(smem + kBufBytes);
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_stage_desc& D = B.d[di];
    const int64_t row0 = (int64_t)((int)blockIdx.x - B.blk_start[di]) * TM;
    // rows that exist (include/cwn_hip.h, "device-side row counts"): D.M is then the capacity of the buffers -- it bounds the
    // addresses below, so that no load waits for this one -- and Mv what is stored and counted
    const int64_t Mv = D.m_dev != nullptr ? *D.m_dev : D.M;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ct = wave % S::kNCT, rt0 = (wave / S::kNCT) * kRT;
    const int l15 = lane & 15, kq = lane >> 4;
    const bool two = D.X2 != nullptr;
    // (EX) the third / fourth K-block: tiles requested with the others, staged into the SAME two LDS buffers once the first
    // two products have left them
    const cwn_stage_extra* E = nullptr;
    bool three = false, four = false;
    if constexpr (EX) {
        E = B.more[di];
        three = E[0].X != nullptr;
        four = three && E[1].X != nullptr;
    }

    // workgroup barrier that orders LDS traffic only (__syncthreads() also waits for every outstanding global load: here
    // the tiles and the weight, which keep streaming across the barrier)
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // an input tile: TM rows x F / 4 float4, two per thread, row-contiguous; rows past M are clamped, not guarded (their
    // outputs are neither stored nor counted)
    typedef float4 RowRegs[kV];
    RowRegs v0, v1;
    auto request_rows = [&](RowRegs& v, const float* X, int64_t ld) {
#pragma unroll
        for (int i = 0; i < kV; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx / (F / 4), c4 = idx % (F / 4);
            const int64_t row = row0 + r < D.M ? row0 + r : D.M - 1;
            v[i] = reinterpret_cast<const float4*>(X + row * ld)[c4];
        }
    };
    // the prologue of an input: per-column affine (the producing stage's BatchNorm) and ReLU.  A thread's columns are the
    // same for every row it stages (kThreads is a multiple of F / 4): its constants are loaded once.
    struct Pro { float4 sc, sh; bool affine, relu; };
    auto request_pro = [&](const float* scale, const float* shift, bool relu, int live_slot) {
        Pro p;
        p.sc = make_float4(1.f, 1.f, 1.f, 1.f);
        p.sh = make_float4(0.f, 0.f, 0.f, 0.f);
        p.affine = scale != nullptr || live_slot >= 0;
        p.relu = relu;
        const int c4 = threadIdx.x % (F / 4);
        if (live_slot >= 0) {              // derived in this workgroup (below): [input][scale | shift][F] floats in LDS
            const float4* aff = reinterpret_cast<const float4*>(live_scratch) + live_slot * 2 * (F / 4);
            p.sc = aff[c4];
            p.sh = aff[F / 4 + c4];
        } else if (p.affine) {
            p.sc = reinterpret_cast<const float4*>(scale)[c4];
            p.sh = reinterpret_cast<const float4*>(shift)[c4];
        }
        return p;
    };
    auto stage_rows = [&](const RowRegs& v, const Pro& p, uint16_t* buf) {   // prologue, then split ONCE per element
#pragma unroll
        for (int i = 0; i < kV; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx / (F / 4), c4 = idx % (F / 4);
            float4 x = v[i];
            if (p.affine) x = make_float4(x.x * p.sc.x + p.sh.x, x.y * p.sc.y + p.sh.y, x.z * p.sc.z + p.sh.z, x.w * p.sc.w + p.sh.w);
            if (p.relu) x = make_float4(fmaxf(x.x, 0.f), fmaxf(x.y, 0.f), fmaxf(x.z, 0.f), fmaxf(x.w, 0.f));
            uint2 ph, pm, pl;
            cwn::split4(x, ph, pm, pl);
            uint16_t* dst = buf + (size_t)r * kRowStride + c4 * 4;
            *reinterpret_cast<uint2*>(dst) = ph;
            *reinterpret_cast<uint2*>(dst + kPlaneElems) = pm;
            *reinterpret_cast<uint2*>(dst + 2 * kPlaneElems) = pl;
        }
    };
    // the stationary operand: this wave's 16 output columns of one F x F block, [k step][plane]; two sets
    typedef uint4 WeightRegs[kKS][3];
    WeightRegs wfA, wfB;
    auto request_kstep = [&](WeightRegs& wf, const void* packed, int ks) {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(packed) + (size_t)ct * kChunksPerTile * 1024 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[ks][pl] = *reinterpret_cast<const uint4*>(wp + (ks * 3 + pl) * 1024);
    };
    typedef frag_cd AccRegs[kRT];
    AccRegs acc;
    // acc += buf x W^T (k steps in order, six terms each: cwn_split.h).  `next` != NULL: the k steps of that block are
    // requested into the OTHER register set one by one between this block's MFMAs.
    auto multiply = [&](const uint16_t* buf, const WeightRegs& wf, WeightRegs& wnext, const void* next) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
#pragma unroll
            for (int rt = 0; rt < kRT; ++rt) {
                const uint16_t* p = buf + (size_t)((rt0 + rt) * 16 + l15) * kRowStride + ks * 32 + kq * 8;
                const uint4 xh = *reinterpret_cast<const uint4*>(p);
                const uint4 xm = *reinterpret_cast<const uint4*>(p + kPlaneElems);
                const uint4 xl = *reinterpret_cast<const uint4*>(p + 2 * kPlaneElems);
                acc[rt] = cwn::mfma_split6(wf[ks][0], wf[ks][1], wf[ks][2], xh, xm, xl, acc[rt]);
            }
            if (next != nullptr) request_kstep(wnext, next, ks);
        }
    };

    // requests: rows and the constants behind them first, then the weight (loads return in order: a constant behind 96 KB of
    // weight is a wait for the weight)
    // live BatchNorm prologue (cwn_bn_live.h): thread c < F the column c of X's record, thread F + c of X2's; their slot sums
    // are requested AHEAD of the tiles (loads return in order: behind the tiles they would wait for them)
    const bool live0 = D.in_bn.slots != nullptr, live1 = D.in_bn2.slots != nullptr;
    bool live2 = false, live3 = false;
    if constexpr (EX) {
        live2 = three && E[0].bn.slots != nullptr;
        live3 = four && E[1].bn.slots != nullptr;
    }
    const int live_which = threadIdx.x / F, live_col = threadIdx.x % F;
    bool live_mine = (live_which == 0 && live0) || (live_which == 1 && live1);
    if constexpr (EX) live_mine = live_mine || (live_which == 2 && live2) || (live_which == 3 && live3);
    auto live_rec = [&]() -> const cwn_bn_live& {
        if constexpr (EX) {
            if (live_which >= 2) return E[live_which == 2 ? 0 : 1].bn;
        }
        return live_which == 0 ? D.in_bn : D.in_bn2;
    };
    cwn::BnLiveRegs live_regs;
    if (live_mine && !(B.dbg & 2))
        cwn::bn_live_request(live_rec(), F, live_col, (int)blockIdx.x == B.blk_start[di], live_regs);
    __builtin_amdgcn_sched_barrier(0);              // (the slot requests leave FIRST: the compiler would hoist the tile loads)
    request_rows(v0, D.X, D.ldx);
    if (row0 >= Mv) return;                         // (uniform) a tile past the batch's own rows: nothing to store or count
    if (two) request_rows(v1, D.X2, D.ldx2);
    RowRegs v2, v3;
    if constexpr (EX) {
        if (three) request_rows(v2, E[0].X, E[0].ldx);
        if (four) request_rows(v3, E[1].X, E[1].ldx);
    }
    if (live0 || live1 || live2 || live3) {
        // ... the first workgroup of the descriptor writes what the backward reads and the running statistics
        float* const aff = reinterpret_cast<float*>(live_scratch);
        if (live_mine) {
            float sc = 1.f, sh = 0.f;
            if (!(B.dbg & 3)) {
                cwn::bn_live_finish(live_rec(), live_regs, F, Mv, live_col,
                                    (int)blockIdx.x == B.blk_start[di] && !(B.dbg & 8), sc, sh);
            } else if (!(B.dbg & 2)) {               // every load consumed, none of the arithmetic
                double a = 0.0;
#pragma unroll
                for (int q = 0; q < CWN_BN_SLOTS; ++q) a += live_regs.t[q] + live_regs.u[q];
                sc = (float)a + live_regs.g + live_regs.beta;
            }
            aff[(live_which * 2) * F + live_col] = sc;
            aff[(live_which * 2 + 1) * F + live_col] = sh;
        }
        lds_barrier();                              // (LDS traffic only: the tiles stay in flight)
    }
    const Pro p0 = request_pro(D.in_scale, D.in_shift, (D.in_relu & 1) != 0, live0 ? 0 : -1);
    Pro p1 = p0;
    if (two) p1 = request_pro(D.in_scale2, D.in_shift2, (D.in_relu & 2) != 0, live1 ? 1 : -1);
    const int n0 = ct * 16 + kq * 4;                 // D[i][j]: i = output column (lane >> 4) * 4 + reg, j = row (lane & 15)
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (D.bias != nullptr) b4 = *reinterpret_cast<const float4*>(D.bias + n0);
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) request_kstep(wfA, D.w_packed, ks);
    stage_rows(v0, p0, buf0);
    if (two) stage_rows(v1, p1, buf1);
    lds_barrier();
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt) acc[rt] = frag_cd{0.f, 0.f, 0.f, 0.f};
    multiply(buf0, wfA, wfB, two ? D.w2_packed : nullptr);
    if constexpr (!EX) {
        if (two) multiply(buf1, wfB, wfA, nullptr);
    } else {
        if (two) multiply(buf1, wfB, wfA, three ? E[0].w_packed : nullptr);
        if (three) {
            const Pro p2 = request_pro(nullptr, nullptr, E[0].relu != 0, live2 ? 2 : -1);
            lds_barrier();                          // every wave has read the first two tiles
            stage_rows(v2, p2, buf0);
            if (four) {
                const Pro p3 = request_pro(nullptr, nullptr, E[1].relu != 0, live3 ? 3 : -1);
                stage_rows(v3, p3, buf1);
            }
            lds_barrier();
            multiply(buf0, wfA, wfB, four ? E[1].w_packed : nullptr);
            if (four) multiply(buf1, wfB, wfA, nullptr);
        }
    }

    // epilogue: + bias; the band's column statistics of the pre-normalisation value (fp64, the 16 rows of a lane group
    // through DPP, both row tiles summed in registers first, ONE plain store per column and band: no atomics, no zero
    // fill, deterministic -- as cwn_gemm_f32's); the rows of Z
    const bool slotted = D.stat_slots != nullptr;
    const bool stats = D.col_sum != nullptr || slotted;
    double csum[4] = {0., 0., 0., 0.}, csq[4] = {0., 0., 0., 0.};
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt) {
        const int r = (rt0 + rt) * 16 + l15;
        const bool ok = row0 + r < Mv;
        const float y[4] = {acc[rt][0] + b4.x, acc[rt][1] + b4.y, acc[rt][2] + b4.z, acc[rt][3] + b4.w};
        if (stats && ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                csum[q] += (double)y[q];
                csq[q] += (double)y[q] * (double)y[q];
            }
        }
        if (ok) cwn::store_result4(D.Y + (row0 + r) * D.ldy + n0, y[0], y[1], y[2], y[3]);
    }
    if (stats) {
        const int64_t slot = (row0 + rt0 * 16) / 32;          // this wave's two row tiles are one 32-row band
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double a = csum[q], b = csq[q];
            a += row_ror_f64<0x128>(a); b += row_ror_f64<0x128>(b);
            a += row_ror_f64<0x124>(a); b += row_ror_f64<0x124>(b);
            a += row_ror_f64<0x122>(a); b += row_ror_f64<0x122>(b);
            a += row_ror_f64<0x121>(a); b += row_ror_f64<0x121>(b);
            if (slotted) {
                if (l15 == 0) {                       // (F = 64: two waves per column tile, one per 32-row half)
                    const int half = wave / S::kNCT;
                    live_scratch[(half * 2) * F + n0 + q] = a;
                    live_scratch[(half * 2 + 1) * F + n0 + q] = b;
                }
            } else if (l15 == 0 && slot < CWN_STAT_ROWS(Mv)) {
                D.col_sum[slot * F + n0 + q] = a;
                D.col_sumsq[slot * F + n0 + q] = b;
            }
        }
        if (slotted) {
            // the workgroup's 2 F sums leave as 2 F / 64 fully coalesced fp64 atomic instructions
            lds_barrier();                        // (LDS traffic only: the stores of Y above are not waited for)
            if (threadIdx.x < 2 * F && !(B.dbg & 4)) {
                const int which = threadIdx.x / F, col = threadIdx.x % F;
                double v = live_scratch[which * F + col];
                if constexpr (TM > 32) v += live_scratch[(2 + which) * F + col];
                unsafeAtomicAdd(D.stat_slots + (size_t)(((int)blockIdx.x % CWN_BN_SLOTS) * 2 + which) * F + col, v);
            }
        }
    }
}

template <int F>
__global__ __launch_bounds__(kThreads) void dense_stage_kernel(StageBatch B) {
    dense_stage_body<F, false>(B);
}

template <int F>
__global__ __launch_bounds__(kThreads) void dense_stage_ex_kernel(StageBatchEx B) {
    dense_stage_body<F, true>(B);
}

// ---- the same stage BACKWARD: dX = dz W, with dz formed on the way in ---------------------------------------------------------
//     dz = scale * (dyh - s1 / M - xhat * s2 / M),   dyh = dy * [z * scale + shift > 0],   xhat = (z - mean) * rstd
// (BatchNorm(train) + ReLU backward given the column sums s1 = sum dyh, s2 = sum dyh * xhat of cwn_norm_bwd_reduce_f32;
// without a norm: dz = dy * [z > 0]).  The tile of dz goes to global memory (the weight-gradient GEMM reads it) and,
// split once, into the LDS planes; one or two products leave from the SAME planes: dX = dz W for a Linear(F -> F), the two
// halves dz W[:, :F] and dz W[:, F:] for combine_nn's Linear(2F -> F).  Weights: the transposed blocks
// (cwn_update_mlp_pack_weights_t_many_f32).  Replaces the transposed-weight launch of cwn_gemm_f32 with the cwn_gemm_bnb
// prologue (fp32 MFMA, weights staged through LDS per workgroup: 16.2 us per launch at the ZINC batch of 128, twelve per
// training step).
struct StageBwdBatch {
    cwn_stage_bwd_desc d[CWN_MAX_DESCS];
    int32_t blk_start[CWN_MAX_DESCS + 1];
    int32_t n;
};

template <int F>
__global__ __launch_bounds__(kThreads, 4) void dense_stage_bwd_kernel(StageBwdBatch B) {
    using S = Shape<F>;
    constexpr int TM = S::kTM, kRowStride = S::kRowStride, kChunksPerTile = S::kChunksPerTile, kKS = S::kKS;
    constexpr size_t kPlaneElems = S::kPlaneElems;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // 1 KiB next to the tile: the slot sums s1 | s2 a live consumer takes in its prologue, and the workgroup's column sums of
    // a live producer on their way to one coalesced atomic per column ([half][s1 | s2][F])
    __shared__ __attribute__((aligned(16))) float bn_scratch[256];
    uint16_t* const buf0 = reinterpret_cast<uint16_t*>(smem);
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_stage_bwd_desc& D = B.d[di];
    const bool first_block = (int)blockIdx.x == B.blk_start[di];
    const int64_t row0 = (int64_t)((int)blockIdx.x - B.blk_start[di]) * TM;
    const int64_t Mv = D.m_dev != nullptr ? *D.m_dev : D.M;      // rows that exist (D.M: the capacity, bounds the addresses)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ct = wave % S::kNCT, rt0 = (wave / S::kNCT) * kRT;
    const int l15 = lane & 15, kq = lane >> 4;
    const bool two = D.wt2_packed != nullptr;

    typedef float4 RowRegs[kV];
    RowRegs vy, vz;
    auto request_rows = [&](RowRegs& v, const float* X, int64_t ld) {
#pragma unroll
        for (int i = 0; i < kV; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx / (F / 4), c4 = idx % (F / 4);
            const int64_t row = row0 + r < D.M ? row0 + r : D.M - 1;
            v[i] = reinterpret_cast<const float4*>(X + row * ld)[c4];
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // ONE set of weight registers; the second block of a two-output stage is requested when the first product is done (its
    // latency is covered by the CU's other workgroup).  Two sets, two accumulator sets and the fragments of both row tiles
    // were 164 registers -- one workgroup per CU where two overlap each other's loads -- and bounded to 128 the compiler
    // put 56 of them in scratch; requesting the second block k step by k step into the registers the first had just
    // finished with was allocated as a second set all the same (44 in scratch).
    typedef uint4 WeightRegs[kKS][3];
    WeightRegs wf;
    auto request_kstep = [&](const void* packed, int ks) {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(packed) + (size_t)ct * kChunksPerTile * 1024 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[ks][pl] = *reinterpret_cast<const uint4*>(wp + (ks * 3 + pl) * 1024);
    };
    typedef frag_cd AccRegs[kRT];
    auto multiply = [&](AccRegs& acc) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
#pragma unroll
            for (int rt = 0; rt < kRT; ++rt) {
                const uint16_t* p = buf0 + (size_t)((rt0 + rt) * 16 + l15) * kRowStride + ks * 32 + kq * 8;
                const uint4 xh = *reinterpret_cast<const uint4*>(p);
                const uint4 xm = *reinterpret_cast<const uint4*>(p + kPlaneElems);
                const uint4 xl = *reinterpret_cast<const uint4*>(p + 2 * kPlaneElems);
                acc[rt] = cwn::mfma_split6(wf[ks][0], wf[ks][1], wf[ks][2], xh, xm, xl, acc[rt]);
            }
        }
    };

    // requests: (a live consumer: the slot sums of its s1 | s2, thread t < 2 F one column, AHEAD of the tiles,) the two tiles,
    // the constants of this thread's four columns (the same in every row it stages), the weight
    float slot_t[CWN_BN_SLOTS];
    const bool slot_mine = D.s_slots != nullptr && (int)threadIdx.x < 2 * F;
    if (slot_mine) {
#pragma unroll
        for (int q = 0; q < CWN_BN_SLOTS; ++q) slot_t[q] = D.s_slots[(size_t)q * 2 * F + threadIdx.x];
    }
    request_rows(vy, D.dy, D.lddy);
    request_rows(vz, D.z, D.ldz);
    const int c4 = threadIdx.x % (F / 4);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f), mu = sh, c0 = sh, c1 = sh;
    const bool norm = D.scale != nullptr;
    if (norm) {
        sc = reinterpret_cast<const float4*>(D.scale)[c4];
        sh = reinterpret_cast<const float4*>(D.shift)[c4];
        mu = reinterpret_cast<const float4*>(D.mean)[c4];
        const float4 rs = reinterpret_cast<const float4*>(D.rstd)[c4];
        float4 s1, s2;
        if (D.s_slots != nullptr) {
            // the sums the producing launch left in the slots, in slot order
            if (slot_mine) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < CWN_BN_SLOTS; ++q) a += slot_t[q];
                bn_scratch[threadIdx.x] = a;
                if (first_block) ((int)threadIdx.x < F ? D.s1 : D.s2 - F)[threadIdx.x] = a;
            }
            lds_barrier();                       // (LDS traffic only: the tiles stay in flight)
            s1 = reinterpret_cast<const float4*>(bn_scratch)[c4];
            s2 = reinterpret_cast<const float4*>(bn_scratch + F)[c4];
            // (the producer side stages through the same scratch -- behind the barrier in front of the product below)
        } else {
            s1 = reinterpret_cast<const float4*>(D.s1)[c4];
            s2 = reinterpret_cast<const float4*>(D.s2)[c4];
        }
        const float invM = 1.0f / (float)(Mv > 0 ? Mv : 1);
        c0 = make_float4(-sc.x * (s1.x * invM), -sc.y * (s1.y * invM), -sc.z * (s1.z * invM), -sc.w * (s1.w * invM));
        c1 = make_float4(-sc.x * (rs.x * (s2.x * invM)), -sc.y * (rs.y * (s2.y * invM)), -sc.z * (rs.z * (s2.z * invM)),
                         -sc.w * (rs.w * (s2.w * invM)));
        if (first_block && threadIdx.x < F / 4) {       // the sums go on to beta.grad / gamma.grad: one writer per column
            if (D.acc1 != nullptr) {
                float4* a = reinterpret_cast<float4*>(D.acc1) + c4;
                const float4 o = *a;
                *a = make_float4(o.x + s1.x, o.y + s1.y, o.z + s1.z, o.w + s1.w);
            }
            if (D.acc2 != nullptr) {
                float4* a = reinterpret_cast<float4*>(D.acc2) + c4;
                const float4 o = *a;
                *a = make_float4(o.x + s2.x, o.y + s2.y, o.z + s2.z, o.w + s2.w);
            }
        }
    }
    // (a tile past the batch's own rows leaves here -- behind the hand-over of the sums above, which the FIRST workgroup of a
    // descriptor does whatever the batch holds)
    if (row0 >= Mv) return;
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) request_kstep(D.wt_packed, ks);
    const bool relu = D.relu != 0;
#pragma unroll
    for (int i = 0; i < kV; ++i) {
        const int idx = threadIdx.x + i * kThreads, r = idx / (F / 4);
        const float4 dy = vy[i], z = vz[i];
        auto one = [&](float dyv, float zv, float s, float h, float m, float a0, float a1) {
            const float y = zv * s + h;
            const float dyh = (!relu || y > 0.f) ? dyv : 0.f;
            return s * dyh + a0 + a1 * (zv - m);
        };
        const float4 d = make_float4(one(dy.x, z.x, sc.x, sh.x, mu.x, c0.x, c1.x), one(dy.y, z.y, sc.y, sh.y, mu.y, c0.y, c1.y),
                                     one(dy.z, z.z, sc.z, sh.z, mu.z, c0.z, c1.z), one(dy.w, z.w, sc.w, sh.w, mu.w, c0.w, c1.w));
        if (D.dz != nullptr && row0 + r < Mv) cwn::store_result4(D.dz + (row0 + r) * D.lddz + c4 * 4, d.x, d.y, d.z, d.w);
        uint2 ph, pm, pl;
        cwn::split4(d, ph, pm, pl);
        uint16_t* dst = buf0 + (size_t)r * kRowStride + c4 * 4;
        *reinterpret_cast<uint2*>(dst) = ph;
        *reinterpret_cast<uint2*>(dst + kPlaneElems) = pm;
        *reinterpret_cast<uint2*>(dst + 2 * kPlaneElems) = pl;
    }
    lds_barrier();
    const int n0 = ct * 16 + kq * 4;                 // D[i][j]: i = output column (lane >> 4) * 4 + reg, j = row (lane & 15)
    AccRegs acc;
    // what a live producer needs of the stage that receives its dx: that stage's z at this lane's rows / columns -- requested
    // BEFORE the product whose result it meets (behind it: an exposed round trip to HBM at the end of every workgroup)
    struct LiveIn { float4 zz[kRT]; };
    auto request_live = [&](const cwn_bn_bwd_live& L, LiveIn& R) {
        if (L.slots == nullptr) return;
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) {
            const int64_t row = row0 + (rt0 + rt) * 16 + l15;
            R.zz[rt] = *reinterpret_cast<const float4*>(L.z + (row < D.M ? row : D.M - 1) * L.ldz + n0);
        }
    };
    auto store = [&](float* out, int64_t ld, const cwn_bn_bwd_live& L, const LiveIn& R) {
        const bool live = L.slots != nullptr;
        const float4 (&zz)[kRT] = R.zz;
        // (the constants behind the product: with them in flight across it the F = 128 kernel spills past its 128 registers)
        float4 lsc, lsh, lmu, lrs;
        if (live) {
            lsc = *reinterpret_cast<const float4*>(L.aff + n0);
            lsh = *reinterpret_cast<const float4*>(L.aff + F + n0);
            lmu = *reinterpret_cast<const float4*>(L.aff + 2 * F + n0);
            lrs = *reinterpret_cast<const float4*>(L.aff + 3 * F + n0);
        }
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) {
            const int r = (rt0 + rt) * 16 + l15;
            if (row0 + r < Mv) cwn::store_result4(out + (row0 + r) * ld + n0, acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
        }
        if (!live) return;
        // dyh = dx * [z * scale + shift > 0]; column sums of dyh and dyh * xhat over the workgroup's rows
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        const float sc[4] = {lsc.x, lsc.y, lsc.z, lsc.w}, sh[4] = {lsh.x, lsh.y, lsh.z, lsh.w};
        const float mu[4] = {lmu.x, lmu.y, lmu.z, lmu.w}, rs[4] = {lrs.x, lrs.y, lrs.z, lrs.w};
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) {
            const bool ok = row0 + (rt0 + rt) * 16 + l15 < Mv;
            const float zv[4] = {zz[rt].x, zz[rt].y, zz[rt].z, zz[rt].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float y = zv[q] * sc[q] + sh[q];
                const float dyh = (ok && y > 0.f) ? acc[rt][q] : 0.f;
                a1[q] += dyh;
                a2[q] += dyh * ((zv[q] - mu[q]) * rs[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = a1[q], b = a2[q];
            a += row_ror_f32<0x128>(a); b += row_ror_f32<0x128>(b);
            a += row_ror_f32<0x124>(a); b += row_ror_f32<0x124>(b);
            a += row_ror_f32<0x122>(a); b += row_ror_f32<0x122>(b);
            a += row_ror_f32<0x121>(a); b += row_ror_f32<0x121>(b);
            if (l15 == 0) {                       // (F = 64: two waves per column tile, one per 32-row half)
                const int half = wave / S::kNCT;
                bn_scratch[(half * 2) * F + n0 + q] = a;
                bn_scratch[(half * 2 + 1) * F + n0 + q] = b;
            }
        }
        lds_barrier();                            // (LDS traffic only: the stores of dx above are not waited for)
        if ((int)threadIdx.x < 2 * F) {
            float v = bn_scratch[threadIdx.x];
            if constexpr (TM > 32) v += bn_scratch[2 * F + threadIdx.x];
            unsafeAtomicAdd(L.slots + (size_t)((int)blockIdx.x % CWN_BN_SLOTS) * 2 * F + threadIdx.x, v);
        }
        lds_barrier();                            // (a second product of the workgroup stages through the same scratch)
    };
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt) acc[rt] = frag_cd{0.f, 0.f, 0.f, 0.f};
    LiveIn live_in;
    request_live(D.out_bn, live_in);
    multiply(acc);
    if (two) {
        __builtin_amdgcn_sched_barrier(0);               // (the requests below not hoisted into the product above)
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) request_kstep(D.wt2_packed, ks);
    }
    store(D.dx, D.lddx, D.out_bn, live_in);
    if (two) {
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) acc[rt] = frag_cd{0.f, 0.f, 0.f, 0.f};
        request_live(D.out_bn2, live_in);
        multiply(acc);
        store(D.dx2, D.lddx2, D.out_bn2, live_in);
    }
}

inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; }

template <int F>
int launch_stage(const StageBatch& B, int64_t blocks, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_stage_kernel<F>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Shape<F>::kLdsBytes);
    });
    if (attr_err != hipSuccess) return CWN_ERR_LAUNCH;
    dense_stage_kernel<F><<<dim3((unsigned)blocks), dim3(kThreads), Shape<F>::kLdsBytes, stream>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

template <int F>
int launch_stage_ex(const StageBatchEx& B, int64_t blocks, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_stage_ex_kernel<F>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Shape<F>::kLdsBytes);
    });
    if (attr_err != hipSuccess) return CWN_ERR_LAUNCH;
    dense_stage_ex_kernel<F><<<dim3((unsigned)blocks), dim3(kThreads), Shape<F>::kLdsBytes, stream>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

template <int F>
int launch_stage_bwd(const StageBwdBatch& B, int64_t blocks, hipStream_t stream) {
    dense_stage_bwd_kernel<F><<<dim3((unsigned)blocks), dim3(kThreads), Shape<F>::kBufBytes, stream>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

// n weights -> packed blocks in ONE launch: entry e is the F x F block W[e][:, col0[e] : col0[e] + F] (row stride ldw[e])
struct PackTable {
    const float* W[CWN_STAGE_PACK_MAX];
    unsigned char* out[CWN_STAGE_PACK_MAX];
    int64_t ldw[CWN_STAGE_PACK_MAX];
    uint8_t trans[CWN_STAGE_PACK_MAX];        // per entry: the block of the transposed weight
};
static_assert(sizeof(PackTable) <= 4096, "kernel-argument segment");

// (the layout of cwn_update_mlp_pack_weights_f32: chunk ((tile * KS + ks) * 3 + plane), lane l = kq * 16 + n holds
// W[tile * 16 + n][ks * 32 + kq * 8 ..] of that plane)
// TRANS: the block of the TRANSPOSED weight (dX = dz W: output column = input feature of the Linear, reduction over its
// outputs): lane l = kq * 16 + n of chunk (tile, ks) holds W[ks * 32 + kq * 8 ..][tile * 16 + n]
template <int F>
__global__ __launch_bounds__(256) void pack_stage_weights_kernel(PackTable T) {
    constexpr int KS = F / 32;
    constexpr int kPerWeight = (F / 16) * KS * 64;                 // one thread per (tile, ks, lane)
    const int e = blockIdx.y;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= kPerWeight) return;
    const int lane = g & 63, ks = (g >> 6) % KS, tile = (g >> 6) / KS;
    float4 a, b;
    if (T.trans[e]) {
        const float* src = T.W[e] + (int64_t)(ks * 32 + (lane >> 4) * 8) * T.ldw[e] + tile * 16 + (lane & 15);
        const int64_t ld = T.ldw[e];
        a = make_float4(src[0], src[ld], src[2 * ld], src[3 * ld]);
        b = make_float4(src[4 * ld], src[5 * ld], src[6 * ld], src[7 * ld]);
    } else {
        const float* src = T.W[e] + (int64_t)(tile * 16 + (lane & 15)) * T.ldw[e] + ks * 32 + (lane >> 4) * 8;
        a = make_float4(src[0], src[1], src[2], src[3]);
        b = make_float4(src[4], src[5], src[6], src[7]);
    }
    uint4 ph, pm, pl;
    cwn::split8(a, b, ph, pm, pl);
    unsigned char* dst = T.out[e] + ((size_t)(tile * KS + ks) * 3) * 1024 + lane * 16;
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + 1024) = pm;
    *reinterpret_cast<uint4*>(dst + 2048) = pl;
}

}  // namespace

namespace {
// entries: n blocks in the plain form into out[] (when given) and the same n in the transposed form into out_t[] (when given)
int pack_stage_many(const float* const* W, const int64_t* ldw, int32_t F, void* const* out, void* const* out_t, int32_t n,
                    cwn_stream_t stream_) {
    if ((F != 64 && F != 128) || n < 0) return CWN_ERR_BAD_ARG;
    const int forms = (out != nullptr) + (out_t != nullptr);
    if (n == 0) return CWN_OK;
    if (W == nullptr || ldw == nullptr || forms == 0 || (int64_t)n * forms > CWN_STAGE_PACK_MAX) return CWN_ERR_BAD_ARG;
    PackTable T{};
    int m = 0;
    for (int form = 0; form < 2; ++form) {
        void* const* o = form == 0 ? out : out_t;
        if (o == nullptr) continue;
        for (int e = 0; e < n; ++e, ++m) {
            if (W[e] == nullptr || o[e] == nullptr || ldw[e] < F) return CWN_ERR_BAD_ARG;
            if (((uintptr_t)W[e] & 3u) || ((uintptr_t)o[e] & 15u)) return CWN_ERR_ALIGN;
            T.W[m] = W[e];
            T.out[m] = (unsigned char*)o[e];
            T.ldw[m] = ldw[e];
            T.trans[m] = (uint8_t)form;
        }
    }
    const int threads = (F / 16) * (F / 32) * 64;
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((threads + 255) / 256, m);
    if (F == 128) pack_stage_weights_kernel<128><<<grid, dim3(256), 0, stream>>>(T);
    else pack_stage_weights_kernel<64><<<grid, dim3(256), 0, stream>>>(T);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
}  // namespace

extern "C" int cwn_update_mlp_pack_weights_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out,
                                                    int32_t n, cwn_stream_t stream) {
    return out == nullptr ? CWN_ERR_BAD_ARG : pack_stage_many(W, ldw, F, out, nullptr, n, stream);
}

extern "C" int cwn_update_mlp_pack_weights_t_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out,
                                                      int32_t n, cwn_stream_t stream) {
    return out == nullptr ? CWN_ERR_BAD_ARG : pack_stage_many(W, ldw, F, nullptr, out, n, stream);
}

extern "C" int cwn_update_mlp_pack_weights_both_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out,
                                                         void* const* out_t, int32_t n, cwn_stream_t stream) {
    return (out == nullptr || out_t == nullptr) ? CWN_ERR_BAD_ARG : pack_stage_many(W, ldw, F, out, out_t, n, stream);
}

static int fill_stage_batch(StageBatch& B, const cwn_stage_desc* descs, int n, int32_t F, int64_t& blocks) {
    if (descs == nullptr || n < 1 || n > CWN_MAX_DESCS || (F != 64 && F != 128)) return CWN_ERR_BAD_ARG;
    const int TM = 4096 / F;
    B.n = n;
    blocks = 0;
    for (int i = 0; i < n; ++i) {
        const cwn_stage_desc& D = descs[i];
        if (D.M < 0) return CWN_ERR_BAD_ARG;
        B.blk_start[i] = (int32_t)blocks;
        B.d[i] = D;
        if (D.M == 0) continue;
        if (D.X == nullptr || D.Y == nullptr || D.w_packed == nullptr) return CWN_ERR_BAD_ARG;
        if ((D.X2 == nullptr) != (D.w2_packed == nullptr)) return CWN_ERR_BAD_ARG;
        if ((D.in_scale == nullptr) != (D.in_shift == nullptr) || (D.in_scale2 == nullptr) != (D.in_shift2 == nullptr))
            return CWN_ERR_BAD_ARG;
        if ((D.col_sum == nullptr) != (D.col_sumsq == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.stat_slots != nullptr && (D.col_sum != nullptr || ((uintptr_t)D.stat_slots & 7u))) return CWN_ERR_BAD_ARG;
        if (D.in_bn.slots != nullptr && (D.in_scale != nullptr || D.in_bn.aff == nullptr || !al16(D.in_bn.aff))) return CWN_ERR_BAD_ARG;
        if (D.in_bn2.slots != nullptr && (D.X2 == nullptr || D.in_scale2 != nullptr || D.in_bn2.aff == nullptr || !al16(D.in_bn2.aff)))
            return CWN_ERR_BAD_ARG;
        if ((D.in_bn.running_mean == nullptr) != (D.in_bn.running_var == nullptr)) return CWN_ERR_BAD_ARG;
        if ((D.in_bn2.running_mean == nullptr) != (D.in_bn2.running_var == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.ldx < F || D.ldy < F || D.ldx % 4 || D.ldy % 4 || (D.X2 != nullptr && (D.ldx2 < F || D.ldx2 % 4)))
            return CWN_ERR_BAD_ARG;
        if (!(al16(D.X) && al16(D.X2) && al16(D.Y) && al16(D.w_packed) && al16(D.w2_packed) && al16(D.bias) && al16(D.in_scale) &&
              al16(D.in_shift) && al16(D.in_scale2) && al16(D.in_shift2)))
            return CWN_ERR_ALIGN;
        if (((uintptr_t)D.col_sum & 7u) || ((uintptr_t)D.col_sumsq & 7u)) return CWN_ERR_ALIGN;
        blocks += (D.M + TM - 1) / TM;
        if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    }
    for (int i = n; i <= CWN_MAX_DESCS; ++i) B.blk_start[i] = (int32_t)blocks;
    static const int dbg = getenv("CWN_STAGE_DBG") ? atoi(getenv("CWN_STAGE_DBG")) : 0;
    B.dbg = dbg;
    return CWN_OK;
}

extern "C" int cwn_dense_stage_f32(const cwn_stage_desc* descs, int n, int32_t F, cwn_stream_t stream_) {
    StageBatch B{};
    int64_t blocks = 0;
    const int rc = fill_stage_batch(B, descs, n, F, blocks);
    if (rc != CWN_OK || blocks == 0) return rc;
    return F == 128 ? launch_stage<128>(B, blocks, (hipStream_t)stream_) : launch_stage<64>(B, blocks, (hipStream_t)stream_);
}

extern "C" int cwn_dense_stage_ex_f32(const cwn_stage_desc* descs, const cwn_stage_extra* extras, int n, int32_t F,
                                      cwn_stream_t stream_) {
    if (extras == nullptr) return cwn_dense_stage_f32(descs, n, F, stream_);
    StageBatchEx B{};
    int64_t blocks = 0;
    const int rc = fill_stage_batch(B, descs, n, F, blocks);
    if (rc != CWN_OK) return rc;
    bool any = false;
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 2; ++k) {
            const cwn_stage_extra& E = extras[2 * i + k];
            B.more[i][k] = E;
            if (descs[i].M == 0 || E.X == nullptr) continue;
            if (k == 1 && extras[2 * i].X == nullptr) return CWN_ERR_BAD_ARG;        // a fourth block needs the third
            if (descs[i].X2 == nullptr || E.w_packed == nullptr || E.ldx < F || E.ldx % 4) return CWN_ERR_BAD_ARG;
            if (E.bn.slots != nullptr && (E.bn.aff == nullptr || !al16(E.bn.aff))) return CWN_ERR_BAD_ARG;
            if ((E.bn.running_mean == nullptr) != (E.bn.running_var == nullptr)) return CWN_ERR_BAD_ARG;
            if (!(al16(E.X) && al16(E.w_packed))) return CWN_ERR_ALIGN;
            any = true;
        }
    }
    if (blocks == 0) return CWN_OK;
    if (!any) return F == 128 ? launch_stage<128>(B, blocks, (hipStream_t)stream_) : launch_stage<64>(B, blocks, (hipStream_t)stream_);
    return F == 128 ? launch_stage_ex<128>(B, blocks, (hipStream_t)stream_) : launch_stage_ex<64>(B, blocks, (hipStream_t)stream_);
}

extern "C" int cwn_dense_stage_bwd_f32(const cwn_stage_bwd_desc* descs, int n, int32_t F, cwn_stream_t stream_) {
    if (descs == nullptr || n < 1 || n > CWN_MAX_DESCS || (F != 64 && F != 128)) return CWN_ERR_BAD_ARG;
    const int TM = 4096 / F;
    StageBwdBatch B{};
    B.n = n;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        const cwn_stage_bwd_desc& D = descs[i];
        if (D.M < 0) return CWN_ERR_BAD_ARG;
        B.blk_start[i] = (int32_t)blocks;
        B.d[i] = D;
        if (D.M == 0) continue;
        if (D.dy == nullptr || D.z == nullptr || D.wt_packed == nullptr || D.dx == nullptr) return CWN_ERR_BAD_ARG;
        if ((D.wt2_packed == nullptr) != (D.dx2 == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.scale != nullptr && (D.shift == nullptr || D.mean == nullptr || D.rstd == nullptr || D.s1 == nullptr || D.s2 == nullptr))
            return CWN_ERR_BAD_ARG;
        if (D.lddy < F || D.ldz < F || D.lddx < F || D.lddy % 4 || D.ldz % 4 || D.lddx % 4) return CWN_ERR_BAD_ARG;
        if (D.s_slots != nullptr && (D.scale == nullptr || !al16(D.s_slots))) return CWN_ERR_BAD_ARG;
        for (const cwn_bn_bwd_live* L : {&D.out_bn, &D.out_bn2}) {
            if (L->slots == nullptr) continue;
            if (L == &D.out_bn2 && D.dx2 == nullptr) return CWN_ERR_BAD_ARG;
            if (L->z == nullptr || L->aff == nullptr || L->ldz < F || L->ldz % 4) return CWN_ERR_BAD_ARG;
            if (!(al16(L->z) && al16(L->aff) && al16(L->slots))) return CWN_ERR_ALIGN;
        }
        if (D.dz != nullptr && (D.lddz < F || D.lddz % 4)) return CWN_ERR_BAD_ARG;
        if (D.dx2 != nullptr && (D.lddx2 < F || D.lddx2 % 4)) return CWN_ERR_BAD_ARG;
        if (!(al16(D.dy) && al16(D.z) && al16(D.dz) && al16(D.dx) && al16(D.dx2) && al16(D.wt_packed) && al16(D.wt2_packed) &&
              al16(D.scale) && al16(D.shift) && al16(D.mean) && al16(D.rstd) && al16(D.s1) && al16(D.s2) && al16(D.acc1) && al16(D.acc2)))
            return CWN_ERR_ALIGN;
        blocks += (D.M + TM - 1) / TM;
        if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    }
    for (int i = n; i <= CWN_MAX_DESCS; ++i) B.blk_start[i] = (int32_t)blocks;
    if (blocks == 0) return CWN_OK;
    return F == 128 ? launch_stage_bwd<128>(B, blocks, (hipStream_t)stream_) : launch_stage_bwd<64>(B, blocks, (hipStream_t)stream_);
}
