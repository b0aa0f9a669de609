// cwn_mlp.hip -- the update / combine networks of a SparseCIN layer in ONE launch (inference).
//
// What it replaces (reference): mp/layers.py:193-199
//     out_up = update_up_nn(out_up); out_b = update_boundaries_nn(out_b); return combine_nn(cat(out_up, out_b))
// with update_*_nn = [Linear(128->128), BatchNorm, ReLU] x 2 (mp/layers.py:303-321) and combine_nn =
// [Linear(256->128), BatchNorm, ReLU] (:322-325), BatchNorm in eval mode folded into a per-column affine.
// The library ran this as three grouped GEMM launches (cwn_gemm_f32: stage 1 and 2 of both branches on the
// bf16-split kernel, the combine with K = 256 on the fp32-MFMA kernel), ~34 us of a 52 us layer at the ZINC
// batch of 128, the two intermediate activations of every branch written to and read back from HBM.
//
// The five Linear layers are ROW-LOCAL, so a workgroup takes 32 rows of one cochain dimension through
// all of them without leaving the CU: the activations of a stage go from the accumulators straight
// into the bf16 planes (LDS) the next stage multiplies, and the only thing streamed per stage is its
// weight -- packed once per weight version by cwn_gemm_pack_weights_f32, 96 KB of fragment-ordered bf16
// planes, 1 KiB per load instruction.  Arithmetic: the exact three-way bf16 split of csrc/cwn_split.h
// for every product (fp32 in, fp32 accumulate, fp32 out; the same split, MFMA order and epilogue as
// cwn_gemm_split.hip, so stage 1 and 2 are bit-identical to the launches they replace; the combine,
// formerly on fp32 MFMA, is now on the same path: fp32 accuracy, not bit-identical to an fmaf chain).
//
//   x_up --W1u--> relu(bn) --W2u--> relu(bn) = h_up --+
//                                                      +-- Wc[:, :128] h_up + Wc[:, 128:] h_b --> relu(bn) = y
//   x_b  --W1b--> relu(bn) --W2b--> relu(bn) = h_b  --+
//
// Every workgroup streams all six weights (576 KB out of L2); measured against the three weight-stationary
// grouped GEMM launches on the full forward (tools/mlp_crossover.sh, ms): batch 128 0.172 vs 0.245, 1024
// 0.71 vs 0.96, 8192 4.57 vs 5.79 -- the HBM round trips of the intermediates cost more than the weight
// stream at every size measured.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include "../../include/cwn_hip.h"
#include "cwn_split.h"
#include "cwn_mem.h"

namespace {

using cwn::frag_cd;

constexpr int kThreads = 512;

// F = the width of every Linear (64 or 128).  A workgroup takes TM = 4096 / F rows (32 / 64): 8 waves x two 16 x 16 tiles
// (RT = 2) cover its TM x F output -- wave w owns column tile w % (F / 16) and the row tiles 2 (w / (F / 16)), + 1.
// Two schedules of the same arithmetic (bit-identical results):
//   SEQ = false (rounds 2 - 4): the two branches ALTERNATE, the epilogue of a stage under the other branch's MFMAs; five
//     plane buffers, 130 - 138 KB of LDS: the workgroup owns its CU.  The shortest chain per workgroup -- the form of a
//     launch that is one round of workgroups (ZINC-128: 214 on 256 CUs).
//   SEQ = true (round 5): the branches one after the other through THREE buffers (78 KB) within 128 registers: TWO
//     workgroups per CU.  The stamps (tools/time_mlp_phases.py) say a workgroup is a chain of latencies -- load + split
//     4.5 k cycles, six multiplications of ~1.1 k (the matrix pipe would need 0.77 k), four epilogues of ~1.1 k, at
//     width 64 -- of which the pipe is busy for 29 %: in a launch of several rounds (REDDIT-like batches: 1142 workgroups;
//     ZINC-2048: 3400) a second workgroup on the CU fills the rest.  (Half-SIZE workgroups, two per CU, were measured first:
//     a half-size workgroup is the same chain of latencies, 15.5 k cycles against 16.1 k -- no gain.)
template <int F, int RT, bool SEQ> struct Shape {
    static constexpr int kRT = RT, kV = RT;
    static constexpr int kTM = RT * 2048 / F;
    static constexpr int kNCT = F / 16;
    static constexpr int kKS = F / 32;
    // bf16 elements per LDS row.  F + 8: 16 bytes of padding make the fragment reads (16 rows x 16 B per quarter wave)
    // conflict-free.  Sequential schedule at width 64: three padded buffers of 64 rows are 2 KB beyond half a CU, so the
    // rows are unpadded and the 16-byte chunks of a row XOR-swizzled with (row >> 1) & 7 instead (col(): rows r, r + 1 differ
    // in bank half, the eight row pairs of a fragment read in chunk).
    static constexpr bool kSwizzle = SEQ && F == 64;
    static constexpr int kRowStride = kSwizzle ? F : F + 8;
    static __device__ __forceinline__ int col(int row, int c) {
        if constexpr (kSwizzle) return (((c >> 3) ^ ((row >> 1) & 7)) << 3) | (c & 7);
        else return c;
    }
    static constexpr int kChunksPerTile = kKS * 3;    // packed weight: 1-KiB chunks per 16-column tile (k steps x planes)
    static constexpr size_t kPlaneElems = (size_t)kTM * kRowStride;
    static constexpr size_t kBufBytes = 3 * kPlaneElems * 2;   // three planes
    static constexpr size_t kLdsBytes = (SEQ ? 3 : 5) * kBufBytes;   // alternating: x_up / h_b, x_b, h1_up, h1_b, h_up
    static_assert(kTM * (F / 4) == kV * kThreads && (kTM / 16) * kNCT == 8 * kRT, "tile shape");
    static_assert(!SEQ || 2 * kLdsBytes <= 160 * 1024, "two workgroups per CU");
};

struct MlpBatch {
    cwn_mlp_dim d[CWN_LAYER_MAX_DIMS];
    int32_t blk_start[CWN_LAYER_MAX_DIMS + 1];
    int32_t n;
#ifdef CWN_MLP_TIMING
    unsigned long long* stamps;              // [workgroups][16] shader-clock stamps (instrumented build only)
#endif
};

#ifdef CWN_MLP_TIMING
#define MLP_STAMP(k)                                                                           \
    do {                                                                                       \
        if (threadIdx.x == 0 && B.stamps != nullptr)                                           \
            B.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();            \
    } while (0)
unsigned long long* g_mlp_stamps = nullptr;
#else
#define MLP_STAMP(k) do { } while (0)
#endif

// One workgroup per CU, EIGHT waves: wave w owns the 16 output columns of column tile w for both 16-row tiles
// and the registers of that column tile's weight, TWICE (the weight of stage k + 1 is requested into the
// second set before the MFMAs of stage k).  What was measured on the way (per launch at the ZINC batch of
// 128, 214 workgroups): four waves x 32 columns, one or two register sets: 20.7 - 21.7 us (one wave per SIMD:
// every split / epilogue instruction and every dependent MFMA waits out its own latency); sixteen waves x one
// 16 x 16 tile: 21.0 us -- two waves per column tile request the same weight, 192 KB a stage, and the address
// unit (64 B per clock and CU) became the bound; and in every form the epilogue constants requested BEHIND
// the next weight made each stage wait for that weight (loads return in order).
// NARROW: some dimension's inputs have fewer than F columns (cwn_mlp_dim.in_width) -- a build of its own, so that the row loads
// of the common form stay one unconditional 16-byte load (with the test inside, uniform as it is, the ZINC-128 launch went
// 10.7 -> 12.0 us: the compiler schedules loads behind a branch differently).
template <int F, int RT, bool SEQ, bool NARROW>
// (HIP's second __launch_bounds__ argument is the minimum number of WAVES per SIMD, not CUDA's blocks per multiprocessor: two
// resident 8-wave workgroups are four waves a SIMD -- the compiler then holds the sequential build to 128 registers)
// (n_pre, bs1_pre, bs2_pre: copies of B.n, B.blk_start[1], B.blk_start[2] as LEADING scalar arguments -- preloaded into SGPRs by the
//  dispatcher, csrc/Makefile PRELOAD -- so the workgroup knows its dimension without a scalar load of the argument segment)
__global__ __launch_bounds__(kThreads, SEQ ? 4 : 1) void update_mlp_kernel(int32_t n_pre, int32_t bs1_pre, int32_t bs2_pre, MlpBatch B) {
    using S = Shape<F, RT, SEQ>;
    constexpr int TM = S::kTM, kRowStride = S::kRowStride, kChunksPerTile = S::kChunksPerTile, kKS = S::kKS;
    constexpr int kRT = S::kRT, kV = S::kV;
    constexpr size_t kPlaneElems = S::kPlaneElems, kBufBytes = S::kBufBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* const bufA = reinterpret_cast<uint16_t*>(smem);                   // x_up, later h_b
    uint16_t* const bufC = reinterpret_cast<uint16_t*>(smem + kBufBytes);       // x_b
    uint16_t* const bufB = reinterpret_cast<uint16_t*>(smem + 2 * kBufBytes);   // stage-1 output, upper branch
    uint16_t* const bufD = reinterpret_cast<uint16_t*>(smem + 3 * kBufBytes);   // stage-1 output, boundary branch
    uint16_t* const bufU = reinterpret_cast<uint16_t*>(smem + 4 * kBufBytes);   // h_up
    static_assert(CWN_LAYER_MAX_DIMS == 3, "two preloaded block starts");
    int di = 0, blk0 = 0;
    if (1 < n_pre && (int)blockIdx.x >= bs1_pre) { di = 1; blk0 = bs1_pre; }
    if (2 < n_pre && (int)blockIdx.x >= bs2_pre) { di = 2; blk0 = bs2_pre; }
    const cwn_mlp_dim& D = B.d[di];
    const int64_t row0 = (int64_t)((int)blockIdx.x - blk0) * TM;
    // rows that exist (include/cwn_hip.h, "device-side row counts"; D.M is then the capacity: it bounds the addresses)
    const int64_t Mv = D.m_dev != nullptr ? *D.m_dev : D.M;
    if (row0 >= Mv) return;                         // (uniform) a tile past the batch's own rows
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ct = wave % S::kNCT, rt0 = (wave / S::kNCT) * kRT;
    const int l15 = lane & 15, kq = lane >> 4;

    // an input tile: TM rows x 32 float4, two per thread, row-contiguous; rows past M are clamped, not guarded
    typedef float4 RowRegs[kV];
    RowRegs vU, vB;
    // (narrow input -- cwn_mlp_dim.in_width columns, the rest of the tile zero: the first weight is zero-padded to match)
    const int in_w = NARROW && D.in_width > 0 && D.in_width < F ? D.in_width : F;          // (uniform)
    auto request_rows = [&](RowRegs& v, const float* X, int64_t ld) {
#pragma unroll
        for (int i = 0; i < kV; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx / (F / 4), c4 = idx % (F / 4);
            const int64_t row = row0 + r < D.M ? row0 + r : D.M - 1;
            if (!NARROW || in_w == F) {
                v[i] = reinterpret_cast<const float4*>(X + row * ld)[c4];
            } else {
                const float* src = X + row * ld + c4 * 4;
                const int left = in_w - c4 * 4;
                v[i] = make_float4(left > 0 ? src[0] : 0.f, left > 1 ? src[1] : 0.f, left > 2 ? src[2] : 0.f, left > 3 ? src[3] : 0.f);
            }
        }
    };
    auto stage_rows = [&](const RowRegs& v, uint16_t* buf) {   // split the tile ONCE per element into the three planes
#pragma unroll
        for (int i = 0; i < kV; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx / (F / 4), c4 = idx % (F / 4);
            uint2 ph, pm, pl;
            cwn::split4(v[i], ph, pm, pl);
            uint16_t* dst = buf + (size_t)r * kRowStride + S::col(r, c4 * 4);
            *reinterpret_cast<uint2*>(dst) = ph;
            *reinterpret_cast<uint2*>(dst + kPlaneElems) = pm;
            *reinterpret_cast<uint2*>(dst + 2 * kPlaneElems) = pl;
        }
    };
    // workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL
    // load (s_waitcnt vmcnt(0)) -- here the next stage's weight, which is meant to keep streaming across the
    // barrier (measured: 3.3 k cycles per stage spent in that wait)
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // the stationary operand of a stage: this wave's 16 output columns, [k step][plane]; two sets
    typedef uint4 WeightRegs[kKS][3];
    WeightRegs wfA, wfB;
    auto request_kstep = [&](WeightRegs& wf, int k, int ks) {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(D.w_packed[k]) +
                                  (size_t)ct * kChunksPerTile * 1024 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[ks][pl] = *reinterpret_cast<const uint4*>(wp + (ks * 3 + pl) * 1024);
    };
    auto request_weight = [&](WeightRegs& wf, int k) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) request_kstep(wf, k, ks);
    };
    typedef frag_cd AccRegs[kRT];
    AccRegs accU, accB;                          // the two branches alternate (see the chain below)
    auto clear = [&](AccRegs& acc) {
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) acc[rt] = frag_cd{0.f, 0.f, 0.f, 0.f};
    };
    // acc += buf x W^T (k steps in order, six terms each: cwn_split.h).  `next` >= 0: the k steps of weight
    // `next` are requested into the OTHER register set one by one between this stage's MFMAs -- three 1-KiB
    // loads a wave at a time, which the address unit takes without making the wave wait (requested twelve at
    // once after the stage, the waves sat in the issue of their loads for 1.5 k cycles while the matrix pipe
    // idled, and then multiplied while the address unit idled)
    auto multiply = [&](AccRegs& acc, const uint16_t* buf, const WeightRegs& wf, WeightRegs& wnext, int next) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
#pragma unroll
            for (int rt = 0; rt < kRT; ++rt) {
                const int row = (rt0 + rt) * 16 + l15;
                const uint16_t* p = buf + (size_t)row * kRowStride + S::col(row, ks * 32 + kq * 8);
                const uint4 xh = *reinterpret_cast<const uint4*>(p);
                const uint4 xm = *reinterpret_cast<const uint4*>(p + kPlaneElems);
                const uint4 xl = *reinterpret_cast<const uint4*>(p + 2 * kPlaneElems);
                acc[rt] = cwn::mfma_split6(wf[ks][0], wf[ks][1], wf[ks][2], xh, xm, xl, acc[rt]);
            }
            if (next >= 0) request_kstep(wnext, next, ks);
        }
    };
    // the epilogue constants of a stage are requested BEFORE its MFMAs (and so before the next weight): loads
    // return in order, a constant behind 96 KB of weight is a wait for the weight (measured: 4.4 k cycles a stage)
    struct Consts { float4 b, sc, sh; bool affine; };
    Consts cU, cB;
    auto request_consts = [&](Consts& c, int s) {
        const int n0 = ct * 16 + kq * 4;
        c.b = make_float4(0.f, 0.f, 0.f, 0.f);
        c.sc = make_float4(1.f, 1.f, 1.f, 1.f);
        c.sh = c.b;
        if (D.bias[s] != nullptr) c.b = *reinterpret_cast<const float4*>(D.bias[s] + n0);
        c.affine = D.scale[s] != nullptr;
        if (c.affine) {
            c.sc = *reinterpret_cast<const float4*>(D.scale[s] + n0);
            c.sh = *reinterpret_cast<const float4*>(D.shift[s] + n0);
        }
    };
    // epilogue of stage s: + bias, folded BatchNorm, ReLU; then either into the planes of `buf` (the next
    // stage's operand) or, for the last stage, to y.  D[i][j]: i = output column (lane >> 4) * 4 + reg,
    // j = row (lane & 15): a lane holds 4 consecutive columns of one row.
    auto finish = [&](const AccRegs& acc, const Consts& c, uint16_t* buf) {
        const int n0 = ct * 16 + kq * 4;
        const float4 b4 = c.b, sc = c.sc, sh = c.sh;
        const bool affine = c.affine;
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) {
            float y[4] = {acc[rt][0] + b4.x, acc[rt][1] + b4.y, acc[rt][2] + b4.z, acc[rt][3] + b4.w};
            if (affine) {
                y[0] = y[0] * sc.x + sh.x;
                y[1] = y[1] * sc.y + sh.y;
                y[2] = y[2] * sc.z + sh.z;
                y[3] = y[3] * sc.w + sh.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q] = fmaxf(y[q], 0.0f);
            const int r = (rt0 + rt) * 16 + l15;
            if (buf != nullptr) {
                uint2 ph, pm, pl;
                cwn::split4(make_float4(y[0], y[1], y[2], y[3]), ph, pm, pl);
                uint16_t* dst = buf + (size_t)r * kRowStride + S::col(r, n0);
                *reinterpret_cast<uint2*>(dst) = ph;
                *reinterpret_cast<uint2*>(dst + kPlaneElems) = pm;
                *reinterpret_cast<uint2*>(dst + 2 * kPlaneElems) = pl;
            } else if (row0 + r < Mv) {
                cwn::store_result4(D.y + (row0 + r) * D.ldy + n0, y[0], y[1], y[2], y[3]);
            }
        }
    };

    // ---- the chain ----------------------------------------------------------------------------------------------
    // The two branches are independent until the combine, so their stages ALTERNATE: the epilogue of a stage
    // (VALU + LDS: bias, folded norm, ReLU, split into planes) sits in the instruction stream behind the MFMAs
    // of the OTHER branch's stage and runs while the matrix pipe works on them; the weight of the next
    // multiplication streams in k step by k step meanwhile.  Weight order: 1u, 1b, 2u, 2b, c(up half), c(b half).
    // (round 5, tried at width 64 and dropped: all six weights requested up front into registers of their own -- a wave's
    // slice of a weight is 24 registers, 214 in all without scratch -- instead of streaming one multiplication ahead: SLOWER,
    // 23.3 -> 26.2 us per launch at the molhiv batch, 0.376 -> 0.391 ms REDDIT forward: the stages do not wait for weights.)
    if constexpr (SEQ) {
        // one branch after the other through three buffers; weights in the order they are packed: 1u, 2u, 1b, 2b, c(up), c(b)
        uint16_t* const b0 = bufA;         // x_up, later h_up
        uint16_t* const b1 = bufC;         // x_b, later h_b
        uint16_t* const b2 = bufB;         // the stage-1 outputs
        MLP_STAMP(0);
        request_rows(vU, D.x_up, D.ldx_up);
        request_consts(cU, 0);
        request_weight(wfA, 0);
        request_rows(vB, D.x_b, D.ldx_b);
        stage_rows(vU, b0);
        stage_rows(vB, b1);
        lds_barrier();
        MLP_STAMP(1);
        clear(accU); multiply(accU, b0, wfA, wfB, 1);       // stage 1, upper branch     (W2u streams in)
        MLP_STAMP(2);
        finish(accU, cU, b2);
        request_consts(cU, 1);
        lds_barrier();
        clear(accU); multiply(accU, b2, wfB, wfA, 2);       // stage 2, upper branch     (W1b streams in)
        finish(accU, cU, b0);                               // h_up (x_up's planes are dead)
        request_consts(cU, 2);
        lds_barrier();
        MLP_STAMP(3);
        clear(accU); multiply(accU, b1, wfA, wfB, 3);       // stage 1, boundary branch  (W2b streams in)
        finish(accU, cU, b2);
        request_consts(cU, 3);
        lds_barrier();
        MLP_STAMP(4);
        clear(accU); multiply(accU, b2, wfB, wfA, 4);       // stage 2, boundary branch  (Wc, upper half)
        finish(accU, cU, b1);                               // h_b
        request_consts(cU, 4);
        lds_barrier();
        MLP_STAMP(5);
        clear(accU); multiply(accU, b0, wfA, wfB, 5);       // combine: Wc[:, :F] h_up   (Wc, boundary half) ...
        MLP_STAMP(6);
        multiply(accU, b1, wfB, wfA, -1);                   // ... + Wc[:, F:] h_b (cat order of mp/layers.py:199)
        MLP_STAMP(7);
        finish(accU, cU, nullptr);
        MLP_STAMP(8);
        return;
    }
    MLP_STAMP(0);
    request_rows(vU, D.x_up, D.ldx_up);
    request_consts(cU, 0);
    request_weight(wfA, 0);
    request_rows(vB, D.x_b, D.ldx_b);
    request_consts(cB, 2);
    stage_rows(vU, bufA);
    stage_rows(vB, bufC);
    lds_barrier();
    MLP_STAMP(1);
    clear(accU); multiply(accU, bufA, wfA, wfB, 2);     // stage 1, upper branch     (W1b streams in)
    MLP_STAMP(2);
    clear(accB); multiply(accB, bufC, wfB, wfA, 1);     // stage 1, boundary branch  (W2u streams in) ...
    finish(accU, cU, bufB);                             // ... over the epilogue of stage 1, upper branch
    request_consts(cU, 1);
    lds_barrier();
    MLP_STAMP(3);
    clear(accU); multiply(accU, bufB, wfA, wfB, 3);     // stage 2, upper branch     (W2b streams in)
    finish(accB, cB, bufD);
    request_consts(cB, 3);
    lds_barrier();
    MLP_STAMP(4);
    clear(accB); multiply(accB, bufD, wfB, wfA, 4);     // stage 2, boundary branch  (Wc, upper half)
    finish(accU, cU, bufU);                             // h_up
    request_consts(cU, 4);
    lds_barrier();
    MLP_STAMP(5);
    clear(accU); multiply(accU, bufU, wfA, wfB, 5);     // combine: Wc[:, :128] h_up (Wc, boundary half) ...
    finish(accB, cB, bufA);                             // h_b (x_up's planes are long dead)
    lds_barrier();
    MLP_STAMP(6);
    multiply(accU, bufA, wfB, wfA, -1);                 // ... + Wc[:, 128:] h_b (cat order of mp/layers.py:199)
    MLP_STAMP(7);
    finish(accU, cU, nullptr);
    MLP_STAMP(8);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

}  // namespace

#ifdef CWN_MLP_TIMING
extern "C" void cwn_mlp_debug_stamps(unsigned long long* buf) { g_mlp_stamps = buf; }
#endif

extern "C" int64_t cwn_update_mlp_max_rows(void) { return 1 << 20; }

extern "C" size_t cwn_update_mlp_packed_weight_bytes(int32_t F) { return (F == 64 || F == 128) ? (size_t)F * F * 6 : 0; }

namespace {

// fp32 [F, F] weight (row stride ldw) -> bf16 hi / mid / lo planes in MFMA-fragment order: the 1-KiB chunk number
// ((tile * KS + ks) * 3 + plane) holds, for lane l = kq * 16 + n, the eight k-values
// W[tile * 16 + n][ks * 32 + kq * 8 ..] of that plane (F = 128: the layout of cwn_gemm_pack_weights_f32).
template <int F>
__global__ __launch_bounds__(256) void pack_mlp_weights_kernel(const float* __restrict__ W, int64_t ldw,
                                                               unsigned char* __restrict__ out) {
    constexpr int KS = F / 32;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (tile, ks, lane)
    if (g >= (F / 16) * KS * 64) return;
    const int lane = g & 63, ks = (g >> 6) % KS, tile = (g >> 6) / KS;
    const float* src = W + (int64_t)(tile * 16 + (lane & 15)) * ldw + ks * 32 + (lane >> 4) * 8;
    const float4 a = make_float4(src[0], src[1], src[2], src[3]), b = make_float4(src[4], src[5], src[6], src[7]);
    uint4 ph, pm, pl;
    cwn::split8(a, b, ph, pm, pl);
    unsigned char* dst = out + ((size_t)(tile * KS + ks) * 3) * 1024 + lane * 16;
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + 1024) = pm;
    *reinterpret_cast<uint4*>(dst + 2048) = pl;
}

template <int F, int RT, bool SEQ, bool NARROW>
int launch_mlp(MlpBatch& B, int64_t blocks, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&update_mlp_kernel<F, RT, SEQ, NARROW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Shape<F, RT, SEQ>::kLdsBytes);
    });
    if (attr_err != hipSuccess) return CWN_ERR_LAUNCH;
#ifdef CWN_MLP_TIMING
    B.stamps = g_mlp_stamps;
#endif
    update_mlp_kernel<F, RT, SEQ, NARROW><<<dim3((unsigned)blocks), dim3(kThreads), Shape<F, RT, SEQ>::kLdsBytes, stream>>>(
        B.n, B.blk_start[1], B.blk_start[2], B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

// which schedule: CWN_MLP_FORM = 5 (alternating, five buffers) | 3 (sequential, three buffers, two per CU) | auto (default;
// read once): the sequential form for launches of more workgroups than the chip has CUs
int mlp_form_mode() {
    static const int mode = [] {
        const char* e = getenv("CWN_MLP_FORM");
        if (e == nullptr) return 0;
        return e[0] == '3' ? 3 : (e[0] == '5' ? 5 : 0);
    }();
    return mode;
}

}  // namespace

extern "C" int cwn_update_mlp_pack_weights_f32(const float* W, int64_t ldw, int32_t F, void* out, cwn_stream_t stream_) {
    if ((F != 64 && F != 128) || W == nullptr || out == nullptr || ldw < F) return CWN_ERR_BAD_ARG;
    if (((uintptr_t)W & 3u) || !al16(out)) return CWN_ERR_ALIGN;
    const int threads = (F / 16) * (F / 32) * 64;
    hipStream_t stream = (hipStream_t)stream_;
    if (F == 128) pack_mlp_weights_kernel<128><<<dim3((threads + 255) / 256), dim3(256), 0, stream>>>(W, ldw, (unsigned char*)out);
    else pack_mlp_weights_kernel<64><<<dim3((threads + 255) / 256), dim3(256), 0, stream>>>(W, ldw, (unsigned char*)out);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_update_mlp_f32(const cwn_mlp_dim* dims, int n_dims, int32_t F, cwn_stream_t stream_) {
    if (dims == nullptr || n_dims < 1 || n_dims > CWN_LAYER_MAX_DIMS || (F != 64 && F != 128)) return CWN_ERR_BAD_ARG;
    const int TM = 4096 / F;
    MlpBatch B{};
    B.n = n_dims;
    int64_t blocks = 0;
    for (int i = 0; i < n_dims; ++i) {
        const cwn_mlp_dim& D = dims[i];
        if (D.M < 0) return CWN_ERR_BAD_ARG;
        if (D.M > cwn_update_mlp_max_rows()) return CWN_ERR_TOO_LARGE;
        B.blk_start[i] = (int32_t)blocks;
        B.d[i] = D;
        if (D.M == 0) continue;
        if (D.x_up == nullptr || D.x_b == nullptr || D.y == nullptr) return CWN_ERR_BAD_ARG;
        if (D.in_width < 0 || D.in_width > F) return CWN_ERR_BAD_ARG;
        const bool narrow = D.in_width > 0 && D.in_width < F;
        if (D.ldy < F || D.ldy % 4) return CWN_ERR_BAD_ARG;
        if (narrow) {
            if (D.ldx_up < D.in_width || D.ldx_b < D.in_width) return CWN_ERR_BAD_ARG;
            if (((uintptr_t)D.x_up & 3u) || ((uintptr_t)D.x_b & 3u) || !al16(D.y)) return CWN_ERR_ALIGN;
        } else {
            if (D.ldx_up < F || D.ldx_b < F || D.ldx_up % 4 || D.ldx_b % 4) return CWN_ERR_BAD_ARG;
            if (!(al16(D.x_up) && al16(D.x_b) && al16(D.y))) return CWN_ERR_ALIGN;
        }
        for (int k = 0; k < 6; ++k) {
            if (D.w_packed[k] == nullptr) return CWN_ERR_BAD_ARG;
            if (!al16(D.w_packed[k])) return CWN_ERR_ALIGN;
        }
        for (int s = 0; s < 5; ++s) {
            if ((D.scale[s] == nullptr) != (D.shift[s] == nullptr)) return CWN_ERR_BAD_ARG;
            if (!(al16(D.bias[s]) && al16(D.scale[s]) && al16(D.shift[s]))) return CWN_ERR_ALIGN;
        }
        blocks += (D.M + TM - 1) / TM;
    }
    for (int i = n_dims; i <= CWN_LAYER_MAX_DIMS; ++i) B.blk_start[i] = (int32_t)blocks;
    if (blocks == 0) return CWN_OK;
    const int mode = mlp_form_mode();
    // more workgroups than the chip has CUs (one round of the alternating form): the sequential two-per-CU form
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    const bool seq = mode == 3 || (mode == 0 && blocks > n_cu);
    hipStream_t stream = (hipStream_t)stream_;
    bool narrow = false;
    for (int i = 0; i < n_dims; ++i) narrow = narrow || (dims[i].M > 0 && dims[i].in_width > 0 && dims[i].in_width < F);
    if (narrow) {
        if (F == 128) return seq ? launch_mlp<128, 2, true, true>(B, blocks, stream) : launch_mlp<128, 2, false, true>(B, blocks, stream);
        return seq ? launch_mlp<64, 2, true, true>(B, blocks, stream) : launch_mlp<64, 2, false, true>(B, blocks, stream);
    }
    if (F == 128) return seq ? launch_mlp<128, 2, true, false>(B, blocks, stream) : launch_mlp<128, 2, false, false>(B, blocks, stream);
    return seq ? launch_mlp<64, 2, true, false>(B, blocks, stream) : launch_mlp<64, 2, false, false>(B, blocks, stream);
}
