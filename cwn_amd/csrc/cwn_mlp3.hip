// cwn_mlp3.hip -- the update / combine networks of a CIN++ layer in ONE launch (inference).
//
// What it replaces (reference): mp/layers.py:252-260, CINppCochainConv.forward behind propagate()
//     out_up = update_up_nn(out_up); out_down = update_down_nn(out_down); out_boundaries = update_boundaries_nn(out_boundaries)
//     return combine_nn(torch.cat([out_up, out_down, out_boundaries], dim=-1))
// with update_*_nn = [Linear(F->F), BatchNorm, ReLU] x 2 (mp/layers.py:383-410) and combine_nn = [Linear(3F->F), BatchNorm,
// ReLU] (:411-414), BatchNorm in eval mode folded into a per-column affine.  Until round 6 the library ran this as two grouped
// GEMM launches over nine descriptors + torch.cat + the combine as torch modules (~0.13 ms of a 0.16-ms layer at the ZINC
// batch of 128: profiles/r5_*cinpp*).
//
// Same design as csrc/cwn_mlp.hip (the two-branch kernel of SparseCIN layers; read its header first): a workgroup of 8 waves
// takes 4096 / F rows of one cochain dimension through all seven Linear layers without leaving the CU, the activations of a
// stage go from the accumulators through the epilogue straight into the bf16 planes the next stage multiplies, the weights
// stream through two register sets one multiplication ahead.  What three branches change:
//   * the combine is accumulated BRANCH BY BRANCH: as soon as h_k = update_k(x_k) is in LDS its block of the combine product,
//     Wc[:, kF:(k+1)F] h_k, is added to a second accumulator set -- in the order of the cat (up, down, boundaries), i.e. the
//     k steps of a K = 3F product in their natural order -- so h_k's buffer is free for the next branch and THREE plane buffers
//     (78 KB at F = 128) serve three branches: two workgroups per CU, as the sequential form of the two-branch kernel;
//   * weight order (packed by cwn_update_mlp_pack_weights_f32): per branch k: W1_k, W2_k, Wc[:, kF:(k+1)F] -- the order the
//     launch multiplies in, so "the next weight" is always the next pointer.
//
//   x_up   --W1u--> relu(bn) --W2u--> relu(bn) = h_up --Wc[:, 0:F]----+
//   x_down --W1d--> relu(bn) --W2d--> relu(bn) = h_d  --Wc[:, F:2F]---+--> (+ bc) relu(bn) = y
//   x_b    --W1b--> relu(bn) --W2b--> relu(bn) = h_b  --Wc[:, 2F:3F]--+
//
// Arithmetic: the exact three-way bf16 split of csrc/cwn_split.h for every product (fp32 in, fp32 accumulate, fp32 out): the
// stages of the update networks are bit-identical to the grouped launches they replace (same split, MFMA order, epilogue).
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
constexpr int kBranches = 3;

template <int F> struct Shape3 {
    static constexpr int kRT = 2, kV = 2;
    static constexpr int kTM = kRT * 2048 / F;          // rows per workgroup: 32 (F = 128) / 64 (F = 64)
    static constexpr int kNCT = F / 16;
    static constexpr int kKS = F / 32;
    // as Shape<F, 2, SEQ = true> of cwn_mlp.hip: padded rows at 128, XOR-swizzled unpadded rows at 64 (three buffers of 64 padded
    // rows are 2 KB beyond half a CU)
    static constexpr bool kSwizzle = F == 64;
    static constexpr int kRowStride = kSwizzle ? F : F + 8;
    static __device__ __forceinline__ int col(int row, int c) {
        if constexpr (kSwizzle) return (((c >> 3) ^ ((row >> 1) & 7)) << 3) | (c & 7);
        else return c;
    }
    static constexpr int kChunksPerTile = kKS * 3;
    static constexpr size_t kPlaneElems = (size_t)kTM * kRowStride;
    static constexpr size_t kBufBytes = 3 * kPlaneElems * 2;
    static constexpr size_t kLdsBytes = 3 * kBufBytes;
    static_assert(kTM * (F / 4) == kV * kThreads && (kTM / 16) * kNCT == 8 * kRT, "tile shape");
    static_assert(2 * kLdsBytes <= 160 * 1024, "two workgroups per CU");
};

struct Mlp3Batch {
    cwn_mlp3_dim d[CWN_LAYER_MAX_DIMS];
    int32_t blk_start[CWN_LAYER_MAX_DIMS + 1];
    int32_t n;
};

// (second __launch_bounds__ argument on HIP: minimum WAVES per SIMD -- two resident 8-wave workgroups are four)
template <int F>
__global__ __launch_bounds__(kThreads, 4) void update_mlp3_kernel(Mlp3Batch B) {
    using S = Shape3<F>;
    constexpr int TM = S::kTM, kRowStride = S::kRowStride, kChunksPerTile = S::kChunksPerTile, kKS = S::kKS;
    constexpr int kRT = S::kRT, kV = S::kV;
    constexpr size_t kPlaneElems = S::kPlaneElems, kBufBytes = S::kBufBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* const P0 = reinterpret_cast<uint16_t*>(smem);
    uint16_t* const P1 = reinterpret_cast<uint16_t*>(smem + kBufBytes);
    uint16_t* const P2 = reinterpret_cast<uint16_t*>(smem + 2 * kBufBytes);
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_LAYER_MAX_DIMS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_mlp3_dim& D = B.d[di];
    const int64_t row0 = (int64_t)((int)blockIdx.x - B.blk_start[di]) * TM;
    const int64_t Mv = D.m_dev != nullptr ? *D.m_dev : D.M;      // rows that exist (D.M is then the capacity)
    if (row0 >= Mv) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ct = wave % S::kNCT, rt0 = (wave / S::kNCT) * kRT;
    const int l15 = lane & 15, kq = lane >> 4;

    typedef float4 RowRegs[kV];
    RowRegs vA;                          // ONE set of row registers: the next branch's rows are requested when the last were staged
    auto request_rows = [&](RowRegs& v, const float* X, int64_t ld) {
#pragma unroll
        for (int i = 0; i < kV; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx / (F / 4), c4 = idx % (F / 4);
            const int64_t row = row0 + r < D.M ? row0 + r : D.M - 1;       // rows past M are clamped, not guarded
            v[i] = reinterpret_cast<const float4*>(X + row * ld)[c4];
        }
    };
    auto stage_rows = [&](const RowRegs& v, uint16_t* buf) {
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
    auto lds_barrier = [&]() {          // orders LDS traffic only: the streaming weight stays in flight across it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    typedef uint4 WeightRegs[kKS][3];
    WeightRegs wfA, wfB;
    auto request_kstep = [&](WeightRegs& wf, int k, int ks) {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(D.w_packed[k]) +
                                  (size_t)ct * kChunksPerTile * 1024 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[ks][pl] = *reinterpret_cast<const uint4*>(wp + (ks * 3 + pl) * 1024);
    };
    typedef frag_cd AccRegs[kRT];
    AccRegs accS, accC;                 // the running stage / the combine, accumulated branch by branch
    auto clear = [&](AccRegs& acc) {
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) acc[rt] = frag_cd{0.f, 0.f, 0.f, 0.f};
    };
    // acc += buf x W^T; the k steps of weight `next` (>= 0) go into the OTHER register set between this product's MFMAs
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
    struct Consts { float4 b, sc, sh; bool affine; };
    Consts cS;
    auto request_consts = [&](Consts& c, int s) {       // BEFORE the MFMAs of the stage (and so before the next weight)
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

    // ---- the chain: weights 0 .. 8 in order, register sets alternating (weight k in set k % 2); every step ends in a barrier;
    // a buffer is overwritten at the earliest one barrier after its last read.
    //   step          reads      writes
    //   0             -          P0 = x_up
    //   1  W1u        P0         P1 = s1(up),   P2 = x_down
    //   2  W2u        P1         P0 = h_up
    //   3  Wc_u, W1d  P0, P2     C += .,        P1 = s1(down)
    //   4  W2d        P1         P2 = h_down,   P0 = x_b
    //   5  Wc_d, W1b  P2, P0     C += .,        P1 = s1(b)
    //   6  W2b        P1         P2 = h_b
    //   7  Wc_b       P2         C += . -> y
    request_rows(vA, D.x[0], D.ldx[0]);
    request_consts(cS, 0);
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) request_kstep(wfA, 0, ks);
    stage_rows(vA, P0);
    request_rows(vA, D.x[1], D.ldx[1]);                  //    x_down: in flight under step 1
    lds_barrier();
    clear(accS); multiply(accS, P0, wfA, wfB, 1);        // 1: W1u                 (W2u streams in)
    finish(accS, cS, P1);
    request_consts(cS, 1);
    stage_rows(vA, P2);                                  //    x_down
    lds_barrier();
    clear(accS); multiply(accS, P1, wfB, wfA, 2);        // 2: W2u                 (Wc_u)
    finish(accS, cS, P0);                                //    h_up
    request_consts(cS, 2);
    lds_barrier();
    clear(accC); multiply(accC, P0, wfA, wfB, 3);        // 3: C = Wc[:, 0:F] h_up (W1d)
    clear(accS); multiply(accS, P2, wfB, wfA, 4);        //    W1d                 (W2d)
    finish(accS, cS, P1);
    request_consts(cS, 3);
    lds_barrier();
    request_rows(vA, D.x[2], D.ldx[2]);                  //    x_b: in flight under step 4 (requested earlier its eight registers
                                                         //    are live across step 3, the widest one: four spilled at F = 128)
    clear(accS); multiply(accS, P1, wfA, wfB, 5);        // 4: W2d                 (Wc_d)
    finish(accS, cS, P2);                                //    h_down
    request_consts(cS, 4);
    stage_rows(vA, P0);                                  //    x_b
    lds_barrier();
    multiply(accC, P2, wfB, wfA, 6);                     // 5: C += Wc[:, F:2F] h_d (W1b)
    clear(accS); multiply(accS, P0, wfA, wfB, 7);        //    W1b                 (W2b)
    finish(accS, cS, P1);
    request_consts(cS, 5);
    lds_barrier();
    clear(accS); multiply(accS, P1, wfB, wfA, 8);        // 6: W2b                 (Wc_b)
    finish(accS, cS, P2);                                //    h_b
    request_consts(cS, 6);
    lds_barrier();
    multiply(accC, P2, wfA, wfB, -1);                    // 7: C += Wc[:, 2F:3F] h_b (cat order of mp/layers.py:260)
    finish(accC, cS, nullptr);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <int F>
int launch_mlp3(Mlp3Batch& B, int64_t blocks, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&update_mlp3_kernel<F>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Shape3<F>::kLdsBytes);
    });
    if (attr_err != hipSuccess) return CWN_ERR_LAUNCH;
    update_mlp3_kernel<F><<<dim3((unsigned)blocks), dim3(kThreads), Shape3<F>::kLdsBytes, stream>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

}  // namespace

extern "C" int cwn_update_mlp3_f32(const cwn_mlp3_dim* dims, int n_dims, int32_t F, cwn_stream_t stream_) {
    if (dims == nullptr || n_dims < 1 || n_dims > CWN_LAYER_MAX_DIMS || (F != 64 && F != 128)) return CWN_ERR_BAD_ARG;
    const int TM = 4096 / F;
    Mlp3Batch B{};
    B.n = n_dims;
    int64_t blocks = 0;
    for (int i = 0; i < n_dims; ++i) {
        const cwn_mlp3_dim& D = dims[i];
        if (D.M < 0) return CWN_ERR_BAD_ARG;
        if (D.M > cwn_update_mlp_max_rows()) return CWN_ERR_TOO_LARGE;
        B.blk_start[i] = (int32_t)blocks;
        B.d[i] = D;
        if (D.M == 0) continue;
        if (D.y == nullptr || D.ldy < F || D.ldy % 4) return CWN_ERR_BAD_ARG;
        if (!al16(D.y)) return CWN_ERR_ALIGN;
        for (int k = 0; k < kBranches; ++k) {
            if (D.x[k] == nullptr || D.ldx[k] < F || D.ldx[k] % 4) return CWN_ERR_BAD_ARG;
            if (!al16(D.x[k])) return CWN_ERR_ALIGN;
        }
        for (int k = 0; k < 3 * kBranches; ++k) {
            if (D.w_packed[k] == nullptr) return CWN_ERR_BAD_ARG;
            if (!al16(D.w_packed[k])) return CWN_ERR_ALIGN;
        }
        for (int s = 0; s < 2 * kBranches + 1; ++s) {
            if ((D.scale[s] == nullptr) != (D.shift[s] == nullptr)) return CWN_ERR_BAD_ARG;
            if (!(al16(D.bias[s]) && al16(D.scale[s]) && al16(D.shift[s]))) return CWN_ERR_ALIGN;
        }
        blocks += (D.M + TM - 1) / TM;
    }
    for (int i = n_dims; i <= CWN_LAYER_MAX_DIMS; ++i) B.blk_start[i] = (int32_t)blocks;
    if (blocks == 0) return CWN_OK;
    hipStream_t stream = (hipStream_t)stream_;
    return F == 128 ? launch_mlp3<128>(B, blocks, stream) : launch_mlp3<64>(B, blocks, stream);
}
