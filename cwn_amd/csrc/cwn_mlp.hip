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
//   x_up --W1u--> relu(bn) --W2u--> relu(bn) = h_up \
//                                                     Wc[:, :128] h_up + Wc[:, 128:] h_b --> relu(bn) = y
//   x_b  --W1b--> relu(bn) --W2b--> relu(bn) = h_b  /
//
// For launches of at most a few thousand rows per dimension (every workgroup streams all six weights:
// 576 KB out of L2); larger ones keep the weight-stationary grouped GEMMs (the caller decides:
// cwn_update_mlp_max_rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/cwn_hip.h"
#include "cwn_split.h"
#include "cwn_mem.h"

namespace {

using cwn::frag_cd;

constexpr int K = 128, N = 128, TM = 32, kThreads = 256;
constexpr int kRowStride = K + 8;                 // bf16 elements per LDS row (272 B: fragment reads conflict-free)
constexpr int kV = TM * 32 / kThreads;            // float4 of an input tile per thread
constexpr int kRT = TM / 16;                      // 16-row tiles per workgroup
constexpr int kChunksPerTile = 4 * 3;             // packed weight: 1-KiB chunks per 16-column tile (k steps x planes)
constexpr size_t kPlaneElems = (size_t)TM * kRowStride;
constexpr size_t kBufBytes = 3 * kPlaneElems * 2;  // three planes
constexpr size_t kLdsBytes = 3 * kBufBytes;        // ping, pong, kept h_up

struct MlpBatch {
    cwn_mlp_dim d[CWN_LAYER_MAX_DIMS];
    int32_t blk_start[CWN_LAYER_MAX_DIMS + 1];
    int32_t n;
};

__global__ __launch_bounds__(kThreads, 2) void update_mlp_kernel(MlpBatch B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* const bufA = reinterpret_cast<uint16_t*>(smem);
    uint16_t* const bufB = reinterpret_cast<uint16_t*>(smem + kBufBytes);
    uint16_t* const bufU = reinterpret_cast<uint16_t*>(smem + 2 * kBufBytes);
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_LAYER_MAX_DIMS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_mlp_dim& D = B.d[di];
    const int64_t row0 = (int64_t)((int)blockIdx.x - B.blk_start[di]) * TM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;

    // an input tile: TM rows x 32 float4, kV per thread, row-contiguous; rows past M are clamped, not guarded
    float4 v[kV];
    auto request_rows = [&](const float* X, int64_t ld) {
#pragma unroll
        for (int i = 0; i < kV; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx >> 5, c4 = idx & 31;
            const int64_t row = row0 + r < D.M ? row0 + r : D.M - 1;
            v[i] = reinterpret_cast<const float4*>(X + row * ld)[c4];
        }
    };
    auto stage_rows = [&](uint16_t* buf) {          // split the tile ONCE per element into the three planes
#pragma unroll
        for (int i = 0; i < kV; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx >> 5, c4 = idx & 31;
            uint2 ph, pm, pl;
            cwn::split4(v[i], ph, pm, pl);
            uint16_t* dst = buf + (size_t)r * kRowStride + c4 * 4;
            *reinterpret_cast<uint2*>(dst) = ph;
            *reinterpret_cast<uint2*>(dst + kPlaneElems) = pm;
            *reinterpret_cast<uint2*>(dst + 2 * kPlaneElems) = pl;
        }
    };
    // the stationary operand of a stage: this wave's 32 output columns, [column tile][k step][plane]
    uint4 wf[2][4][3];
    auto request_weight = [&](int k) {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(D.w_packed[k]) +
                                  (size_t)wave * 2 * kChunksPerTile * 1024 + lane * 16;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    wf[ct][ks][pl] = *reinterpret_cast<const uint4*>(wp + ((ct * 4 + ks) * 3 + pl) * 1024);
    };
    frag_cd acc[kRT][2];
    auto clear = [&]() {
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = frag_cd{0.f, 0.f, 0.f, 0.f};
    };
    auto multiply = [&](const uint16_t* buf) {      // acc += buf x W^T: same loop order as cwn_gemm_split.hip
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int rt = 0; rt < kRT; ++rt) {
                const uint16_t* p = buf + (size_t)(rt * 16 + l15) * kRowStride + ks * 32 + kq * 8;
                const uint4 xh = *reinterpret_cast<const uint4*>(p);
                const uint4 xm = *reinterpret_cast<const uint4*>(p + kPlaneElems);
                const uint4 xl = *reinterpret_cast<const uint4*>(p + 2 * kPlaneElems);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[rt][ct] = cwn::mfma_split6(wf[ct][ks][0], wf[ct][ks][1], wf[ct][ks][2], xh, xm, xl, acc[rt][ct]);
            }
        }
    };
    // epilogue of stage s: + bias, folded BatchNorm, ReLU; then either into the planes of `buf` (the next
    // stage's operand) or, for the last stage, to y.  D[i][j]: i = output column (lane >> 4) * 4 + reg,
    // j = row (lane & 15): a lane holds 4 consecutive columns of one row.
    auto finish = [&](int s, uint16_t* buf) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int n0 = wave * 32 + ct * 16 + kq * 4;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = b4;
            if (D.bias[s] != nullptr) b4 = *reinterpret_cast<const float4*>(D.bias[s] + n0);
            const bool affine = D.scale[s] != nullptr;
            if (affine) {
                sc = *reinterpret_cast<const float4*>(D.scale[s] + n0);
                sh = *reinterpret_cast<const float4*>(D.shift[s] + n0);
            }
#pragma unroll
            for (int rt = 0; rt < kRT; ++rt) {
                float y[4] = {acc[rt][ct][0] + b4.x, acc[rt][ct][1] + b4.y, acc[rt][ct][2] + b4.z, acc[rt][ct][3] + b4.w};
                if (affine) {
                    y[0] = y[0] * sc.x + sh.x;
                    y[1] = y[1] * sc.y + sh.y;
                    y[2] = y[2] * sc.z + sh.z;
                    y[3] = y[3] * sc.w + sh.w;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) y[q] = fmaxf(y[q], 0.0f);
                const int r = rt * 16 + l15;
                if (buf != nullptr) {
                    uint2 ph, pm, pl;
                    cwn::split4(make_float4(y[0], y[1], y[2], y[3]), ph, pm, pl);
                    uint16_t* dst = buf + (size_t)r * kRowStride + n0;
                    *reinterpret_cast<uint2*>(dst) = ph;
                    *reinterpret_cast<uint2*>(dst + kPlaneElems) = pm;
                    *reinterpret_cast<uint2*>(dst + 2 * kPlaneElems) = pl;
                } else if (row0 + r < D.M) {
                    cwn::store_result4(D.y + (row0 + r) * D.ldy + n0, y[0], y[1], y[2], y[3]);
                }
            }
        }
    };

    // ---- the chain; a stage's weight is requested as soon as the previous stage's MFMAs have been issued ----
    request_rows(D.x_up, D.ldx_up);
    request_weight(0);
    stage_rows(bufA);
    __syncthreads();
    clear(); multiply(bufA);                 // stage 1, upper branch
    request_weight(1);
    request_rows(D.x_b, D.ldx_b);            // the boundary branch's rows fly under two stages
    finish(0, bufB);
    __syncthreads();
    clear(); multiply(bufB);                 // stage 2, upper branch
    request_weight(2);
    finish(1, bufU);                         // h_up stays in LDS until the combine
    stage_rows(bufA);                        // (bufA was last read before the previous barrier)
    __syncthreads();
    clear(); multiply(bufA);                 // stage 1, boundary branch
    request_weight(3);
    finish(2, bufB);
    __syncthreads();
    clear(); multiply(bufB);                 // stage 2, boundary branch
    request_weight(4);
    finish(3, bufA);                         // h_b
    __syncthreads();
    clear(); multiply(bufU);                 // combine: Wc[:, :128] h_up ...
    request_weight(5);
    multiply(bufA);                          // ... + Wc[:, 128:] h_b (cat order of mp/layers.py:199)
    finish(4, nullptr);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

}  // namespace

extern "C" int64_t cwn_update_mlp_max_rows(void) { return 32768; }

extern "C" int cwn_update_mlp_f32(const cwn_mlp_dim* dims, int n_dims, cwn_stream_t stream_) {
    if (dims == nullptr || n_dims < 1 || n_dims > CWN_LAYER_MAX_DIMS) return CWN_ERR_BAD_ARG;
    MlpBatch B{};
    B.n = n_dims;
    int64_t blocks = 0, rows = 0;
    for (int i = 0; i < n_dims; ++i) {
        const cwn_mlp_dim& D = dims[i];
        if (D.M < 0) return CWN_ERR_BAD_ARG;
        B.blk_start[i] = (int32_t)blocks;
        B.d[i] = D;
        if (D.M == 0) continue;
        if (D.x_up == nullptr || D.x_b == nullptr || D.y == nullptr) return CWN_ERR_BAD_ARG;
        if (D.ldx_up < K || D.ldx_b < K || D.ldy < N || D.ldx_up % 4 || D.ldx_b % 4 || D.ldy % 4) return CWN_ERR_BAD_ARG;
        if (!(al16(D.x_up) && al16(D.x_b) && al16(D.y))) return CWN_ERR_ALIGN;
        for (int k = 0; k < 6; ++k) {
            if (D.w_packed[k] == nullptr) return CWN_ERR_BAD_ARG;
            if (!al16(D.w_packed[k])) return CWN_ERR_ALIGN;
        }
        for (int s = 0; s < 5; ++s) {
            if ((D.scale[s] == nullptr) != (D.shift[s] == nullptr)) return CWN_ERR_BAD_ARG;
            if (!(al16(D.bias[s]) && al16(D.scale[s]) && al16(D.shift[s]))) return CWN_ERR_ALIGN;
        }
        blocks += (D.M + TM - 1) / TM;
        rows += D.M;
    }
    for (int i = n_dims; i <= CWN_LAYER_MAX_DIMS; ++i) B.blk_start[i] = (int32_t)blocks;
    if (rows > cwn_update_mlp_max_rows() * CWN_LAYER_MAX_DIMS) return CWN_ERR_TOO_LARGE;
    if (blocks == 0) return CWN_OK;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&update_mlp_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    });
    if (attr_err != hipSuccess) return CWN_ERR_LAUNCH;
    update_mlp_kernel<<<dim3((unsigned)blocks), dim3(kThreads), kLdsBytes, (hipStream_t)stream_>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
