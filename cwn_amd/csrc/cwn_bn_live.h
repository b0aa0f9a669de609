// cwn_bn_live.h -- the consumer half of a cwn_bn_live record (include/cwn_hip.h): one column's batch statistics from the
// CWN_BN_SLOTS slot sums, in slot order, with cwn_bn_finalize_f32's arithmetic (fp64 mean / biased variance, fp32 affine).
// `writer` (the first workgroup of the consuming descriptor, one thread per column) also stores what the backward reads and
// updates the module's running statistics (torch semantics: unbiased variance, momentum; a batch without rows leaves them
// alone).  Shared by cwn_stage.hip (the next stage's prologue) and cwn_norm.hip (the activation of a layer's last stage).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/cwn_hip.h"

namespace cwn {

// the loads of one column (requested early: ahead of the consumer's own tile loads, which return behind them)
struct BnLiveRegs {
    double t[CWN_BN_SLOTS], u[CWN_BN_SLOTS];
    float g, beta;
    float rm, rv;       // (the writer) the running statistics it is about to update
    int64_t nbt;        // ... and the batch counter (column 0)
};

// `writer`: also the module's running statistics -- requested HERE, with everything else: behind the derive they were a
// second dependent round trip in the one workgroup that every launch then waited for (+1 us per consuming launch, measured)
__device__ __forceinline__ void bn_live_request(const cwn_bn_live& L, int N, int n, bool writer, BnLiveRegs& R) {
#pragma unroll
    for (int q = 0; q < CWN_BN_SLOTS; ++q) {
        R.t[q] = L.slots[(size_t)(2 * q) * N + n];
        R.u[q] = L.slots[(size_t)(2 * q + 1) * N + n];
    }
    R.g = L.gamma != nullptr ? L.gamma[n] : 1.0f;
    R.beta = L.beta != nullptr ? L.beta[n] : 0.0f;
    R.rm = R.rv = 0.f;
    R.nbt = 0;
    if (writer && L.running_mean != nullptr) {
        R.rm = L.running_mean[n];
        R.rv = L.running_var[n];
    }
    if (writer && n == 0 && L.num_batches_tracked != nullptr) R.nbt = *L.num_batches_tracked;
}

__device__ __forceinline__ void bn_live_finish(const cwn_bn_live& L, const BnLiveRegs& R, int N, int64_t Mv, int n, bool writer,
                                               float& scale, float& shift) {
    double s = 0.0, sq = 0.0;
#pragma unroll
    for (int q = 0; q < CWN_BN_SLOTS; ++q) {
        s += R.t[q];
        sq += R.u[q];
    }
    // 1 / M and 1 / sqrt(var + eps) in fp64 from fp32 seeds and two Newton steps each (relative error ~1e-7 -> 1e-14 -> 1e-28:
    // the last bit of a double) -- the library's fp64 division and square root are ~150 dependent instructions, 1.3 us of every
    // consuming launch (measured: CWN_STAGE_DBG); these are ~25
    const double Md = (double)(Mv > 0 ? Mv : 1);
    double invM = (double)(1.0f / (float)Md);
    invM = invM * (2.0 - Md * invM);
    invM = invM * (2.0 - Md * invM);
    const double mean = s * invM;
    double var = sq * invM - mean * mean;     // biased, as BatchNorm normalises
    var = var > 0.0 ? var : 0.0;
    const double x = var + (double)L.eps;
    double y = (double)rsqrtf((float)x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    const float rstd = (float)y;
    scale = R.g * rstd;
    shift = R.beta - (float)mean * scale;
    if (!writer) return;
    L.aff[n] = scale;
    L.aff[N + n] = shift;
    L.aff[2 * N + n] = (float)mean;
    L.aff[3 * N + n] = rstd;
    if (Mv < 1) return;
    if (L.num_batches_tracked != nullptr && n == 0) *L.num_batches_tracked = R.nbt + 1;
    if (L.running_mean != nullptr) {
        const float mom = L.momentum;
        const float unbiased = Mv > 1 ? (float)var * ((float)Mv / (float)(Mv - 1)) : (float)var;
        L.running_mean[n] = (1.0f - mom) * R.rm + mom * (float)mean;
        L.running_var[n] = (1.0f - mom) * R.rv + mom * unbiased;
    }
}

__device__ __forceinline__ void bn_live_column(const cwn_bn_live& L, int N, int64_t Mv, int n, bool writer, float& scale,
                                               float& shift) {
    BnLiveRegs R;
    bn_live_request(L, N, n, writer, R);
    bn_live_finish(L, R, N, Mv, n, writer, scale, shift);
}

}  // namespace cwn
