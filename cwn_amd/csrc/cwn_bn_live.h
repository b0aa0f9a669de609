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
};

__device__ __forceinline__ void bn_live_request(const cwn_bn_live& L, int N, int n, BnLiveRegs& R) {
#pragma unroll
    for (int q = 0; q < CWN_BN_SLOTS; ++q) {
        R.t[q] = L.slots[(size_t)(2 * q) * N + n];
        R.u[q] = L.slots[(size_t)(2 * q + 1) * N + n];
    }
    R.g = L.gamma != nullptr ? L.gamma[n] : 1.0f;
    R.beta = L.beta != nullptr ? L.beta[n] : 0.0f;
}

__device__ __forceinline__ void bn_live_finish(const cwn_bn_live& L, const BnLiveRegs& R, int N, int64_t Mv, int n, bool writer,
                                               float& scale, float& shift) {
    double s = 0.0, sq = 0.0;
#pragma unroll
    for (int q = 0; q < CWN_BN_SLOTS; ++q) {
        s += R.t[q];
        sq += R.u[q];
    }
    const double invM = 1.0 / (double)(Mv > 0 ? Mv : 1);
    const double mean = s * invM;
    double var = sq * invM - mean * mean;     // biased, as BatchNorm normalises
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)L.eps));
    scale = R.g * rstd;
    shift = R.beta - (float)mean * scale;
    if (!writer) return;
    L.aff[n] = scale;
    L.aff[N + n] = shift;
    L.aff[2 * N + n] = (float)mean;
    L.aff[3 * N + n] = rstd;
    if (Mv < 1) return;
    if (L.num_batches_tracked != nullptr && n == 0) *L.num_batches_tracked += 1;
    if (L.running_mean != nullptr) {
        const float mom = L.momentum;
        const double unbiased = Mv > 1 ? var * ((double)Mv / (double)(Mv - 1)) : var;
        L.running_mean[n] = (1.0f - mom) * L.running_mean[n] + mom * (float)mean;
        L.running_var[n] = (1.0f - mom) * L.running_var[n] + mom * (float)unbiased;
    }
}

__device__ __forceinline__ void bn_live_column(const cwn_bn_live& L, int N, int64_t Mv, int n, bool writer, float& scale,
                                               float& shift) {
    BnLiveRegs R;
    bn_live_request(L, N, n, R);
    bn_live_finish(L, R, N, Mv, n, writer, scale, shift);
}

}  // namespace cwn
