// cwn_collate.hip -- device-side batching of complexes resident in HBM.
//
// One launch fills every array of a ComplexBatch: workgroup (descriptor d, segment s) copies the
// s-th selected complex's slice of array d to its place in the batched array, adding that
// complex's running cell offset to index values.  Byte / integer work, HBM-bound; segments are
// tens to hundreds of elements, so a 64-thread workgroup per (array, complex) with coalesced
// element-wise accesses is the natural grain (B = 128 complexes x ~14 arrays = ~1.8 k workgroups).
#include <hip/hip_runtime.h>
#include "../../include/cwn_hip.h"

namespace {

constexpr int kThreads = 64;

struct CollateBatch {
    cwn_collate_desc d[CWN_MAX_COLLATE_DESCS];
    int32_t n;
};

__global__ __launch_bounds__(kThreads) void collate_kernel(CollateBatch B, int64_t n_seg) {
    const int di = blockIdx.y;
    const int64_t s = blockIdx.x;
    const cwn_collate_desc& D = B.d[di];
    const int64_t d0 = D.dst_start[s], len = D.dst_start[s + 1] - d0;
    if (len <= 0) return;
    const int64_t s0 = D.op == CWN_COLLATE_SEGID64 ? 0 : D.src_start[s];
    for (int r = 0; r < D.n_rows; ++r) {
        const int64_t add = D.add != nullptr ? D.add[(int64_t)r * n_seg + s] : 0;
        if (D.op == CWN_COLLATE_COPY32) {
            const int32_t* src = (const int32_t*)D.src + r * D.src_row_stride + s0;
            int32_t* dst = (int32_t*)D.dst + r * D.dst_row_stride + d0;
            for (int64_t q = threadIdx.x; q < len; q += kThreads) dst[q] = src[q];
        } else if (D.op == CWN_COLLATE_SEGID64) {
            int64_t* dst = (int64_t*)D.dst + r * D.dst_row_stride + d0;
            for (int64_t q = threadIdx.x; q < len; q += kThreads) dst[q] = s;
        } else {
            const int64_t* src = (const int64_t*)D.src + r * D.src_row_stride + s0;
            int64_t* dst = (int64_t*)D.dst + r * D.dst_row_stride + d0;
            const int64_t a = D.op == CWN_COLLATE_ADD64 ? add : 0;
            for (int64_t q = threadIdx.x; q < len; q += kThreads) dst[q] = src[q] + a;
        }
    }
}

}  // namespace

extern "C" int cwn_collate(const cwn_collate_desc* descs, int n, int64_t n_seg, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_COLLATE_DESCS || n_seg < 0) return CWN_ERR_BAD_ARG;
    if (n_seg == 0) return CWN_OK;
    if (n_seg >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    CollateBatch B{};
    B.n = n;
    for (int i = 0; i < n; ++i) {
        const cwn_collate_desc& D = descs[i];
        if (D.dst == nullptr || D.dst_start == nullptr || D.n_rows < 1 || D.n_rows > 2) return CWN_ERR_BAD_ARG;
        if (D.op < CWN_COLLATE_COPY32 || D.op > CWN_COLLATE_SEGID64) return CWN_ERR_BAD_ARG;
        if (D.op != CWN_COLLATE_SEGID64 && (D.src == nullptr || D.src_start == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.op == CWN_COLLATE_ADD64 && D.add == nullptr) return CWN_ERR_BAD_ARG;
        B.d[i] = D;
    }
    collate_kernel<<<dim3((unsigned)n_seg, (unsigned)n), dim3(kThreads), 0, (hipStream_t)stream_>>>(B, n_seg);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
