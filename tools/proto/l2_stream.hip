// Prototype: how fast can G workgroups of 512 threads each stream the SAME `bytes` out of L2 (the packed weights of
// update_mlp_kernel: 576 KB per workgroup)?  mode 0: every workgroup walks the buffer from 0; mode 1: workgroup b starts
// at chunk (b * 37) % n_chunks (the walks are spread over the buffer).  1-KiB loads per wave, 12 in flight.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/proto/l2_stream.hip -o tools/proto/libproto_l2.so
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace {
__global__ __launch_bounds__(512) void stream_kernel(const uint4* __restrict__ w, int64_t n_chunks /* of 8 KiB: one per workgroup step */,
                                                     int mode, uint4* out) {
    const int64_t start = mode == 1 ? ((int64_t)blockIdx.x * 37) % n_chunks : 0;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int64_t c0 = 0; c0 < n_chunks; c0 += 12) {
        uint4 v[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int64_t c = (start + c0 + q) % n_chunks;
            v[q] = w[c * 512 + threadIdx.x];          // 512 threads x 16 B = 8 KiB per chunk
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) { acc.x ^= v[q].x; acc.y ^= v[q].y; acc.z ^= v[q].z; acc.w ^= v[q].w; }
    }
    if (acc.x == 0x12345678u) out[blockIdx.x * 512 + threadIdx.x] = acc;
}
}  // namespace
extern "C" int proto_l2_stream(const void* w, int64_t bytes, int G, int mode, void* out, void* stream) {
    stream_kernel<<<dim3(G), dim3(512), 0, (hipStream_t)stream>>>((const uint4*)w, bytes / 8192, mode, (uint4*)out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
