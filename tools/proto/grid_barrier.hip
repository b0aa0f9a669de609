// Prototype: what does a grid-wide barrier cost inside one launch on MI355X?  G workgroups of 512 threads, each does a
// little work (one 16-KB read) between NB barriers on global counters (bounded spin: never hangs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/proto/grid_barrier.hip -o tools/proto/libproto_bar.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr unsigned kSpinLimit = 1u << 22;

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) { ok = false; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(512) void bar_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned* counters,
                                                  int nb, unsigned epoch, int* err) {
    float acc = 0.f;
    const int G = gridDim.x;
    for (int b = 0; b < nb; ++b) {
        // a little work: every workgroup reads 16 KB that ANOTHER workgroup wrote in the previous phase
        const int src = (blockIdx.x + b + 1) % G;
        const float4 v = reinterpret_cast<const float4*>(b == 0 ? x : y)[(size_t)src * 1024 + threadIdx.x * 2 % 1024];
        acc += v.x + v.y + v.z + v.w;
        reinterpret_cast<float4*>(y)[(size_t)blockIdx.x * 1024 + threadIdx.x * 2 % 1024] = make_float4(acc, acc, acc, acc);
        if (!grid_barrier(counters + b, epoch * (unsigned)G)) {
            if (threadIdx.x == 0) atomicExch(err, 1);
            return;
        }
    }
    if (threadIdx.x == 0) y[(size_t)blockIdx.x * 4096] = acc;
}
}  // namespace

// counters are monotone: launch number `epoch` (1-based) waits for epoch * G arrivals, so nothing is ever reset
extern "C" int proto_grid_barrier(const float* x, float* y, unsigned* counters, int G, int nb, unsigned epoch, int* err,
                                  void* stream) {
    bar_kernel<<<dim3(G), dim3(512), 0, (hipStream_t)stream>>>(x, y, counters, nb, epoch, err);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
