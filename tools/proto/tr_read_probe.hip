#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(int* out, int stride_el) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 80];
    for (int i = threadIdx.x; i < 64 * 80; i += 64) lds[i] = (uint16_t)((i / stride_el) * 100 + (i % stride_el));   // value = row*100 + col
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    // 16-lane group g reads the 4x16 block rows g*4 .. g*4+3, cols 0..15: lane i supplies the address of [row g*4 + i/4][(i%4)*4]
    const uint16_t* p = lds + (g * 4 + i / 4) * stride_el + (i % 4) * 4;
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (int)(uint16_t)v[j];
}
int main() {
    int* d; hipMalloc(&d, 64 * 4 * 4);
    int h[256];
    for (int stride : {16, 72}) {
        probe<<<1, 64>>>(d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d\n", stride);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
