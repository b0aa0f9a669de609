// PROTOTYPE (not part of libcwn_hip.so): Y[M,128] = X[M,128] * W[128,128]^T in fp32 accuracy on the
// bf16 matrix pipe of gfx950.
//
// fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA rate, and
// is the largest single phase of the grouped GEMM (3.0 of 8.8 us at ZINC-128, ~150 of 249 us at
// batch 8192).  An fp32 number splits EXACTLY into three bf16 numbers by truncation
//     x = hi + mid + lo,   hi = top 8 significant bits, mid = the next 8, lo = the last 8
// (each subtraction is exact), so x*w = sum of 9 bf16 products, of which the three smallest
// (mid*lo, lo*mid, lo*lo: <= 2^-22 |x||w| each, ~2^-24 typically) are dropped: 6 v_mfma_f32_16x16x32_bf16 per 32 k-values
// instead of 8 v_mfma_f32_16x16x4_f32 -> 6*16 cycles against 8*32: 2.67x on the MFMA phase, error
// a few 2^-24 relative to |x||w| per term before fp32 accumulation (the accumulator is fp32 in both).
//
// Shape of the experiment: roles swapped as in the production kernel (A = W rows, B = X rows) so a
// lane ends up with 4 consecutive output columns of one X row; W split once per workgroup and kept
// in registers (96 VGPRs), the X tile split ONCE per element while it is staged into LDS (three
// bf16 planes, 272-B row stride: conflict-free ds_read_b128 fragments).  No prologue / epilogue /
// K-concat / double buffering: this measures the arithmetic and its accuracy, nothing else.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int K = 128, N = 128, TM = 64, kThreads = 256;
constexpr int kRowStride = K + 8;            // bf16 elements: 272 B, shifts rows by 4 banks

typedef __bf16 frag_ab __attribute__((ext_vector_type(8)));
typedef float frag_cd __attribute__((ext_vector_type(4)));

struct Split { uint32_t h, m, l; };          // bf16 bit patterns in the UPPER 16 bits

__device__ __forceinline__ Split split3(float x) {
    Split s;
    s.h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(s.h);
    s.m = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(s.m);
    s.l = __float_as_uint(r2) & 0xFFFF0000u;
    return s;
}

__device__ __forceinline__ uint32_t pack2(uint32_t even_hi16, uint32_t odd_hi16) {
    return (even_hi16 >> 16) | odd_hi16;     // element k in the low half, k+1 in the high half
}

// 8 consecutive fp32 -> three planes of 8 bf16
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& ph, uint4& pm, uint4& pl) {
    const Split s0 = split3(a.x), s1 = split3(a.y), s2 = split3(a.z), s3 = split3(a.w);
    const Split s4 = split3(b.x), s5 = split3(b.y), s6 = split3(b.z), s7 = split3(b.w);
    ph = make_uint4(pack2(s0.h, s1.h), pack2(s2.h, s3.h), pack2(s4.h, s5.h), pack2(s6.h, s7.h));
    pm = make_uint4(pack2(s0.m, s1.m), pack2(s2.m, s3.m), pack2(s4.m, s5.m), pack2(s6.m, s7.m));
    pl = make_uint4(pack2(s0.l, s1.l), pack2(s2.l, s3.l), pack2(s4.l, s5.l), pack2(s6.l, s7.l));
}

__device__ __forceinline__ frag_ab as_frag(const uint4& v) {
    return __builtin_bit_cast(frag_ab, v);
}

// flags: 1 = skip the MFMAs (timing split), 2 = only the hi*hi product (plain bf16 accuracy, for scale)
__global__ __launch_bounds__(kThreads) void gemm_bf16x3_kernel(const float* __restrict__ X,
                                                               const float* __restrict__ W,
                                                               float* __restrict__ Y, int64_t M, int flags) {
    __shared__ __attribute__((aligned(16))) uint16_t xs[3][TM][kRowStride];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;

    // stationary W fragments of this wave's 32 output columns: [col tile][k step][plane]
    uint4 wf[2][4][3];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int n = wave * 32 + ct * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4* p = reinterpret_cast<const float4*>(W + (int64_t)n * K + ks * 32 + kq * 8);
            split8(p[0], p[1], wf[ct][ks][0], wf[ct][ks][1], wf[ct][ks][2]);
        }
    }

    const int64_t tiles = (M + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row0 = tile * TM;
        // stage + split the X tile: 64 rows x 32 float4, 8 per thread, coalesced
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx >> 5, c4 = idx & 31;
            const int64_t row = row0 + r < M ? row0 + r : M - 1;      // clamp: no guarded loads
            v[i] = reinterpret_cast<const float4*>(X + row * K)[c4];
        }
        __syncthreads();                          // the previous tile's fragments are consumed
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx >> 5, c4 = idx & 31;
            const Split s0 = split3(v[i].x), s1 = split3(v[i].y), s2 = split3(v[i].z), s3 = split3(v[i].w);
            *reinterpret_cast<uint2*>(&xs[0][r][c4 * 4]) = make_uint2(pack2(s0.h, s1.h), pack2(s2.h, s3.h));
            *reinterpret_cast<uint2*>(&xs[1][r][c4 * 4]) = make_uint2(pack2(s0.m, s1.m), pack2(s2.m, s3.m));
            *reinterpret_cast<uint2*>(&xs[2][r][c4 * 4]) = make_uint2(pack2(s0.l, s1.l), pack2(s2.l, s3.l));
        }
        __syncthreads();

        frag_cd acc[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = frag_cd{0.f, 0.f, 0.f, 0.f};
        if (!(flags & 1)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const int r = rt * 16 + l15, k0 = ks * 32 + kq * 8;
                    const frag_ab xh = as_frag(*reinterpret_cast<const uint4*>(&xs[0][r][k0]));
                    const frag_ab xm = as_frag(*reinterpret_cast<const uint4*>(&xs[1][r][k0]));
                    const frag_ab xl = as_frag(*reinterpret_cast<const uint4*>(&xs[2][r][k0]));
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const frag_ab wh = as_frag(wf[ct][ks][0]), wm = as_frag(wf[ct][ks][1]),
                                      wl = as_frag(wf[ct][ks][2]);
                        frag_cd c = acc[rt][ct];
                        if (!(flags & 2)) {          // smallest terms first
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
                        acc[rt][ct] = c;
                    }
                }
            }
        }
        // D[i][j]: i = W row (output column) = (lane >> 4) * 4 + reg, j = X row = lane & 15
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int64_t row = row0 + rt * 16 + l15;
            if (row < M) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int n0 = wave * 32 + ct * 16 + kq * 4;
                    *reinterpret_cast<float4*>(Y + row * N + n0) =
                        make_float4(acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]);
                }
            }
        }
    }
}

}  // namespace

extern "C" int proto_gemm_bf16x3(const float* X, const float* W, float* Y, int64_t M, int flags,
                                 int max_blocks, void* stream) {
    if (M <= 0) return 0;
    const int64_t tiles = (M + TM - 1) / TM;
    const int blocks = (int)(tiles < max_blocks ? tiles : max_blocks);
    gemm_bf16x3_kernel<<<dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream>>>(X, W, Y, M, flags);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
