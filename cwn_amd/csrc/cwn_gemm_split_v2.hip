// cwn_gemm_split_v2.hip -- CANDIDATE successor of cwn_gemm_split.hip, NOT in the default build and
// not yet run on a GPU: the same kernel plus the pieces the training path and the input-gradient
// GEMMs need -- the BatchNorm(+ReLU) prologue on X, the per-32-row-band column statistics of the
// epilogue, and the transposed weight layout (w_trans).  `make -C cwn_amd/csrc v2` builds
// cwn_amd/libcwn_hip_v2.so with this file in place of cwn_gemm_split.hip; validate with
//     CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_v2.so python -m pytest tests -m gpu -q
//     CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_v2.so python tools/check_gemm_split.py
// (the suite's BatchNorm-statistics, prologue, w_trans and training tests then run through it) and
// promote it by renaming.  Everything below the dashed line is the text of cwn_gemm_split.hip.
// ------------------------------------------------------------------------------------------------
// the grouped GEMM of cwn_gemm_f32 on the BF16 matrix pipe, at fp32 accuracy.
//
// v_mfma_f32_16x16x4_f32 (cwn_gemm.hip) runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA rate,
// and its MFMA phase is the largest single piece of the dense launches (3.0 of 8.8 us at ZINC-128,
// ~150 of 249 us at batch 8192).  An fp32 number splits EXACTLY into three bf16 numbers by
// truncation,
//     x = hi + mid + lo      hi = its top 8 significant bits, mid = the next 8, lo = the last 8
// (every subtraction below is exact), so x * w is the sum of nine bf16 products of which the three
// smallest (mid*lo, lo*mid, lo*lo: <= 2^-22 |x||w| each, ~2^-24 typically) are dropped: six v_mfma_f32_16x16x32_bf16 per
// 32 k-values instead of eight v_mfma_f32_16x16x4_f32 -- 6 x 16 cycles against 8 x 32, 2.67x on
// the MFMA phase -- accumulated in fp32 like the exact kernel.  Measured against float64
// (tools/proto/run_gemm_bf16x3.py): max error 1.0-3.6e-7 of |x|.|w|, the same as the fp32-MFMA
// kernel (1.2-3.5e-7); 649 664 x 128 x 128: 289 -> 135 us (the launch becomes HBM-bound, 4.9 TB/s).
// NOT bit-identical to an fmaf chain, and non-finite inputs give NaN where fp32 gives inf
// (inf - inf in the split): cwn_gemm_set_split(0) / CWN_GEMM_SPLIT=0 select the exact kernel.
//
// Served here (cwn_gemm.hip routes, everything else stays on the exact kernel): every descriptor
// has N == 128, K == 128, K2 == 0 and 16-B aligned operands; v2 adds the X prologue, the band
// statistics and w_trans to bias / output affine / ReLU.
//
// Mapping: operand roles swapped as in the exact kernel (A = W rows, B = X rows), so a lane ends up
// with 4 consecutive output columns of one X row.  W is split once per workgroup and stays in
// registers (2 column tiles x 4 k-steps x 3 planes x 4 VGPRs = 96); the 64-row X tile is split ONCE
// per element while it is staged into LDS as three bf16 planes (row stride 272 B: the 16 rows of a
// fragment read land on different banks, ds_read_b128 conflict-free).  Persistent workgroups, two
// per CU: one stages while the other multiplies.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/cwn_hip.h"

namespace {

constexpr int K = 128, N = 128, kThreads = 256;
// rows per workgroup tile = 16 * RT: 64 (RT = 4) by default; 32 (RT = 2) when 64-row tiles would leave
// CUs idle (ZINC-128: 161 tiles of 64 rows on 256 CUs; 318 of 32 rows, two co-resident per CU) --
// CWN_SPLIT_TM32=1, to be measured (tools/check_gemm_split.py prints both).
constexpr int kRowStride = K + 8;            // bf16 elements per LDS row
constexpr int kMaxBlocks = 512;              // 2 per CU (measured: 256 -> 204 us, 512 -> 135, one per tile -> 210)

typedef __bf16 frag_ab __attribute__((ext_vector_type(8)));
typedef float frag_cd __attribute__((ext_vector_type(4)));

struct SplitBatch {
    cwn_gemm_desc d[CWN_MAX_DESCS];
    int32_t blk_start[CWN_MAX_DESCS + 1];    // first workgroup of each descriptor
    int32_t n_tiles[CWN_MAX_DESCS];          // 64-row tiles
    int32_t n;
};

struct Split { uint32_t h, m, l; };          // bf16 bit patterns in the UPPER 16 bits

__device__ __forceinline__ Split split3(float x) {
    Split s;
    s.h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(s.h);          // exact: the low 16 significant bits
    s.m = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(s.m);         // exact: at most 8 significant bits left
    s.l = __float_as_uint(r2) & 0xFFFF0000u;
    return s;
}

__device__ __forceinline__ uint32_t pack2(uint32_t even_hi16, uint32_t odd_hi16) {
    return (even_hi16 >> 16) | odd_hi16;     // element k in the low half, k + 1 in the high half
}

// 8 consecutive fp32 -> three planes of 8 bf16
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& ph, uint4& pm, uint4& pl) {
    const Split s0 = split3(a.x), s1 = split3(a.y), s2 = split3(a.z), s3 = split3(a.w);
    const Split s4 = split3(b.x), s5 = split3(b.y), s6 = split3(b.z), s7 = split3(b.w);
    ph = make_uint4(pack2(s0.h, s1.h), pack2(s2.h, s3.h), pack2(s4.h, s5.h), pack2(s6.h, s7.h));
    pm = make_uint4(pack2(s0.m, s1.m), pack2(s2.m, s3.m), pack2(s4.m, s5.m), pack2(s6.m, s7.m));
    pl = make_uint4(pack2(s0.l, s1.l), pack2(s2.l, s3.l), pack2(s4.l, s5.l), pack2(s6.l, s7.l));
}

__device__ __forceinline__ frag_ab as_frag(const uint4& v) { return __builtin_bit_cast(frag_ab, v); }

// MULTI: workgroups walk several tiles (more tiles than resident workgroups): the next tile's
// loads are issued right after the current one has been written to LDS and fly under its MFMAs
// (batch 8192 in tools/check_gemm_split.py: 199 -> 175 us).  Not for one-tile workgroups, where the
// same restructuring measured 8.7 -> 9.2 us on the ZINC-128 launch: that launch keeps the plain
// load -> split -> multiply order the compiler schedules best.
// PRO / STATS: the prologue and the statistics epilogue are separate instantiations -- as run-time
// branches they cost the plain kernel its registers (256 VGPRs + 36 B of scratch in the MULTI form).
template <int RT, bool MULTI, bool PRO, bool STATS>
__global__ __launch_bounds__(kThreads, 2) void gemm_split_kernel(SplitBatch B) {
    constexpr int TM = 16 * RT;              // rows per tile
    constexpr int U = TM * 32 / kThreads;    // float4 per thread of a staged tile (8 or 4)
    static_assert(RT == 4 || RT == 2, "tiles are one or two 32-row statistics bands");
    __shared__ __attribute__((aligned(16))) uint16_t xs[3][TM][kRowStride];
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_gemm_desc D = B.d[di];         // by value (see cwn_aggregate.hip)
    const int blk = blockIdx.x - B.blk_start[di];
    const int nblk = B.blk_start[di + 1] - B.blk_start[di];
    const int tiles = B.n_tiles[di];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;

    // stage: 64 rows x 32 float4, 8 per thread, row-contiguous; rows past M are clamped, not
    // guarded (guarded loads serialise; their outputs are never stored)
    float4 v[U];
    auto request_tile = [&](int tile) {
        const int64_t row0 = (int64_t)tile * TM;
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx >> 5, c4 = idx & 31;
            const int64_t row = row0 + r < D.M ? row0 + r : D.M - 1;
            v[i] = reinterpret_cast<const float4*>(D.X + row * D.ldx)[c4];
        }
    };
    // stationary W fragments of this wave's 32 output columns: [column tile][k step][plane]
    uint4 wf[2][4][3];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int n = wave * 32 + ct * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k0 = ks * 32 + kq * 8;
            float4 a, b;
            if (D.w_trans == 0) {            // W[n][k]: 8 consecutive k of row n
                const float4* p = reinterpret_cast<const float4*>(D.W + (int64_t)n * D.ldw + k0);
                a = p[0];
                b = p[1];
            } else {                         // W[k][n]: column n of 8 consecutive rows (16 lanes = 64 B each)
                const float* p = D.W + (int64_t)k0 * D.ldw + n;
                a = make_float4(p[0], p[D.ldw], p[2 * D.ldw], p[3 * D.ldw]);
                b = make_float4(p[4 * D.ldw], p[5 * D.ldw], p[6 * D.ldw], p[7 * D.ldw]);
            }
            split8(a, b, wf[ct][ks][0], wf[ct][ks][1], wf[ct][ks][2]);
        }
    }
    const bool affine = D.out_scale != nullptr, relu = D.relu != 0;
    // prologue (BatchNorm apply + ReLU of the producing stage) on this thread's 4 input columns:
    // c4 = idx & 31 = threadIdx.x & 31 for every i
    const bool pro_affine = PRO && D.in_scale != nullptr, pro_relu = PRO && (D.in_relu & 1) != 0;
    if constexpr (MULTI) {
        if (blk < tiles) request_tile(blk);
    }

    for (int tile = blk; tile < tiles; tile += nblk) {
        const int64_t row0 = (int64_t)tile * TM;
        if constexpr (!MULTI) request_tile(tile);
        __syncthreads();                     // the previous tile's fragments have been read
        // split the tile ONCE per element into the three bf16 planes (the prologue constants are
        // re-read per tile, like the epilogue's: 8 registers the MFMA loop needs more)
        float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pro_affine) {
            psc = reinterpret_cast<const float4*>(D.in_scale)[threadIdx.x & 31];
            psh = reinterpret_cast<const float4*>(D.in_shift)[threadIdx.x & 31];
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx >> 5, c4 = idx & 31;
            float4 x = v[i];
            if (pro_affine) {                // same operation order as the exact kernel's stage_store
                x.x = x.x * psc.x + psh.x; x.y = x.y * psc.y + psh.y;
                x.z = x.z * psc.z + psh.z; x.w = x.w * psc.w + psh.w;
            }
            if (pro_relu) {
                x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
            }
            const Split s0 = split3(x.x), s1 = split3(x.y), s2 = split3(x.z), s3 = split3(x.w);
            *reinterpret_cast<uint2*>(&xs[0][r][c4 * 4]) = make_uint2(pack2(s0.h, s1.h), pack2(s2.h, s3.h));
            *reinterpret_cast<uint2*>(&xs[1][r][c4 * 4]) = make_uint2(pack2(s0.m, s1.m), pack2(s2.m, s3.m));
            *reinterpret_cast<uint2*>(&xs[2][r][c4 * 4]) = make_uint2(pack2(s0.l, s1.l), pack2(s2.l, s3.l));
        }
        __syncthreads();
        if constexpr (MULTI) {
            if (tile + nblk < tiles) request_tile(tile + nblk);     // in flight under this tile's MFMAs
        }

        frag_cd acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = frag_cd{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int r = rt * 16 + l15, k0 = ks * 32 + kq * 8;
                const frag_ab xh = as_frag(*reinterpret_cast<const uint4*>(&xs[0][r][k0]));
                const frag_ab xm = as_frag(*reinterpret_cast<const uint4*>(&xs[1][r][k0]));
                const frag_ab xl = as_frag(*reinterpret_cast<const uint4*>(&xs[2][r][k0]));
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const frag_ab wh = as_frag(wf[ct][ks][0]), wm = as_frag(wf[ct][ks][1]),
                                  wl = as_frag(wf[ct][ks][2]);
                    frag_cd c = acc[rt][ct];             // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
                    acc[rt][ct] = c;
                }
            }
        }
        // D[i][j]: i = W row (output column) = (lane >> 4) * 4 + reg, j = X row = lane & 15
        // Epilogue per column tile: bias, statistics of the pre-normalisation value (fp64 partial
        // sums of each 32-row band = two row tiles, reduced over the 16 lanes that hold the band's
        // rows and stored with plain stores: [CWN_STAT_ROWS(M), N], no atomics, nothing to zero,
        // deterministic -- the exact kernel's layout), output affine, ReLU, store.
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            // (epilogue constants are re-read per tile from L1/L2: held in registers across the tile
            //  loop they cost 24 VGPRs and the second resident workgroup of the CU)
            const int n0 = wave * 32 + ct * 16 + kq * 4;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f),
                   sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (D.bias != nullptr) b4 = *reinterpret_cast<const float4*>(D.bias + n0);
            if (affine) {
                sc = *reinterpret_cast<const float4*>(D.out_scale + n0);
                sh = *reinterpret_cast<const float4*>(D.out_shift + n0);
            }
#pragma unroll
            for (int band = 0; band < RT / 2; ++band) {
                double csum[4] = {0., 0., 0., 0.}, csq[4] = {0., 0., 0., 0.};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rt = band * 2 + h;
                    const int64_t row = row0 + rt * 16 + l15;
                    float y[4] = {acc[rt][ct][0] + b4.x, acc[rt][ct][1] + b4.y, acc[rt][ct][2] + b4.z,
                                  acc[rt][ct][3] + b4.w};
                    if (STATS && D.col_sum != nullptr && row < D.M) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            csum[q] += (double)y[q];
                            csq[q] += (double)y[q] * (double)y[q];
                        }
                    }
                    if (affine) {
                        y[0] = y[0] * sc.x + sh.x;
                        y[1] = y[1] * sc.y + sh.y;
                        y[2] = y[2] * sc.z + sh.z;
                        y[3] = y[3] * sc.w + sh.w;
                    }
                    if (relu) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[q] = fmaxf(y[q], 0.0f);
                    }
                    if (row < D.M)
                        *reinterpret_cast<float4*>(D.Y + row * D.ldy + n0) = make_float4(y[0], y[1], y[2], y[3]);
                }
                if (STATS && D.col_sum != nullptr) {
                    const int64_t slot = (row0 >> 5) + band;        // tiles align to 32-row bands
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        double a = csum[q], b = csq[q];
#pragma unroll
                        for (int o = 8; o >= 1; o >>= 1) {           // over the 16 lanes l15 = 0..15 (same kq)
                            a += __shfl_xor(a, o, 16);
                            b += __shfl_xor(b, o, 16);
                        }
                        if (l15 == 0 && slot < CWN_STAT_ROWS(D.M)) {
                            D.col_sum[slot * N + n0 + q] = a;
                            D.col_sumsq[slot * N + n0 + q] = b;
                        }
                    }
                }
            }
        }
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

}  // namespace

// 1 when every descriptor fits this kernel (see the header comment); arguments already validated
// by cwn_gemm_f32.
int cwn_gemm_split_eligible(const cwn_gemm_desc* descs, int n) {
    for (int i = 0; i < n; ++i) {
        const cwn_gemm_desc& D = descs[i];
        if (D.N != N || D.K != K || D.K2 != 0 || D.reserved != 0) return 0;
        if (D.in_scale2 != nullptr || (D.in_relu & 2) != 0) return 0;
        if (!(al16(D.X) && al16(D.W) && al16(D.Y) && al16(D.bias) && al16(D.out_scale) && al16(D.out_shift) &&
              al16(D.in_scale) && al16(D.in_shift)))
            return 0;
        if (D.ldx % 4 != 0 || D.ldw % 4 != 0 || D.ldy % 4 != 0) return 0;
        if ((D.M + 31) / 32 >= INT32_MAX) return 0;
    }
    return 1;
}

int cwn_gemm_split_launch(const cwn_gemm_desc* descs, int n, hipStream_t stream) {
    SplitBatch B{};
    B.n = n;
    static const char* tm32_env = getenv("CWN_SPLIT_TM32");
    int64_t tiles64 = 0;
    for (int i = 0; i < n; ++i) tiles64 += (descs[i].M + 63) / 64;
    // 32-row tiles only for launches whose 64-row tiles do not fill the chip
    const bool tm32 = tm32_env != nullptr && tm32_env[0] == '1' && tiles64 < 256;
    const int TM = tm32 ? 32 : 64;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        B.d[i] = descs[i];
        B.n_tiles[i] = (int32_t)((descs[i].M + TM - 1) / TM);
        total += B.n_tiles[i];
    }
    if (total == 0) return CWN_OK;
    // persistent workgroups, shared between the descriptors in proportion to their tile counts and
    // NEVER more than the 2 x 256 that are resident at once: rounding the shares up gave 514 for the
    // four GEMMs of a batch-8192 layer, and the two workgroups that had to wait for a free slot
    // walked their 19 tiles after everyone else had finished (205 us instead of ~140)
    const int64_t cap = kMaxBlocks - n;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        int64_t nb = B.n_tiles[i];
        if (total > cap) {
            nb = nb * cap / total;
            if (nb < 1 && B.n_tiles[i] > 0) nb = 1;
        }
        B.blk_start[i] = (int32_t)blocks;
        blocks += nb;
    }
    for (int i = n; i <= CWN_MAX_DESCS; ++i) B.blk_start[i] = (int32_t)blocks;
    bool pro = false, stats = false;
    for (int i = 0; i < n; ++i) {
        pro = pro || descs[i].in_scale != nullptr || (descs[i].in_relu & 1) != 0;
        stats = stats || descs[i].col_sum != nullptr;
    }
    using Kern = void (*)(SplitBatch);
#define CWN_SPLIT_KERNS(RT)                                                                            \
    {{{gemm_split_kernel<RT, false, false, false>, gemm_split_kernel<RT, false, false, true>},         \
      {gemm_split_kernel<RT, false, true, false>, gemm_split_kernel<RT, false, true, true>}},          \
     {{gemm_split_kernel<RT, true, false, false>, gemm_split_kernel<RT, true, false, true>},           \
      {gemm_split_kernel<RT, true, true, false>, gemm_split_kernel<RT, true, true, true>}}}
    static const Kern kerns[2][2][2][2] = {CWN_SPLIT_KERNS(4), CWN_SPLIT_KERNS(2)};
#undef CWN_SPLIT_KERNS
    hipLaunchKernelGGL(kerns[tm32 ? 1 : 0][total > blocks ? 1 : 0][pro ? 1 : 0][stats ? 1 : 0],
                       dim3((unsigned)blocks), dim3(kThreads), 0, stream, B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
