// cwn_gemm_split.hip -- the grouped GEMM of cwn_gemm_f32 on the BF16 matrix pipe, at fp32 accuracy.
//
// v_mfma_f32_16x16x4_f32 (cwn_gemm.hip) runs at the fp32 VECTOR rate, 1/16 of the bf16 MFMA rate,
// and its MFMA phase is the largest single piece of the dense launches (3.0 of 8.8 us at ZINC-128,
// ~150 of 249 us at batch 8192).  An fp32 number splits EXACTLY into three bf16 numbers,
//     x = hi + mid + lo      (csrc/cwn_split.h: round-to-nearest at every step, every subtraction exact)
// so x * w is the sum of nine bf16 products of which the three smallest (mid*lo, lo*mid, lo*lo:
// <= 2^-26 |x||w| together) are dropped: six v_mfma_f32_16x16x32_bf16 per 32 k-values instead of
// eight v_mfma_f32_16x16x4_f32 -- 6 x 16 cycles against 8 x 32, 2.67x on the MFMA phase --
// accumulated in fp32 like the exact kernel.  Measured against float64: max error 1-4e-7 of
// |x|.|w|, the same as the fp32-MFMA kernel; 649 664 x 128 x 128: 289 -> 135 us (the launch becomes
// HBM-bound).  NOT bit-identical to an fmaf chain, and non-finite inputs give NaN where fp32 gives
// inf (inf - inf in the split): the per-call CWN_GEMM_EXACT flag (cwn_gemm_desc.flags)
// selects the exact kernel.
//
// Served here (cwn_gemm.hip routes, everything else stays on the exact kernel): every descriptor
// has N == 128, K == 128, K2 == 0, no prologue, no statistics, natural weight layout, 16-B
// aligned operands.  Bias, output affine and ReLU are applied in the epilogue.
//
// Mapping: operand roles swapped as in the exact kernel (A = W rows, B = X rows), so a lane ends up
// with 4 consecutive output columns of one X row.  W is split once per workgroup and stays in
// registers (2 column tiles x 4 k-steps x 3 planes x 4 VGPRs = 96); the 64-row X tile is split ONCE
// per element while it is staged into LDS as three bf16 planes (row stride 272 B: the 16 rows of a
// fragment read land on different banks, ds_read_b128 conflict-free).  Persistent workgroups, two
// per CU: one stages while the other multiplies.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cwn_hip.h"
#include "cwn_split.h"
#include "cwn_mem.h"

namespace {

constexpr int K = 128, N = 128, TM = 64, kThreads = 256;
constexpr int kRowStride = K + 8;            // bf16 elements per LDS row
constexpr int kChunksPerTile = 4 * 3;        // packed weight: 1-KiB chunks per 16-column tile (k steps x planes)
constexpr int kMaxBlocks = 512;              // 2 per CU (measured: 256 -> 204 us, 512 -> 135, one per tile -> 210)

using cwn::frag_cd;

struct SplitBatch {
    cwn_gemm_desc d[CWN_MAX_DESCS];
    int32_t blk_start[CWN_MAX_DESCS + 1];    // first workgroup of each descriptor
    int32_t n_tiles[CWN_MAX_DESCS];          // 64-row tiles
    int32_t n;
};

// MULTI: workgroups walk several tiles (more tiles than resident workgroups): the next tile's
// loads are issued right after the current one has been written to LDS and fly under its MFMAs
// (batch 8192 in tools/check_gemm_split.py: 199 -> 175 us).  Not for one-tile workgroups, where the
// same restructuring measured 8.7 -> 9.2 us on the ZINC-128 launch: that launch keeps the plain
// load -> split -> multiply order the compiler schedules best.
template <bool MULTI>
__global__ __launch_bounds__(kThreads, 2) void gemm_split_kernel(SplitBatch B) {
    __shared__ __attribute__((aligned(16))) uint16_t xs[3][TM][kRowStride];
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_gemm_desc D = B.d[di];         // by value (see cwn_aggregate.hip)
    const int blk = blockIdx.x - B.blk_start[di];
    const int nblk = B.blk_start[di + 1] - B.blk_start[di];
    int tiles = B.n_tiles[di];
    if (D.m_dev != nullptr) {                // (uniform) a static batch: the 64-row tiles below the rows that exist
        const int64_t mv = *D.m_dev;
        const int64_t live = (mv < 0 ? 0 : (mv < D.M ? mv : D.M));
        const int64_t t = (live + TM - 1) / TM;
        tiles = t < tiles ? (int)t : tiles;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;

    // stage: 64 rows x 32 float4, 8 per thread, row-contiguous; rows past M are clamped, not
    // guarded (guarded loads serialise; their outputs are never stored)
    float4 v[8];
    auto request_tile = [&](int tile) {
        const int64_t row0 = (int64_t)tile * TM;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx >> 5, c4 = idx & 31;
            const int64_t row = row0 + r < D.M ? row0 + r : D.M - 1;
            v[i] = reinterpret_cast<const float4*>(D.X + row * D.ldx)[c4];
        }
    };
    // stationary W fragments of this wave's 32 output columns: [column tile][k step][plane]
    uint4 wf[2][4][3];
    if (D.flags & CWN_GEMM_W_PACKED) {
        // the weight was split and laid out in fragment order once per weight version
        // (cwn_gemm_pack_weights_f32): 1 KiB contiguous per instruction and no split arithmetic, where the
        // fp32 form costs fragment-shaped loads (16 rows x 32 B per quarter wave) and ~320 VALU
        // instructions per wave in EVERY workgroup
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(D.W) + (size_t)wave * 2 * kChunksPerTile * 1024 + lane * 16;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    wf[ct][ks][pl] = *reinterpret_cast<const uint4*>(wp + ((ct * 4 + ks) * 3 + pl) * 1024);
    } else {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int n = wave * 32 + ct * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float4* p = reinterpret_cast<const float4*>(D.W + (int64_t)n * D.ldw + ks * 32 + kq * 8);
                cwn::split8(p[0], p[1], wf[ct][ks][0], wf[ct][ks][1], wf[ct][ks][2]);
            }
        }
    }
    const bool affine = D.out_scale != nullptr, relu = D.relu != 0;
    if constexpr (MULTI) {
        if (blk < tiles) request_tile(blk);
    }

    for (int tile = blk; tile < tiles; tile += nblk) {
        const int64_t row0 = (int64_t)tile * TM;
        if constexpr (!MULTI) request_tile(tile);
        __syncthreads();                     // the previous tile's fragments have been read
        // split the tile ONCE per element into the three bf16 planes
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + i * kThreads, r = idx >> 5, c4 = idx & 31;
            uint2 ph, pm, pl;
            cwn::split4(v[i], ph, pm, pl);
            *reinterpret_cast<uint2*>(&xs[0][r][c4 * 4]) = ph;
            *reinterpret_cast<uint2*>(&xs[1][r][c4 * 4]) = pm;
            *reinterpret_cast<uint2*>(&xs[2][r][c4 * 4]) = pl;
        }
        __syncthreads();
        if constexpr (MULTI) {
            if (tile + nblk < tiles) request_tile(tile + nblk);     // in flight under this tile's MFMAs
        }

        frag_cd acc[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = frag_cd{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const int r = rt * 16 + l15, k0 = ks * 32 + kq * 8;
                const uint4 xh = *reinterpret_cast<const uint4*>(&xs[0][r][k0]);
                const uint4 xm = *reinterpret_cast<const uint4*>(&xs[1][r][k0]);
                const uint4 xl = *reinterpret_cast<const uint4*>(&xs[2][r][k0]);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[rt][ct] = cwn::mfma_split6(wf[ct][ks][0], wf[ct][ks][1], wf[ct][ks][2], xh, xm, xl,
                                                   acc[rt][ct]);
            }
        }
        // D[i][j]: i = W row (output column) = (lane >> 4) * 4 + reg, j = X row = lane & 15
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int64_t row = row0 + rt * 16 + l15;
            if (row < D.M) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    // (epilogue constants are re-read per tile from L1/L2: held in registers across
                    //  the tile loop they cost 24 VGPRs and the second resident workgroup of the CU)
                    const int n0 = wave * 32 + ct * 16 + kq * 4;
                    float y[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};
                    if (D.bias != nullptr) {
                        const float4 b4 = *reinterpret_cast<const float4*>(D.bias + n0);
                        y[0] += b4.x; y[1] += b4.y; y[2] += b4.z; y[3] += b4.w;
                    }
                    if (affine) {
                        const float4 sc = *reinterpret_cast<const float4*>(D.out_scale + n0);
                        const float4 sh = *reinterpret_cast<const float4*>(D.out_shift + n0);
                        y[0] = y[0] * sc.x + sh.x;
                        y[1] = y[1] * sc.y + sh.y;
                        y[2] = y[2] * sc.z + sh.z;
                        y[3] = y[3] * sc.w + sh.w;
                    }
                    if (relu) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[q] = fmaxf(y[q], 0.0f);
                    }
                    cwn::store_result4(D.Y + row * D.ldy + n0, y[0], y[1], y[2], y[3]);
                }
            }
        }
    }
}

// fp32 [128, 128] weight (row stride ldw) -> bf16 hi / mid / lo planes in MFMA-fragment order: the 1-KiB chunk
// number ((tile * 4 + ks) * 3 + plane) holds, for lane l = kq * 16 + n, the eight k-values
// W[tile * 16 + n][ks * 32 + kq * 8 ..] of that plane.  One thread per (tile, ks, lane).
__global__ __launch_bounds__(256) void pack_gemm_weights_kernel(const float* __restrict__ W, int64_t ldw,
                                                                unsigned char* __restrict__ out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (N / 16) * 4 * 64) return;
    const int lane = g & 63, ks = (g >> 6) & 3, tile = g >> 8;
    const float* src = W + (int64_t)(tile * 16 + (lane & 15)) * ldw + ks * 32 + (lane >> 4) * 8;
    const float4 a = make_float4(src[0], src[1], src[2], src[3]), b = make_float4(src[4], src[5], src[6], src[7]);
    uint4 ph, pm, pl;
    cwn::split8(a, b, ph, pm, pl);
    unsigned char* dst = out + ((size_t)(tile * 4 + ks) * 3) * 1024 + lane * 16;
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + 1024) = pm;
    *reinterpret_cast<uint4*>(dst + 2048) = pl;
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

}  // namespace

extern "C" size_t cwn_gemm_packed_weight_bytes(void) { return (size_t)N * K * 6; }

extern "C" int cwn_gemm_pack_weights_f32(const float* W, int64_t ldw, void* out, cwn_stream_t stream_) {
    if (W == nullptr || out == nullptr || ldw < K) return CWN_ERR_BAD_ARG;
    if (((uintptr_t)W & 3u) || !al16(out)) return CWN_ERR_ALIGN;
    const int threads = (N / 16) * 4 * 64;
    pack_gemm_weights_kernel<<<dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream_>>>(W, ldw, (unsigned char*)out);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

// 1 when every descriptor fits this kernel (see the header comment); arguments already validated
// by cwn_gemm_f32.
int cwn_gemm_split_eligible(const cwn_gemm_desc* descs, int n) {
    for (int i = 0; i < n; ++i) {
        const cwn_gemm_desc& D = descs[i];
        if (D.N != N || D.K != K || D.K2 != 0 || D.w_trans != 0 || (D.flags >> 8) != 0) return 0;
        if (D.in_scale != nullptr || D.in_scale2 != nullptr || D.in_relu != 0 || D.col_sum != nullptr) return 0;
        if (!(al16(D.X) && al16(D.W) && al16(D.Y) && al16(D.bias) && al16(D.out_scale) && al16(D.out_shift)))
            return 0;
        if (D.ldx % 4 != 0 || (D.ldw % 4 != 0 && !(D.flags & CWN_GEMM_W_PACKED)) || D.ldy % 4 != 0) return 0;
        if ((D.M + TM - 1) / TM >= INT32_MAX) return 0;
    }
    return 1;
}

int cwn_gemm_split_launch(const cwn_gemm_desc* descs, int n, hipStream_t stream) {
    SplitBatch B{};
    B.n = n;
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
    if (total > blocks) gemm_split_kernel<true><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
    else gemm_split_kernel<false><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
