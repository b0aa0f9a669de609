// cwn_gemm_tn.hip -- weight gradients of the dense layers on the gfx950 matrix cores:
//
//     dW[n, k] += sum_m dZ[m, n] * prologue([X | X2])[m, k]        db[n] += sum_m dZ[m, n]
//
// (the backward of torch.nn.Linear inside msg_up_nn / update_up_nn / update_boundaries_nn /
// combine_nn, mp/layers.py:290-325).  The reduction runs over the M cells of a dimension
// (thousands) while the output is only hidden x hidden, the opposite shape of cwn_gemm_f32:
//   * a workgroup owns a 64 x 64 tile of dW and a band of 128 rows of M; its four waves hold
//     32 x 32 each (2 x 2 v_mfma_f32_16x16x4_f32 tiles, 16 accumulator VGPRs);
//   * both operands are read ROW-CONTIGUOUSLY from global memory ([m][n] and [m][k] tiles of 64
//     rows, 16-B loads, next chunk in flight during the MFMAs), staged in LDS with an 80-float row
//     stride, and the MFMA fragments are 4-B reads of (row 4s + g, column j): lanes j walk
//     consecutive words and the four lane groups g land 16 banks apart -- conflict-free without a
//     transposition, because the reduction index m is the ROW of both tiles;
//   * the bands are combined either by a second tiny launch that sums per-band partial tiles in
//     band order into dW (workspace given: deterministic, no atomics) or with fp32 atomics
//     straight into dW (no workspace; ~27 same-address adds per element at M ~3 k);
//   * the normalisation + ReLU of the producing layer is applied to X on the way into LDS (the
//     activation itself is never materialised in the forward pass);
//   * up to CWN_GEMM_TN_MAX_DESCS (24) weight gradients per launch: a training step defers them to the end of its backward
//     (nothing reads a weight gradient before the optimizer) and runs the ~100 of a 4-layer model in 5 launches that
//     fill the chip, where 17 launches of ~440 workgroups each spent 18 us apiece on their own latency chain.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "../../include/cwn_hip.h"
#include "cwn_split.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kTile = 64;        // rows of dW (n) and columns of dW (k) per workgroup
constexpr int kChunk = 64;       // rows of M per LDS stage
constexpr int kBandRows = 128;   // rows of M per workgroup at least; the launcher picks a multiple of kChunk from here up (see there)
constexpr int kMaxBandRows = 1024;
constexpr int kLd = 80;          // LDS row stride in floats
constexpr int kU = kChunk * (kTile / 4) / kThreads;   // 16-B loads per thread per tile (= 4)
constexpr double kTnFixed2 = 4.0; // split kernel: prologue + epilogue of a workgroup in 64-row chunk-times (band choice below)

struct TnBatch {       // a kernel argument: must stay under 4 KiB (static_assert below)
    cwn_gemm_tn_desc d[CWN_GEMM_TN_MAX_DESCS];
    int32_t blk_start[CWN_GEMM_TN_MAX_DESCS + 1];
    int16_t tiles_n[CWN_GEMM_TN_MAX_DESCS], tiles_k[CWN_GEMM_TN_MAX_DESCS];
    int32_t band_rows_of[CWN_GEMM_TN_MAX_DESCS];   // rows of M per workgroup of descriptor i (multiple of the kernel's chunk)
    int32_t n;
    float* ws[CWN_GEMM_TN_MAX_DESCS];      // per-band partials [bands][N][K + K2] then [bands][N] (db), or NULL
    int32_t bands[CWN_GEMM_TN_MAX_DESCS];
    int32_t dbg;       // timing experiments (CWN_TN_DBG): 1 no output, 2 no MFMA, 4 no bias sum
};

static_assert(sizeof(TnBatch) <= 4096, "kernel-argument segment");

struct Src {                 // one logical [M, C1 + C2] operand made of up to two matrices
    const float* p1;
    const float* p2;
    int64_t ld1, ld2;
    int c1, c2;
    const float* scale1;     // prologue, or NULL
    const float* shift1;
    const float* scale2;
    const float* shift2;
    int relu;                // bit 0: first matrix, bit 1: second
};

template <bool FAST, int TW = kTile>
__device__ __forceinline__ void tile_load(f32x4 (&v)[kU], const Src& S, int col0, int64_t row0,
                                          int64_t row_hi) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        const int q = u * kThreads + threadIdx.x;
        const int r = q / (TW / 4), c = q % (TW / 4);
        const int64_t row = row0 + r < row_hi ? row0 + r : row_hi - 1;   // clamped, never faults
        const int col = col0 + 4 * c;
        const bool second = S.c2 > 0 && col >= S.c1;
        const float* base = second ? S.p2 : S.p1;
        const int64_t ld = second ? S.ld2 : S.ld1;
        const int cc = second ? col - S.c1 : col;
        const int cmax = second ? S.c2 : S.c1;
        if constexpr (FAST) {
            v[u] = *reinterpret_cast<const f32x4*>(base + row * ld + (cc < cmax ? cc : 0));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[u][t] = base[row * ld + (cc + t < cmax ? cc + t : 0)];
        }
    }
}

// The prologue constants of a thread: q = u * kThreads + tid walks rows only (kThreads is a multiple of kTile / 4), so its
// four columns are the same in every tile of the band -- loaded ONCE per workgroup (inside tile_store they were global loads
// in front of every LDS store: 23 of the 71 us of the merged ZINC-128 launch, tools/ubench_tn24.py).
struct Pro {
    f32x4 sc, sh;
    bool affine, relu;
    int cc, cmax;
    bool second;     // (the split kernel) the thread's columns lie in the second matrix of the operand
};

// the prologue constants of a thread whose columns (second, cc, cmax) are set
__device__ __forceinline__ void fill_pro(Pro& P, const Src& S) {
    const float* sc = P.second ? S.scale2 : S.scale1;
    const float* sh = P.second ? S.shift2 : S.shift1;
    P.relu = (S.relu & (P.second ? 2 : 1)) != 0;
    P.affine = sc != nullptr;
    P.sc = (f32x4){1.f, 1.f, 1.f, 1.f};
    P.sh = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (P.affine) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (P.cc + t < P.cmax) {
                P.sc[t] = sc[P.cc + t];
                P.sh[t] = sh[P.cc + t];
            }
    }
}

template <int TW = kTile>
__device__ __forceinline__ Pro make_pro(const Src& S, int col0) {
    static_assert(kThreads % (TW / 4) == 0, "a thread keeps its columns");
    const int c = threadIdx.x % (TW / 4);
    const int col = col0 + 4 * c;
    const bool second = S.c2 > 0 && col >= S.c1;
    Pro P;
    P.cc = second ? col - S.c1 : col;
    P.cmax = second ? S.c2 : S.c1;
    const float* sc = second ? S.scale2 : S.scale1;
    const float* sh = second ? S.shift2 : S.shift1;
    P.relu = (S.relu & (second ? 2 : 1)) != 0;
    P.affine = sc != nullptr;
    P.sc = (f32x4){1.f, 1.f, 1.f, 1.f};
    P.sh = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (P.affine) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (P.cc + t < P.cmax) {
                P.sc[t] = sc[P.cc + t];
                P.sh[t] = sh[P.cc + t];
            }
    }
    return P;
}

__device__ __forceinline__ void tile_store(float* lds, const f32x4 (&v)[kU], const Pro& P, int64_t row0, int64_t row_hi) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        const int q = u * kThreads + threadIdx.x;
        const int r = q / (kTile / 4), c = q % (kTile / 4);
        const bool row_ok = row0 + r < row_hi;
        f32x4 x = v[u];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float y = x[t];
            const bool ok = row_ok && P.cc + t < P.cmax;
            if (P.affine) y = y * P.sc[t] + P.sh[t];
            if (P.relu) y = fmaxf(y, 0.f);
            x[t] = ok ? y : 0.f;      // rows past the band and columns past the matrix add nothing
        }
        *reinterpret_cast<f32x4*>(lds + r * kLd + 4 * c) = x;
    }
}

// Rows of M per band.  The host cuts a descriptor into `bands` bands of equal length over D.M; with a device-side row count
// (a static batch: D.M is the CAPACITY of the buffers) the batch's own rows would all fall into the first bands -- the others
// idle, the launch as long as a full-capacity one (REDDIT-32, capacity 1.46 x the batch: 87 against 67 us, round 6).  The same
// number of bands over the rows that exist instead: shorter bands, every workgroup busy; never longer than the host's (the
// workspace of the deterministic form is laid out by band NUMBER).
__device__ __forceinline__ int64_t dev_band_rows(int64_t host_rows, int bands, bool dynamic, int64_t M, int chunk) {
    if (!dynamic || bands < 1) return host_rows;
    const int64_t per = (M + bands - 1) / bands;
    int64_t r = (per + chunk - 1) / chunk * chunk;
    if (r < chunk) r = chunk;
    return r < host_rows ? r : host_rows;
}

template <bool FAST>
__global__ __launch_bounds__(kThreads, 4) void gemm_tn_kernel(TnBatch B) {
    __shared__ __attribute__((aligned(16))) float zt[kChunk * kLd];
    __shared__ __attribute__((aligned(16))) float xt[kChunk * kLd];
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_GEMM_TN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_gemm_tn_desc& D = B.d[di];
    const int tn = B.tiles_n[di], tk = B.tiles_k[di];
    int b = blockIdx.x - B.blk_start[di];
    const int tile_k = b % tk;
    b /= tk;
    const int tile_n = b % tn;
    const int band = b / tn;
    // rows of the reduction that exist (include/cwn_hip.h, "device-side row counts"): D.M is then the capacity -- it shaped the
    // grid and bounds the addresses of the loads, which do not wait for this one
    const int64_t Mcap = D.M;
    const int64_t M = D.m_dev != nullptr ? (*D.m_dev < Mcap ? *D.m_dev : Mcap) : Mcap;
    const int64_t band_rows = dev_band_rows(B.band_rows_of[di], B.bands[di], D.m_dev != nullptr, M, kChunk);
    const int64_t row_lo = (int64_t)band * band_rows;
    const int64_t cap_hi = row_lo + band_rows < Mcap ? row_lo + band_rows : Mcap;
    const int64_t row_hi = row_lo + band_rows < M ? row_lo + band_rows : M;
    const int n0 = tile_n * kTile, k0 = tile_k * kTile;
    const int N = D.N, Ktot = D.K + D.K2;
    const Src SZ{D.dZ, nullptr, D.lddz, 0, N, 0, nullptr, nullptr, nullptr, nullptr, 0};
    const Src SX{D.X, D.X2, D.ldx, D.ldx2, D.K, D.K2, D.in_scale, D.in_shift, D.in_scale2, D.in_shift2,
                 D.in_relu};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool do_bias = D.db != nullptr && tile_k == 0;

    f32x4 vz[kU], vx[kU];
    tile_load<FAST>(vz, SZ, n0, row_lo, cap_hi);
    tile_load<FAST>(vx, SX, k0, row_lo, cap_hi);
    // a band past the batch's own rows adds nothing (the deterministic form still writes its partial tile: zeros)
    if (row_lo >= row_hi && B.ws[di] == nullptr) return;
    const Pro PZ = make_pro(SZ, n0), PX = make_pro(SX, k0);
    for (int64_t row0 = row_lo; row0 < row_hi; row0 += kChunk) {
        __syncthreads();                          // everyone is done reading the previous chunk
        tile_store(zt, vz, PZ, row0, row_hi);
        tile_store(xt, vx, PX, row0, row_hi);
        __syncthreads();
        if (row0 + kChunk < row_hi) {             // next chunk in flight during the MFMAs
            tile_load<FAST>(vz, SZ, n0, row0 + kChunk, cap_hi);
            tile_load<FAST>(vx, SX, k0, row0 + kChunk, cap_hi);
        }
        if (do_bias && threadIdx.x < kTile && !(B.dbg & 4)) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < kChunk; ++r) s += zt[r * kLd + threadIdx.x];
            bsum += s;
        }
        if (!(B.dbg & 2))
#pragma unroll
        for (int s = 0; s < kChunk / 4; ++s) {
            const int row = 4 * s + g;
            float a[2], bb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = zt[row * kLd + wn * 32 + t * 16 + j];
                bb[t] = xt[row * kLd + wk * 32 + t * 16 + j];
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt], bb[kt], acc[nt][kt], 0, 0, 0);
        }
    }
    // acc[nt][kt][r] = partial dW[n0 + wn*32 + nt*16 + 4g + r][k0 + wk*32 + kt*16 + j]
    if (B.dbg & 1) return;
    float* const ws = B.ws[di];
    float* const wsW = ws != nullptr ? ws + (int64_t)band * N * Ktot : nullptr;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 32 + nt * 16 + 4 * g + r;
                const int k = k0 + wk * 32 + kt * 16 + j;
                if (n < N && k < Ktot) {
                    if (wsW != nullptr) wsW[(int64_t)n * Ktot + k] = acc[nt][kt][r];   // summed by tn_reduce_kernel
                    else atomicAdd(D.dW + (int64_t)n * D.lddw + k, acc[nt][kt][r]);
                }
            }
    if (do_bias && threadIdx.x < kTile && n0 + (int)threadIdx.x < N) {
        if (ws != nullptr) ws[(int64_t)B.bands[di] * N * Ktot + (int64_t)band * N + n0 + threadIdx.x] = bsum;
        else atomicAdd(D.db + n0 + threadIdx.x, bsum);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix pipe (round 4): both operands split three ways (cwn_split.h: x = hi + mid + lo, exact),
// six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block instead of eight v_mfma_f32_16x16x4_f32 at twice their issue time each
// -- 96 MFMA cycles where the fp32 pipe takes 256 -- with the dropped terms (<= 2^-26 |dz||x|) below the rounding of the fp32
// accumulation, like the forward kernels.  With the multiply that cheap the launch is a memory pipeline, and shaped as one:
//   * a workgroup owns a 128 x 128 tile of dW (all of it at hidden 128: dZ and X are read ONCE per band, where 64 x 64
//     tiles read each twice) and a band of rows; EIGHT waves, 64 (n) x 32 (k) each (8 accumulator tiles), one workgroup
//     per CU;
//   * a chunk is 32 rows of M = 32 KB of the two operands, two 16-B loads per thread and operand -- and THREE chunks are in
//     flight in registers (a ring of three named buffers, the loop unrolled by three): a chunk's MFMAs take ~0.4 us, a load
//     from HBM / MALL 2 us, and with one chunk in flight (the first form of this kernel: four waves, two workgroups per CU)
//     every chunk waited ~1.5 us for its data: 45 us per merged ZINC-128 launch where the fp32 kernel, four workgroups deep
//     per CU, took 38;
//   * a thread splits the float4s it loaded (prologue applied first) and stores 4 bf16 per plane ROW-MAJOR, [m][n] / [m][k]
//     with a row stride of 144 bf16: a wave writes two full rows, every bank once;
//   * an MFMA operand wants 8 CONSECUTIVE m for one n (or k): the column of a row-major tile.  gfx950's transposing LDS read
//     does exactly that (ds_read_b64_tr_b16: a 16-lane group reads a 4 x 16 block, lane i supplying the address of
//     [i / 4][4 (i % 4)] and receiving column i; semantics pinned by tools/proto/tr_read_probe.hip on the hardware): two of them
//     per fragment.  The 8 reduction indices of lane group g are chunk rows {4g .. 4g+3} and {16+4g .. 16+4g+3} -- any
//     bijection works as long as both operands use the same one, and this one makes the 32 lanes the LDS serves together read
//     8 CONSECUTIVE rows, which at 72 dwords per row land on all 64 banks once (rows 8 apart would collide);
//   * the bias gradient is summed from the registers of the load (a thread keeps its 4 columns for the whole band) and
//     reduced once at the end of the band.
// 110 KB of LDS (two sets of planes).  Chosen when every operand is 16-B aligned (the FAST test below) unless CWN_TN_SPLIT=0; the fp32-MFMA kernel
// above stays for the rest and for the comparison (tools/ubench_tn24.py).
constexpr int kThreads2 = 512;
constexpr int kTile2 = 128;      // rows and columns of dW per workgroup
constexpr int kChunk2 = 32;      // rows of M per LDS stage = the k of one MFMA
constexpr int kLd2 = 144;        // LDS row stride in bf16 (72 dwords = 8 * 9)
constexpr int kPlane = kChunk2 * kLd2;
constexpr int kU2 = kChunk2 * (kTile2 / 4) / kThreads2;     // 16-B loads per thread, operand and chunk (= 2)
#ifndef CWN_TN_RING
#define CWN_TN_RING 3
#endif
#ifndef CWN_TN_WGS
#define CWN_TN_WGS 1
#endif
constexpr int kRing = CWN_TN_RING;         // chunks in flight
static_assert(kThreads2 % (kTile2 / 4) == 0, "a thread keeps its columns");

typedef short v4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint2 lds_tr(const uint16_t* p) {
    const v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
    return __builtin_bit_cast(uint2, v);
}

// one operand fragment (8 reduction indices of this lane group for column `col`) of the three planes at `base`
__device__ __forceinline__ void frag3(const uint16_t* base, uint4& h, uint4& m, uint4& l) {
    const uint2 h0 = lds_tr(base), h1 = lds_tr(base + 16 * kLd2);
    const uint2 m0 = lds_tr(base + kPlane), m1 = lds_tr(base + kPlane + 16 * kLd2);
    const uint2 l0 = lds_tr(base + 2 * kPlane), l1 = lds_tr(base + 2 * kPlane + 16 * kLd2);
    h = make_uint4(h0.x, h0.y, h1.x, h1.y);
    m = make_uint4(m0.x, m0.y, m1.x, m1.y);
    l = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

struct Chunk { f32x4 z[kU2], x[kU2]; };      // one chunk of both operands as a thread holds it

__device__ __forceinline__ void chunk_load_one(f32x4 (&v)[kU2], const Src& S, const Pro& P, int64_t row0, int64_t row_hi) {
#pragma unroll
    for (int u = 0; u < kU2; ++u) {
        const int r = (u * kThreads2 + (int)threadIdx.x) / (kTile2 / 4);
        const int64_t row = row0 + r < row_hi ? row0 + r : row_hi - 1;      // clamped, never faults
        const bool second = P.second;
        const float* base = second ? S.p2 : S.p1;
        const int64_t ld = second ? S.ld2 : S.ld1;
        v[u] = *reinterpret_cast<const f32x4*>(base + row * ld + (P.cc < P.cmax ? P.cc : 0));
    }
}

// prologue + three-way split of the chunk a thread loaded, 4 bf16 per plane at [r][4c]; returns the column sums it saw
template <bool PLAIN>
__device__ __forceinline__ f32x4 chunk_store_one(uint16_t* planes, const f32x4 (&v)[kU2], const Pro& P, int64_t row0,
                                                 int64_t row_hi) {
    f32x4 colsum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kU2; ++u) {
        const int q = u * kThreads2 + threadIdx.x;
        const int r = q / (kTile2 / 4), c = q % (kTile2 / 4);
        const bool row_ok = row0 + r < row_hi;
        f32x4 x = v[u];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float y = x[t];
            const bool ok = row_ok && P.cc + t < P.cmax;
            if constexpr (!PLAIN) {
                if (P.affine) y = y * P.sc[t] + P.sh[t];
                if (P.relu) y = fmaxf(y, 0.f);
            }
            x[t] = ok ? y : 0.f;
        }
        colsum += x;
        uint2 ph, pm, pl;
        cwn::split4(make_float4(x[0], x[1], x[2], x[3]), ph, pm, pl);
        uint16_t* dst = planes + r * kLd2 + 4 * c;
        *reinterpret_cast<uint2*>(dst) = ph;
        *reinterpret_cast<uint2*>(dst + kPlane) = pm;
        *reinterpret_cast<uint2*>(dst + 2 * kPlane) = pl;
    }
    return colsum;
}

__global__ __launch_bounds__(kThreads2, CWN_TN_WGS) void gemm_tn_split_kernel(TnBatch B) {
    // two sets of planes: chunk c is staged in set c & 1 while slower waves still multiply chunk c - 1 out of the other --
    // ONE barrier per chunk (a wave that has passed the barrier of chunk c is done with the MFMAs of chunk c - 1, whose set
    // chunk c + 1 overwrites)
    __shared__ __attribute__((aligned(16))) uint16_t zp[2][3 * kPlane];
    __shared__ __attribute__((aligned(16))) uint16_t xp[2][3 * kPlane];
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_GEMM_TN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    const cwn_gemm_tn_desc& D = B.d[di];
    const int tn = B.tiles_n[di], tk = B.tiles_k[di];
    int b = blockIdx.x - B.blk_start[di];
    const int tile_k = b % tk;
    b /= tk;
    const int tile_n = b % tn;
    const int band = b / tn;
    const int64_t Mcap = D.M;
    // (a static batch: the bands are cut from the batch's OWN rows -- dev_band_rows -- so this launch's first loads wait for the
    //  row count; a prepared batch: the host's cut, the count not waited for)
    const int64_t M = D.m_dev != nullptr ? (*D.m_dev < Mcap ? *D.m_dev : Mcap) : Mcap;
    const int64_t band_rows = dev_band_rows(B.band_rows_of[di], B.bands[di], D.m_dev != nullptr, M, kChunk2);
    const int64_t row_lo = (int64_t)band * band_rows;
    const int64_t cap_hi = row_lo + band_rows < Mcap ? row_lo + band_rows : Mcap;
    const int n0 = tile_n * kTile2, k0 = tile_k * kTile2;
    const int N = D.N, Ktot = D.K + D.K2;
    const Src SZ{D.dZ, nullptr, D.lddz, 0, N, 0, nullptr, nullptr, nullptr, nullptr, 0};
    const Src SX{D.X, D.X2, D.ldx, D.ldx2, D.K, D.K2, D.in_scale, D.in_shift, D.in_scale2, D.in_shift2,
                 D.in_relu};
    // which matrix / columns this thread loads (the same for every chunk); the prologue's constants come after the first loads
    Pro PZ, PX;
    {
        const int c = threadIdx.x % (kTile2 / 4);
        PZ.second = false; PZ.cc = n0 + 4 * c; PZ.cmax = N;
        const int col = k0 + 4 * c;
        PX.second = SX.c2 > 0 && col >= SX.c1;
        PX.cc = PX.second ? col - SX.c1 : col;
        PX.cmax = PX.second ? SX.c2 : SX.c1;
    }
    // the ring: chunks 0 .. 2 of the band requested at once (bounded by the CAPACITY of the band: the row count on the device
    // is not waited for)
    Chunk ring0, ring1, ring2;
    chunk_load_one(ring0.z, SZ, PZ, row_lo, cap_hi);
    chunk_load_one(ring0.x, SX, PX, row_lo, cap_hi);
    if (row_lo + kChunk2 < cap_hi) {
        chunk_load_one(ring1.z, SZ, PZ, row_lo + kChunk2, cap_hi);
        chunk_load_one(ring1.x, SX, PX, row_lo + kChunk2, cap_hi);
    }
    if (kRing > 2 && row_lo + 2 * kChunk2 < cap_hi) {
        chunk_load_one(ring2.z, SZ, PZ, row_lo + 2 * kChunk2, cap_hi);
        chunk_load_one(ring2.x, SX, PX, row_lo + 2 * kChunk2, cap_hi);
    }
    const int64_t row_hi = row_lo + band_rows < M ? row_lo + band_rows : M;
    if (row_lo >= row_hi && B.ws[di] == nullptr) return;
    fill_pro(PX, SX);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int wn = wave >> 2, wk = wave & 3;

    cwn::frag_cd acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = (cwn::frag_cd){0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = D.db != nullptr && tile_k == 0;
    // this lane's corner of a 4 x 16 block: row 4g + j / 4 (+ 16 for the second half), column 4 (j % 4)
    const int frag_off = (4 * g + (j >> 2)) * kLd2 + 4 * (j & 3);
    const int frag_z = frag_off + wn * 64, frag_x = frag_off + wk * 32;

    // one chunk: planes <- the ring slot, the slot re-requested three chunks ahead, the MFMAs
    int set = 0;
    auto step = [&](Chunk& C, int64_t row0) {
        const f32x4 cs = chunk_store_one<true>(zp[set], C.z, PZ, row0, row_hi);
        if (do_bias && !(B.dbg & 4)) bsum += cs;
        chunk_store_one<false>(xp[set], C.x, PX, row0, row_hi);
        __syncthreads();
        if (row0 + kRing * kChunk2 < row_hi) {
            chunk_load_one(C.z, SZ, PZ, row0 + kRing * kChunk2, cap_hi);
            chunk_load_one(C.x, SX, PX, row0 + kRing * kChunk2, cap_hi);
        }
        const uint16_t* const za = zp[set] + frag_z;
        const uint16_t* const xa = xp[set] + frag_x;
        set ^= 1;
        if (B.dbg & 2) return;
        // all six fragments of the wave, then the six terms of cwn::mfma_split6 (same terms, same order per tile) ACROSS the
        // eight tiles: consecutive MFMAs never wait for each other's accumulator
        uint4 xh[2], xm[2], xl[2], zh[4], zm[4], zl[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) frag3(xa + kt * 16, xh[kt], xm[kt], xl[kt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) frag3(za + nt * 16, zh[nt], zm[nt], zl[nt]);
#define CWN_TN_TERM(ZP, XP)                                                                                              \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                   \
        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cwn::as_frag(ZP[nt]), cwn::as_frag(XP[kt]), acc[nt][kt], 0, 0, 0);
        CWN_TN_TERM(zl, xh)
        CWN_TN_TERM(zh, xl)
        CWN_TN_TERM(zm, xm)
        CWN_TN_TERM(zm, xh)
        CWN_TN_TERM(zh, xm)
        CWN_TN_TERM(zh, xh)
#undef CWN_TN_TERM
    };
    for (int64_t row0 = row_lo; row0 < row_hi; row0 += kRing * kChunk2) {
        step(ring0, row0);
        if (row0 + kChunk2 < row_hi) step(ring1, row0 + kChunk2);
        if (kRing > 2 && row0 + 2 * kChunk2 < row_hi) step(ring2, row0 + 2 * kChunk2);
    }
    // acc[nt][kt][r] = partial dW[n0 + wn*64 + nt*16 + 4g + r][k0 + wk*32 + kt*16 + j]
    if (B.dbg & 1) return;
    float* const ws = B.ws[di];
    float* const wsW = ws != nullptr ? ws + (int64_t)band * N * Ktot : nullptr;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + nt * 16 + 4 * g + r;
                const int k = k0 + wk * 32 + kt * 16 + j;
                if (n < N && k < Ktot) {
                    if (wsW != nullptr) wsW[(int64_t)n * Ktot + k] = acc[nt][kt][r];
                    else atomicAdd(D.dW + (int64_t)n * D.lddw + k, acc[nt][kt][r]);
                }
            }
    if (do_bias) {
        // thread t summed columns 4 (t % 32) .. +3 over the rows t / 32 + 16 u of every chunk: sixteen partial sums per column
        __syncthreads();
        float* red = reinterpret_cast<float*>(zp[0]);       // [16][128]
        *reinterpret_cast<f32x4*>(red + (threadIdx.x / (kTile2 / 4)) * kTile2 + 4 * (threadIdx.x % (kTile2 / 4))) = bsum;
        __syncthreads();
        if (threadIdx.x < kTile2 && n0 + (int)threadIdx.x < N) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < kThreads2 / (kTile2 / 4); ++i) s += red[i * kTile2 + threadIdx.x];
            if (ws != nullptr) ws[(int64_t)B.bands[di] * N * Ktot + (int64_t)band * N + n0 + threadIdx.x] = s;
            else atomicAdd(D.db + n0 + threadIdx.x, s);
        }
    }
}

// second stage of the deterministic form: dW[n, k] += sum over the bands (in band order) of the
// partial tiles; db likewise.  One thread per element, bands walked four at a time.
__global__ __launch_bounds__(kThreads) void tn_reduce_kernel(TnBatch B) {
    const int di = blockIdx.y;
    const cwn_gemm_tn_desc& D = B.d[di];
    const float* ws = B.ws[di];
    if (ws == nullptr) return;
    const int N = D.N, Ktot = D.K + D.K2, bands = B.bands[di];
    const int64_t nk = (int64_t)N * Ktot;
    const int64_t total = nk + (D.db != nullptr ? N : 0);
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const bool is_b = e >= nk;
        const float* src = is_b ? ws + (int64_t)bands * nk + (e - nk) : ws + e;
        const int64_t stride = is_b ? N : nk;
        float s = 0.f;
        int b = 0;
        for (; b + 4 <= bands; b += 4) {
            const float t0 = src[(int64_t)b * stride], t1 = src[(int64_t)(b + 1) * stride];
            const float t2 = src[(int64_t)(b + 2) * stride], t3 = src[(int64_t)(b + 3) * stride];
            s = (((s + t0) + t1) + t2) + t3;
        }
        for (; b < bands; ++b) s += src[(int64_t)b * stride];
        if (is_b) D.db[e - nk] += s;
        else D.dW[(e / Ktot) * D.lddw + (e % Ktot)] += s;
    }
}

inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; }

inline size_t tn_ws_floats(const cwn_gemm_tn_desc& D) {
    const size_t bands = (size_t)((D.M + kBandRows - 1) / kBandRows);
    return bands * (size_t)D.N * (size_t)(D.K + D.K2) + bands * (size_t)D.N;
}

}  // namespace

extern "C" size_t cwn_gemm_tn_workspace_bytes(const cwn_gemm_tn_desc* descs, int n) {
    if (descs == nullptr || n <= 0 || n > CWN_GEMM_TN_MAX_DESCS) return 0;
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (tn_ws_floats(descs[i]) * 4 + 255) / 256 * 256;
    return total;
}

extern "C" int cwn_gemm_tn_f32(const cwn_gemm_tn_desc* descs, int n, void* workspace, size_t ws_bytes,
                               cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_GEMM_TN_MAX_DESCS) return CWN_ERR_BAD_ARG;
    if (workspace != nullptr && ws_bytes < cwn_gemm_tn_workspace_bytes(descs, n)) return CWN_ERR_WORKSPACE;
    TnBatch B{};
    B.n = n;
    int64_t blocks = 0;
    bool fast = true;
    for (int i = 0; i < n; ++i) {
        const cwn_gemm_tn_desc& D = descs[i];
        if (D.M < 0 || D.N <= 0 || D.K <= 0 || D.K2 < 0) return CWN_ERR_BAD_ARG;
        if (D.dW == nullptr || D.lddw < D.K + D.K2) return CWN_ERR_BAD_ARG;
        if (D.M > 0 && (D.dZ == nullptr || D.X == nullptr || D.lddz < D.N || D.ldx < D.K)) return CWN_ERR_BAD_ARG;
        if (D.K2 > 0 && (D.X2 == nullptr || D.ldx2 < D.K2 || D.K % 4 != 0)) return CWN_ERR_BAD_ARG;
        if ((D.in_scale == nullptr) != (D.in_shift == nullptr)) return CWN_ERR_BAD_ARG;
        if ((D.in_scale2 == nullptr) != (D.in_shift2 == nullptr)) return CWN_ERR_BAD_ARG;
        const void* ptrs[] = {D.dZ, D.X, D.X2, D.dW, D.db};
        for (const void* p : ptrs)
            if (p != nullptr && ((uintptr_t)p & 3u)) return CWN_ERR_ALIGN;
        fast = fast && al16(D.dZ) && al16(D.X) && al16(D.X2) && D.lddz % 4 == 0 && D.ldx % 4 == 0 &&
               (D.K2 == 0 || D.ldx2 % 4 == 0) && D.N % 4 == 0 && D.K % 4 == 0 && D.K2 % 4 == 0;
        B.d[i] = D;
    }
    static const bool split_off = getenv("CWN_TN_SPLIT") != nullptr && atoi(getenv("CWN_TN_SPLIT")) == 0;
    // the bf16-split kernel (128 x 128 tiles, 32-row chunks) when most descriptors fill its tile: at hidden 64 -- every dW at most
    // 64 rows -- a 128 x 128 tile is a quarter full and the launch is the same memory pipeline for half the useful bytes; the
    // fp32-MFMA kernel's 64 x 64 tiles, four workgroups deep per CU, are faster there (round 5, graph-replayed training steps:
    // molhiv-512 0.802 -> 0.756 ms, reddit-32 1.597 -> 1.449 ms with CWN_TN_SPLIT=0; ZINC-128 at hidden 128 keeps the split form)
    // (by majority: a hidden-64 model's launch holds forty gradients of 64 rows and the head's three of 128)
    int n_wide = 0;
    for (int i = 0; i < n; ++i) n_wide += descs[i].N > 64 ? 1 : 0;
    static const bool split_force = getenv("CWN_TN_SPLIT") != nullptr && atoi(getenv("CWN_TN_SPLIT")) == 2;
    const bool split = fast && !split_off && (2 * n_wide >= n || split_force);
    const int tile = split ? kTile2 : kTile;
    for (int i = 0; i < n; ++i) {
        if ((descs[i].N + tile - 1) / tile > INT16_MAX || (descs[i].K + descs[i].K2 + tile - 1) / tile > INT16_MAX) return CWN_ERR_TOO_LARGE;
        B.tiles_n[i] = (int16_t)((descs[i].N + tile - 1) / tile);
        B.tiles_k[i] = (int16_t)((descs[i].K + descs[i].K2 + tile - 1) / tile);
    }
    // Rows of M per workgroup (workspace layouts are sized for kBandRows: an upper bound on the bands).  The chip holds
    // kResident workgroups of this kernel at once (4 per CU: 40 KB of LDS, 100 registers); a launch costs about
    // rounds x (a fixed ~6 chunk-times of prologue / atomics + its chunks), and a second round that is mostly empty is
    // the expensive case: the merged ZINC-128 launch (24 descriptors) measured 48.7 / 44.7 / 47.9 / 41.9 / 40.0 / 44.4 /
    // 48.8 us at 128 / 192 / 256 / 320 / 384 / 448 / 512 rows (2200 ... 600 workgroups; tools/ubench_tn24.py with
    // CWN_TN_BAND), where doubling from 128 until <= 2048 workgroups had picked 256.
    static const int band_env = getenv("CWN_TN_BAND") ? atoi(getenv("CWN_TN_BAND")) : 0;     // tuning: fixed rows per workgroup
    const int64_t kResident = split ? 256 : 1024;
    const double fixed = split ? kTnFixed2 : 6.0, per_chunk = split ? 0.6 : 1.0;    // (in 64-row chunk-times of the fp32 kernel)
    const int chunk = split ? kChunk2 : kChunk;
    // a target of `cand` rows per workgroup; descriptor i is cut into ceil(M_i / cand) bands of EQUAL length (rounded up to the
    // chunk): the workgroups of a launch end together instead of a short last band per descriptor idling its CU
    auto bands_of = [&](int i, int cand) { return (descs[i].M + cand - 1) / cand; };
    auto rows_of = [&](int i, int cand) {
        const int64_t nb = bands_of(i, cand);
        if (!split || nb == 0) return (int64_t)cand;
        return ((descs[i].M + nb - 1) / nb + chunk - 1) / chunk * chunk;
    };
    int band_rows = kBandRows;
    {
        double best = 0.0;
        for (int cand = kBandRows; cand <= kMaxBandRows; cand += kChunk) {
            if (band_env >= kBandRows && band_env % kChunk == 0 && cand != band_env) continue;
            int64_t nb = 0, longest = 0;
            for (int i = 0; i < n; ++i) {
                nb += bands_of(i, cand) * B.tiles_n[i] * B.tiles_k[i];
                longest = rows_of(i, cand) > longest ? rows_of(i, cand) : longest;
            }
            const double cost = (double)((nb + kResident - 1) / kResident) * (fixed + per_chunk * (double)longest / (double)kChunk);
            if (cand == kBandRows || cost < best || (band_env == cand)) {
                best = cost;
                band_rows = cand;
            }
        }
    }
    blocks = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t bands = bands_of(i, band_rows);
        B.bands[i] = (int32_t)bands;
        B.band_rows_of[i] = (int32_t)rows_of(i, band_rows);
        B.blk_start[i] = (int32_t)blocks;
        blocks += bands * B.tiles_n[i] * B.tiles_k[i];
        if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    }
    for (int i = n; i <= CWN_GEMM_TN_MAX_DESCS; ++i) B.blk_start[i] = (int32_t)blocks;
    if (blocks == 0) return CWN_OK;
    static const int dbg = getenv("CWN_TN_DBG") ? atoi(getenv("CWN_TN_DBG")) : 0;
    B.dbg = dbg;
    if (workspace != nullptr) {
        size_t off = 0;
        for (int i = 0; i < n; ++i) {
            B.ws[i] = descs[i].M > 0 ? (float*)((char*)workspace + off) : nullptr;
            off += (tn_ws_floats(descs[i]) * 4 + 255) / 256 * 256;
        }
    }
    hipStream_t stream = (hipStream_t)stream_;
    if (split) gemm_tn_split_kernel<<<dim3((unsigned)blocks), dim3(kThreads2), 0, stream>>>(B);
    else if (fast) gemm_tn_kernel<true><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
    else gemm_tn_kernel<false><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
    if (workspace != nullptr) tn_reduce_kernel<<<dim3(64, n), dim3(kThreads), 0, stream>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
