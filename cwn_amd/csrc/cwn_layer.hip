// cwn_layer.hip -- one SparseCIN propagate step of a layer in ONE launch, complex-blocked.
//
// What it replaces (reference): the three propagate calls of SparseCINConv.forward
// (mp/layers.py:333-342 -> :184-192 -> mp/cell_mp.py:357-392) with the coboundary message
// ReLU(Linear(cat(x_j, up_attr))) (mp/layers.py:290-295), the up_attr gather of
// data/complex.py:579-580, the zero fills of mp/cell_mp.py:517-522 and the self terms of
// mp/layers.py:191-192.  Round 1 ran this as CSR build + grouped GEMM (Y1, Y2 to HBM) + aggregate
// (Y1, Y2 read back): 2 launches per layer + 1 per batch, 15 us per layer at ZINC-128, the GEMM on
// 161 of 256 CUs.
//
// A batched complex is block-diagonal and per-complex contiguous (data/complex.py:148-169), a
// ZINC-like complex is ~55 cells x 512 B.  So a workgroup (1024 threads, 16 waves) OWNS a range of
// complexes for one "GEMM dimension" g (plus, as a second task, the top dimension that has no
// upper adjacency) and never leaves the CU's LDS between the dense and the sparse half.  The
// kernel is a latency chain, built around one rule: after the item record, every global load of
// the item is issued in ONE branch-free run, and nothing after that touches memory again until
// the output rows are stored --
//
//   1. item record and set record (all pointers of this item's GEMM dimension): one lane-parallel
//      load each, fields by v_readlane; while they travel, half of this wave's slice of the PRE-PACKED
//      weight (bf16 hi/mid/lo planes in MFMA-fragment order: 1 KiB contiguous per instruction)
//   2. loads, in order of use: COO entries (int64, as delivered; or the item's cached CSR), eps, the
//      rows of x_g and x_{g+1} (GEMM operands; also the self terms, see below), the rows of x_{g-1}
//      the boundary stream reads, the other half of the weight slice
//   3. COO -> per-item CSR in LDS (rowptr, col, aux as 16-bit LOCAL row numbers), stable by destination: a
//      bucket sort -- an LDS atomic per entry counts its row, a scan of the counters gives the row
//      pointers, the entries of a row order themselves by entry number
//   4. GEMM rows -> exact 3-way bf16 split (cwn_split.h) -> three bf16 planes in LDS; boundary-source
//      rows -> fp32 in LDS
//   5. boundary stream + self terms of both tasks in one pass, out of LDS and registers
//   6. Y1 = x_g W[:, :F]^T + b,  Y2 = x_{g+1} W[:, F:]^T : v_mfma_f32_16x16x32_bf16, six per 32
//      k-values; a wave owns 16 output columns of ONE of the two products for every row tile; the
//      fp32 result overwrites the planes (dead by then)
//   7. out_up[i] = sum_p relu(Y1[col[p]] + Y2[aux[p]]) + (1 + eps1) x_i out of LDS, in entry order
//
// Thread t loads float4 number t of row (t / (F/4)) + NG*i of the staged block, and the lane group
// that later FINISHES output row r is exactly the one that loaded x row r (same rows, same columns):
// the self terms are the load registers themselves, nothing is read twice.
//
// What the measurements said on the way here (tools/time_layer_phases.py, shader-clock stamps per
// phase and workgroup; ZINC-128, one complex per workgroup):
//   * 19 us/launch: item fields read as `it[k]` behind conditionals -> one global load + vmcnt(0) per
//     field, twelve serial round trips;
//   * pointers rebuilt from integers dereference as FLAT loads, and loads behind branches make the
//     compiler wait with vmcnt(0) (= for the 200 KB issued later) wherever an early result is used;
//   * fragment-shaped loads of the fp32 weight (16 rows x 32 B per quarter wave) run the address unit
//     at 1/8 rate: issuing the loads alone took 6 us.  Hence the packed weight.
//   (the rest of the list: docs/history/DESIGN_rounds1-5.md 4.0)
// HBM traffic = the item's x rows once, its COO entries once, the two output streams once.  No
// global atomics, no zero-fill pass; sums are sequential in the original entry order (deterministic, the
// order a sequential index_add_ visits them).  Results are bit-identical to the two-kernel path
// (cwn_gemm_split.hip + cwn_aggregate.hip): same split, same MFMA order, same epilogue arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include <type_traits>
#include "../../include/cwn_hip.h"
#include "cwn_split.h"

namespace {

using cwn::frag_cd;

// 1024 threads = 16 waves = four per SIMD, at most 128 VGPRs each: the kernel is a chain of phases that
// are each bound by instruction latency, not by a pipe, so twice the waves finish every thread-parallel
// phase (split, boundary stream, reduce) in about half the time.  What pays for it: a wave holds the
// packed weight of ONE of the two products (Y1 or Y2) -- 48 registers instead of 96.
// CWN_LAYER_W8 (a second compilation of this file, csrc/Makefile: cwn_layer_w8.o): the TWO-PER-CU form.  512
// threads = 8 waves at <= 128 VGPRs and <= 80 KiB of LDS, so that two workgroups are resident on a CU and one
// item's load / sort phases run under the other's matrix-core / reduce phases.  The 16-wave form owns a CU
// (register file and LDS) for the whole of its ~6 us chain: fine while a launch has no more items than the chip has
// CUs, 0.54 - 0.62 of the achievable rate beyond that (863 M cells/s at batch 8192 against 981 M of the
// streaming CSR path).  What differs from the 16-wave form is marked `kW8`: a wave multiplies BOTH products of its
// column tile, one after the other, through ONE set of weight registers (the k steps of the second weight are
// requested into the registers the first product has finished with), and the self terms of the upper reduce are
// read again instead of being held in registers across the matrix-core phase.
#ifdef CWN_LAYER_W8
#undef CWN_LAYER_THREADS
#define CWN_LAYER_THREADS 512
#define CWN_W8 1
#else
#define CWN_W8 0
#endif
#ifndef CWN_LAYER_THREADS
#define CWN_LAYER_THREADS 1024
#endif
#ifndef CWN_LAYER_WEARLY
#define CWN_LAYER_WEARLY -1                 // k-steps of the packed weight requested before the item record arrives (-1: half)
#endif
// Output rows leave with non-temporal stores: they are not read again by this launch, and the lines an
// ordinary store leaves dirty in L2 are written back at the END of the kernel -- part of the ~1.3 us between
// two dependent launches (measured: 41.9 -> 39.8 us per step of four launches; non-temporal LOADS of the
// rows: 40.7).
#ifndef CWN_LAYER_NT_STORE
#define CWN_LAYER_NT_STORE 1
#endif
#ifndef CWN_LAYER_WBAR
#define CWN_LAYER_WBAR 1                    // line the waves up between the row loads and the weight loads
#endif
constexpr int kThreads = CWN_LAYER_THREADS;
static_assert(kThreads == 512 || kThreads == 1024, "8 or 16 waves");
constexpr int kWaves = kThreads / 64;
constexpr bool kW8 = CWN_W8 != 0;
constexpr int kHS = kThreads == 1024 ? 2 : 1;   // 2: a wave computes Y1 OR Y2; 1: both (kW8: one after the other)
constexpr int kWSets = kW8 ? 1 : 2 / kHS;       // register sets of packed weight a wave holds
// caps of an item (include/cwn_hip.h): the 16-wave form's, or the two-per-CU form's
constexpr size_t kLdsBudget = kW8 ? CWN_LAYER_W8_LDS_BYTES : 160 * 1024;
__host__ __device__ constexpr int gemm_rows_cap(int F) { return kW8 ? CWN_LAYER_W8_GEMM_ROWS(F) : CWN_LAYER_GEMM_ROWS(F); }
__host__ __device__ constexpr int source_rows_cap(int F) { return kW8 ? CWN_LAYER_W8_SOURCE_ROWS(F) : CWN_LAYER_SOURCE_ROWS(F); }
constexpr int kEcap = CWN_LAYER_MAX_ENTRIES;
constexpr int kEI = kEcap / kThreads;       // COO entries per thread
constexpr int kTaskRows = CWN_LAYER_TASK_ROWS;

// item record fields (include/cwn_hip.h)
enum { I_FLAGS = 0, I_G, I_GR0, I_GN, I_CR0, I_CN, I_UE0, I_UNE, I_NT, I_TASK0, I_R1 = 23, I_ROWS, I_B1, I_B2, I_TOTAL };
enum { T_DIM = 0, T_R0, T_N, T_BE0, T_BNE, T_SR0, T_SN, T_INTS };

// One SET = the items that share a GEMM dimension (cwn_amd/blockplan.py): the launcher resolves every
// pointer such an item needs into one flat record of 8-byte fields, and a workgroup reads the record
// of ITS set from the kernel-argument segment at a run-time offset (scalar loads).  With the
// per-dimension descriptors selected field by field in the kernel (dim == 0 ? .. : ..) the compiler
// ran out of SGPRs and reloaded kernel arguments ten times, one wait each.
enum { S_XG = 0, S_XC, S_UP_INDEX, S_UP_SHARED, S_UP_E, S_WP, S_MSG_BIAS, S_PAD, S_TASK0 };
// (ST_OUT_D / ST_EPS3: the third output of a CIN++ layer, out_down = (1 + eps) x -- mp/layers.py:253 with the lower stream off)
enum { ST_X = 0, ST_XS, ST_OUT_UP, ST_OUT_B, ST_B_INDEX, ST_B_E, ST_EPS1, ST_EPS2, ST_OUT_D, ST_EPS3, ST_FIELDS };
constexpr int kSetFields = S_TASK0 + 2 * ST_FIELDS;         // 28 fields of 8 bytes
constexpr int kMaxSets = CWN_LAYER_MAX_DIMS;

// what a BIG item (include/cwn_hip.h) reads besides its set record: the caller's CSR of the big complexes' entries
// (global cell numbers) and the scratch matrices for Y1 / Y2
struct BigSet {
    const int32_t* up_rowptr; const int32_t* up_col; const int32_t* up_aux;
    const int32_t* b_rowptr[2]; const int32_t* b_col[2];
    float* y1; float* y2;
};

struct LayerArgs {
    uint64_t set[kMaxSets][kSetFields];      // MUST stay first: read through the kernarg segment pointer
    const int32_t* items;
    int32_t* err;
    unsigned char* csr_cache;                // [n_items][kCsrSlot] per-item CSR images, or NULL
    int32_t rows_cap;                        // staged rows (GEMM operands) the LDS of this launch holds
    int32_t xrows_cap;                       // boundary-source rows it holds
    int32_t set_start1, set_start2;          // first workgroup of set 1 / set 2 (items are ordered by set)
    int32_t lds_limit;                       // dynamic LDS bytes of this launch (kW8: every item lays out its own rows inside it)
    int32_t store_y;                         // CWN_LAYER_STORE_Y: Y1 / Y2 of every item also go to big[set].y1 / y2
    BigSet big[kMaxSets];                    // big items only
#ifdef CWN_LAYER_TIMING
    unsigned long long* stamps;              // [n_items][96]: [0, 16) phase ends seen by wave 0, [16 + 16 k + w] point k of wave w
                                             // (shader clock: per XCD); [64 + w] / [80 + w]: wave w's first / last instruction on
                                             // the chip-wide 100-MHz clock (s_memrealtime: comparable across workgroups and launches)
#endif
};

#ifdef CWN_LAYER_TIMING
#define CWN_STAMP_REC 96
#define CWN_STAMP(k)                                                                   \
    do {                                                                               \
        if (threadIdx.x == 0 && A.stamps != nullptr)                                   \
            A.stamps[(size_t)blockIdx.x * CWN_STAMP_REC + (k)] = __builtin_amdgcn_s_memtime();    \
    } while (0)
#define CWN_WSTAMP(k)                                                                  \
    do {                                                                               \
        if ((threadIdx.x & 63) == 0 && A.stamps != nullptr)                            \
            A.stamps[(size_t)blockIdx.x * CWN_STAMP_REC + 16 + 16 * (k) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define CWN_RSTAMP(k)                                                                  \
    do {                                                                               \
        if ((threadIdx.x & 63) == 0 && A.stamps != nullptr)                            \
            A.stamps[(size_t)blockIdx.x * CWN_STAMP_REC + 64 + 16 * (k) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define CWN_STAMP(k) do { } while (0)
#define CWN_WSTAMP(k) do { } while (0)
#define CWN_RSTAMP(k) do { } while (0)
#endif

template <int F> struct Geo {
    static constexpr int kPlaneStride = F + 8;              // bf16 elements per plane row
    static constexpr int kYStride = F + 4;                  // floats per Y row
    static constexpr int kKS = F / 32;                      // k-steps of 32
    static constexpr int kNCT = F / 16;                     // column tiles
    static constexpr int kWPC = kWaves / kNCT / kHS;        // waves sharing a column tile of a product (row-tile parity)
    static constexpr int kG = F / 4;                        // lanes per row
    static constexpr int kNG = kThreads / kG;               // rows per round (1024 threads: 32 at F = 128, 64 at F = 64)
    static constexpr int kNX = gemm_rows_cap(F) / kNG;           // float4 of staged rows per thread at the row cap
    static constexpr int kNE = source_rows_cap(F) / kNG;         // float4 of boundary-source rows per thread at the cap
    static_assert(gemm_rows_cap(F) % kNG == 0 && source_rows_cap(F) % kNG == 0, "caps are whole rounds");
    // 16-row tiles of ONE product a wave multiplies (its accumulators wait in registers until every wave has read its
    // fragments); the two-per-CU form bounds the rows of each product separately (CWN_LAYER_W8_HALF_ROWS)
    static constexpr int kMaxT = kW8 ? CWN_LAYER_W8_HALF_ROWS(F) / 16 / kWPC : gemm_rows_cap(F) / 16 / kWPC;
    static constexpr int kWChunks = 2 * kKS * 3;            // 1-KiB chunks of the packed weight per column tile
    // planes [3][rows][F + 8] bf16, overwritten by Y [rows][F + 4] fp32 once the MFMAs have read them
    __host__ __device__ static constexpr size_t planes_bytes(int rows) { return (size_t)3 * rows * kPlaneStride * 2; }
    // fp32 boundary-source rows + one row of zeros (a reduce slot past the end of a row reads it)
    __host__ __device__ static constexpr size_t xrows_bytes(int rows) { return (size_t)(rows + 1) * F * 4; }
};

// index scratch: [row counters u32 [3][kRpStride] | entry slots u16 [kEcap] | scol | saux | rowptr].  The tail
// [scol | saux | rowptr] is the item's finished CSR: one contiguous image, kCsrSlot bytes, that the first
// layer of a batch can store (CWN_LAYER_CSR_STORE) and the following layers load back (CWN_LAYER_CSR_LOAD)
// instead of sorting the same entries again.
constexpr int kRpStride = kTaskRows + 2;
constexpr size_t kCntBytes = ((size_t)3 * kRpStride * 4 + 15) & ~(size_t)15;
constexpr size_t kIdxBytes = kCntBytes + (size_t)3 * kEcap * 2 + (size_t)3 * kRpStride * 2;
constexpr int kCsrSlot = CWN_LAYER_CSR_SLOT_BYTES;
static_assert(kCsrSlot >= 2 * kEcap * 2 + 3 * kRpStride * 2 && kCsrSlot % 16 == 0, "slot holds scol, saux, rowptr");
enum { kSort = 0, kSortStore = 1, kLoad = 2 };

template <int F>
__host__ __device__ constexpr size_t lds_bytes(int rows, int xrows) {
    return Geo<F>::planes_bytes(rows) + Geo<F>::xrows_bytes(xrows) + ((kIdxBytes + 15) & ~(size_t)15);
}

// Pointers rebuilt from the 8-byte fields of a set record carry no address space: dereferenced as
// plain C++ pointers they compile to FLAT loads (which also count against lgkmcnt and made the
// compiler wait for vmcnt(0) AND lgkmcnt(0) between them).  These types pin them to global memory.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float* gcf_p;
typedef __attribute__((address_space(1))) float* gf_p;
typedef const __attribute__((address_space(1))) int64_t* gci64_p;
typedef const __attribute__((address_space(1))) v4f* gcv4_p;
typedef const __attribute__((address_space(1))) v4u* gcu4_p;
typedef __attribute__((address_space(1))) v4f* gv4_p;
typedef const __attribute__((address_space(1))) unsigned char* gcb_p;
typedef __attribute__((address_space(1))) unsigned char* gb_p;
typedef const __attribute__((address_space(1))) uint64_t* gcu64_p;
typedef __attribute__((address_space(1))) v4u* gu4_p;

__device__ __forceinline__ float4 ldg4(gcf_p p) {
    const v4f v = *(gcv4_p)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 ldgu4(gcb_p p) {
    const v4u v = *(gcu4_p)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stg4(gf_p p, const float4& a) {
    const v4f v = {a.x, a.y, a.z, a.w};
    *(gv4_p)p = v;
}
// uniform base (SGPR pair) + per-lane 32-bit byte offset: one address instruction instead of a 64-bit add chain
__device__ __forceinline__ float4 ldg4o(gcb_p base, uint32_t off) {
    const v4f v = *(gcv4_p)(base + off);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 ldgu4o(gcb_p base, uint32_t off) {
    const v4u v = *(gcu4_p)(base + off);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stg4o(gb_p base, uint32_t off, const float4& a) {
    const v4f v = {a.x, a.y, a.z, a.w};
#if CWN_LAYER_NT_STORE
    __builtin_nontemporal_store(v, (gv4_p)(base + off));     // outputs are not read again by this launch
#else
    *(gv4_p)(base + off) = v;
#endif
}
__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 sel4(bool c, const float4& a, const float4& b) {
    return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
// xv[k] for a run-time k: a chain of selects on registers (an indexed array would live in scratch)
template <int NX>
__device__ __forceinline__ float4 pick(const float4 (&xv)[NX], int k) {
    float4 r = xv[0];
#pragma unroll
    for (int i = 1; i < NX; ++i) r = sel4(k == i, xv[i], r);
    return r;
}

// Reduce phases: a lane group (F/4 lanes, one float4 each) finishes NR destination rows AT ONCE -- rows
// r0, r0 + NG, ... -- so that the chain row pointers -> source numbers -> source rows (three LDS round
// trips a step) runs NR x 2 times in parallel instead of once: these phases are latency-bound (one
// complex gives a lane group two or three rows of two to six entries).  Entries are added in entry
// order; a slot past the end of its row reads a row of zeros (+0: exact).
template <int NR>
struct RowSet { int s[NR], e[NR]; bool on[NR]; };

template <int NR, int NG>
__device__ __forceinline__ int row_ranges(RowSet<NR>& R, const uint16_t* rp, int r0, int n_rows) {
    int steps = 0;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int r = r0 + u * NG;
        R.on[u] = r < n_rows;
        const int rr = R.on[u] ? r : 0;
        const int s = rp[rr], e = rp[rr + 1];
        R.s[u] = R.on[u] ? s : 0;
        R.e[u] = R.on[u] ? e : 0;
        steps = max(steps, R.e[u] - R.s[u]);
    }
    return steps;
}

// sum of rows src[col[p]] over each row's entries
template <int NR, int F>
__device__ __forceinline__ void gather_sum(float4 (&acc)[NR], const RowSet<NR>& R, int steps,
                                           const uint16_t* const (&cols)[NR], const float* src, int zero_row, int f) {
#pragma unroll
    for (int u = 0; u < NR; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < steps; q += 2) {
        int c[NR][2];
        float4 a[NR][2];
#pragma unroll
        for (int u = 0; u < NR; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int p = R.s[u] + q + v;
                const int cv = cols[u][min(p, max(R.e[u] - 1, 0))];
                c[u][v] = p < R.e[u] ? cv : zero_row;
            }
#pragma unroll
        for (int u = 0; u < NR; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) a[u][v] = lds4(src + (size_t)c[u][v] * F + f);
#pragma unroll
        for (int u = 0; u < NR; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                acc[u].x += a[u][v].x; acc[u].y += a[u][v].y; acc[u].z += a[u][v].z; acc[u].w += a[u][v].w;
            }
    }
}

// Two-per-CU form of the boundary pass: chain 0 (task 0) reads fp32 source rows as above; chain 1 (task 1: the top
// dimension, whose sources are the STAGED cells of g) reads them out of the three bf16 planes -- source number
// v >= first1 is staged row v - first1, and hi + mid + lo is the fp32 value bit for bit (cwn_split.h: every step of
// the split is exact) -- so that no second fp32 copy of those rows occupies LDS.  Same entry order, same sums.
template <int F>
__device__ __forceinline__ void gather_sum_planes(float4 (&acc)[2], const RowSet<2>& R, int steps,
                                                  const uint16_t* const (&cols)[2], const float* src, int zero_row,
                                                  const uint16_t* planes, size_t plane, int first1, int f) {
    constexpr int kPS = F + 8;
    acc[0] = acc[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto up = [](uint32_t w, bool hi) { return __uint_as_float(hi ? (w & 0xFFFF0000u) : (w << 16)); };
    for (int q = 0; q < steps; q += 2) {
        int c0[2], c1[2];
        bool on1[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int p0 = R.s[0] + q + v, p1 = R.s[1] + q + v;
            const int a = cols[0][min(p0, max(R.e[0] - 1, 0))], b = cols[1][min(p1, max(R.e[1] - 1, 0))];
            c0[v] = p0 < R.e[0] ? a : zero_row;
            on1[v] = p1 < R.e[1];
            c1[v] = on1[v] ? b - first1 : 0;
        }
        float4 a0[2];
        uint2 h[2], m[2], l[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            a0[v] = lds4(src + (size_t)c0[v] * F + f);
            const uint16_t* pr = planes + (size_t)c1[v] * kPS + f;
            h[v] = *reinterpret_cast<const uint2*>(pr);
            m[v] = *reinterpret_cast<const uint2*>(pr + plane);
            l[v] = *reinterpret_cast<const uint2*>(pr + 2 * plane);
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            acc[0].x += a0[v].x; acc[0].y += a0[v].y; acc[0].z += a0[v].z; acc[0].w += a0[v].w;
            float4 x;
            x.x = (up(h[v].x, false) + up(m[v].x, false)) + up(l[v].x, false);
            x.y = (up(h[v].x, true) + up(m[v].x, true)) + up(l[v].x, true);
            x.z = (up(h[v].y, false) + up(m[v].y, false)) + up(l[v].y, false);
            x.w = (up(h[v].y, true) + up(m[v].y, true)) + up(l[v].y, true);
            x = sel4(on1[v], x, make_float4(0.f, 0.f, 0.f, 0.f));
            acc[1].x += x.x; acc[1].y += x.y; acc[1].z += x.z; acc[1].w += x.w;
        }
    }
}

// sum of relu(Y1[col[p]] + Y2[aux[p]]) over each row's entries (y2_off = first Y2 row - 0)
template <int NR, int NG, int YS>
__device__ __forceinline__ void gather_relu_sum(float4 (&acc)[NR], const RowSet<NR>& R, int steps, const uint16_t* col,
                                                const uint16_t* aux, const float* Y1, const float* Y2, int zero1,
                                                int zero2, int f) {
#pragma unroll
    for (int u = 0; u < NR; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < steps; q += 2) {
        int cj[NR][2], cc[NR][2];
        float4 a[NR][2], b[NR][2];
#pragma unroll
        for (int u = 0; u < NR; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int p = R.s[u] + q + v, pc = min(p, max(R.e[u] - 1, 0));
                const int j = col[pc], c = aux[pc];
                cj[u][v] = p < R.e[u] ? j : zero1;
                cc[u][v] = p < R.e[u] ? c : zero2;
            }
#pragma unroll
        for (int u = 0; u < NR; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                a[u][v] = lds4(Y1 + (size_t)cj[u][v] * YS + f);
                b[u][v] = lds4(Y2 + (size_t)cc[u][v] * YS + f);
            }
#pragma unroll
        for (int u = 0; u < NR; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                acc[u].x += fmaxf(a[u][v].x + b[u][v].x, 0.0f);
                acc[u].y += fmaxf(a[u][v].y + b[u][v].y, 0.0f);
                acc[u].z += fmaxf(a[u][v].z + b[u][v].z, 0.0f);
                acc[u].w += fmaxf(a[u][v].w + b[u][v].w, 0.0f);
            }
    }
}

__device__ __forceinline__ float4 axpy4(const float4& acc, float s, const float4& x) {   // acc + s * x, unfused
    return make_float4(acc.x + s * x.x, acc.y + s * x.y, acc.z + s * x.z, acc.w + s * x.w);
}

#if CWN_W8
#define CWN_LAYER_OCCUPANCY __attribute__((amdgpu_waves_per_eu(4, 4)))      // 128 VGPRs: two workgroups of 8 waves a CU
#else
#define CWN_LAYER_OCCUPANCY
#endif
// The first three arguments are COPIES of A.items / A.set_start1 / A.set_start2: everything the workgroup needs to request its
// item record.  As leading scalar arguments they can be PRELOADED into SGPRs by the dispatcher (csrc/Makefile:
// -mllvm -amdgpu-kernarg-preload-count; a by-value struct is never preloaded -- round 4's experiment with the flag moved
// nothing because nothing was preloaded), so the record request does not wait for a scalar load of the kernel-argument
// segment first: one dependent memory round trip less in front of the chain.  kArgsOff = where A then starts.
constexpr int kArgsOff = 16;
template <int F, int MODE>
__global__ __launch_bounds__(kThreads) CWN_LAYER_OCCUPANCY void layer_kernel(const int32_t* items_pre, int32_t set_start1_pre,
                                                                             int32_t set_start2_pre, LayerArgs A) {
    using G = Geo<F>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;

    CWN_STAMP(0);
    CWN_WSTAMP(0);           // every wave: when it starts
    CWN_RSTAMP(0);
    // ---- 1. item record and set record: ONE round trip ------------------------------------------------
    // lane l reads word l of the item and 8-byte field l of the set record (the kernel-argument segment
    // is ordinary global memory); fields are broadcast with v_readlane where they are used, so they
    // occupy two VGPRs instead of ~60 SGPRs (the scalar-load form of this spilled SGPRs to VGPR lanes
    // and back: 130 v_readlane / v_writelane in the load phase alone).  The set follows from the
    // workgroup index (items are ordered by set), so the two loads do not depend on each other.
    const int set = ((int)blockIdx.x >= set_start1_pre ? 1 : 0) + ((int)blockIdx.x >= set_start2_pre ? 1 : 0);
    const int32_t itv = items_pre[(size_t)blockIdx.x * CWN_LAYER_ITEM_INTS + (lane & (CWN_LAYER_ITEM_INTS - 1))];
    const uint64_t srec = ((gcu64_p)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + kArgsOff))[set * kSetFields + min(lane, kSetFields - 1)];
    // The packed weight depends on the SET only, and while the two records travel (and are taken apart) the
    // address unit of this CU has nothing to do: the first kWEarly k-steps of this wave's slice are requested
    // now, their address from ONE scalar load of the set record's weight field.  (All of it here was
    // measured worse: the rows then queue behind 192 KB.)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);       // wave-uniform: scalar registers from here on
    const int ct = wave_u % G::kNCT, w2 = wave_u / G::kNCT;
    const int my_h = kHS == 2 ? (w2 & 1) : 0, rt_par = w2 / kHS;   // this wave's product (kHS == 2), row-tile parity
    uint4 wsp[kWSets][G::kKS][3];
    constexpr int kWEarly = CWN_LAYER_WEARLY < 0 ? G::kKS / 2 : (CWN_LAYER_WEARLY < G::kKS ? CWN_LAYER_WEARLY : G::kKS);
    __builtin_amdgcn_sched_barrier(0);       // both record loads leave before the scalar load below is waited for
    const uint64_t wp_bits =
        ((const __attribute__((address_space(4))) uint64_t*)((const __attribute__((address_space(4))) unsigned char*)
                                                             __builtin_amdgcn_kernarg_segment_ptr() + kArgsOff))[set * kSetFields + S_WP];
    // chunk (ks, plane) of product h and column tile ct is the 1-KiB block number ((ks * 3 + plane) * 2 + h) * kNCT + ct:
    // the waves of a workgroup, which walk their chunks in step, read CONSECUTIVE kilobytes (all L2 channels)
    // instead of sixteen blocks 24 KB apart (a multiple of the channel interleave: the same few channels)
    // address = scalar base of (column tile, product) + one 32-bit offset per lane and chunk (an add each)
    const gcb_p wbase = wp_bits != 0 ? (gcb_p)wp_bits + (size_t)ct * 1024 : (gcb_p)A.items;
    const uint32_t won = wp_bits != 0 ? 1024u : 0u;
    const uint32_t wlane = (uint32_t)lane * 16;
    auto wload = [&](int h, int ks, int pl) {
#if defined(CWN_LAYER_EXP_NOWEIGHT) && CWN_LAYER_EXP_NOWEIGHT == 2      // headroom experiment (wrong results): no weight traffic at all
        return make_uint4(wlane + (uint32_t)ks, wlane ^ (uint32_t)pl, wlane + (uint32_t)h, wlane);
#elif defined(CWN_LAYER_EXP_NOWEIGHT)                                    // ... every chunk = the same KiB (L1 hits; the address unit still works)
        return ldgu4o(wbase, wlane + (uint32_t)(((ks * 3 + pl) * 2 + h) * G::kNCT) * 0u * won);
#else
        return ldgu4o(wbase, wlane + (uint32_t)(((ks * 3 + pl) * 2 + h) * G::kNCT) * won);
#endif
    };
#pragma unroll
    for (int hh = 0; hh < kWSets; ++hh) {
        const int h = kHS == 2 ? my_h : hh;
#pragma unroll
        for (int ks = 0; ks < kWEarly; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wsp[hh][ks][pl] = wload(h, ks, pl);
    }
    __builtin_amdgcn_sched_barrier(0);       // ... and the early weight requests leave before the records are waited for
    const int srec_lo = (int)(uint32_t)srec, srec_hi = (int)(uint32_t)(srec >> 32);
    auto fld = [&](int k) { return __builtin_amdgcn_readlane(itv, k); };
    auto sfld = [&](int k) {
        return (uint64_t)(uint32_t)__builtin_amdgcn_readlane(srec_lo, k) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(srec_hi, k) << 32);
    };
    // The record carries every number the workgroup would otherwise DERIVE (first coface row, staged rows,
    // entry segments: written by the table builder, checked on the host by cwn_layer_items_check): with
    // four waves per SIMD an instruction costs an issue slot in each of them, and 16 waves deriving the
    // same scalars took longer than the loads they lead to.  Absent tasks / products are all-zero fields.
    const bool has_gemm = (fld(I_FLAGS) & 1) != 0;
    // an EMPTY record (no task): a table of fixed capacity that this batch does not fill (cwn_amd/static_graph.py,
    // cwn_layer_items_build_dev) -- uniform over the workgroup, before any barrier; the early weight requests are dropped
    if (fld(I_NT) == 0) { CWN_RSTAMP(1); return; }
    CWN_STAMP(9);
    int t_r0[2], t_n[2], t_bne[2], t_sn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int o = I_TASK0 + t * T_INTS;
        t_r0[t] = fld(o + T_R0);
        t_n[t] = fld(o + T_N);
        t_bne[t] = fld(o + T_BNE);
        t_sn[t] = fld(o + T_SN);                 // 0 unless the task has boundary entries (sources are staged)
    }
    // staged rows: [0, g_n) the cells of task 0 (the GEMM dimension g when the item has one), then, from
    // row R1 (a multiple of 16: whole MFMA tiles, whole waves of the load rounds), the c_n cells of g + 1
    const int g_n = t_n[0], c_n = fld(I_CN);
    const int R1 = fld(I_R1), rows_pad = fld(I_ROWS);
    // segments of the item's combined entry list, each starting at a multiple of 4 (ds_read_b128 of
    // four keys): [0, u_ne) upper, [b1, b1 + t_bne[0]) boundary of task 0, [b2, b2 + t_bne[1]) of task 1
    const int b1 = fld(I_B1), b2 = fld(I_B2), total = fld(I_TOTAL);
    // boundary sources in LDS: [0, t_sn[0]) cells of dim g-1 (loaded), then the g_n cells of dim g (copied
    // from the staged rows) when task 1 reads them
    // (two-per-CU form: task 1 reads the cells of g out of the bf16 planes -- hi + mid + lo is x, exactly -- so that
    // only the loaded sources take LDS; 80 KiB do not hold a second fp32 copy of the edges)
    const int x_rows = kW8 ? t_sn[0] : t_sn[0] + t_sn[1];
    // LDS layout.  16-wave form: one layout per LAUNCH (the largest staged block and the most sources of any item).
    // Two-per-CU form: every item lays out ITS OWN rows (a vertex item has many staged rows and no sources, an
    // edges + rings item the other way round: one shared layout would need the sum of both maxima).
    const int rows_cap = kW8 ? rows_pad : A.rows_cap;
    const int xrows_cap = kW8 ? x_rows : A.xrows_cap;
    uint16_t* const planes = reinterpret_cast<uint16_t*>(smem);                    // [3][rows_cap][F + 8]
    float* const Y = reinterpret_cast<float*>(smem);                               // [rows_cap][F + 4], later
    float* const xsrc = reinterpret_cast<float*>(smem + G::planes_bytes(rows_cap)); // [xrows_cap][F]
    unsigned char* const idx = smem + G::planes_bytes(rows_cap) + G::xrows_bytes(xrows_cap);
    uint32_t* const ecnt = reinterpret_cast<uint32_t*>(idx);        // [3][kRpStride] entries per destination row (sort modes)
    uint16_t* const eslot = reinterpret_cast<uint16_t*>(idx + kCntBytes);   // [kEcap] entry numbers grouped by row (sort modes)
    uint16_t* const scol = eslot + kEcap;                           // sorted by destination, stable: local source row
    uint16_t* const saux = eslot + 2 * kEcap;                       //                                 local shared (coface) row
    uint16_t* const rowptr = eslot + 3 * kEcap;  // [3][kRpStride]: upper, boundary of task 0, of task 1
    const int gl = tid % G::kG, gq = tid / G::kG, f = gl * 4;  // lane group gq finishes rows gq, gq + kNG, ...
    constexpr uint32_t kRowB = F * 4;                          // bytes per row
    const uint32_t fB = (uint32_t)f * 4;
    CWN_STAMP(10);
    // ---- BIG item: a complex that no workgroup's LDS holds is STREAMED by this workgroup (include/cwn_hip.h) ---------
    // Uniform over the workgroup.  Y1 / Y2 of the complex go row tile by row tile through the matrix cores straight
    // from the fp32 rows (fragment-shaped loads, split on the fly: this wave's weight slice is the stationary operand
    // as ever) into the scratch matrices in global memory; then every lane group walks the rows of the complex through
    // the caller's CSR of its entries (global cell numbers) like the streaming path's aggregation does.  Same split,
    // same MFMA order per tile, same entry order, same epilogue arithmetic: bit-identical to every other path.  It
    // is the slow way to do a complex (~2.5 x the matrix-pipe time for the redundant splits, L2 round trips instead
    // of LDS) -- and it keeps ONE oversized molecule from sending its whole batch to the two-kernel path.
    if (__builtin_expect((fld(I_FLAGS) & CWN_LAYER_ITEM_BIG) != 0, 0)) {
        if constexpr (kW8) {
            if (tid == 0) atomicOr(A.err, CWN_ERR_BIT_BLOCK);      // big items ride in the 16-wave form only
            return;
        } else {
            const BigSet& Bg = A.big[set];
            if (Bg.y1 == nullptr && has_gemm) {                    // the caller gave no scratch / CSR: refuse, do not fault
                if (tid == 0) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
                return;
            }
#pragma unroll
            for (int hh = 0; hh < kWSets; ++hh) {                  // the rest of this wave's weight slice
                const int h = kHS == 2 ? my_h : hh;
#pragma unroll
                for (int ks = kWEarly; ks < G::kKS; ++ks)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wsp[hh][ks][pl] = wload(h, ks, pl);
            }
            const int c_r0 = fld(I_CR0);
            constexpr int kCH = 32;                                   // rows of EACH product per chunk
            uint16_t* const bplanes = reinterpret_cast<uint16_t*>(smem);      // [3][2 * kCH][F + 8]
            constexpr size_t bplane = (size_t)2 * kCH * G::kPlaneStride;
            if (has_gemm) {
                // Y1 | Y2 in chunks of 32 + 32 rows: coalesced row loads (the next chunk's are in flight during this
                // chunk's MFMAs), ONE split per element into bf16 planes in LDS, fragments out of LDS -- the normal
                // path's phases 2 / 4 / 6 in a loop, the results to the scratch matrices instead of LDS
                const gcf_p bias_p = (gcf_p)sfld(S_MSG_BIAS);
                float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias_p != (gcf_p)0) bb = ldg4(bias_p + ct * 16 + kq * 4);
                const gcb_p xg_b = (gcb_p)sfld(S_XG) + (size_t)t_r0[0] * kRowB, xc_b = (gcb_p)sfld(S_XC) + (size_t)c_r0 * kRowB;
                constexpr int kPer = 2 * kCH / G::kNG;                // float4 per thread per chunk (2 at F = 128, 1 at F = 64)
                // chunk-local row of this thread's load i: rows [0, 32) belong to product 0, [32, 64) to product 1
                auto request = [&](float4 (&v)[kPer], int c0) {
#pragma unroll
                    for (int i = 0; i < kPer; ++i) {
                        const int lr = gq + i * G::kNG, hh_ = lr / kCH, rr = c0 + (lr - hh_ * kCH);
                        const int n_h = hh_ == 0 ? g_n : c_n;
                        v[i] = ldg4o(hh_ == 0 ? xg_b : xc_b, (uint32_t)min(rr, max(n_h - 1, 0)) * kRowB + fB);
                    }
                };
                float4 cur[kPer], nxt[kPer];
                const int n_max = max(g_n, c_n);
                request(cur, 0);
                for (int c0 = 0; c0 < n_max; c0 += kCH) {
                    __syncthreads();                                  // the fragments of the previous chunk have been read
#pragma unroll
                    for (int i = 0; i < kPer; ++i) {
                        const int lr = gq + i * G::kNG;
                        uint2 ph, pm, pl;
                        cwn::split4(cur[i], ph, pm, pl);
                        uint16_t* dst = bplanes + (size_t)lr * G::kPlaneStride + f;
                        *reinterpret_cast<uint2*>(dst) = ph;
                        *reinterpret_cast<uint2*>(dst + bplane) = pm;
                        *reinterpret_cast<uint2*>(dst + 2 * bplane) = pl;
                    }
                    if (c0 + kCH < n_max) request(nxt, c0 + kCH);
                    __syncthreads();
#pragma unroll
                    for (int hh = 0; hh < kWSets; ++hh) {
                        const int h = kHS == 2 ? my_h : hh;
                        const int rows_h = h == 0 ? g_n : c_n;
                        float* const yb = h == 0 ? Bg.y1 + (size_t)t_r0[0] * F : Bg.y2 + (size_t)c_r0 * F;
                        for (int tl = rt_par; tl < kCH / 16; tl += G::kWPC) {
                            if (c0 + tl * 16 >= rows_h) break;
                            const uint16_t* p0 = bplanes + (size_t)(h * kCH + tl * 16 + l15) * G::kPlaneStride + kq * 8;
                            frag_cd c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int ks = 0; ks < G::kKS; ++ks) {
                                const uint4 xh = *reinterpret_cast<const uint4*>(p0 + ks * 32);
                                const uint4 xm = *reinterpret_cast<const uint4*>(p0 + bplane + ks * 32);
                                const uint4 xl = *reinterpret_cast<const uint4*>(p0 + 2 * bplane + ks * 32);
                                c = cwn::mfma_split6(wsp[hh][ks][0], wsp[hh][ks][1], wsp[hh][ks][2], xh, xm, xl, c);
                            }
                            if (h == 0) { c[0] += bb.x; c[1] += bb.y; c[2] += bb.z; c[3] += bb.w; }
                            const int row = c0 + tl * 16 + l15;
                            if (row < rows_h)
                                *reinterpret_cast<float4*>(yb + (size_t)row * F + ct * 16 + kq * 4) = make_float4(c[0], c[1], c[2], c[3]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < kPer; ++i) cur[i] = nxt[i];
                }
            }
            // the Y rows of this workgroup's other waves: written to global memory, read back below
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // the complex's slices of the caller's CSR (its rows are contiguous, so are their segments) come into LDS in
            // one coalesced pass when they fit; the row loops below then chase LDS instead of global memory
            int32_t* const lidx = reinterpret_cast<int32_t*>(smem);
            const int lcap = A.lds_limit / 4;
            int lused = 0;
            // (positions are the caller's CSR positions; an LDS copy starts at position `off` -- subtracted at every access: a
            // pointer moved BELOW its LDS array is a 32-bit wrap that the cast to a generic pointer does not undo)
            struct Seg { const int32_t* rp; const int32_t* cl; const int32_t* ax; int off; };
            auto stage = [&](const int32_t* rp_g, const int32_t* cl_g, const int32_t* ax_g, int first, int n_rows) {
                if (rp_g == nullptr || n_rows <= 0) return Seg{nullptr, nullptr, nullptr, 0};
                Seg sg{rp_g + first, cl_g, ax_g, 0};
                const int s0 = rp_g[first], e0 = rp_g[first + n_rows], ne = e0 - s0;
                const int need = (n_rows + 1) + ne * (ax_g != nullptr ? 2 : 1);
#ifdef CWN_BIG_NO_STAGE
                if (true) return sg;
#endif
                if (ne < 0 || lused + need > lcap) return sg;          // does not fit: chase global memory (correct, slower)
                int32_t* lrp = lidx + lused;
                int32_t* lcl = lrp + n_rows + 1;
                int32_t* lax = lcl + ne;
                for (int i = tid; i <= n_rows; i += kThreads) lrp[i] = rp_g[first + i];
                for (int i = tid; i < ne; i += kThreads) {
                    lcl[i] = cl_g[s0 + i];
                    if (ax_g != nullptr) lax[i] = ax_g[s0 + i];
                }
                lused += need;
                return Seg{lrp, lcl, ax_g != nullptr ? lax : nullptr, s0};
            };
            const Seg su = (has_gemm) ? stage(Bg.up_rowptr, Bg.up_col, Bg.up_aux, t_r0[0], g_n) : Seg{nullptr, nullptr, nullptr, 0};
            Seg sb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
                sb[t] = t_bne[t] > 0 ? stage(Bg.b_rowptr[t], Bg.b_col[t], nullptr, t_r0[t], t_n[t]) : Seg{nullptr, nullptr, nullptr, 0};
            __syncthreads();
            bool bad_big = false;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t_n[t] == 0) continue;                        // uniform
                const gcf_p e1p = (gcf_p)sfld(S_TASK0 + t * ST_FIELDS + ST_EPS1), e2p = (gcf_p)sfld(S_TASK0 + t * ST_FIELDS + ST_EPS2);
                const float eps1 = e1p != (gcf_p)0 ? *e1p : 0.f, eps2 = e2p != (gcf_p)0 ? *e2p : 0.f;
                const gcf_p e3p = (gcf_p)sfld(S_TASK0 + t * ST_FIELDS + ST_EPS3);
                const float eps3 = e3p != (gcf_p)0 ? *e3p : 0.f;
                const gb_p o_d = (gb_p)sfld(S_TASK0 + t * ST_FIELDS + ST_OUT_D);
                const gcb_p xt = (gcb_p)sfld(S_TASK0 + t * ST_FIELDS + ST_X), xs = (gcb_p)sfld(S_TASK0 + t * ST_FIELDS + ST_XS);
                const gb_p o_up = (gb_p)sfld(S_TASK0 + t * ST_FIELDS + ST_OUT_UP), o_b = (gb_p)sfld(S_TASK0 + t * ST_FIELDS + ST_OUT_B);
                const Seg B_ = sb[t];
                const bool upper = t == 0 && has_gemm && su.rp != nullptr;
                const int s_lo = fld(I_TASK0 + t * T_INTS + T_SR0), s_n = t_sn[t];
                for (int r = gq; r < t_n[t]; r += G::kNG) {
                    const int64_t row = (int64_t)t_r0[t] + r;
                    const float4 xi = ldg4o(xt, (uint32_t)row * kRowB + fB);
                    float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (B_.rp != nullptr) {
                        const int s_ = B_.rp[r], e_ = B_.rp[r + 1];
                        for (int p = s_; p < e_; p += 4) {              // four source rows in flight, added in entry order
                            int v[4];
                            float4 a[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                v[u] = B_.cl[min(p + u, e_ - 1) - B_.off];
                                if ((unsigned)(v[u] - s_lo) >= (unsigned)s_n) { bad_big = true; v[u] = s_lo; }
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) a[u] = ldg4o(xs, (uint32_t)v[u] * kRowB + fB);
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (p + u < e_) { ab.x += a[u].x; ab.y += a[u].y; ab.z += a[u].z; ab.w += a[u].w; }
                        }
                    }
                    stg4o(o_b, (uint32_t)row * kRowB + fB, axpy4(ab, 1.0f + eps2, xi));
                    if (o_d != (gb_p)0)
                        stg4o(o_d, (uint32_t)row * kRowB + fB, axpy4(make_float4(0.f, 0.f, 0.f, 0.f), 1.0f + eps3, xi));
                    float4 au = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (upper) {
                        const int s_ = su.rp[r], e_ = su.rp[r + 1];
                        for (int p = s_; p < e_; p += 4) {
                            int j[4], ci[4];
                            float4 y[4], z[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                j[u] = su.cl[min(p + u, e_ - 1) - su.off];
                                ci[u] = su.ax[min(p + u, e_ - 1) - su.off];
                                if ((unsigned)(j[u] - t_r0[0]) >= (unsigned)g_n || (unsigned)(ci[u] - c_r0) >= (unsigned)c_n) {
                                    bad_big = true;
                                    j[u] = t_r0[0]; ci[u] = c_r0;
                                }
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                y[u] = *reinterpret_cast<const float4*>(Bg.y1 + (size_t)j[u] * F + f);
                                z[u] = *reinterpret_cast<const float4*>(Bg.y2 + (size_t)ci[u] * F + f);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (p + u < e_) {
                                    au.x += fmaxf(y[u].x + z[u].x, 0.0f); au.y += fmaxf(y[u].y + z[u].y, 0.0f);
                                    au.z += fmaxf(y[u].z + z[u].z, 0.0f); au.w += fmaxf(y[u].w + z[u].w, 0.0f);
                                }
                        }
                    }
                    stg4o(o_up, (uint32_t)row * kRowB + fB, axpy4(au, 1.0f + eps1, xi));
                }
            }
            if (bad_big) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
            CWN_RSTAMP(1);
            return;
        }
    }
    // What stays checked here: that the record fits the LDS of this launch (memory safety inside the
    // workgroup).  Uniform over the workgroup; unsigned compares also catch negative fields.
    const bool fits = kW8 ? ((unsigned)rows_pad <= (unsigned)gemm_rows_cap(F) && (unsigned)x_rows <= (unsigned)source_rows_cap(F) &&
                             lds_bytes<F>(rows_pad, x_rows) <= (size_t)A.lds_limit)
                          : ((unsigned)rows_pad <= (unsigned)rows_cap && (unsigned)x_rows <= (unsigned)xrows_cap);
    if (!fits || (unsigned)total > (unsigned)kEcap ||
        (unsigned)t_n[0] > (unsigned)kTaskRows || (unsigned)t_n[1] > (unsigned)kTaskRows || (unsigned)R1 > (unsigned)rows_pad) {
        if (tid == 0) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
        return;
    }
    bool bad = false;
    // sort modes: the per-row entry counters of the three adjacencies are zeroed here and visible to every wave
    // behind the barrier in front of the late weight requests
    if constexpr (MODE != kLoad) {
        for (int i = tid; i < 3 * kRpStride; i += kThreads) ecnt[i] = 0u;
    }

    // ---- 2. every global load of the item, in one run; no load behind a DIVERGENT branch ---------------
    // A lane with nothing to fetch reads a harmless valid address (the item table) instead.  Behind
    // divergent branches the compiler cannot count the loads in flight and waits with vmcnt(0) -- i.e.
    // for the 200 KB of rows and weights issued after the entries -- wherever an early result is used
    // (measured: entries -> LDS and the rank each sat 2 us on such waits).  The row loads that only
    // large items need sit behind UNIFORM guards at the very end of the run, where no earlier load
    // has to be counted past them.
    const gcf_p dummy = (gcf_p)A.items;                      // >= 128 readable bytes, 16-B aligned
    int64_t ek[kEI], ev[kEI], ea[kEI];
    int s1 = 0, s2 = 0, s3 = 0;
    if constexpr (MODE != kLoad) {
        const int u_e0 = fld(I_UE0), t_be0[2] = {fld(I_TASK0 + T_BE0), fld(I_TASK0 + T_INTS + T_BE0)};
        s1 = fld(I_UNE), s2 = b1 + t_bne[0], s3 = b2 + t_bne[1];
        const gci64_p up_index = (gci64_p)sfld(S_UP_INDEX), up_shared = (gci64_p)sfld(S_UP_SHARED);
        const gci64_p b_index0 = (gci64_p)sfld(S_TASK0 + ST_B_INDEX);
        const gci64_p b_index1 = (gci64_p)sfld(S_TASK0 + ST_FIELDS + ST_B_INDEX);
        const int64_t up_E = (int64_t)sfld(S_UP_E);
        const int64_t b_E0 = (int64_t)sfld(S_TASK0 + ST_B_E), b_E1 = (int64_t)sfld(S_TASK0 + ST_FIELDS + ST_B_E);
#pragma unroll
        for (int i = 0; i < kEI; ++i) {
            const int w = tid + i * kThreads;
            const bool in_up = w < s1, in_b0 = w >= b1 && w < s2, in_b1 = w >= b2 && w < s3;
            const gci64_p src = in_up ? up_index : in_b0 ? b_index0 : in_b1 ? b_index1 : (gci64_p)A.items;
            const int64_t E = in_up ? up_E : in_b0 ? b_E0 : in_b1 ? b_E1 : 0;
            const int64_t e = in_up ? (int64_t)u_e0 + w : in_b0 ? (int64_t)t_be0[0] + (w - b1)
                                                       : in_b1 ? (int64_t)t_be0[1] + (w - b2) : 0;
            ev[i] = src[e];
            ek[i] = src[E + e];
            ea[i] = (in_up ? up_shared : (gci64_p)A.items)[in_up ? e : 0];
        }
    }
    uint4 csr_img = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (MODE == kLoad)      // the item's finished CSR, stored by an earlier layer of this batch
        csr_img = ldgu4((gcb_p)A.csr_cache + (size_t)blockIdx.x * kCsrSlot + (size_t)min(tid, kCsrSlot / 16 - 1) * 16);
    float epsv;
    {   // lane 0..3 of every wave -> eps1, eps2 of task 0, eps1, eps2 of task 1; lanes 4, 5 -> eps3 of task 0, 1 (NULL = 0)
        static_assert(ST_EPS2 == ST_EPS1 + 1, "eps1, eps2 are consecutive fields");
        const int k = lane < 4 ? S_TASK0 + ((lane & 2) ? ST_FIELDS : 0) + ST_EPS1 + (lane & 1)
                               : S_TASK0 + ((lane & 1) ? ST_FIELDS : 0) + ST_EPS3;
        const uint64_t bits = (uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(4 * k, srec_lo) |
                              ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(4 * k, srec_hi) << 32);
        const gcf_p ep = lane < 6 ? (gcf_p)bits : (gcf_p)0;
        const float v = *(ep != (gcf_p)0 ? ep : dummy);
        epsv = ep != (gcf_p)0 ? v : 0.0f;
    }
    const gcf_p bias = (gcf_p)sfld(S_MSG_BIAS);
    const bool has_bias = has_gemm && bias != (gcf_p)0;
    CWN_STAMP(11);
    // the staged rows (rows past the real ones re-read the last real row, never used), then the rows the
    // boundary stream of task 0 gathers from; round i is skipped when no item row falls into it
    // (no zero fill: merging a constant with a load result made the compiler copy the loaded register right
    // behind the load -- s_waitcnt vmcnt(0) in the middle of the run)
    constexpr int kNX = G::kNX, kNE = G::kNE;
    float4 xv[kNX], ev4[kNE];
    {
        // uniform base of the block (SGPR pair) + a 32-bit byte offset per lane.  Round i belongs to the
        // coface block iff i >= R1 / kNG (R1 is a multiple of kNG): a scalar decision per round.
        const gcb_p xg = (gcb_p)sfld(S_XG) + (size_t)t_r0[0] * kRowB, xc = (gcb_p)sfld(S_XC) + (size_t)fld(I_CR0) * kRowB;
        const gcb_p xs = (gcb_p)sfld(S_TASK0 + ST_XS) + (size_t)fld(I_TASK0 + T_SR0) * kRowB;
        const int nxr = (rows_pad + G::kNG - 1) / G::kNG, ner = (t_sn[0] + G::kNG - 1) / G::kNG;
        // The staged rows of a WAVE in round i (64 / kG consecutive ones from i * kNG + wave_row0) lie in the coface block
        // iff the first of them is >= R1 -- R1 is a multiple of 16, a wave's rows never straddle it: base, row count and
        // first row of the round are chosen by scalar selects, one compare per round.  (Round 4: R1 used to be a multiple
        // of kNG, the decision one per WORKGROUP and round; 33 vertices then padded to 64 staged rows and a molecule of
        // more than 32 atoms did not fit the 96 rows of F = 128 -- the real ZINC subset has ~2 % of them.)
        const int wave_row0 = wave_u * (64 / G::kG);
        const uint64_t xg_b = (uint64_t)(uintptr_t)xg, xc_b = (uint64_t)(uintptr_t)xc;
#pragma unroll
        for (int i = 0; i < kNX; ++i) {
            if (i < nxr) {           // a skipped round leaves xv[i] unset: it is never read (rows >= rows_pad)
                // (16 rows per round -- the two-per-CU form at F = 128: R1 is whole rounds, nothing per wave)
                const bool second = c_n > 0 && i * G::kNG + (G::kNG > 16 ? wave_row0 : 0) >= R1;
                const gcb_p base = (gcb_p)(second ? xc_b : xg_b);
                const int last = (second ? c_n : g_n) - 1, first = second ? i * G::kNG - R1 : i * G::kNG;
                xv[i] = ldg4o(base, (uint32_t)max(min(gq + first, last), 0) * kRowB + fB);
            }
        }
#pragma unroll
        for (int i = 0; i < kNE; ++i) {
            if (i < ner) ev4[i] = ldg4o(xs, (uint32_t)min(gq + i * G::kNG, t_sn[0] - 1) * kRowB + fB);
        }
    }
    CWN_STAMP(12);
    // the rest of this wave's slice of the packed weight: chunks of 1 KiB, lane l takes bytes 16 l .. 16 l + 15.
    // LAST: it is the bulk of the bytes (192 KB per workgroup at F = 128) and only the matrix cores need it,
    // so the rows are split and phase 5 runs while it is still landing.  The barrier (no memory wait: it only
    // lines the waves up) keeps every wave's ROW requests ahead of every wave's weight requests in the
    // address unit's queue -- without it the rows of the wave that issues last arrive behind the weights
    // of the fifteen others, and the split phase waits for most of the weight (measured: 3.3 k cycles).
    CWN_WSTAMP(1);           // every wave: its row requests are out
    if constexpr (MODE != kLoad) {           // the zeroed counters must have landed before any wave adds to them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else if (CWN_LAYER_WBAR) {
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int hh = 0; hh < kWSets; ++hh) {
        const int h = kHS == 2 ? my_h : hh;
#pragma unroll
        for (int ks = kWEarly; ks < G::kKS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wsp[hh][ks][pl] = wload(h, ks, pl);
    }
    CWN_STAMP(1);

    unsigned char* const csr_lds = reinterpret_cast<unsigned char*>(scol);     // [scol | saux | rowptr]
    if constexpr (MODE == kLoad) {
        // ---- 3'. the finished CSR of this item comes back from the cache: one 16-byte load per thread ---
        if (tid < kCsrSlot / 16) *reinterpret_cast<uint4*>(csr_lds + (size_t)tid * 16) = csr_img;
        CWN_STAMP(2);
        CWN_STAMP(3);
    } else {
        // ---- 3a. entries as local row numbers, range-checked; counted per destination row --------------------
        // Stable sort by destination as a BUCKET sort: an LDS atomic per entry counts its row and hands the
        // entry a slot in it (arrival order); a scan of the counters gives the row pointers; the entries of
        // a row then order themselves by entry number (a row has a handful of entries: each compares itself
        // with its row mates).  The first form ranked every entry against its whole adjacency -- ~200
        // instructions per wave where this takes ~50; the order of the result is the same.
        const int g_r0 = t_r0[0], c_r0 = fld(I_CR0);
        const int t_sr0[2] = {fld(I_TASK0 + T_SR0), fld(I_TASK0 + T_INTS + T_SR0)};
        int e_seg[kEI], e_row[kEI], e_val[kEI], e_aux[kEI], e_pos[kEI];
    #pragma unroll
        for (int i = 0; i < kEI; ++i) {
            const int w = tid + i * kThreads;
            e_seg[i] = -1;
            e_row[i] = e_val[i] = e_aux[i] = e_pos[i] = 0;
            if (w < total) {
                int64_t k = 0, v = 0, a = 0, nk = 1, nv = 1, na = 1;
                int seg = -1;
                if (w < s1) {
                    seg = 0;
                    k = ek[i] - g_r0; v = ev[i] - g_r0; a = ea[i] - c_r0;
                    nk = g_n; nv = g_n; na = c_n;
                } else if (w >= b1 && w < s2) {
                    seg = 1;
                    k = ek[i] - t_r0[0]; v = ev[i] - t_sr0[0];
                    nk = t_n[0]; nv = t_sn[0];
                } else if (w >= b2 && w < s3) {
                    seg = 2;
                    k = ek[i] - t_r0[1]; v = ev[i] - t_sr0[1];
                    nk = t_n[1];
                    v = (v >= 0 && v < t_sn[1]) ? v + t_sn[0] : -1;   // row in the staged source block
                    nv = t_sn[0] + t_sn[1];
                }                                  // else: a padding slot between two segments
                if (seg >= 0 && (k < 0 || k >= nk || v < 0 || v >= nv || a < 0 || a >= na)) {
                    bad = true;
                    k = 0; v = 0; a = 0;
                }
                if (seg >= 0) {
                    e_seg[i] = seg; e_row[i] = (int)k; e_val[i] = (int)v; e_aux[i] = (int)a;
                    e_pos[i] = (int)atomicAdd(&ecnt[seg * kRpStride + (int)k], 1u);
                }
            }
        }
        if (bad) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
        __syncthreads();
        CWN_STAMP(2);

        // ---- 3b. row pointers: exclusive scan of the counters, one wave per adjacency ---------------------
        if (wave < 3) {
            static_assert(kRpStride <= 4 * 64, "four rows per lane");
            const int n_rows = wave == 0 ? g_n : t_n[wave - 1];
            const uint32_t* c = ecnt + wave * kRpStride;
            uint16_t* rp = rowptr + wave * kRpStride;
            int cj[4], sum = 0;
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * lane + j;
                cj[j] = r < n_rows ? (int)c[r] : 0;
                sum += cj[j];
            }
            int incl = sum;
    #pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            int run = incl - sum;
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * lane + j;
                if (r <= n_rows) rp[r] = (uint16_t)run;      // rp[n_rows] = number of entries
                run += cj[j];
            }
        }
        __syncthreads();

        // ---- 3c. entries into their rows (arrival order) ... ------------------------------------------------
    #pragma unroll
        for (int i = 0; i < kEI; ++i) {
            if (e_seg[i] >= 0) {
                const int seg0 = e_seg[i] == 0 ? 0 : (e_seg[i] == 1 ? b1 : b2);
                eslot[seg0 + rowptr[e_seg[i] * kRpStride + e_row[i]] + e_pos[i]] = (uint16_t)(tid + i * kThreads);
            }
        }
        __syncthreads();
        // ---- ... and into entry order inside the row: its position = the row mates with a smaller number ---
    #pragma unroll
        for (int i = 0; i < kEI; ++i) {
            if (e_seg[i] >= 0) {
                const int w = tid + i * kThreads;
                const int seg0 = e_seg[i] == 0 ? 0 : (e_seg[i] == 1 ? b1 : b2);
                const uint16_t* rp = rowptr + e_seg[i] * kRpStride + e_row[i];
                const int lo = seg0 + rp[0], hi = seg0 + rp[1];
                int less = 0;
                for (int p = lo; p < hi; ++p) less += (int)eslot[p] < w ? 1 : 0;
                scol[lo + less] = (uint16_t)e_val[i];
                saux[lo + less] = (uint16_t)e_aux[i];
            }
        }
        CWN_STAMP(3);
    }

    // ---- 4. GEMM rows -> three bf16 planes; boundary-source rows -> fp32 -------------------------------
    {
        const size_t plane = (size_t)rows_cap * G::kPlaneStride;
        const bool copy_src = !kW8 && t_sn[1] > 0;      // task 1 gathers the staged cells of g: keep them as fp32 too
#pragma unroll
        for (int i = 0; i < kNX; ++i) {
            const int row = gq + i * G::kNG;
            // padding rows of a tile are left as they are: a column of the X operand reaches only its own
            // column of the product, and no entry names a padding row
            if (has_gemm && (row < g_n || (row >= R1 && row < R1 + c_n))) {
                uint2 ph, pm, pl;
                cwn::split4(xv[i], ph, pm, pl);
                uint16_t* dst = planes + (size_t)row * G::kPlaneStride + f;
                *reinterpret_cast<uint2*>(dst) = ph;
                *reinterpret_cast<uint2*>(dst + plane) = pm;
                *reinterpret_cast<uint2*>(dst + 2 * plane) = pl;
            }
            if (copy_src && row < g_n) *reinterpret_cast<float4*>(xsrc + (size_t)(t_sn[0] + row) * F + f) = xv[i];
        }
#pragma unroll
        for (int i = 0; i < kNE; ++i) {
            const int row = gq + i * G::kNG;
            if (row < t_sn[0]) *reinterpret_cast<float4*>(xsrc + (size_t)row * F + f) = ev4[i];
        }
        if (gq == 0) *reinterpret_cast<float4*>(xsrc + (size_t)x_rows * F + f) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    CWN_WSTAMP(2);           // every wave: its rows are split and staged
    __syncthreads();
    if constexpr (MODE == kSortStore) {     // the finished CSR goes to the cache for the next layers of this batch
        if (tid < kCsrSlot / 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(csr_lds + (size_t)tid * 16);
            const v4u vv = {v.x, v.y, v.z, v.w};
#if CWN_LAYER_NT_STORE
            __builtin_nontemporal_store(vv, (gu4_p)((gb_p)A.csr_cache + (size_t)blockIdx.x * kCsrSlot + (size_t)tid * 16));
#else
            *(gu4_p)((gb_p)A.csr_cache + (size_t)blockIdx.x * kCsrSlot + (size_t)tid * 16) = vv;
#endif
        }
    }
    CWN_STAMP(4);

    float eps1[2], eps2[2];
    eps1[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, epsv), 0));
    eps2[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, epsv), 1));
    eps1[1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, epsv), 2));
    eps2[1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, epsv), 3));
    const float eps3[2] = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, epsv), 4)),
                           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, epsv), 5))};
    // ---- 5. boundary stream and self terms of BOTH tasks in one pass, out of LDS (W is still landing) ---
    // Lane group gq finishes row gq + k kNG of task 0 and of task 1 together: two independent chains
    // (row pointers -> source numbers -> source rows) in flight, where two passes ran them one after the
    // other (measured: 4.1 k cycles for an edges + rings item, most of it LDS round trips).
#ifndef CWN_LAYER_NR
#define CWN_LAYER_NR (CWN_LAYER_THREADS == 1024 ? 1 : 2)
#endif
    constexpr int kNR = CWN_LAYER_NR;                          // destination rows per lane group in flight (phase 7)
    {
        gb_p out_up[2], out_b[2], out_d[2];
        bool has_d[2];                               // (uniform) the layer wants out_down = (1 + eps3) x of this task's rows
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            out_up[t] = (gb_p)sfld(S_TASK0 + t * ST_FIELDS + ST_OUT_UP) + (size_t)t_r0[t] * kRowB;
            out_b[t] = (gb_p)sfld(S_TASK0 + t * ST_FIELDS + ST_OUT_B) + (size_t)t_r0[t] * kRowB;
            const uint64_t od = sfld(S_TASK0 + t * ST_FIELDS + ST_OUT_D);
            has_d[t] = od != 0;
            out_d[t] = (gb_p)od + (size_t)t_r0[t] * kRowB;
        }
        const uint16_t* const cols[2] = {scol + b1, scol + b2};
        // Task 1's cells are the staged rows from R1 on, and a lane group finishes the rows IT loaded (their self terms wait in
        // its registers): cell r of task 1 is staged row R1 + r, loaded by lane group (R1 + r) % kNG in round (R1 + r) / kNG.
        // So lane group gq takes the cells r = (gq - R1) mod kNG + k kNG of task 1 -- the cells gq + k kNG when R1 is a whole
        // number of rounds (every item before round 4).
        const int sh = G::kNG > 16 ? R1 % G::kNG : 0, k0 = R1 / G::kNG;
        const int gq1 = gq >= sh ? gq - sh : gq - sh + G::kNG, carry1 = gq >= sh ? 0 : 1;
        const bool any_b = t_bne[0] + t_bne[1] > 0; // no boundary entries at all (vertices): self terms only
        const int rounds = (max(t_n[0], t_n[1]) + G::kNG - 1) / G::kNG;
        for (int k = 0; k < rounds; ++k) {
            const int r = gq + k * G::kNG;
            const int rt_[2] = {r, gq1 + k * G::kNG};
            RowSet<2> R;
            int steps = 0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                R.on[t] = rt_[t] < t_n[t];
                R.s[t] = R.e[t] = 0;
                if (any_b) {
                    const uint16_t* rp = rowptr + (t + 1) * kRpStride;
                    const int rr = R.on[t] ? rt_[t] : 0;
                    const int s_ = rp[rr], e_ = rp[rr + 1];
                    R.s[t] = R.on[t] ? s_ : 0;
                    R.e[t] = R.on[t] ? e_ : 0;
                    steps = max(steps, R.e[t] - R.s[t]);
                }
            }
            float4 acc[2];
            const float4 xi[2] = {pick(xv, k), pick(xv, k0 + k + carry1)};     // self terms: this group loaded them
            if constexpr (kW8)
                gather_sum_planes<F>(acc, R, steps, cols, xsrc, x_rows, planes, (size_t)rows_cap * G::kPlaneStride, t_sn[0], f);
            else
                gather_sum<2, F>(acc, R, steps, cols, xsrc, x_rows, f);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t off = (uint32_t)rt_[t] * kRowB + fB;
                if (R.on[t]) {
                    stg4o(out_b[t], off, axpy4(acc[t], 1.0f + eps2[t], xi[t]));
                    if (!(has_gemm && t == 0))   // no upper adjacency in this dimension: zeros + self term
                        stg4o(out_up[t], off, axpy4(make_float4(0.f, 0.f, 0.f, 0.f), 1.0f + eps1[t], xi[t]));
                    if (!kW8 && has_d[t])
                        stg4o(out_d[t], off, axpy4(make_float4(0.f, 0.f, 0.f, 0.f), 1.0f + eps3[t], xi[t]));
                }
            }
        }
        if constexpr (kW8) {
            // the two-per-CU form lives on exactly 128 registers (the store above, inline, spilled four of them): a pass of its
            // own over the task's rows, re-read from L2 -- taken by CIN++ layers only (uniform branch), same values, same product
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (!has_d[t]) continue;
                const gcb_p xt = (gcb_p)sfld(S_TASK0 + t * ST_FIELDS + ST_X) + (size_t)t_r0[t] * kRowB;
                for (int r = gq; r < t_n[t]; r += G::kNG) {
                    const float4 v = ldg4o(xt, (uint32_t)r * kRowB + fB);
                    stg4o(out_d[t], (uint32_t)r * kRowB + fB, axpy4(make_float4(0.f, 0.f, 0.f, 0.f), 1.0f + eps3[t], v));
                }
            }
        }
    }
    CWN_STAMP(5);
    if (!has_gemm) { CWN_RSTAMP(1); return; }

    // ---- 6. Y1 | Y2 on the matrix cores ---------------------------------------------------------------
    // the bias of this wave's output columns: needed after the MFMAs, requested here (four registers that
    // would otherwise be held through the whole load run, where the register file is full)
    float4 b4 = ldg4(has_bias ? bias + ct * 16 + kq * 4 : dummy);
    if (!has_bias) b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int T1 = (g_n + 15) >> 4, T2 = (c_n + 15) >> 4;      // 16-row tiles of Y1, Y2
    if constexpr (kW8) {
        // Two-per-CU form: this wave multiplies BOTH products of its column tile, Y1 first.  k steps outermost: when
        // the row tiles of a k step are done, its three weight registers are free and the same k step of the SECOND
        // weight is requested into them -- it lands under the remaining MFMAs of the first product.  Per tile the
        // accumulation order (k steps ascending, six terms each) is the 16-wave form's: bit-identical results.
        constexpr int kMaxT = G::kMaxT;
        const size_t plane = (size_t)rows_cap * G::kPlaneStride;
        frag_cd acc[2][kMaxT];
        if ((unsigned)T1 > (unsigned)(kMaxT * G::kWPC) || (unsigned)T2 > (unsigned)(kMaxT * G::kWPC)) {   // table / kernel mismatch
            if (tid == 0) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rt0 = h == 0 ? 0 : R1 / 16, rt1 = rt0 + (h == 0 ? T1 : T2);
            const int first = rt0 + ((rt_par - rt0) % G::kWPC + G::kWPC) % G::kWPC;
#pragma unroll
            for (int j = 0; j < kMaxT; ++j) acc[h][j] = frag_cd{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < G::kKS; ++ks) {
#pragma unroll
                for (int j = 0; j < kMaxT; ++j) {
                    const int rt = first + j * G::kWPC;
                    if (rt < rt1) {                  // uniform over the wave
                        const uint16_t* p0 = planes + (size_t)(rt * 16 + l15) * G::kPlaneStride + kq * 8 + ks * 32;
                        const uint4 xh0 = *reinterpret_cast<const uint4*>(p0);
                        const uint4 xm0 = *reinterpret_cast<const uint4*>(p0 + plane);
                        const uint4 xl0 = *reinterpret_cast<const uint4*>(p0 + 2 * plane);
                        acc[h][j] = cwn::mfma_split6(wsp[0][ks][0], wsp[0][ks][1], wsp[0][ks][2], xh0, xm0, xl0, acc[h][j]);
                    }
                }
                if (h == 0) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wsp[0][ks][pl] = wload(1, ks, pl);
                }
            }
        }
        __syncthreads();                     // every wave has read its fragments: Y may overwrite the planes
        CWN_STAMP(6);
        if (gq == 0) *reinterpret_cast<float4*>(Y + (size_t)rows_cap * G::kYStride + f) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rt0 = h == 0 ? 0 : R1 / 16, rt1 = rt0 + (h == 0 ? T1 : T2);
            const int first = rt0 + ((rt_par - rt0) % G::kWPC + G::kWPC) % G::kWPC;
#pragma unroll
            for (int j = 0; j < kMaxT; ++j) {
                const int rt = first + j * G::kWPC;
                if (rt < rt1) {
                    frag_cd c = acc[h][j];
                    if (h == 0 && has_bias) {
                        c[0] += b4.x; c[1] += b4.y; c[2] += b4.z; c[3] += b4.w;
                    }
                    float* y0 = Y + (size_t)(rt * 16 + l15) * G::kYStride + ct * 16 + kq * 4;
                    *reinterpret_cast<float4*>(y0) = make_float4(c[0], c[1], c[2], c[3]);
                }
            }
        }
    } else {
        const size_t plane = (size_t)rows_cap * G::kPlaneStride;
        // this wave's row tiles: at most kMaxT per half (six 16-row tiles in all at the row cap); the
        // accumulators wait in registers until every wave has read its fragments, because Y
        // overwrites the planes.  Tiles go in PAIRS (two independent MFMA chains in flight).
        constexpr int kMaxT = G::kMaxT;
        static_assert(kMaxT % 2 == 0, "tiles are processed in pairs");
        static_assert(offsetof(LayerArgs, set) == 0, "read through the kernarg segment pointer");
        static_assert(kSetFields <= 64 && CWN_LAYER_ITEM_INTS <= 64, "one lane per field");
        frag_cd acc[kWSets][kMaxT];
#pragma unroll
        for (int hh = 0; hh < kWSets; ++hh) {
            const int h = kHS == 2 ? my_h : hh;
            const uint4 (&wf)[G::kKS][3] = wsp[hh];
            const int rt0 = h == 0 ? 0 : R1 / 16, rt1 = rt0 + (h == 0 ? T1 : T2);
            // row tiles of this half that are this wave's (F = 64: two waves share a column tile)
            const int first = rt0 + ((rt_par - rt0) % G::kWPC + G::kWPC) % G::kWPC;
#pragma unroll
            for (int j = 0; j < kMaxT; j += 2) {
                const int rta = first + j * G::kWPC, rtb = rta + G::kWPC;
                if (rtb < rt1) {                 // two tiles: two independent MFMA chains (uniform over the wave)
                    frag_cd c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
                    const uint16_t* p0 = planes + (size_t)(rta * 16 + l15) * G::kPlaneStride + kq * 8;
                    const uint16_t* p1 = planes + (size_t)(rtb * 16 + l15) * G::kPlaneStride + kq * 8;
#pragma unroll
                    for (int ks = 0; ks < G::kKS; ++ks) {
                        const uint4 xh0 = *reinterpret_cast<const uint4*>(p0 + ks * 32);
                        const uint4 xm0 = *reinterpret_cast<const uint4*>(p0 + plane + ks * 32);
                        const uint4 xl0 = *reinterpret_cast<const uint4*>(p0 + 2 * plane + ks * 32);
                        const uint4 xh1 = *reinterpret_cast<const uint4*>(p1 + ks * 32);
                        const uint4 xm1 = *reinterpret_cast<const uint4*>(p1 + plane + ks * 32);
                        const uint4 xl1 = *reinterpret_cast<const uint4*>(p1 + 2 * plane + ks * 32);
                        c0 = cwn::mfma_split6(wf[ks][0], wf[ks][1], wf[ks][2], xh0, xm0, xl0, c0);
                        c1 = cwn::mfma_split6(wf[ks][0], wf[ks][1], wf[ks][2], xh1, xm1, xl1, c1);
                    }
                    acc[hh][j] = c0;
                    acc[hh][j + 1] = c1;
                } else if (rta < rt1) {          // the odd tile of this half
                    frag_cd c0 = {0.f, 0.f, 0.f, 0.f};
                    const uint16_t* p0 = planes + (size_t)(rta * 16 + l15) * G::kPlaneStride + kq * 8;
#pragma unroll
                    for (int ks = 0; ks < G::kKS; ++ks) {
                        const uint4 xh0 = *reinterpret_cast<const uint4*>(p0 + ks * 32);
                        const uint4 xm0 = *reinterpret_cast<const uint4*>(p0 + plane + ks * 32);
                        const uint4 xl0 = *reinterpret_cast<const uint4*>(p0 + 2 * plane + ks * 32);
                        c0 = cwn::mfma_split6(wf[ks][0], wf[ks][1], wf[ks][2], xh0, xm0, xl0, c0);
                    }
                    acc[hh][j] = c0;
                }
            }
        }
        __syncthreads();                     // every wave has read its fragments: Y may overwrite the planes
        CWN_STAMP(6);
        // one row of zeros behind the Y rows (row rows_cap of the region: the planes are 816 B a row, Y
        // 528 B): a reduce slot past the end of its row reads it for Y1 and for Y2, relu(0 + 0) = 0
        if (gq == 0) *reinterpret_cast<float4*>(Y + (size_t)rows_cap * G::kYStride + f) = make_float4(0.f, 0.f, 0.f, 0.f);
        // D[i][j]: i = output column (lane >> 4) * 4 + reg, j = x row (lane & 15); Y1 carries the bias
#pragma unroll
        for (int hh = 0; hh < kWSets; ++hh) {
            const int h = kHS == 2 ? my_h : hh;
            const int rt0 = h == 0 ? 0 : R1 / 16, rt1 = rt0 + (h == 0 ? T1 : T2);
            const int first = rt0 + ((rt_par - rt0) % G::kWPC + G::kWPC) % G::kWPC;
#pragma unroll
            for (int j = 0; j < kMaxT; ++j) {
                const int rt = first + j * G::kWPC;
                if (rt < rt1) {
                    frag_cd c = acc[hh][j];
                    if (h == 0 && has_bias) {
                        c[0] += b4.x; c[1] += b4.y; c[2] += b4.z; c[3] += b4.w;
                    }
                    float* y0 = Y + (size_t)(rt * 16 + l15) * G::kYStride + ct * 16 + kq * 4;
                    *reinterpret_cast<float4*>(y0) = make_float4(c[0], c[1], c[2], c[3]);
                }
            }
        }
    }
    __syncthreads();
    CWN_STAMP(7);
    if (A.store_y) {
        // training forward (CWN_LAYER_STORE_Y): the backward pass needs Y1 / Y2 (ReLU mask of the message); they are
        // complete in LDS here -- every lane group copies the rows it is about to reduce anyway
        const BigSet& Bs = A.big[set];
        float* const y1g = Bs.y1 + (size_t)t_r0[0] * F;
        float* const y2g = Bs.y2 + (size_t)fld(I_CR0) * F;
        const float* const Y2l = Y + (size_t)R1 * G::kYStride;
        for (int r = gq; r < g_n; r += G::kNG)
            *reinterpret_cast<float4*>(y1g + (size_t)r * F + f) = *reinterpret_cast<const float4*>(Y + (size_t)r * G::kYStride + f);
        for (int r = gq; r < c_n; r += G::kNG)
            *reinterpret_cast<float4*>(y2g + (size_t)r * F + f) = *reinterpret_cast<const float4*>(Y2l + (size_t)r * G::kYStride + f);
    }

    // ---- 7. upper stream out of LDS: out_up[i] = sum_p relu(Y1[col[p]] + Y2[aux[p]]) + (1 + eps1) x_i -
    {
        const gb_p out_up0 = (gb_p)sfld(S_TASK0 + ST_OUT_UP) + (size_t)t_r0[0] * kRowB;
        const float scale1 = 1.0f + eps1[0];
        const float* Y2 = Y + (size_t)R1 * G::kYStride;
        for (int k = 0; gq + k * G::kNG < g_n; k += kNR) {
            RowSet<kNR> R;
            float4 acc[kNR], xi[kNR];
            const int steps = row_ranges<kNR, G::kNG>(R, rowptr, gq + k * G::kNG, g_n);
            if constexpr (kW8) {
                // the rows this lane group loaded in phase 2 are not held across the matrix-core phase (128 registers):
                // requested again here (L2), they land under the reduce's LDS chain
                const gcb_p xg = (gcb_p)sfld(S_XG) + (size_t)t_r0[0] * kRowB;
#pragma unroll
                for (int u = 0; u < kNR; ++u)
                    xi[u] = ldg4o(xg, (uint32_t)min(gq + (k + u) * G::kNG, g_n - 1) * kRowB + fB);
            } else {
#pragma unroll
                for (int u = 0; u < kNR; ++u) xi[u] = pick(xv, k + u);
            }
            gather_relu_sum<kNR, G::kNG, G::kYStride>(acc, R, steps, scol, saux, Y, Y2, rows_cap, rows_cap - R1, f);
#pragma unroll
            for (int u = 0; u < kNR; ++u)
                if (R.on[u])
                    stg4o(out_up0, (uint32_t)(gq + (k + u) * G::kNG) * kRowB + fB, axpy4(acc[u], scale1, xi[u]));
        }
    }
    CWN_STAMP(8);
    CWN_RSTAMP(1);
}

#if !CWN_W8
// fp32 [F, 2F] weight of the message Linear -> bf16 hi / mid / lo planes in MFMA-fragment order: the 1-KiB
// chunk number ((ks * 3 + plane) * 2 + h) * NCT + ct holds, for lane l = kq * 16 + n, the eight k-values
// W[ct * 16 + n][h * F + ks * 32 + kq * 8 ..] of that plane (16 bytes per lane).
struct PackMany {            // up to CWN_LAYER_PACK_MAX weights of one width in one launch (blockIdx.y = the weight)
    const float* W[CWN_LAYER_PACK_MAX];
    unsigned char* out[CWN_LAYER_PACK_MAX];
    int64_t ldw[CWN_LAYER_PACK_MAX];
    uint8_t trans[CWN_LAYER_PACK_MAX];   // 1: the transposed halves (the backward launch's operand: cwn_layer_bwd.hip)
};

template <int F>
__global__ __launch_bounds__(256) void pack_weights_kernel(PackMany P) {
    constexpr int KS = F / 32, NCT = F / 16;
    const float* __restrict__ W = P.W[blockIdx.y];
    unsigned char* __restrict__ out = P.out[blockIdx.y];
    const int64_t ldw = P.ldw[blockIdx.y];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (ct, h, ks, lane)
    if (g >= NCT * 2 * KS * 64) return;
    const int lane = g & 63, chunk3 = g >> 6;                     // chunk3 = (ct * 2 + h) * KS + ks
    const int ks = chunk3 % KS, h = (chunk3 / KS) & 1, ct = chunk3 / (2 * KS);
    float e[8];
    if (P.trans[blockIdx.y]) {           // element k of lane (kq, n): W[ks * 32 + kq * 8 + k][h * F + ct * 16 + n]
        const float* src = W + (int64_t)(ks * 32 + (lane >> 4) * 8) * ldw + h * F + ct * 16 + (lane & 15);
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = src[(int64_t)k * ldw];
    } else {
        const float* src = W + (int64_t)(ct * 16 + (lane & 15)) * ldw + h * F + ks * 32 + (lane >> 4) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = src[k];
    }
    const float4 a = make_float4(e[0], e[1], e[2], e[3]), b = make_float4(e[4], e[5], e[6], e[7]);
    uint4 ph, pm, pl;
    cwn::split8(a, b, ph, pm, pl);
    unsigned char* dst = out + ((size_t)(ks * 3 * 2 + h) * NCT + ct) * 1024 + lane * 16;
    constexpr size_t kPlane = (size_t)2 * NCT * 1024;     // from a chunk to the same chunk of the next plane
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + kPlane) = pm;
    *reinterpret_cast<uint4*>(dst + 2 * kPlane) = pl;
}
#endif

inline bool al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

#ifdef CWN_LAYER_TIMING
unsigned long long* g_stamps = nullptr;
long long g_stamp_launch = 0, g_stamp_launches = 1;     // the next launch's slot / the slots the buffer holds (round robin)
#endif

template <int F, int MODE>
int launch(LayerArgs& A, int64_t n_items, hipStream_t stream) {
    static std::once_flag once;            // raise the dynamic-LDS limit of this instantiation, once
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel<F, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
    });
    if (attr_err != hipSuccess) return CWN_ERR_LAUNCH;
#ifdef CWN_LAYER_TIMING
    A.stamps = g_stamps == nullptr ? nullptr : g_stamps + (size_t)(g_stamp_launch++ % g_stamp_launches) * (size_t)n_items * CWN_STAMP_REC;
#endif
    size_t lds = kW8 ? (size_t)A.lds_limit : lds_bytes<F>(A.rows_cap, A.xrows_cap);
    if (!kW8) {
        // BIG items stage 64-row chunks as planes and then the complex's CSR slices in the same LDS: at least 96 KiB
        if (A.big[0].y1 != nullptr || A.big[1].y1 != nullptr || A.big[2].y1 != nullptr) lds = lds > 96 * 1024 ? lds : 96 * 1024;
        A.lds_limit = (int32_t)lds;
    }
    static_assert(alignof(LayerArgs) == 8 && sizeof(const int32_t*) + 2 * sizeof(int32_t) == kArgsOff, "where A starts in the kernel-argument segment");
    layer_kernel<F, MODE><<<dim3((unsigned)n_items), dim3(kThreads), lds, stream>>>(A.items, A.set_start1, A.set_start2, A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

}  // namespace

// The entry points of THIS compilation of the file: the public names (16-wave form, include/cwn_hip.h), or -- in the
// second compilation (CWN_LAYER_W8) -- the same four functions of the two-per-CU form under internal names, which
// the public ones dispatch to on cwn_layer_plan.variant.
#if CWN_W8
#define CWN_FN(name) cwn_layer_w8_##name
#else
#define CWN_FN(name) cwn_layer_v0_##name
extern "C" int32_t cwn_layer_w8_round_rows(int32_t F);
extern "C" size_t cwn_layer_w8_lds_bytes(int32_t F, int32_t max_gemm_rows, int32_t max_source_rows);
extern "C" int cwn_layer_w8_items_check(const int32_t* items, int64_t n_items, int32_t F, const cwn_layer_plan* plan);
extern "C" int cwn_layer_w8_launch(const cwn_layer_dim* dims, int n_dims, int32_t F, const cwn_layer_plan* plan,
                                   int32_t flags, int32_t* err_flag, cwn_stream_t stream_);
#endif

#if defined(CWN_LAYER_TIMING) && !CWN_W8
extern "C" void cwn_layer_debug_stamps(unsigned long long* buf) { g_stamps = buf; g_stamp_launch = 0; g_stamp_launches = 1; }
// buf holds `launches` x n_items x CWN_STAMP_REC stamps: consecutive launches write consecutive slots (round robin)
extern "C" void cwn_layer_debug_stamps_many(unsigned long long* buf, long long launches) {
    g_stamps = buf;
    g_stamp_launch = 0;
    g_stamp_launches = launches > 0 ? launches : 1;
}
#endif

extern "C" int32_t CWN_FN(round_rows)(int32_t F) {
    return (F == 64 || F == 128) ? kThreads / (F / 4) : 0;
}

extern "C" size_t CWN_FN(lds_bytes)(int32_t F, int32_t max_gemm_rows, int32_t max_source_rows) {
    if (max_gemm_rows < 0 || max_gemm_rows % 16 != 0 || max_source_rows < 0) return 0;
    size_t b = 0;
    if (F == 128 && max_gemm_rows <= gemm_rows_cap(128) && max_source_rows <= 2 * source_rows_cap(128))
        b = lds_bytes<128>(max_gemm_rows, max_source_rows);
    if (F == 64 && max_gemm_rows <= gemm_rows_cap(64) && max_source_rows <= 2 * source_rows_cap(64))
        b = lds_bytes<64>(max_gemm_rows, max_source_rows);
    return b <= kLdsBudget ? b : 0;      // (two-per-CU form: of ONE item with these rows; a launch takes the largest)
}

// Host-side check of a HOST copy of the item table against its plan: every derived field is what
// include/cwn_hip.h defines, every range lies inside the plan's summary, every item fits the caps.  The
// kernel itself re-checks only what keeps a workgroup inside its LDS.
extern "C" int CWN_FN(items_check)(const int32_t* items, int64_t n_items, int32_t F, const cwn_layer_plan* plan) {
    if (plan == nullptr || n_items < 0 || (n_items > 0 && items == nullptr) || (F != 64 && F != 128)) return CWN_ERR_BAD_ARG;
    if (plan->n_items != n_items) return CWN_ERR_BAD_ARG;
    const int ng = kThreads / (F / 4);
    auto pad16 = [](int64_t n) { return (n + 15) / 16 * 16; };
    auto pad4 = [](int64_t n) { return (n + 3) / 4 * 4; };
    int64_t n_big = 0;
    for (int64_t it = 0; it < n_items; ++it) {
        const int32_t* r = items + it * CWN_LAYER_ITEM_INTS;
        const bool has_gemm = (r[I_FLAGS] & 1) != 0;
        const int nt = r[I_NT];
        if (nt < 0 || nt > 2) return CWN_ERR_BAD_ARG;
        int64_t bne[2] = {0, 0}, sn = 0;
        for (int t = 0; t < 2; ++t) {
            const int32_t* T = r + I_TASK0 + t * T_INTS;
            if (t >= nt) {
                for (int k = 0; k < T_INTS; ++k)
                    if (T[k] != 0) return CWN_ERR_BAD_ARG;
                continue;
            }
            const int d = T[T_DIM];
            const bool is_big = (r[I_FLAGS] & CWN_LAYER_ITEM_BIG) != 0;        // a streamed complex is not bound by the LDS caps
            if (d < 0 || d >= CWN_LAYER_MAX_DIMS || T[T_R0] < 0 || T[T_N] < 0 || (!is_big && T[T_N] > CWN_LAYER_TASK_ROWS) ||
                T[T_BE0] < 0 || T[T_BNE] < 0 || T[T_SR0] < 0 || T[T_SN] < 0)
                return CWN_ERR_BAD_ARG;
            if ((int64_t)T[T_R0] + T[T_N] > plan->cells_end[d] || (int64_t)T[T_BE0] + T[T_BNE] > plan->b_end[d]) return CWN_ERR_BAD_ARG;
            if (T[T_BNE] == 0 ? T[T_SN] != 0 : (d == 0 || (int64_t)T[T_SR0] + T[T_SN] > plan->cells_end[d - 1])) return CWN_ERR_BAD_ARG;
            bne[t] = T[T_BNE];
            sn += T[T_SN];
        }
        const int32_t* T0 = r + I_TASK0;
        const int32_t* T1r = T0 + T_INTS;
        int64_t n0 = nt > 0 ? T0[T_N] : 0, nc = 0, une = 0;
        if (has_gemm) {
            const int g = r[I_G];
            if (nt < 1 || g != T0[T_DIM] || g + 1 >= CWN_LAYER_MAX_DIMS || r[I_GR0] != T0[T_R0] || r[I_GN] != T0[T_N] || n0 <= 0 ||
                r[I_CR0] < 0 || r[I_CN] < 0 || r[I_UE0] < 0 || r[I_UNE] < 0)
                return CWN_ERR_BAD_ARG;
            nc = r[I_CN];
            une = r[I_UNE];
            if ((int64_t)r[I_CR0] + nc > plan->cells_end[g + 1] || (int64_t)r[I_UE0] + une > plan->up_end[g]) return CWN_ERR_BAD_ARG;
            // a second task is the coface block, and gathers from the staged cells of g
            if (nt > 1 && (T1r[T_DIM] != g + 1 || T1r[T_R0] != r[I_CR0] || T1r[T_N] != nc ||
                           (T1r[T_BNE] > 0 && (T1r[T_SR0] != T0[T_R0] || T1r[T_SN] != T0[T_N]))))
                return CWN_ERR_BAD_ARG;
            // two-per-CU form: the rows of EACH product are bounded (a wave holds the accumulators of both)
            if (kW8 && (pad16(n0) > CWN_LAYER_W8_HALF_ROWS(F) || pad16(nc) > CWN_LAYER_W8_HALF_ROWS(F))) return CWN_ERR_BAD_ARG;
        } else {
            if (r[I_CN] != 0 || r[I_UNE] != 0 || nt > 1) return CWN_ERR_BAD_ARG;
        }
        if ((r[I_FLAGS] & CWN_LAYER_ITEM_BIG) != 0) {      // a streamed complex: ranges only (checked above), no derived fields
            if (kW8) return CWN_ERR_BAD_ARG;
            for (int k = I_R1; k < CWN_LAYER_ITEM_INTS; ++k)
                if (r[k] != 0) return CWN_ERR_BAD_ARG;
            ++n_big;
            continue;
        }
        const int64_t r1 = pad16(n0);                      // (round 4: no longer a whole number of rounds of `ng` rows)
        (void)ng;
        const int64_t rows = nc > 0 ? r1 + pad16(nc) : pad16(n0);
        const int64_t b1 = pad4(une), b2 = pad4(b1 + bne[0]), total = pad4(b2 + bne[1]);
        if (r[I_R1] != r1 || r[I_ROWS] != rows || r[I_B1] != b1 || r[I_B2] != b2 || r[I_TOTAL] != total) return CWN_ERR_BAD_ARG;
        for (int k = I_TOTAL + 1; k < CWN_LAYER_ITEM_INTS; ++k)
            if (r[k] != 0) return CWN_ERR_BAD_ARG;
        // sources that take LDS: both tasks' in the 16-wave form, task 0's in the two-per-CU form
        const int64_t sn_lds = kW8 ? (nt > 0 ? (int64_t)T0[T_SN] : 0) : sn;
        if (rows > plan->max_gemm_rows || sn_lds > plan->max_source_rows || total > CWN_LAYER_MAX_ENTRIES) return CWN_ERR_BAD_ARG;
        if (kW8) {       // every item lays out its own rows: staged rows + the LOADED sources (task 0's) within the launch's LDS
            const int64_t src0 = nt > 0 ? T0[T_SN] : 0;
            const size_t need = CWN_FN(lds_bytes)(F, (int32_t)rows, (int32_t)src0);
            if (need == 0 || (int64_t)need > plan->lds_bytes) return CWN_ERR_BAD_ARG;
        }
    }
    if (n_big != plan->n_big) return CWN_ERR_BAD_ARG;
    if (kW8) return plan->lds_bytes <= (int64_t)kLdsBudget ? CWN_OK : CWN_ERR_BAD_ARG;
    return CWN_FN(lds_bytes)(F, plan->max_gemm_rows, plan->max_source_rows) != 0 ? CWN_OK : CWN_ERR_BAD_ARG;
}

extern "C" int CWN_FN(launch)(const cwn_layer_dim* dims, int n_dims, int32_t F, const cwn_layer_plan* plan,
                              int32_t flags, int32_t* err_flag, cwn_stream_t stream_) {
    if (dims == nullptr || plan == nullptr || n_dims < 1 || n_dims > CWN_LAYER_MAX_DIMS || plan->n_items < 0)
        return CWN_ERR_BAD_ARG;
    if ((flags & ~(CWN_LAYER_CSR_STORE | CWN_LAYER_CSR_LOAD | CWN_LAYER_STORE_Y)) != 0 ||
        (flags & (CWN_LAYER_CSR_STORE | CWN_LAYER_CSR_LOAD)) == (CWN_LAYER_CSR_STORE | CWN_LAYER_CSR_LOAD))
        return CWN_ERR_BAD_ARG;
    const bool store_y = (flags & CWN_LAYER_STORE_Y) != 0;
    if (F != 64 && F != 128) return CWN_ERR_BAD_ARG;
    const int64_t n_items = plan->n_items;
    if (n_items == 0) return CWN_OK;
    if (plan->items == nullptr || err_flag == nullptr) return CWN_ERR_BAD_ARG;
    if ((flags & (CWN_LAYER_CSR_STORE | CWN_LAYER_CSR_LOAD)) != 0 && plan->csr_cache == nullptr) return CWN_ERR_BAD_ARG;
    if (n_items >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    if (!kW8 && CWN_FN(lds_bytes)(F, plan->max_gemm_rows, plan->max_source_rows) == 0) return CWN_ERR_BAD_ARG;
    if (kW8 && (plan->max_gemm_rows > gemm_rows_cap(F) || plan->max_source_rows > 2 * source_rows_cap(F))) return CWN_ERR_BAD_ARG;
    if (!al16(plan->items) || !al16(plan->csr_cache)) return CWN_ERR_ALIGN;
    LayerArgs A{};
    bool has_up[CWN_LAYER_MAX_DIMS] = {false, false, false};
    for (int d = 0; d < n_dims; ++d) {
        const cwn_layer_dim& D = dims[d];
        if (D.n_cells < 0 || D.e_up < 0 || D.n_b < 0) return CWN_ERR_BAD_ARG;
        if (D.n_cells >= INT32_MAX || D.e_up >= INT32_MAX || D.n_b >= INT32_MAX) return CWN_ERR_TOO_LARGE;
        if (D.n_cells > 0 && (D.x == nullptr || D.out_up == nullptr || D.out_b == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.e_up > 0 && (D.up_index == nullptr || D.up_shared == nullptr || D.msg_w_packed == nullptr ||
                           d + 1 >= n_dims))
            return CWN_ERR_BAD_ARG;
        if (D.n_b > 0 && (D.b_index == nullptr || d == 0)) return CWN_ERR_BAD_ARG;
        if (!(al16(D.x) && al16(D.out_up) && al16(D.out_b) && al16(D.out_down) && al16(D.msg_w_packed) && al16(D.msg_bias)))
            return CWN_ERR_ALIGN;
        // the table may not address more than the tensors hold: the kernel forms addresses from it
        if (plan->cells_end[d] < 0 || plan->cells_end[d] > D.n_cells || plan->up_end[d] < 0 ||
            plan->up_end[d] > D.e_up || plan->b_end[d] < 0 || plan->b_end[d] > D.n_b)
            return CWN_ERR_BAD_ARG;
        has_up[d] = D.e_up > 0;
        if (store_y && D.e_up > 0) {
            if (D.big_y1 == nullptr || dims[d + 1].big_y2 == nullptr) return CWN_ERR_BAD_ARG;
            if (!(al16(D.big_y1) && al16(dims[d + 1].big_y2))) return CWN_ERR_ALIGN;
        }
        // BIG records: the streamed complexes need their CSR and the scratch matrices (include/cwn_hip.h)
        if (plan->n_big > 0) {
            if (kW8) return CWN_ERR_BAD_ARG;
            if (D.e_up > 0 && (D.big_up_rowptr == nullptr || D.big_up_col == nullptr || D.big_up_aux == nullptr ||
                               D.big_y1 == nullptr || dims[d + 1].big_y2 == nullptr))
                return CWN_ERR_BAD_ARG;
            if (D.n_b > 0 && (D.big_b_rowptr == nullptr || D.big_b_col == nullptr)) return CWN_ERR_BAD_ARG;
            if (!(al16(D.big_y1) && al16(D.big_y2))) return CWN_ERR_ALIGN;
        }
    }
    // the sets, in the order the item table numbers them (include/cwn_hip.h): every dimension with an
    // upper adjacency is the GEMM dimension of a set; the top dimension without one rides as its
    // neighbour's second task; any other dimension without one is a set of its own
    int n_sets = 0;
    for (int d = 0; d < n_dims;) {
        uint64_t* S = A.set[n_sets++];
        int tasks[2] = {d, -1};
        S[S_XG] = (uint64_t)(uintptr_t)dims[d].x;               // the staged block is task 0's cells
        if (has_up[d]) {
            const cwn_layer_dim& D = dims[d];
            S[S_XC] = (uint64_t)(uintptr_t)dims[d + 1].x;
            S[S_UP_INDEX] = (uint64_t)(uintptr_t)D.up_index;
            S[S_UP_SHARED] = (uint64_t)(uintptr_t)D.up_shared;
            S[S_UP_E] = (uint64_t)D.e_up;
            S[S_WP] = (uint64_t)(uintptr_t)D.msg_w_packed;
            S[S_MSG_BIAS] = (uint64_t)(uintptr_t)D.msg_bias;
            if (d + 1 < n_dims && !has_up[d + 1] && d + 2 >= n_dims) tasks[1] = d + 1;
        }
        BigSet& Bg = A.big[n_sets - 1];
        if (has_up[d]) {
            Bg.up_rowptr = dims[d].big_up_rowptr; Bg.up_col = dims[d].big_up_col; Bg.up_aux = dims[d].big_up_aux;
            Bg.y1 = dims[d].big_y1;
            Bg.y2 = dims[d + 1].big_y2;
        }
        for (int t = 0; t < 2; ++t) {
            if (tasks[t] < 0) continue;
            const cwn_layer_dim& D = dims[tasks[t]];
            Bg.b_rowptr[t] = D.big_b_rowptr;
            Bg.b_col[t] = D.big_b_col;
            uint64_t* T = S + S_TASK0 + t * ST_FIELDS;
            T[ST_X] = (uint64_t)(uintptr_t)D.x;
            T[ST_XS] = tasks[t] > 0 ? (uint64_t)(uintptr_t)dims[tasks[t] - 1].x : 0;
            T[ST_OUT_UP] = (uint64_t)(uintptr_t)D.out_up;
            T[ST_OUT_B] = (uint64_t)(uintptr_t)D.out_b;
            T[ST_B_INDEX] = (uint64_t)(uintptr_t)D.b_index;
            T[ST_B_E] = (uint64_t)D.n_b;
            T[ST_EPS1] = (uint64_t)(uintptr_t)D.eps1;
            T[ST_EPS2] = (uint64_t)(uintptr_t)D.eps2;
            T[ST_OUT_D] = (uint64_t)(uintptr_t)D.out_down;
            T[ST_EPS3] = (uint64_t)(uintptr_t)D.eps3;
        }
        d += tasks[1] >= 0 ? 2 : 1;
    }
    // items are ordered by set: set_start[s] is the first item of set s
    for (int s_ = 0; s_ <= n_sets; ++s_) {
        const int32_t lo = s_ == 0 ? 0 : plan->set_start[s_ - 1];
        const int32_t v = s_ == n_sets ? (int32_t)n_items : plan->set_start[s_];
        if ((s_ == 0 && plan->set_start[0] != 0) || v < lo || v > n_items) return CWN_ERR_BAD_ARG;
    }
    A.set_start1 = n_sets > 1 ? plan->set_start[1] : INT32_MAX;
    A.set_start2 = n_sets > 2 ? plan->set_start[2] : INT32_MAX;
    A.items = plan->items;
    A.err = err_flag;
    A.csr_cache = static_cast<unsigned char*>(plan->csr_cache);
    A.rows_cap = plan->max_gemm_rows;
    A.xrows_cap = plan->max_source_rows;
    A.lds_limit = (int32_t)plan->lds_bytes;
    A.store_y = store_y ? 1 : 0;
    if (kW8 && (plan->lds_bytes < (int64_t)lds_bytes<128>(16, 0) || plan->lds_bytes > (int64_t)kLdsBudget)) return CWN_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int mode = (flags & CWN_LAYER_CSR_LOAD) ? kLoad : (flags & CWN_LAYER_CSR_STORE) ? kSortStore : kSort;
    if (F == 128) {
        if (mode == kLoad) return launch<128, kLoad>(A, n_items, stream);
        if (mode == kSortStore) return launch<128, kSortStore>(A, n_items, stream);
        return launch<128, kSort>(A, n_items, stream);
    }
    if (mode == kLoad) return launch<64, kLoad>(A, n_items, stream);
    if (mode == kSortStore) return launch<64, kSortStore>(A, n_items, stream);
    return launch<64, kSort>(A, n_items, stream);
}

#if !CWN_W8
// ---- the public entry points (include/cwn_hip.h): variant 0 = the 16-wave form above, variant 1 = the two-per-CU form
extern "C" size_t cwn_layer_packed_weight_bytes(int32_t F) {
    return (F == 64 || F == 128) ? (size_t)F * 2 * F * 6 : 0;
}

static int pack_many(const float* const* W, const int64_t* ldw, int32_t F, void* const* out, void* const* out_t, int32_t n,
                     cwn_stream_t stream_) {
    if ((F != 64 && F != 128) || W == nullptr || (out == nullptr && out_t == nullptr) || ldw == nullptr || n < 0) return CWN_ERR_BAD_ARG;
    const int threads = (F / 16) * 2 * (F / 32) * 64;
    hipStream_t stream = (hipStream_t)stream_;
    // entries: (weight, form); CWN_LAYER_PACK_MAX per launch
    const int forms = (out != nullptr) + (out_t != nullptr);
    const int64_t total = (int64_t)n * forms;
    for (int64_t e0 = 0; e0 < total; e0 += CWN_LAYER_PACK_MAX) {
        const int m = (int)(total - e0 < CWN_LAYER_PACK_MAX ? total - e0 : CWN_LAYER_PACK_MAX);
        PackMany P{};
        for (int i = 0; i < m; ++i) {
            const int64_t e = e0 + i;
            const int form = forms == 2 ? (int)(e / n) : (out != nullptr ? 0 : 1);
            const int w = (int)(e % n);
            void* o = form == 0 ? out[w] : out_t[w];
            if (W[w] == nullptr || o == nullptr || ldw[w] < 2 * F) return CWN_ERR_BAD_ARG;
            if (((uintptr_t)W[w] & 3u) || !al16(o)) return CWN_ERR_ALIGN;
            P.W[i] = W[w];
            P.out[i] = (unsigned char*)o;
            P.ldw[i] = ldw[w];
            P.trans[i] = (uint8_t)form;
        }
        if (F == 128) pack_weights_kernel<128><<<dim3((threads + 255) / 256, m), dim3(256), 0, stream>>>(P);
        else pack_weights_kernel<64><<<dim3((threads + 255) / 256, m), dim3(256), 0, stream>>>(P);
    }
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_layer_pack_weights_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out, int32_t n,
                                               cwn_stream_t stream_) {
    return pack_many(W, ldw, F, out, nullptr, n, stream_);
}

extern "C" int cwn_layer_pack_weights_t_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out, int32_t n,
                                                 cwn_stream_t stream_) {
    return pack_many(W, ldw, F, nullptr, out, n, stream_);
}

extern "C" int cwn_layer_pack_weights_both_many_f32(const float* const* W, const int64_t* ldw, int32_t F, void* const* out,
                                                    void* const* out_t, int32_t n, cwn_stream_t stream_) {
    if (out == nullptr || out_t == nullptr) return CWN_ERR_BAD_ARG;
    return pack_many(W, ldw, F, out, out_t, n, stream_);
}

extern "C" int cwn_layer_pack_weights_f32(const float* W, int64_t ldw, int32_t F, void* out, cwn_stream_t stream_) {
    return cwn_layer_pack_weights_many_f32(&W, &ldw, F, &out, 1, stream_);
}

extern "C" int32_t cwn_layer_variant_round_rows(int32_t F, int32_t variant) {
    return variant == 0 ? cwn_layer_v0_round_rows(F) : variant == 1 ? cwn_layer_w8_round_rows(F) : 0;
}
extern "C" int32_t cwn_layer_round_rows(int32_t F) { return cwn_layer_v0_round_rows(F); }

extern "C" size_t cwn_layer_variant_lds_bytes(int32_t F, int32_t variant, int32_t max_gemm_rows, int32_t max_source_rows) {
    return variant == 0 ? cwn_layer_v0_lds_bytes(F, max_gemm_rows, max_source_rows)
                        : variant == 1 ? cwn_layer_w8_lds_bytes(F, max_gemm_rows, max_source_rows) : 0;
}
extern "C" size_t cwn_layer_fused_lds_bytes(int32_t F, int32_t max_gemm_rows, int32_t max_source_rows) {
    return cwn_layer_v0_lds_bytes(F, max_gemm_rows, max_source_rows);
}

extern "C" int cwn_layer_items_check(const int32_t* items, int64_t n_items, int32_t F, const cwn_layer_plan* plan) {
    if (plan == nullptr) return CWN_ERR_BAD_ARG;
    return plan->variant == 0 ? cwn_layer_v0_items_check(items, n_items, F, plan)
                              : plan->variant == 1 ? cwn_layer_w8_items_check(items, n_items, F, plan) : CWN_ERR_BAD_ARG;
}

extern "C" int cwn_layer_fused_f32(const cwn_layer_dim* dims, int n_dims, int32_t F, const cwn_layer_plan* plan,
                                   int32_t flags, int32_t* err_flag, cwn_stream_t stream_) {
    if (plan == nullptr) return CWN_ERR_BAD_ARG;
    return plan->variant == 0 ? cwn_layer_v0_launch(dims, n_dims, F, plan, flags, err_flag, stream_)
                              : plan->variant == 1 ? cwn_layer_w8_launch(dims, n_dims, F, plan, flags, err_flag, stream_)
                                                   : CWN_ERR_BAD_ARG;
}
#endif
