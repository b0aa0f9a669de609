// cwn_aggregate.hip -- fused gather -> message -> segmented reduce over destination-sorted CSR
// (K1 + message hook + K2 of SURVEY.md §2.2 in one pass), and the plain row gather (K1).
//
// HBM-bound byte work (≈0.25 FLOP/B): no MFMA here.  Mapping for gfx950:
//   * a GROUP of G lanes (G = power of two, 1..64) owns one destination row; each lane holds a
//     VEC-wide (16 B when F % 4 == 0) slice of the feature row, so a gathered source row is read
//     by one fully coalesced wave instruction (G*16 B contiguous) and the output row is written
//     once, coalesced.  F = 128 -> G = 32: two rows per wavefront; F = 64 -> four rows; F <= 4 ->
//     one lane per row.
//   * the segment's indices are fetched by the group cooperatively (one coalesced load of up to
//     G indices) and broadcast with ds_bpermute (__shfl), so the dependent chain is
//     index-load -> row-load instead of one round trip per entry; row loads are issued four at a
//     time before the first add.
//   * accumulation is sequential in CSR (= original entry) order in registers: deterministic and
//     bit-identical to a sequential index_add_; no atomics, no zero-fill pass, absent
//     adjacencies and empty rows write zeros directly.  Rows longer than CWN_LONG_ROW (hubs) are
//     the one exception: a whole workgroup folds such a row in R chunks combined in chunk order
//     (still deterministic; equal to the sequential sum up to fp32 re-association).
//   * one launch covers up to CWN_MAX_DESCS descriptors (all adjacencies of all dimensions of a
//     layer): blockIdx -> (descriptor, row tile) through a small prefix table in kernel args.
#include <hip/hip_runtime.h>
#include <float.h>
#include "../../include/cwn_hip.h"
#include "cwn_mem.h"

namespace {

constexpr int kThreads = 256;

struct AggBatch {
    cwn_agg_desc d[CWN_MAX_DESCS];
    int32_t blk_start[CWN_MAX_DESCS + 1];
    int32_t group[CWN_MAX_DESCS];  // lanes per destination row
    int32_t fgroup[CWN_MAX_DESCS]; // of which feature lanes (the rest are entry slots, narrow F only)
    int32_t n;
};

template <int VEC> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<1> { using T = float; };

template <int VEC> struct Acc { float v[VEC]; };

template <int VEC>
__device__ __forceinline__ Acc<VEC> ld(const float* p) {
    Acc<VEC> a;
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        a.v[0] = t.x; a.v[1] = t.y; a.v[2] = t.z; a.v[3] = t.w;
    } else if constexpr (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        a.v[0] = t.x; a.v[1] = t.y;
    } else {
        a.v[0] = *p;
    }
    return a;
}

template <int VEC>
__device__ __forceinline__ void st(float* p, const Acc<VEC>& a) {
    if constexpr (VEC == 4) {
        cwn::store_result4(p, a.v[0], a.v[1], a.v[2], a.v[3]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(a.v[0], a.v[1]);
    } else {
        *p = a.v[0];
    }
}

template <int VEC>
__device__ __forceinline__ Acc<VEC> splat(float x) {
    Acc<VEC> a;
#pragma unroll
    for (int k = 0; k < VEC; ++k) a.v[k] = x;
    return a;
}

// Address of columns f.. of row `idx` of a row-major [*, F] fp32 matrix.  SMALL (every operand of
// the launch lies within 4 GiB of its base pointer: CWN_AGG_SMALL_OPERANDS + the output size): a
// 32-bit byte offset on the scalar base -- the global_load saddr form, one address register per
// load in flight instead of two and no 64-bit multiply (quarter rate) per gathered row.
template <bool SMALL>
__device__ __forceinline__ const float* row_at(const float* base, int64_t idx, int F, int f) {
    if constexpr (SMALL) {
        const uint32_t off = ((uint32_t)idx * (uint32_t)F + (uint32_t)f) << 2;
        return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off);
    } else {
        return base + idx * F + f;
    }
}
template <bool SMALL>
__device__ __forceinline__ float* row_at(float* base, int64_t idx, int F, int f) {
    return const_cast<float*>(row_at<SMALL>(const_cast<const float*>(base), idx, F, f));
}

// message for one CSR position; `pre` is self_pre[i, f..] (mask form only)
template <int VEC, int OP>
__device__ __forceinline__ Acc<VEC> message(const Acc<VEC>& a, const Acc<VEC>& b, const Acc<VEC>& pre) {
    Acc<VEC> m;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        if constexpr (OP == CWN_MSG_A) m.v[k] = a.v[k];
        else if constexpr (OP == CWN_MSG_A_PLUS_B) m.v[k] = a.v[k] + b.v[k];
        else if constexpr (OP == CWN_MSG_A_TIMES_B) m.v[k] = a.v[k] * b.v[k];
        else if constexpr (OP == CWN_MSG_RELU_A_PLUS_B) m.v[k] = fmaxf(a.v[k] + b.v[k], 0.0f);
        else if constexpr (OP == CWN_MSG_RELU_A_PLUS_B_SQ) {
            const float r = fmaxf(a.v[k] + b.v[k], 0.0f);
            m.v[k] = r * r;
        } else if constexpr (OP == CWN_MSG_A_TIMES_2RELU) m.v[k] = 2.0f * a.v[k] * fmaxf(pre.v[k] + b.v[k], 0.0f);
        else m.v[k] = (pre.v[k] + b.v[k] > 0.0f) ? a.v[k] : 0.0f;
    }
    return m;
}

template <int VEC, int RED>
__device__ __forceinline__ void combine(Acc<VEC>& acc, const Acc<VEC>& m) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        if constexpr (RED == CWN_REDUCE_MAX) acc.v[k] = fmaxf(acc.v[k], m.v[k]);
        else acc.v[k] = acc.v[k] + m.v[k];
    }
}

// One group (G lanes, lane-in-group `gl`) folds CSR positions [start, end) of one destination
// row into a register accumulator, in CSR order.  Every lane of the group runs every loop with
// the same trip counts (the index fetch and the shuffles need all G lanes); lanes whose feature
// slice starts past F (`!active`) only skip the loads.
template <int VEC, int OP, int RED, bool SMALL>
__device__ __forceinline__ Acc<VEC> fold_range(const cwn_agg_desc& D, int start, int end, int G, int gl,
                                               int f, bool active, const Acc<VEC>& pre) {
    constexpr bool kUsesB = (OP != CWN_MSG_A);
    const int F = D.F;
    const bool b_scalar = kUsesB && D.b_width == 1;
    Acc<VEC> acc = splat<VEC>(RED == CWN_REDUCE_MAX ? -FLT_MAX : 0.0f);
    for (int base = start; base < end; base += G) {
        // cooperative index fetch: lane gl holds the indices of CSR position base+gl
        const int mine = base + gl;
        int my_ia = 0, my_ib = 0;
        if (mine < end) {
            my_ia = D.ia[mine];
            if constexpr (kUsesB) my_ib = D.ib[mine];
        }
        const int cnt = min(G, end - base);
        int t = 0;
        for (; t + 4 <= cnt; t += 4) {
            Acc<VEC> a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ia = __shfl(my_ia, t + u, G);
                int ib = 0;
                if constexpr (kUsesB) ib = __shfl(my_ib, t + u, G);
                a[u] = splat<VEC>(0.0f);
                b[u] = splat<VEC>(0.0f);
                if (active) {
                    a[u] = ld<VEC>(row_at<SMALL>(D.A, ia, F, f));
                    if constexpr (kUsesB)
                        b[u] = b_scalar ? splat<VEC>(*row_at<SMALL>(D.B, ib, 1, 0))
                                        : ld<VEC>(row_at<SMALL>(D.B, ib, F, f));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) combine<VEC, RED>(acc, message<VEC, OP>(a[u], b[u], pre));
        }
        for (; t < cnt; ++t) {
            const int ia = __shfl(my_ia, t, G);
            int ib = 0;
            if constexpr (kUsesB) ib = __shfl(my_ib, t, G);
            Acc<VEC> a = splat<VEC>(0.0f), b = splat<VEC>(0.0f);
            if (active) {
                a = ld<VEC>(row_at<SMALL>(D.A, ia, F, f));
                if constexpr (kUsesB)
                    b = b_scalar ? splat<VEC>(*row_at<SMALL>(D.B, ib, 1, 0)) : ld<VEC>(row_at<SMALL>(D.B, ib, F, f));
            }
            combine<VEC, RED>(acc, message<VEC, OP>(a, b, pre));
        }
    }
    return acc;
}

// Narrow features (F <= 16: REDDIT-like inputs have ONE scalar feature per vertex): the G lanes of a
// group are then S = G / GF entry slots x GF feature lanes, and for rows with more than
// kSplitRow entries every slot folds every S-th entry; the S partials are combined by a fixed
// xor tree.  One lane walking a 60-entry row alone is 15 dependent round trips (the F = 1 layer of
// the REDDIT-like configuration took 30 us, longer than its F = 64 layers).  Rows up to kSplitRow
// entries keep the sequential order (bit-identical to index_add_), like every row of wide layers.
constexpr int kSplitRow = 16;

struct Operands {        // the descriptor fields a fold needs, by value (registers)
    const int32_t* ia;
    const int32_t* ib;
    const float* A;
    const float* B;
    int F, b_width;
};

template <int VEC, int OP, int RED, bool SMALL>
__device__ __forceinline__ Acc<VEC> fold_range_split(const Operands D, int start, int end, int G, int GF,
                                                     int gl, const Acc<VEC>& pre) {
    constexpr bool kUsesB = (OP != CWN_MSG_A);
    const int F = D.F;
    const bool b_scalar = kUsesB && D.b_width == 1;
    const int S = G / GF, e = gl / GF, f = (gl % GF) * VEC;
    const bool active = f < F;
    Acc<VEC> acc = splat<VEC>(RED == CWN_REDUCE_MAX ? -FLT_MAX : 0.0f);
    for (int base = start; base < end; base += G) {
        const int mine = base + gl;
        int my_ia = 0, my_ib = 0;
        if (mine < end) {
            my_ia = D.ia[mine];
            if constexpr (kUsesB) my_ib = D.ib[mine];
        }
        const int cnt = min(G, end - base);
        for (int tb = 0; tb < cnt; tb += 2 * S) {          // uniform trip count over the group
            Acc<VEC> a[2], b[2];
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = tb + u * S + e;
                ok[u] = t < cnt;
                const int ia = __shfl(my_ia, ok[u] ? t : 0, G);
                int ib = 0;
                if constexpr (kUsesB) ib = __shfl(my_ib, ok[u] ? t : 0, G);
                a[u] = splat<VEC>(0.0f);
                b[u] = splat<VEC>(0.0f);
                if (active && ok[u]) {
                    a[u] = ld<VEC>(row_at<SMALL>(D.A, ia, F, f));
                    if constexpr (kUsesB)
                        b[u] = b_scalar ? splat<VEC>(*row_at<SMALL>(D.B, ib, 1, 0))
                                        : ld<VEC>(row_at<SMALL>(D.B, ib, F, f));
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (ok[u]) combine<VEC, RED>(acc, message<VEC, OP>(a[u], b[u], pre));
        }
    }
    for (int off = GF; off < G; off <<= 1) {               // entry slots -> slot 0, fixed tree
        Acc<VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = __shfl_xor(acc.v[k], off, G);
        combine<VEC, RED>(acc, o);
    }
    return acc;
}

// The self terms of a row slice ((1 + eps) x_i; in backward the two GIN self terms of a cell).  For
// the one-operand message they are loaded BEFORE the fold, next to the row pointers they do not
// depend on: fetched at the end they are a fourth dependent round trip (row pointers -> indices ->
// rows -> self) of a lane group that lives for one row.  The two-operand messages set the kernel's
// register count (72 = 7 waves per SIMD, see aggregate_kernel) and have no room for 8 more live
// registers: they load late.
template <int VEC> struct SelfTerms { Acc<VEC> s1, s2; };

template <int VEC, int OP, bool SMALL>
__device__ __forceinline__ SelfTerms<VEC> load_self_early(const cwn_agg_desc& D, int64_t row, int f, bool active) {
    SelfTerms<VEC> t{splat<VEC>(0.0f), splat<VEC>(0.0f)};
    if constexpr (OP == CWN_MSG_A) {
        if (active && D.self_x != nullptr) t.s1 = ld<VEC>(row_at<SMALL>(D.self_x, row, D.F, f));
        if (active && D.self_x2 != nullptr) t.s2 = ld<VEC>(row_at<SMALL>(D.self_x2, row, D.F, f));
    }
    return t;
}

// mean / empty-max fix-up, self terms, one coalesced store of the row slice
template <int VEC, int OP, int RED, bool SMALL>
__device__ __forceinline__ void finish_row(const cwn_agg_desc& D, int64_t row, int f, int len, float scale1,
                                           Acc<VEC> acc, const SelfTerms<VEC>& self) {
    const int F = D.F;
    if constexpr (RED == CWN_REDUCE_MEAN) {
        const float cntf = (float)max(len, 1);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc.v[k] = acc.v[k] / cntf;
    }
    if constexpr (RED == CWN_REDUCE_MAX) {
        if (len == 0) acc = splat<VEC>(0.0f);
    }
    if (D.self_x != nullptr) {
        const Acc<VEC> s1 = OP == CWN_MSG_A ? self.s1 : ld<VEC>(row_at<SMALL>(D.self_x, row, F, f));
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc.v[k] = acc.v[k] + scale1 * s1.v[k];
    }
    if (D.self_x2 != nullptr) {      // backward: the two GIN self terms of a cell, in one pass
        const float scale2 = 1.0f + (D.eps2 != nullptr ? *D.eps2 : 0.0f);
        const Acc<VEC> s2 = OP == CWN_MSG_A ? self.s2 : ld<VEC>(row_at<SMALL>(D.self_x2, row, F, f));
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc.v[k] = acc.v[k] + scale2 * s2.v[k];
    }
    st<VEC>(row_at<SMALL>(D.out, row, F, f), acc);
}

// Workgroup `blk` of the `nblk` that serve descriptor D.
//   1. every lane group reduces its own destination row, sequentially in CSR (= original entry)
//      order: bit-identical to a sequential index_add_;
//   2. rows with more than CWN_LONG_ROW entries (hub cells of REDDIT-like complexes; listed by
//      cwn_csr_build) are skipped in 1 and taken round-robin by whole workgroups here: the R lane
//      groups of the block fold R contiguous chunks of the row, the partials meet in LDS and are
//      combined in chunk order -- deterministic, no atomics, and the kernel no longer waits for
//      one lane group to walk a 300-entry row alone.
template <int VEC, int OP, int RED, bool SMALL>
__device__ __forceinline__ void run_desc(const cwn_agg_desc& D, int blk, int nblk, int G, int GF, float* part) {
    const int F = D.F;
    const int R = kThreads / G;  // lane groups (= rows in flight) per workgroup
    const int gl = threadIdx.x & (G - 1);
    const int gq = threadIdx.x / G;
    const bool has_long = D.long_rows != nullptr && D.n_long != nullptr && D.rowptr != nullptr;
    const int64_t row = (int64_t)blk * R + gq;
    // destination rows that exist (include/cwn_hip.h, "device-side row counts"; D.n_dst is then the capacity)
    const int64_t n_dst = D.m_dev != nullptr ? *D.m_dev : D.n_dst;
    int start = 0, end = 0;
    if (row < n_dst && D.rowptr != nullptr) {
        start = D.rowptr[row];
        end = D.rowptr[row + 1];
    }
    // The long-row counters are only needed after the regular rows, eps at the end of a row.
    // Loaded here as VECTOR loads (per-lane address) issued AFTER the row pointers: vector loads
    // return in order, so waiting for the row pointers does not wait for these, whereas a scalar
    // load joins the kernel-argument loads in the one out-of-order scalar counter and puts a global
    // round trip (~0.5-1 us) in front of every workgroup's first row.
    const int nl_lane = has_long ? D.n_long[threadIdx.x & (CWN_LONG_PARTS - 1)] : 0;
    int z = 0;
    asm volatile("" : "+v"(z));  // a zero the compiler cannot fold: keeps the eps loads in VMEM
    const float self_scale = 1.0f + (D.eps != nullptr ? D.eps[z] : 0.0f);
    if (row < n_dst) {  // whole groups take the branch together (G divides 64)
        if (has_long && end - start > CWN_LONG_ROW) {
            // left to the whole-workgroup pass below
        } else if (GF < G && end - start > kSplitRow) {
            const int f = (gl % GF) * VEC;
            const bool active = f < F;
            Acc<VEC> pre = splat<VEC>(0.0f);
            if constexpr (OP == CWN_MSG_A_MASK_RELU || OP == CWN_MSG_A_TIMES_2RELU) {
                if (active) pre = ld<VEC>(row_at<SMALL>(D.self_pre, row, F, f));
            }
            const SelfTerms<VEC> self = load_self_early<VEC, OP, SMALL>(D, row, f, active && gl < GF);
            const Operands ops{D.ia, D.ib, D.A, D.B, D.F, D.b_width};
            const Acc<VEC> acc = fold_range_split<VEC, OP, RED, SMALL>(ops, start, end, G, GF, gl, pre);
            if (active && gl < GF)
                finish_row<VEC, OP, RED, SMALL>(D, row, f, end - start, self_scale, acc, self);
        } else {
            // feature chunks of G*VEC columns (one chunk when F <= G*VEC, the common case)
            for (int f0 = 0; f0 < F; f0 += G * VEC) {
                const int f = f0 + gl * VEC;
                const bool active = f < F;
                Acc<VEC> pre = splat<VEC>(0.0f);
                if constexpr (OP == CWN_MSG_A_MASK_RELU || OP == CWN_MSG_A_TIMES_2RELU) {
                    if (active) pre = ld<VEC>(row_at<SMALL>(D.self_pre, row, F, f));
                }
                const SelfTerms<VEC> self = load_self_early<VEC, OP, SMALL>(D, row, f, active);
                const Acc<VEC> acc = fold_range<VEC, OP, RED, SMALL>(D, start, end, G, gl, f, active, pre);
                if (active) finish_row<VEC, OP, RED, SMALL>(D, row, f, end - start, self_scale, acc, self);
            }
        }
    }
    int n_long = 0, nl[CWN_LONG_PARTS];
#pragma unroll
    for (int p = 0; p < CWN_LONG_PARTS; ++p) {
        nl[p] = __builtin_amdgcn_readlane(nl_lane, p);
        n_long += nl[p];
    }
    for (int li = blk; li < n_long; li += nblk) {  // uniform over the workgroup
        int p = 0, k = li;
#pragma unroll
        for (int q = 0; q < CWN_LONG_PARTS - 1; ++q)     // li-th entry of the concatenated sub-lists
            if (p == q && k >= nl[q]) { k -= nl[q]; ++p; }
        const int64_t lrow = D.long_rows[(int64_t)p * D.long_cap + k];
        const int start = D.rowptr[lrow], end = D.rowptr[lrow + 1];
        const int chunk = (((end - start + R - 1) / R) + 3) & ~3;
        const int s = min(end, start + gq * chunk), e = min(end, s + chunk);
        for (int f0 = 0; f0 < F; f0 += G * VEC) {
            const int f = f0 + gl * VEC;
            const bool active = f < F;
            Acc<VEC> pre = splat<VEC>(0.0f);
            if constexpr (OP == CWN_MSG_A_MASK_RELU || OP == CWN_MSG_A_TIMES_2RELU) {
                if (active) pre = ld<VEC>(row_at<SMALL>(D.self_pre, lrow, F, f));
            }
            Acc<VEC> acc = fold_range<VEC, OP, RED, SMALL>(D, s, e, G, gl, f, active, pre);
            if (gq != 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) part[threadIdx.x * VEC + k] = acc.v[k];
            }
            __syncthreads();
            if (gq == 0 && active) {
                for (int q = 1; q < R; ++q) {
                    Acc<VEC> m;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) m.v[k] = part[(q * G + gl) * VEC + k];
                    combine<VEC, RED>(acc, m);
                }
                const SelfTerms<VEC> self = load_self_early<VEC, OP, SMALL>(D, lrow, f, true);
                finish_row<VEC, OP, RED, SMALL>(D, lrow, f, end - start, self_scale, acc, self);
            }
            __syncthreads();
        }
    }
}

template <int VEC, int OP, bool SMALL>
__device__ __forceinline__ void run_desc_red(const cwn_agg_desc& D, int blk, int nblk, int G, int GF, float* part) {
    switch (D.reduce) {
        case CWN_REDUCE_MEAN: run_desc<VEC, OP, CWN_REDUCE_MEAN, SMALL>(D, blk, nblk, G, GF, part); break;
        case CWN_REDUCE_MAX: run_desc<VEC, OP, CWN_REDUCE_MAX, SMALL>(D, blk, nblk, G, GF, part); break;
        default: run_desc<VEC, OP, CWN_REDUCE_ADD, SMALL>(D, blk, nblk, G, GF, part); break;
    }
}

// Registers decide this kernel's speed: the gathers are latency-bound, so throughput follows the
// number of loads in flight = waves per SIMD x 4..8.  Measured on the same code, 73 VGPRs (6 waves)
// vs 71 (7 waves): 6.5 vs 5.9 us at ZINC-128, 333 vs 298 us at batch 8192.  Forcing a budget with
// amdgpu_waves_per_eu on the 64-bit-address code is not the answer (12 bytes of scratch cost more
// than the wave gained); the SMALL addressing is: 64 VGPRs against 75, and with the attribute (it
// also holds the SGPRs under 96) 8 waves without a spill.
// NARROW: some descriptor has fewer than 8 feature lanes (F <= 16 with 16-B vectors): rows get 8
// lanes and the entry-parallel fold; a separate instantiation so that the wide-feature kernel (every
// layer of the molecular models) carries none of that code.
template <int VEC, bool NARROW, bool SMALL>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(SMALL ? 8 : 1, 8)))
void aggregate_kernel(AggBatch B) {
    __shared__ float part[kThreads * VEC];
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.blk_start[i]) di = i;
    // By VALUE.  hipcc passes the batch struct through a private copy that it normally folds back
    // into kernarg loads; with `const cwn_agg_desc& D = B.d[di]` and enough inlined uses of D (the
    // NARROW variant) it stopped doing so and the whole 1.2-KB struct landed in scratch: every
    // descriptor field became a scratch load and the kernel ran 20x slower (measured: 23 -> 580 us).
    const cwn_agg_desc D = B.d[di];
    const int G = B.group[di], GF = NARROW ? B.fgroup[di] : G;
    const int blk = blockIdx.x - B.blk_start[di];
    const int nblk = B.blk_start[di + 1] - B.blk_start[di];
    switch (D.msg_op) {
        case CWN_MSG_A_PLUS_B: run_desc_red<VEC, CWN_MSG_A_PLUS_B, SMALL>(D, blk, nblk, G, GF, part); break;
        case CWN_MSG_A_TIMES_B: run_desc_red<VEC, CWN_MSG_A_TIMES_B, SMALL>(D, blk, nblk, G, GF, part); break;
        case CWN_MSG_RELU_A_PLUS_B:
            run_desc<VEC, CWN_MSG_RELU_A_PLUS_B, CWN_REDUCE_ADD, SMALL>(D, blk, nblk, G, GF, part); break;
        case CWN_MSG_A_MASK_RELU:
            run_desc<VEC, CWN_MSG_A_MASK_RELU, CWN_REDUCE_ADD, SMALL>(D, blk, nblk, G, GF, part); break;
        case CWN_MSG_RELU_A_PLUS_B_SQ:
            run_desc<VEC, CWN_MSG_RELU_A_PLUS_B_SQ, CWN_REDUCE_ADD, SMALL>(D, blk, nblk, G, GF, part); break;
        case CWN_MSG_A_TIMES_2RELU:
            run_desc<VEC, CWN_MSG_A_TIMES_2RELU, CWN_REDUCE_ADD, SMALL>(D, blk, nblk, G, GF, part); break;
        default: run_desc_red<VEC, CWN_MSG_A, SMALL>(D, blk, nblk, G, GF, part); break;
    }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const float* __restrict__ src,
                                                               const int64_t* __restrict__ idx,
                                                               float* __restrict__ out, int64_t n_idx,
                                                               int F, int G) {
    const int rows_per_block = kThreads / G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t e = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / G;
    if (e >= n_idx) return;
    const int64_t r = idx[e];
    for (int f = gl * VEC; f < F; f += G * VEC) st<VEC>(out + e * F + f, ld<VEC>(src + r * F + f));
}

inline int pow2_at_least(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

inline int pick_group(int F, int vec) {
    int g = pow2_at_least((F + vec - 1) / vec);
    return g > 64 ? 64 : g;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }
inline bool aligned8(const void* p) { return ((uintptr_t)p & 7u) == 0; }

}  // namespace

extern "C" int cwn_aggregate_f32(const cwn_agg_desc* descs, int n, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_DESCS) return CWN_ERR_BAD_ARG;
    AggBatch B{};
    B.n = n;
    int vec = 4;
    for (int i = 0; i < n; ++i) {
        const cwn_agg_desc& D = descs[i];
        if (D.F <= 0 || D.n_dst < 0 || (D.n_dst > 0 && D.out == nullptr)) return CWN_ERR_BAD_ARG;
        if (D.msg_op < CWN_MSG_A || D.msg_op > CWN_MSG_A_TIMES_2RELU) return CWN_ERR_BAD_ARG;
        if (D.reduce < CWN_REDUCE_ADD || D.reduce > CWN_REDUCE_MAX) return CWN_ERR_BAD_ARG;
        if (D.msg_op >= CWN_MSG_RELU_A_PLUS_B && D.reduce != CWN_REDUCE_ADD) return CWN_ERR_BAD_ARG;
        if (D.rowptr != nullptr) {
            if (D.ia == nullptr || D.A == nullptr) return CWN_ERR_BAD_ARG;
            if (D.msg_op != CWN_MSG_A && (D.ib == nullptr || D.B == nullptr)) return CWN_ERR_BAD_ARG;
            if (D.msg_op != CWN_MSG_A && D.b_width != D.F && D.b_width != 1) return CWN_ERR_BAD_ARG;
            if ((D.msg_op == CWN_MSG_A_MASK_RELU || D.msg_op == CWN_MSG_A_TIMES_2RELU) && D.self_pre == nullptr) return CWN_ERR_BAD_ARG;
        }
        if (D.n_dst >= INT32_MAX) return CWN_ERR_TOO_LARGE;
        // widest vector every pointer and the row stride allow
        int v = (D.F % 4 == 0) ? 4 : (D.F % 2 == 0 ? 2 : 1);
        const void* ptrs[] = {D.A, D.b_width == D.F ? (const void*)D.B : nullptr, D.self_x,
                              D.self_pre, D.out, D.self_x2};
        for (const void* p : ptrs) {
            if (p == nullptr) continue;
            if (((uintptr_t)p & 3u) != 0) return CWN_ERR_ALIGN;
            if (v == 4 && !aligned16(p)) v = aligned8(p) ? 2 : 1;
            if (v == 2 && !aligned8(p)) v = 1;
        }
        if (v < vec) vec = v;
    }
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        B.d[i] = descs[i];
        B.fgroup[i] = pick_group(descs[i].F, vec);
        B.group[i] = B.fgroup[i] < 8 ? 8 : B.fgroup[i];     // narrow features: 8 lanes per row anyway
        const int rows_per_block = kThreads / B.group[i];
        B.blk_start[i] = (int32_t)blocks;
        blocks += (descs[i].n_dst + rows_per_block - 1) / rows_per_block;
        if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    }
    for (int i = n; i <= CWN_MAX_DESCS; ++i) B.blk_start[i] = (int32_t)blocks;
    if (blocks == 0) return CWN_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)blocks), block(kThreads);
    bool narrow = false, small = true;
    for (int i = 0; i < n; ++i) {
        narrow = narrow || B.fgroup[i] < B.group[i];
        // 32-bit byte offsets: the caller vouches for the gathered operands, the row-aligned ones
        // (out, self_x, self_x2, self_pre: [n_dst, F]) are checked here
        small = small && (descs[i].flags & CWN_AGG_SMALL_OPERANDS) != 0 &&
                (uint64_t)descs[i].n_dst * (uint64_t)descs[i].F * 4u < (1ull << 32);
    }
    auto launch = [&](auto kernel) { kernel<<<grid, block, 0, stream>>>(B); };
    const int variant = (vec == 4 ? 0 : vec == 2 ? 1 : 2) * 4 + (narrow ? 2 : 0) + (small ? 1 : 0);
    switch (variant) {
        case 0: launch(aggregate_kernel<4, false, false>); break;
        case 1: launch(aggregate_kernel<4, false, true>); break;
        case 2: launch(aggregate_kernel<4, true, false>); break;
        case 3: launch(aggregate_kernel<4, true, true>); break;
        case 4: launch(aggregate_kernel<2, false, false>); break;
        case 5: launch(aggregate_kernel<2, false, true>); break;
        case 6: launch(aggregate_kernel<2, true, false>); break;
        case 7: launch(aggregate_kernel<2, true, true>); break;
        case 8: launch(aggregate_kernel<1, false, false>); break;
        case 9: launch(aggregate_kernel<1, false, true>); break;
        case 10: launch(aggregate_kernel<1, true, false>); break;
        default: launch(aggregate_kernel<1, true, true>); break;
    }
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_gather_rows_f32(const float* src, int64_t n_src, int64_t F, const int64_t* idx,
                                   int64_t n_idx, float* out, cwn_stream_t stream_) {
    if (F <= 0 || n_idx < 0 || n_src < 0 || F >= INT32_MAX) return CWN_ERR_BAD_ARG;
    if (n_idx == 0) return CWN_OK;
    if (src == nullptr || idx == nullptr || out == nullptr) return CWN_ERR_BAD_ARG;
    if ((((uintptr_t)src) | ((uintptr_t)out)) & 3u) return CWN_ERR_ALIGN;
    int vec = (F % 4 == 0) ? 4 : (F % 2 == 0 ? 2 : 1);
    if (vec == 4 && !(aligned16(src) && aligned16(out))) vec = 2;
    if (vec == 2 && !(aligned8(src) && aligned8(out))) vec = 1;
    const int G = pick_group((int)F, vec);
    const int rows_per_block = kThreads / G;
    const int64_t blocks = (n_idx + rows_per_block - 1) / rows_per_block;
    if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)blocks), block(kThreads);
    if (vec == 4) gather_rows_kernel<4><<<grid, block, 0, stream>>>(src, idx, out, n_idx, (int)F, G);
    else if (vec == 2) gather_rows_kernel<2><<<grid, block, 0, stream>>>(src, idx, out, n_idx, (int)F, G);
    else gather_rows_kernel<1><<<grid, block, 0, stream>>>(src, idx, out, n_idx, (int)F, G);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_abi_version(void) { return CWN_ABI_VERSION; }

extern "C" const char* cwn_target_arch(void) { return "gfx950"; }

extern "C" const char* cwn_error_string(int code) {
    switch (code) {
        case CWN_OK: return "ok";
        case CWN_ERR_BAD_ARG: return "bad argument";
        case CWN_ERR_TOO_LARGE: return "size does not fit int32";
        case CWN_ERR_WORKSPACE: return "workspace too small";
        case CWN_ERR_LAUNCH: return "kernel launch failed";
        case CWN_ERR_ALIGN: return "pointer misses its alignment (4 bytes; 16 for outputs, weights and packed buffers of the vectorised kernels)";
        default: return "unknown error";
    }
}
