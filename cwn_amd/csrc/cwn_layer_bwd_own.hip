// cwn_layer_bwd_own.hip -- the backward of one SparseCIN propagate step in one launch, OWNER form: every row of dx is
// written once, by the workgroup that owns it.
//
// Forward (cwn_layer.hip; mp/layers.py:184-192, 290-295, 333-342), per dimension d:
//     out_up_d[i] = sum_{p: dst_p = i} relu(Y1_d[src_p] + Y2_{d+1}[cof_p]) + (1 + eps1_d) x_d[i]
//     out_b_d[i]  = sum_{b in boundary(i)} x_{d-1}[b]                     + (1 + eps2_d) x_d[i]
//     Y1_d = x_d W_d[:, :F]^T + bias_d   (rows: cells of d),    Y2_{d+1} = x_{d+1} W_d[:, F:]^T   (rows: cells of d+1)
// Backward, given gU_d = dL/d out_up_d and gB_d = dL/d out_b_d:
//     m_p       = gU_d[dst_p] * [Y1_d[src_p] + Y2_{d+1}[cof_p] > 0]                       per entry p of up_index_d
//     gY1_d[j]  = sum_{p: src_p = j} m_p           gY2_{d+1}[c] = sum_{p: cof_p = c} m_p    (-> the weight gradients)
//     dx_d      = (1 + eps1_d) gU_d + (1 + eps2_d) gB_d + gY1_d W_d[:, :F] + gY2_d W_{d-1}[:, F:]
//                 + sum_{i in dim d+1: b in boundary(i)} gB_{d+1}[i]
// The first blocked form (cwn_layer_bwd.hip) ran over the FORWARD's item table: an item computed the pieces its
// forward counterpart had produced, so a row of dx received pieces from up to three workgroups -- fp32 atomics onto a
// zeroed matrix, 25 us + the fill.  Here an item is a range of complexes for the dimension whose rows it OWNS
// (include/cwn_hip.h: cwn_layer_bwd_items_build): it stages what those rows receive -- its own Y1 / gU rows and the
// coface rows of Y2 for the masks of its dimension's entries, the rows of Y1 / gU of the dimension BELOW and its own
// rows of Y2 for the entries that name its cells as cofaces, the rows of gB of the dimension ABOVE for the boundary
// transposes -- gathers per owned row with one key per lane and a ballot (no float atomics, no sort, entry order),
// multiplies [gY1 | gY2] by the two transposed weights on the matrix cores (cwn_split.h: the forward's exact
// three-way bf16 split), adds the products onto the self terms + transposes in LDS in a fixed order and stores the
// rows of dx once.  Deterministic; the caller does not zero anything.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include <stdlib.h>
#include "../../include/cwn_hip.h"
#include "cwn_split.h"
#include "cwn_layer_bwd_own.h"

namespace {

namespace bo = cwn_bwd_own;
using cwn::frag_cd;

constexpr int kThreads = bo::kThreads;
constexpr int kWaves = kThreads / 64;
constexpr int kBnScratch = 160 * 1024 - bo::kLdsCap;   // [4][F] floats behind the item's LDS (cwn_layer_bwd_dim.out_bn)
static_assert(kBnScratch >= 4 * 128 * 4, "scratch of the BatchNorm sums");

struct OwnArgs {
    cwn_layer_bwd_dim d[CWN_LAYER_MAX_DIMS];
    const int32_t* items;
    int32_t* err;
    int32_t lds_bytes;
    int32_t n_dims;
    int32_t dbg;          // timing experiments (CWN_LBWD_DBG): 1 no entry walk, 2 no matrix cores, 4 no gY store, 8 no slot atomics,
                          // 16 no z / constant loads of the BatchNorm sums, 32 no BatchNorm sums at all
#ifdef CWN_LBWD_TIMING
    unsigned long long* stamps;              // [n_items][32]: [0, 16) points seen by wave 0 (first product), [16, 32) by the first wave of the second
#endif
};

// make -C cwn_amd/csrc bwdtiming: s_memtime at the end of every phase, per workgroup (tools/time_layer_bwd_phases.py)
#ifdef CWN_LBWD_TIMING
#define CWN_STAMP(k)                                                                                   \
    do {                                                                                               \
        if (A.stamps != nullptr && (threadIdx.x == 0 || threadIdx.x == 64 * G::kNCT))                  \
            A.stamps[(size_t)blockIdx.x * 32 + (threadIdx.x == 0 ? 0 : 16) + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CWN_STAMP(k) do { } while (0)
#endif

template <int F> struct Geo {
    static constexpr int kPlaneStride = F + 8;          // bf16 elements per plane row (fragment reads conflict-free)
    static constexpr int kYStride = F + 4;              // floats per staged / O row
    static constexpr int kKS = F / 32;
    static constexpr int kNCT = F / 16;
    static constexpr int kWPC = kWaves / kNCT / 2;      // waves sharing a column tile of ONE product (row-tile parity)
    static constexpr int kG = F / 4;                    // lanes per row
    static constexpr int kNG = kThreads / kG;           // rows per round
    static constexpr int kMaxOwn = (CWN_LAYER_GEMM_ROWS(F) + kNG - 1) / kNG;
    static constexpr int kMaxTiles = (CWN_LAYER_GEMM_ROWS(F) / 16 + kWPC - 1) / kWPC;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// The lanes of a group test kG keys AT ONCE (one each; the ballot gives the group its matches) and walk only the
// matches -- a cell's degree, two or three -- in entry order, TWO at a time: the index reads of both and then the row
// reads of both are in flight together (a match is two dependent LDS round trips: ~400 cycles each when walked one by
// one, and a ring of six is the coface of thirty entries).  Every lane of the wave runs the loop; a group without a live
// row asks for a key no entry has.  Chunks e_first, e_first + e_step, ...: the TOP rows are walked in slices.
template <int KG, class Fn>
__device__ __forceinline__ void walk_matches(const int* key, int n4, int e_first, int e_step, int want, int gl, int gsh, Fn&& fn) {
    int kv = e_first + gl < n4 ? key[e_first + gl] : -2;
    for (int e0 = e_first; e0 < n4; e0 += e_step) {
        const int en = e0 + e_step + gl;
        const int kv_next = en < n4 ? key[en] : -2;              // the next chunk's keys travel while this one's matches are walked
        const unsigned long long bal = __ballot(kv == want);
        unsigned m = (unsigned)((bal >> gsh) & (KG == 32 ? 0xffffffffull : 0xffffull));
        while (m != 0u) {
            const int q0 = e0 + __builtin_ctz(m);
            m &= m - 1u;
            const bool two = m != 0u;
            const int q1 = two ? e0 + __builtin_ctz(m) : q0;
            m &= m - 1u;
            fn(q0, q1, two);
        }
        kv = kv_next;
    }
}

__device__ __forceinline__ void add_masked(float4& acc, const float4& a, const float4& b, const float4& m) {
    acc.x += a.x + b.x > 0.f ? m.x : 0.f;
    acc.y += a.y + b.y > 0.f ? m.y : 0.f;
    acc.z += a.z + b.z > 0.f ? m.z : 0.f;
    acc.w += a.w + b.w > 0.f ? m.w : 0.f;
}

template <int F>
// (items_pre = A.items as a LEADING scalar argument: preloaded into SGPRs by the dispatcher, csrc/Makefile PRELOAD)
__global__ __launch_bounds__(kThreads) void layer_bwd_own_kernel(const int32_t* items_pre, OwnArgs A) {
    using G = Geo<F>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    CWN_STAMP(0);
    const int32_t* const it = items_pre + (size_t)blockIdx.x * CWN_LAYER_BWD_ITEM_INTS;
    const int flags = it[bo::R_FLAGS], d = it[bo::R_DIM];
    const int o_r0 = it[bo::R_OWN_R0], n_o = it[bo::R_OWN_N], a_r0 = it[bo::R_ABOVE_R0], n_a = it[bo::R_ABOVE_N];
    const int b_r0 = it[bo::R_BELOW_R0], n_b = it[bo::R_BELOW_N];
    const int ea0 = it[bo::R_UPA_E0], ne_a = it[bo::R_UPA_NE], eb0 = it[bo::R_UPB_E0], ne_b = it[bo::R_UPB_NE];
    const int bd0 = it[bo::R_BND_E0], ne_bd = it[bo::R_BND_NE];
    const bool PA = (flags & bo::F_PA) != 0, PB = (flags & bo::F_PB) != 0, TOP = (flags & bo::F_TOP) != 0;
    // an EMPTY record (all zero: a table of fixed capacity that this batch does not fill, cwn_layer_bwd_items_build_dev): nothing to do
    if (flags == 0 && n_o == 0 && n_a == 0 && n_b == 0 && ne_a == 0 && ne_b == 0 && ne_bd == 0) return;
    // table / launch mismatch (uniform over the workgroup, before any barrier)
    bool bad_rec = d < 0 || d >= A.n_dims || n_o <= 0 || n_o > bo::own_rows_cap(F) || n_a < 0 || n_b < 0 || n_a > 4096 || n_b > 4096 ||
                   ne_a < 0 || ne_b < 0 || ne_bd < 0 || ne_a > CWN_LAYER_MAX_ENTRIES || ne_b > CWN_LAYER_MAX_ENTRIES ||
                   ne_bd > CWN_LAYER_MAX_ENTRIES || (TOP && n_a > bo::top_rows_cap(F)) ||
                   ((PA || TOP || ne_bd > 0 || n_a > 0) && d + 1 >= A.n_dims) || (PB && d == 0) || (!PA && ne_a > 0) ||
                   (!PB && (ne_b > 0 || n_b > 0)) || (ne_bd > 0 && n_a <= 0) || (PA && ne_a > 0 && n_a <= 0);
    const bo::Layout L = bo::layout(F, flags, n_o, n_a, n_b, ne_a, ne_b, ne_bd);
    bad_rec = bad_rec || L.total > A.lds_bytes;
    if (bad_rec) {
        if (tid == 0) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
        return;
    }
    const cwn_layer_bwd_dim& Dd = A.d[d];
    const cwn_layer_bwd_dim& Da = A.d[d + 1 < A.n_dims ? d + 1 : d];
    const cwn_layer_bwd_dim& Db = A.d[d > 0 ? d - 1 : 0];

    float* const S = reinterpret_cast<float*>(smem);                       // staged rows, stride F + 4
    uint16_t* const planes = reinterpret_cast<uint16_t*>(smem);           // [3][pl_rows][F + 8], after the walk
    float* const O = reinterpret_cast<float*>(smem + L.o_off);            // [RO + RT][F + 4]
    // entries of up_index_d: keys (source, coface: local rows) and the LDS offsets (floats) of the rows an entry's term reads
    int* const ej = reinterpret_cast<int*>(smem + L.ent_off);             // key: source
    int* const ec = ej + L.ea4;                                           // key: coface
    int2* const eo = reinterpret_cast<int2*>(ec + L.ea4);                 // {gU row of the destination, Y2 row of the coface}
    int2* const et = eo + L.ea4;                                          // {Y1 row of the source, gU row of the destination} (TOP)
    // entries of up_index_{d-1}: key coface (an owned cell); {Y1 row of the source, gU row of the destination} below
    int* const fc = reinterpret_cast<int*>(et + L.ea4);
    int2* const fo = reinterpret_cast<int2*>(fc + L.eb4);
    // entries of b_index_{d+1}: key boundary cell (owned); gB row of the cell above
    int* const bb = reinterpret_cast<int*>(fo + L.eb4);
    int* const bo = bb + L.bd4;
    float* const P = reinterpret_cast<float*>(smem + L.p_off);           // [kNG][F + 4]: partial sums of the TOP rows (TOP only)
    const size_t plane = (size_t)L.pl_rows * G::kPlaneStride;
    const int gl = tid % G::kG, gq = tid / G::kG, f = gl * 4;
    constexpr int YS = G::kYStride;

    const int ct = wave % G::kNCT, w2 = wave / G::kNCT;
    const int my_h = w2 & 1, rt_par = w2 >> 1;

    CWN_STAMP(1);
    // ---- 1. requests: the entries (one list position per thread, the three lists side by side), the rows, the weight ----
    // ONE round trip: every request of a thread's first round -- its list position, its row of each staged block, its
    // weight slice -- leaves before anything is waited for; the later rounds of larger items follow as ordinary loops.
    // What the phase then costs is its bytes: an edges + rings item stages 167 rows (85 KB; its own Y1 / Y2 / gU / gB, the
    // vertices' Y1 / gU, the rings' Y2 / gU / gB), the launch 22 MB at the ZINC batch of 128 -- ~3.6 us until the rows are
    // in, a third of the workgroup's time (tools/time_layer_bwd_phases.py).  Lining the waves up between row and weight
    // requests (an s_barrier) did not change it: the weights are not what the rows wait behind.
    const int n_ent = L.ea4 + L.eb4 + L.bd4;
    int bad = 0;
    struct Ent { int64_t x, y, z; };
    auto ent_load = [&](int p) {
        Ent e = {0, 0, 0};
        if (p < L.ea4) {
            const int q = min(p, ne_a - 1);
            e.x = Dd.up_index[ea0 + q];
            e.y = Dd.up_index[Dd.e_up + ea0 + q];
            e.z = Dd.up_shared[ea0 + q];
        } else if (p < L.ea4 + L.eb4) {
            const int q = min(p - L.ea4, ne_b - 1);
            e.x = Db.up_index[eb0 + q];
            e.y = Db.up_index[Db.e_up + eb0 + q];
            e.z = Db.up_shared[eb0 + q];
        } else if (p < n_ent) {
            const int q = min(p - L.ea4 - L.eb4, ne_bd - 1);
            e.x = Da.b_index[bd0 + q];
            e.y = Da.b_index[Da.n_b + bd0 + q];
        }
        return e;
    };
    auto ent_store = [&](int p, const Ent& e) {
        if (p < L.ea4) {                          // up_index_d: source and destination among the owned cells, coface above
            const int64_t jj = e.x - o_r0, ii = e.y - o_r0, cc = e.z - a_r0;
            const int ok = (int)((uint64_t)jj < (uint64_t)n_o) & (int)((uint64_t)ii < (uint64_t)n_o) & (int)((uint64_t)cc < (uint64_t)n_a);
            const int live = (int)(p < ne_a), on = live & ok;
            bad |= live & (ok ^ 1);
            const int j = on ? (int)jj : 0, i = on ? (int)ii : 0, c = on ? (int)cc : 0;
            ej[p] = on ? j : -1;                                           // -1: matches no row
            ec[p] = on ? c : -1;
            eo[p] = make_int2((L.guo + i) * YS, (L.y2a + c) * YS);
            et[p] = make_int2((L.y1o + j) * YS, (L.guo + i) * YS);
        } else if (p < L.ea4 + L.eb4) {           // up_index_{d-1}: source and destination below, coface among the owned cells
            const int pp = p - L.ea4;
            const int64_t jj = e.x - b_r0, ii = e.y - b_r0, cc = e.z - o_r0;
            const int ok = (int)((uint64_t)jj < (uint64_t)n_b) & (int)((uint64_t)ii < (uint64_t)n_b) & (int)((uint64_t)cc < (uint64_t)n_o);
            const int live = (int)(pp < ne_b), on = live & ok;
            bad |= live & (ok ^ 1);
            fc[pp] = on ? (int)cc : -1;
            fo[pp] = make_int2((L.y1b + (on ? (int)jj : 0)) * YS, (L.gub + (on ? (int)ii : 0)) * YS);
        } else if (p < n_ent) {                   // b_index_{d+1}: boundary cell among the owned cells, cell above
            const int pp = p - L.ea4 - L.eb4;
            const int64_t bc = e.x - o_r0, ic = e.y - a_r0;
            const int ok = (int)((uint64_t)bc < (uint64_t)n_o) & (int)((uint64_t)ic < (uint64_t)n_a);
            const int live = (int)(pp < ne_bd), on = live & ok;
            bad |= live & (ok ^ 1);
            bb[pp] = on ? (int)bc : -1;
            bo[pp] = (L.gba + (on ? (int)ic : 0)) * YS;
        }
    };
    const bool gu_d = Dd.g_up != nullptr, gb_d = Dd.g_b != nullptr, gu_a = Da.g_up != nullptr, gb_a = Da.g_b != nullptr;
    const bool gu_b = Db.g_up != nullptr;
    const float s1 = 1.0f + (Dd.eps1 != nullptr ? *Dd.eps1 : 0.f), s2 = 1.0f + (Dd.eps2 != nullptr ? *Dd.eps2 : 0.f);
    const float t1 = 1.0f + (Da.eps1 != nullptr ? *Da.eps1 : 0.f), t2 = 1.0f + (Da.eps2 != nullptr ? *Da.eps2 : 0.f);
    const bool need_gba = ne_bd > 0 || TOP;
    struct Own { float4 y1, y2, gu, gb; };
    struct Abv { float4 y2, gu, gb; };
    struct Blw { float4 y1, gu; };
    // owned rows: Y1 (A), gU (A: staged; always: self term), gB (self term), Y2 at d (B); O = self terms
    auto own_load = [&](int r) {
        Own v = {zero4(), zero4(), zero4(), zero4()};
        if (r < n_o) {
            const size_t go = (size_t)(o_r0 + r) * F + f;
            if (PA) v.y1 = ld4(Dd.y1 + go);
            if (PB) v.y2 = ld4(Dd.y2 + go);
            if (gu_d) v.gu = ld4(Dd.g_up + go);
            if (gb_d) v.gb = ld4(Dd.g_b + go);
        }
        return v;
    };
    auto own_store = [&](int r, const Own& v) {
        if (r < n_o) {
            if (PA) {
                st4(S + (size_t)(L.y1o + r) * YS + f, v.y1);
                st4(S + (size_t)(L.guo + r) * YS + f, v.gu);
            }
            if (PB) st4(S + (size_t)(L.y2o + r) * YS + f, v.y2);
            st4(O + (size_t)r * YS + f, make_float4(s1 * v.gu.x + s2 * v.gb.x, s1 * v.gu.y + s2 * v.gb.y, s1 * v.gu.z + s2 * v.gb.z,
                                                    s1 * v.gu.w + s2 * v.gb.w));
        }
    };
    // rows of the dimension above: Y2 at d+1 (masks of A), gB (boundary transposes), TOP: their self terms into O
    auto abv_load = [&](int r) {
        Abv v = {zero4(), zero4(), zero4()};
        if (r < n_a) {
            const size_t go = (size_t)(a_r0 + r) * F + f;
            if (PA) v.y2 = ld4(Da.y2 + go);
            if (need_gba && gb_a) v.gb = ld4(Da.g_b + go);
            if (TOP && gu_a) v.gu = ld4(Da.g_up + go);
        }
        return v;
    };
    auto abv_store = [&](int r, const Abv& v) {
        if (r < n_a) {
            if (PA) st4(S + (size_t)(L.y2a + r) * YS + f, v.y2);
            if (ne_bd > 0) st4(S + (size_t)(L.gba + r) * YS + f, v.gb);
            if (TOP)
                st4(O + (size_t)(L.RO + r) * YS + f, make_float4(t1 * v.gu.x + t2 * v.gb.x, t1 * v.gu.y + t2 * v.gb.y,
                                                                 t1 * v.gu.z + t2 * v.gb.z, t1 * v.gu.w + t2 * v.gb.w));
        }
    };
    // rows of the dimension below (B): Y1 and gU of the entries whose cofaces are owned here
    auto blw_load = [&](int r) {
        Blw v = {zero4(), zero4()};
        if (PB && r < n_b) {
            const size_t go = (size_t)(b_r0 + r) * F + f;
            v.y1 = ld4(Db.y1 + go);
            if (gu_b) v.gu = ld4(Db.g_up + go);
        }
        return v;
    };
    auto blw_store = [&](int r, const Blw& v) {
        if (PB && r < n_b) {
            st4(S + (size_t)(L.y1b + r) * YS + f, v.y1);
            st4(S + (size_t)(L.gub + r) * YS + f, v.gu);
        }
    };
    const Ent e_first = ent_load(tid);
    const Own o_first = own_load(gq);
    const Abv a_first = abv_load(gq);
    const Blw b_first = blw_load(gq);
    // this wave's slice of a TRANSPOSED packed weight (cwn_layer_pack_weights_t_many_f32; the forward's chunk order):
    // waves with my_h = 0 multiply gY1_d by W_d[:, :F] (half 0 of dimension d's), waves with my_h = 1 the gradient of Y2 at d by
    // W_{d-1}[:, F:] (half 1 of the dimension below's).  Requested after the rows (they would queue behind it), used last.
    uint4 wsp[G::kKS][3];
    const bool mine = (my_h == 0 ? PA : PB) && !(A.dbg & 2);
    auto load_w = [&](uint4 (&w)[G::kKS][3], const void* packed, int h) {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(packed) + (size_t)ct * 1024 + lane * 16;
#pragma unroll
        for (int ks = 0; ks < G::kKS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                w[ks][pl] = *reinterpret_cast<const uint4*>(wp + (size_t)(((ks * 3 + pl) * 2 + h) * G::kNCT) * 1024);
    };
    if (mine) load_w(wsp, my_h == 0 ? Dd.wt_packed : Db.wt_packed, my_h);
    __builtin_amdgcn_sched_barrier(0);       // every request above is out before the first result is waited for
    ent_store(tid, e_first);
    own_store(gq, o_first);
    abv_store(gq, a_first);
    blw_store(gq, b_first);
    for (int p = tid + kThreads; p < n_ent; p += kThreads) ent_store(p, ent_load(p));
    for (int r = gq + G::kNG; r < n_o; r += G::kNG) own_store(r, own_load(r));
    for (int r = gq + G::kNG; r < n_a; r += G::kNG) abv_store(r, abv_load(r));
    for (int r = gq + G::kNG; r < n_b; r += G::kNG) blw_store(r, blw_load(r));
    if (bad) atomicOr(A.err, CWN_ERR_BIT_BLOCK);
    CWN_STAMP(2);
    __syncthreads();
    CWN_STAMP(3);

    // ---- 2. every lane group OWNS rows gq, gq + kNG, ...: it gathers what each receives, out of LDS, in entry order -----
    float4 g1[G::kMaxOwn], g2[G::kMaxOwn];
    float4 g3 = zero4();
    const int gsh = (lane / G::kG) * G::kG;                      // this group's bits of the wave's ballot
    const bool walk = !(A.dbg & 1);
#pragma unroll
    for (int k = 0; k < G::kMaxOwn; ++k) {
        g1[k] = zero4();
        g2[k] = zero4();
        const int r = gq + k * G::kNG;
        if (k * G::kNG >= n_o) continue;                         // (uniform) no row of this round is owned
        const bool live = r < n_o;
        const int rc = live ? r : 0, want = live ? r : -3;
        if (PA && walk) {                                        // entries of up_index_d whose SOURCE is the row: gY1_d
            const float4 a = ld4(S + (L.y1o + rc) * YS + f);
            float4 acc = zero4();
            walk_matches<G::kG>(ej, L.ea4, 0, G::kG, want, gl, gsh, [&](int q0, int q1, bool two) {
                const int2 o0 = eo[q0], o1 = eo[q1];
                const float4 m0 = ld4(S + o0.x + f), b0 = ld4(S + o0.y + f), m1 = ld4(S + o1.x + f), b1 = ld4(S + o1.y + f);
                add_masked(acc, a, b0, m0);
                if (two) add_masked(acc, a, b1, m1);
            });
            g1[k] = acc;
        }
        if (PB && walk) {                                        // entries of up_index_{d-1} whose COFACE is the row: gY2 at d
            const float4 b = ld4(S + (L.y2o + rc) * YS + f);
            float4 acc = zero4();
            walk_matches<G::kG>(fc, L.eb4, 0, G::kG, want, gl, gsh, [&](int q0, int q1, bool two) {
                const int2 o0 = fo[q0], o1 = fo[q1];
                const float4 a0 = ld4(S + o0.x + f), m0 = ld4(S + o0.y + f), a1 = ld4(S + o1.x + f), m1 = ld4(S + o1.y + f);
                add_masked(acc, a0, b, m0);
                if (two) add_masked(acc, a1, b, m1);
            });
            g2[k] = acc;
        }
        if (ne_bd > 0 && walk) {                                 // entries of b_index_{d+1} whose BOUNDARY CELL is the row
            float4 acc = zero4();
            walk_matches<G::kG>(bb, L.bd4, 0, G::kG, want, gl, gsh, [&](int q0, int q1, bool two) {
                const float4 m0 = ld4(S + bo[q0] + f), m1 = ld4(S + bo[q1] + f);
                acc.x += m0.x; acc.y += m0.y; acc.z += m0.z; acc.w += m0.w;
                if (two) { acc.x += m1.x; acc.y += m1.y; acc.z += m1.z; acc.w += m1.w; }
            });
            if (live) {
                float* const o = O + r * YS + f;
                const float4 v = ld4(o);
                st4(o, make_float4(v.x + acc.x, v.y + acc.y, v.z + acc.z, v.w + acc.w));
            }
        }
    }
    CWN_STAMP(4);
    // entries of up_index_d whose COFACE is a TOP row: gY2 at d+1.  A TOP row has MANY matches (a ring of six: thirty) and
    // there are few TOP rows: lane group gq gathers row gq mod R2 (R2 = rows rounded up to a power of two) over slice
    // gq / R2 of the entry chunks into P[gq]; after the barrier the row's owner adds the slices in order.
    int top_r2 = 1;
    while (top_r2 < n_a) top_r2 <<= 1;
    const int top_slices = G::kNG / top_r2;                      // a power of two
    const bool do_top = TOP && PA && n_a > 0 && walk;
    if (do_top) {
        // (Slices of the entry CHUNKS did not spread the work: the entries of one ring are neighbours in the list -- one
        // chunk, one slice.  Every group looks at every chunk; a lane whose key matches counts the matches before it
        // (popcount of the ballot below its own bit) and keeps its own when that running number is the group's modulo
        // the slices; a second ballot gives the group the few it walks.  Numbering the matches bit by bit in a loop was
        // ~20 VALU instructions per match: with four waves on a SIMD, slower than the walk it spread.)
        const int row = gq & (top_r2 - 1), sl = gq / top_r2;
        const bool live = row < n_a;
        const int rc = live ? row : 0, want = live ? row : -3;
        const float4 b = ld4(S + (L.y2a + rc) * YS + f);
        float4 part = zero4();
        auto pair = [&](int q0, int q1, bool two) {
            const int2 o0 = et[q0], o1 = et[q1];
            const float4 a0 = ld4(S + o0.x + f), m0 = ld4(S + o0.y + f), a1 = ld4(S + o1.x + f), m1 = ld4(S + o1.y + f);
            add_masked(part, a0, b, m0);
            if (two) add_masked(part, a1, b, m1);
        };
        constexpr unsigned kGroupBits = G::kG == 32 ? 0xffffffffu : 0xffffu;
        int seen = 0, pend = -1;
        for (int e0 = 0; e0 < L.ea4; e0 += G::kG) {
            const int e = e0 + gl;
            const int kv = e < L.ea4 ? ec[e] : -2;
            const bool hit = kv == want;
            const unsigned m = (unsigned)(__ballot(hit) >> gsh) & kGroupBits;
            const int rank = seen + __popc(m & ((1u << gl) - 1u));
            const bool take = hit && (rank & (top_slices - 1)) == sl;
            unsigned m2 = (unsigned)(__ballot(take) >> gsh) & kGroupBits;
            seen += __popc(m);
            while (m2 != 0u) {
                const int q = e0 + __builtin_ctz(m2);
                m2 &= m2 - 1u;
                if (pend < 0) {
                    pend = q;
                } else {
                    pair(pend, q, true);
                    pend = -1;
                }
            }
        }
        if (pend >= 0) pair(pend, pend, false);
        st4(P + gq * YS + f, part);
    }
    CWN_STAMP(5);
    __syncthreads();                     // every group is done with the staged rows: the planes take their place
    CWN_STAMP(6);

    // ---- 3. the gradients of the products out (the weight-gradient GEMMs read them) and into the bf16 planes ----------
    auto to_planes = [&](int prow, const float4& v) {
        uint2 ph, pm, pl;
        cwn::split4(v, ph, pm, pl);
        uint16_t* dst = planes + (size_t)prow * G::kPlaneStride + f;
        *reinterpret_cast<uint2*>(dst) = ph;
        *reinterpret_cast<uint2*>(dst + plane) = pm;
        *reinterpret_cast<uint2*>(dst + 2 * plane) = pl;
    };
    const bool store_gy = !(A.dbg & 4);
    if (do_top && gq < n_a) {
        for (int sl = 0; sl < top_slices; ++sl) {
            const float4 v = ld4(P + (size_t)(sl * top_r2 + gq) * YS + f);
            g3.x += v.x; g3.y += v.y; g3.z += v.z; g3.w += v.w;
        }
    }
#pragma unroll
    for (int k = 0; k < G::kMaxOwn; ++k) {
        const int r = gq + k * G::kNG;
        if (r < n_o) {
            const size_t go = (size_t)(o_r0 + r) * F + f;
            if (PA) {
                if (Dd.gy1 != nullptr && store_gy) st4(Dd.gy1 + go, g1[k]);
                to_planes(L.pl_a + r, g1[k]);
            }
            if (PB) {
                if (Dd.gy2 != nullptr && store_gy) st4(Dd.gy2 + go, g2[k]);
                to_planes(L.pl_b + r, g2[k]);
            }
        }
    }
    if (TOP && gq < n_a) {
        if (Da.gy2 != nullptr && store_gy) st4(Da.gy2 + (size_t)(a_r0 + gq) * F + f, g3);
        to_planes(L.pl_c + gq, g3);
    }
    // the third product (TOP rows x W_d[:, F:]) is the my_h = 1 waves', after their own: its weight slice is requested at
    // the start of their path below, into a second set of registers (the gathered rows have left theirs), and lands under
    // their first product and the first round of adds.  (Requested after that product it was ~2 us of bare L2 latency.)
    const bool do_c = my_h == 1 && TOP && PA && n_a > 0 && !(A.dbg & 2);
    CWN_STAMP(7);
    __syncthreads();
    CWN_STAMP(8);

    // ---- 4. the products on the matrix cores, added onto O in a FIXED order: B, then A (own rows); C (top rows) --------
    // Operand roles not swapped (A operand = rows of gY, B operand = rows of the transposed weight = output columns): a
    // lane holds rows 4 kq + reg of output column l15 of its tile.  Rows of a tile beyond the block's cells hold what the
    // staged rows left there: every output row depends on its own operand row only, and those rows are never added.
    using cwn::as_frag;
    // (One accumulation chain per tile.  Two were tried for the waves the others wait for: the matrix pipe of a SIMD is
    // shared by its four waves and busy throughout this phase -- 14 tiles x 24 MFMAs x 16 cycles at the ZINC sizes -- so a
    // second chain only cost registers: four of the weight registers went to scratch.)
    auto product = [&](const uint4 (&w)[G::kKS][3], int prow0) {
        frag_cd acc = {0.f, 0.f, 0.f, 0.f};
        const uint16_t* p0 = planes + (size_t)(prow0 + l15) * G::kPlaneStride + kq * 8;
#pragma unroll
        for (int ks = 0; ks < G::kKS; ++ks) {
            const uint4 xh = *reinterpret_cast<const uint4*>(p0 + ks * 32);
            const uint4 xm = *reinterpret_cast<const uint4*>(p0 + plane + ks * 32);
            const uint4 xl = *reinterpret_cast<const uint4*>(p0 + 2 * plane + ks * 32);
            const uint4 &wh = w[ks][0], &wm = w[ks][1], &wl = w[ks][2];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xh), as_frag(wl), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xl), as_frag(wh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xm), as_frag(wm), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xh), as_frag(wm), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xm), as_frag(wh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xh), as_frag(wh), acc, 0, 0, 0);
        }
        return acc;
    };
    // D[i][j]: i = row 4 kq + reg of the tile, j = output column l15
    auto add_tile = [&](int orow0, int n_rows_left, const frag_cd& acc) {
        float* const o = O + (size_t)(orow0 + 4 * kq) * YS + ct * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * kq + r < n_rows_left) o[(size_t)r * YS] += acc[r];
    };
    const int T = L.RO / 16;
    // ---- (round 6) the REDUCE half of the BatchNorm backward of the stage whose OUTPUT x_d is (include/cwn_hip.h:
    // cwn_layer_bwd_dim.out_bn): the rows of dx this workgroup owns are that stage's dy, complete -- their column sums of
    // dx * mask and dx * mask * xhat leave from here, and the previous layer's backward launches no reduce of its own.
    // Scratch [own: F sums of dyh | F of dyh * xhat | top rows: the same two] behind the item's LDS, zeroed here (two
    // barriers in front of its first add); z of the owned rows and the stage's constants are requested behind the products,
    // in front of the last barrier (across the products the registers do not exist: 128 of 128).
    // (the records are read where they are used -- behind the products: read here their fields were live, in registers this
    // kernel does not have, across the whole matrix-core phase)
    float* const bn_sc = reinterpret_cast<float*>(smem + A.lds_bytes);
    if (tid < 4 * F) bn_sc[tid] = 0.f;
    float4 zo[G::kMaxOwn], zt = zero4();
    float4 oa[4], ta[4];                                 // scale, shift, mean, rstd at this lane's four columns
    bool live_o = false, live_t = false;
    auto request_bn = [&]() {
        const cwn_bn_bwd_live& LBo = A.d[d].out_bn;
        const cwn_bn_bwd_live& LBt = A.d[d + 1 < A.n_dims ? d + 1 : d].out_bn;
        live_o = LBo.slots != nullptr && !(A.dbg & 32);
        live_t = TOP && n_a > 0 && LBt.slots != nullptr && !(A.dbg & 32);
        if (live_o && !(A.dbg & 16)) {
#pragma unroll
            for (int k = 0; k < G::kMaxOwn; ++k)
                if (k * G::kNG < n_o) zo[k] = ld4(LBo.z + (size_t)(o_r0 + min(gq + k * G::kNG, n_o - 1)) * LBo.ldz + f);
#pragma unroll
            for (int q = 0; q < 4; ++q) oa[q] = ld4(LBo.aff + q * F + f);
        }
        if (live_t) {
            zt = ld4(LBt.z + (size_t)(a_r0 + min(gq, n_a - 1)) * LBt.ldz + f);
#pragma unroll
            for (int q = 0; q < 4; ++q) ta[q] = ld4(LBt.aff + q * F + f);
        }
    };
    // Two code paths (the branch is wave-uniform; both meet the same two barriers): the waves of the SECOND product add
    // first, tile by tile, and then run the third; the waves of the FIRST keep their tiles until the barrier.
    if (my_h == 1) {
        uint4 wsp2[G::kKS][3];
        if (do_c) load_w(wsp2, Dd.wt_packed, 1);
        if (mine) {
            for (int rt = rt_par; rt < T; rt += G::kWPC) {
                const frag_cd acc = product(wsp, L.pl_b + rt * 16);
                add_tile(rt * 16, n_o - rt * 16, acc);
            }
        }
        CWN_STAMP(9);
        __syncthreads();
        CWN_STAMP(10);
        if (do_c) {
            for (int rt = rt_par; rt < L.RT / 16; rt += G::kWPC) {
                const frag_cd acc = product(wsp2, L.pl_c + rt * 16);
                add_tile(L.RO + rt * 16, n_a - rt * 16, acc);
            }
        }
        CWN_STAMP(11);
        request_bn();
        __syncthreads();
    } else {
        frag_cd accs[G::kMaxTiles];
        if (mine) {
#pragma unroll
            for (int t = 0; t < G::kMaxTiles; ++t) {
                const int rt = rt_par + t * G::kWPC;
                if (rt < T) accs[t] = product(wsp, L.pl_a + rt * 16);
            }
        }
        CWN_STAMP(9);
        __syncthreads();
        CWN_STAMP(10);
        if (mine) {
#pragma unroll
            for (int t = 0; t < G::kMaxTiles; ++t) {
                const int rt = rt_par + t * G::kWPC;
                if (rt < T) add_tile(rt * 16, n_o - rt * 16, accs[t]);
            }
        }
        CWN_STAMP(11);
        request_bn();
        __syncthreads();
    }
    CWN_STAMP(12);

    // ---- 5. the rows of dx, once (+ their share of the BatchNorm sums) -------------------------------------------------
    auto bn_add = [&](float (&a1)[4], float (&a2)[4], const float4& dxv, const float4& zv, const float4 (&c)[4]) {
        const float d_[4] = {dxv.x, dxv.y, dxv.z, dxv.w}, z_[4] = {zv.x, zv.y, zv.z, zv.w};
        const float sc[4] = {c[0].x, c[0].y, c[0].z, c[0].w}, sh[4] = {c[1].x, c[1].y, c[1].z, c[1].w};
        const float mu[4] = {c[2].x, c[2].y, c[2].z, c[2].w}, rs[4] = {c[3].x, c[3].y, c[3].z, c[3].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {                    // (cwn_stage.hip's producer epilogue: the same arithmetic)
            const float y = z_[q] * sc[q] + sh[q];
            const float dyh = y > 0.f ? d_[q] : 0.f;
            a1[q] += dyh;
            a2[q] += dyh * ((z_[q] - mu[q]) * rs[q]);
        }
    };
    float o1[4] = {0.f, 0.f, 0.f, 0.f}, o2[4] = {0.f, 0.f, 0.f, 0.f}, t1_[4] = {0.f, 0.f, 0.f, 0.f}, t2_[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < G::kMaxOwn; ++k) {
        const int r = gq + k * G::kNG;
        if (r < n_o) {
            const float4 v = ld4(O + (size_t)r * YS + f);
            st4(Dd.dx + (size_t)(o_r0 + r) * F + f, v);
            if (live_o) bn_add(o1, o2, v, zo[k], oa);
        }
    }
    if (TOP && gq < n_a) {
        const float4 v = ld4(O + (size_t)(L.RO + gq) * YS + f);
        st4(Da.dx + (size_t)(a_r0 + gq) * F + f, v);
        if (live_t) bn_add(t1_, t2_, v, zt, ta);
    }
    if (live_o || live_t) {                              // (uniform over the workgroup)
        // lane groups of a wave -> its first group (xor tree), waves -> workgroup through LDS, 4 F threads -> the slots
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int off = G::kG; off < 64; off <<= 1) {
                o1[q] += __shfl_xor(o1[q], off, 64);
                o2[q] += __shfl_xor(o2[q], off, 64);
                t1_[q] += __shfl_xor(t1_[q], off, 64);
                t2_[q] += __shfl_xor(t2_[q], off, 64);
            }
        }
        // Waves -> workgroup.  LDS float atomics would be the short form -- and measured ~10 us per launch: ds_add_f32 takes its
        // lanes one by one (256 wave instructions, ~100 cycles each).  The region of the staged rows / planes is dead behind the
        // last barrier: every wave stores its 4 F partial sums there, 4 F threads add the sixteen in wave order.  An item whose
        // region is smaller than that (a few cells) keeps the atomics.
        float* const part = reinterpret_cast<float*>(smem);                 // [kWaves][4 F]
        const bool wide = (size_t)L.o_off >= (size_t)kWaves * 4 * F * sizeof(float);
        if (lane < G::kG) {
            if (wide) {
                float* const w = part + (size_t)wave * 4 * F + f;
                st4(w, make_float4(o1[0], o1[1], o1[2], o1[3]));
                st4(w + F, make_float4(o2[0], o2[1], o2[2], o2[3]));
                st4(w + 2 * F, make_float4(t1_[0], t1_[1], t1_[2], t1_[3]));
                st4(w + 3 * F, make_float4(t2_[0], t2_[1], t2_[2], t2_[3]));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (live_o) {
                        __hip_atomic_fetch_add(bn_sc + f + q, o1[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(bn_sc + F + f + q, o2[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (live_t) {
                        __hip_atomic_fetch_add(bn_sc + 2 * F + f + q, t1_[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(bn_sc + 3 * F + f + q, t2_[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        // (a barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global STORE -- the rows of
        // dx and of gY this workgroup has just written)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float total = 0.f;
        if (tid < 4 * F) {
            if (wide) {
#pragma unroll
                for (int w = 0; w < kWaves; ++w) total += part[(size_t)w * 4 * F + tid];
            } else {
                total = bn_sc[tid];
            }
        }
        const int slot = (int)(blockIdx.x % CWN_BN_SLOTS);
        if (live_o && tid < 2 * F && !(A.dbg & 8)) unsafeAtomicAdd(A.d[d].out_bn.slots + (size_t)slot * 2 * F + tid, total);
        if (live_t && tid >= 2 * F && tid < 4 * F && !(A.dbg & 8))
            unsafeAtomicAdd(A.d[d + 1 < A.n_dims ? d + 1 : d].out_bn.slots + (size_t)slot * 2 * F + (tid - 2 * F), total);
    }
    CWN_STAMP(13);
}

#ifdef CWN_LBWD_TIMING
unsigned long long* g_stamps = nullptr;
#endif

inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; }

template <int F>
int launch(const OwnArgs& A, int64_t n_items, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr = hipSuccess;
    std::call_once(once, [] {
        attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_bwd_own_kernel<F>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr != hipSuccess) return CWN_ERR_LAUNCH;
#ifdef CWN_LBWD_TIMING
    OwnArgs B = A;
    B.stamps = g_stamps;
    layer_bwd_own_kernel<F><<<dim3((unsigned)n_items), dim3(kThreads), (size_t)A.lds_bytes + kBnScratch, stream>>>(B.items, B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
#endif
    layer_bwd_own_kernel<F><<<dim3((unsigned)n_items), dim3(kThreads), (size_t)A.lds_bytes + kBnScratch, stream>>>(A.items, A);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

}  // namespace

#ifdef CWN_LBWD_TIMING
extern "C" void cwn_layer_bwd_own_debug_stamps(unsigned long long* buf) { g_stamps = buf; }
#endif

extern "C" int cwn_layer_bwd_own_f32(const cwn_layer_bwd_dim* dims, int n_dims, int32_t F, const cwn_layer_bwd_plan* plan,
                                     int32_t* err_flag, cwn_stream_t stream_) {
    if (dims == nullptr || plan == nullptr || n_dims < 1 || n_dims > CWN_LAYER_MAX_DIMS || plan->n_items < 0)
        return CWN_ERR_BAD_ARG;
    if (F != 64 && F != 128) return CWN_ERR_BAD_ARG;
    const int64_t n_items = plan->n_items;
    if (n_items == 0) return CWN_OK;
    if (plan->items == nullptr || err_flag == nullptr) return CWN_ERR_BAD_ARG;
    if (n_items >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    if (plan->lds_bytes <= 0 || plan->lds_bytes + kBnScratch > 160 * 1024) return CWN_ERR_TOO_LARGE;
    if (!al16(plan->items)) return CWN_ERR_ALIGN;
    OwnArgs A{};
    for (int d = 0; d < n_dims; ++d) {
        const cwn_layer_bwd_dim& D = dims[d];
        if (D.n_cells < 0 || D.e_up < 0 || D.n_b < 0) return CWN_ERR_BAD_ARG;
        if (D.n_cells > 0 && D.dx == nullptr) return CWN_ERR_BAD_ARG;
        if (plan->up_end[d] > 0 && (D.up_index == nullptr || D.up_shared == nullptr || D.wt_packed == nullptr || D.y1 == nullptr ||
                                    d + 1 >= n_dims || dims[d + 1].y2 == nullptr))
            return CWN_ERR_BAD_ARG;
        if (plan->b_end[d] > 0 && (D.b_index == nullptr || d == 0)) return CWN_ERR_BAD_ARG;
        if (!(al16(D.g_up) && al16(D.g_b) && al16(D.y1) && al16(D.y2) && al16(D.dx) && al16(D.gy1) && al16(D.gy2) &&
              al16(D.wt_packed)))
            return CWN_ERR_ALIGN;
        if (D.out_bn.slots != nullptr) {     // the stage whose output x_d is: z [n_cells, F] (16-byte rows), aff [4][F]
            if (D.out_bn.z == nullptr || D.out_bn.aff == nullptr || D.out_bn.ldz < F || (D.out_bn.ldz & 3)) return CWN_ERR_BAD_ARG;
            if (!(al16(D.out_bn.z) && al16(D.out_bn.aff)) || ((uintptr_t)D.out_bn.slots & 3u)) return CWN_ERR_ALIGN;
        }
        if (plan->cells_end[d] < 0 || plan->cells_end[d] > D.n_cells || plan->up_end[d] < 0 || plan->up_end[d] > D.e_up ||
            plan->b_end[d] < 0 || plan->b_end[d] > D.n_b)
            return CWN_ERR_BAD_ARG;
        A.d[d] = D;
    }
    for (int d = n_dims; d < CWN_LAYER_MAX_DIMS; ++d)
        if (plan->cells_end[d] != 0 || plan->up_end[d] != 0 || plan->b_end[d] != 0) return CWN_ERR_BAD_ARG;
    A.items = plan->items;
    A.err = err_flag;
    A.lds_bytes = (int32_t)((plan->lds_bytes + 15) & ~(int64_t)15);
    A.n_dims = n_dims;
    static const int dbg = getenv("CWN_LBWD_DBG") ? atoi(getenv("CWN_LBWD_DBG")) : 0;
    A.dbg = dbg;
    return F == 128 ? launch<128>(A, n_items, (hipStream_t)stream_) : launch<64>(A, n_items, (hipStream_t)stream_);
}
