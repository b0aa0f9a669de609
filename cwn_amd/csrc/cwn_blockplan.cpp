// cwn_blockplan.cpp -- the item table of the complex-blocked layer kernel, built on the HOST (no GPU code).
//
// A batched complex is a disjoint union: the reference's collate offsets every index per complex
// (data/complex.py:148-169) and records where each complex's cells and index entries lie (`ptr`, data/complex.py:344,
// 432; `__slices__`, :349-394).  From those per-complex prefix sums this file cuts a batch into ITEMS -- contiguous
// ranges of complexes for one GEMM dimension, one workgroup each (record layout: include/cwn_hip.h) -- greedily
// under the caps of the kernel, and chooses how one launch's LDS is split between staged rows and boundary sources
// so that the items are as few as they can be.  A first version did this in Python (11 ms for a ZINC-like batch of
// 128, 90 ms for 1024: more than the forward pass it prepares); this one is a few tens of microseconds.
#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "../../include/cwn_hip.h"
#include "cwn_layer_bwd_own.h"

namespace {

constexpr int kInts = CWN_LAYER_ITEM_INTS;
constexpr int kTargetItemsDefault = 128;      // per GEMM dimension: ~one workgroup per CU over the two sets of a 2-complex
// (CWN_LAYER_TARGET_ITEMS: tuning experiments -- how many complexes an item may hold at most = C / target)
static const int kTargetItems = [] {
    const char* e = getenv("CWN_LAYER_TARGET_ITEMS");
    const int v = e != nullptr ? atoi(e) : 0;
    return v > 0 ? v : kTargetItemsDefault;
}();

inline int64_t pad16(int64_t n) { return (n + 15) / 16 * 16; }
inline int64_t pad4(int64_t n) { return (n + 3) / 4 * 4; }

struct Shape {
    int F, round_rows, variant;
    int64_t kLds;                      // LDS budget of a workgroup of this variant
    int64_t half_cap;                  // bound of the (padded) rows of ONE product, or 0: only the sum is bounded
    // (round 4: a multiple of 16, no longer of the kernel's rows per round -- cwn_layer.hip, the load rounds decide per wave)
    int64_t first_coface_row(int64_t n_g, int64_t /*n_c*/) const { return pad16(n_g); }
    int64_t staged(int64_t n_g, int64_t n_c) const { return n_c > 0 ? first_coface_row(n_g, n_c) + pad16(n_c) : pad16(n_g); }
    // = cwn_layer_fused_lds_bytes without its argument checks
    int64_t lds(int64_t rows, int64_t src) const {
        static const int64_t idx = (int64_t)cwn_layer_fused_lds_bytes(128, 16, 0) - 3 * 16 * (128 + 8) * 2 - 128 * 4;   // same in both variants
        return 3 * rows * (F + 8) * 2 + (src + 1) * F * 4 + idx;
    }
};

struct Set { int g; int tasks[2]; int n_tasks; };

// sets in ascending order of dimension: a dimension with an upper adjacency is the GEMM dimension of a set (the top
// dimension rides as its second task when it has none itself), any other dimension is a set of its own
int make_sets(const cwn_layer_sizes& in, Set (&sets)[CWN_LAYER_MAX_DIMS]) {
    int n = 0;
    for (int d = 0; d < in.n_dims;) {
        Set& s = sets[n++];
        s.tasks[0] = d;
        s.n_tasks = 1;
        if (in.has_up[d]) {
            s.g = d;
            if (d + 1 < in.n_dims && !in.has_up[d + 1] && d + 2 >= in.n_dims) s.tasks[s.n_tasks++] = d + 1;
        } else {
            s.g = -1;
        }
        d += s.n_tasks;
    }
    return n;
}

// one greedy cut under (row_cap, src_cap).  >= 1 items, 0 nothing to do / bad input, -1 a single complex exceeds a cap
int64_t build_with(const cwn_layer_sizes& in, const Shape& sh, int64_t row_cap, int64_t src_cap, std::vector<int32_t>& out,
                   cwn_layer_plan& plan) {
    const int64_t C = in.n_complexes;
    out.clear();
    Set sets[CWN_LAYER_MAX_DIMS];
    const int n_sets = make_sets(in, sets);
    const int64_t gmax = std::max<int64_t>(1, C / (sh.variant == 1 ? 2 * kTargetItems : kTargetItems));   // two workgroups a CU
    int64_t max_rows = 0, max_src = 0, max_item_lds = 0, n_big = 0;
    std::vector<int32_t> recs;
    std::vector<int64_t> weight;
    std::vector<int64_t> order;
    for (int s_ = 0; s_ < n_sets; ++s_) {
        const Set& S = sets[s_];
        const int g = S.g, d0 = S.tasks[0];
        const int64_t* up = g >= 0 ? in.up_ptr[g] : nullptr;
        const int64_t* bp[2] = {nullptr, nullptr};
        for (int t = 0; t < S.n_tasks; ++t)
            if (S.tasks[t] > 0) bp[t] = in.b_ptr[S.tasks[t]];
        auto cells = [&](int d, int64_t a, int64_t b) { return in.cell_ptr[d][b] - in.cell_ptr[d][a]; };
        auto span = [](const int64_t* p, int64_t a, int64_t b) { return p ? p[b] - p[a] : (int64_t)0; };
        recs.clear();
        weight.clear();
        for (int64_t c0 = 0; c0 < C;) {
            if (in.skip != nullptr && in.skip[c0]) { ++c0; continue; }      // not in this table (ranges break at it)
            int64_t c1 = c0;
            while (c1 < C && c1 - c0 < gmax && !(in.skip != nullptr && in.skip[c1])) {
                const int64_t nxt = c1 + 1;
                const int64_t rows = sh.staged(cells(d0, c0, nxt), g >= 0 ? cells(g + 1, c0, nxt) : 0);
                int64_t src = 0, ents = pad4(span(up, c0, nxt));
                bool ok = true;
                for (int t = 0; t < S.n_tasks; ++t) {
                    const int d = S.tasks[t];
                    // (two-per-CU form: only task 0's sources are loaded into LDS; task 1 reads the staged cells of g)
                    if (d > 0 && span(bp[t], c0, nxt) > 0 && (sh.variant == 0 || t == 0)) src += cells(d - 1, c0, nxt);
                    ents += pad4(span(bp[t], c0, nxt));
                    ok = ok && cells(d, c0, nxt) <= CWN_LAYER_TASK_ROWS;
                }
                ok = ok && rows <= row_cap && src <= src_cap && sh.lds(rows, src) <= sh.kLds && ents <= CWN_LAYER_MAX_ENTRIES;
                if (sh.half_cap > 0 && g >= 0)
                    ok = ok && pad16(cells(d0, c0, nxt)) <= sh.half_cap && pad16(cells(g + 1, c0, nxt)) <= sh.half_cap;
                if (!ok) break;
                c1 = nxt;
            }
            bool big = c1 == c0;                // not even this one complex fits a workgroup
            if (big && in.unfit != nullptr) {   // the caller serves it with another launch: marked, left out
                in.unfit[c0] = 1;
                ++c0;
                continue;
            }
            if (big) {
                if (!in.allow_big || sh.variant != 0) return -1;
                c1 = c0 + 1;                    // a BIG record of its own: the workgroup streams it (include/cwn_hip.h)
                ++n_big;
            }
            int32_t r[kInts] = {0};
            r[0] = (s_ << 8) | (big ? CWN_LAYER_ITEM_BIG : 0);
            const int64_t n0 = cells(d0, c0, c1);
            int64_t nc = 0, une = 0;
            int live = S.n_tasks;
            if (g >= 0) {
                r[1] = g;
                if (n0 > 0) {
                    nc = cells(g + 1, c0, c1);
                    une = span(up, c0, c1);
                    r[0] |= 1;
                    r[2] = (int32_t)in.cell_ptr[g][c0];
                    r[3] = (int32_t)n0;
                    r[4] = (int32_t)in.cell_ptr[g + 1][c0];
                    r[5] = (int32_t)nc;
                    r[6] = (int32_t)(up ? up[c0] : 0);
                    r[7] = (int32_t)une;
                } else {
                    for (int t = 1; t < S.n_tasks; ++t)
                        if (cells(S.tasks[t], c0, c1) > 0) return -1;   // cells of g + 1 without cells of g: not a cell complex
                    live = 1;
                }
            }
            if (!big) max_rows = std::max(max_rows, sh.staged(n0, nc));
            r[8] = live;
            int64_t src = 0, src_lds = 0, bne[2] = {0, 0};
            for (int t = 0; t < live; ++t) {
                const int d = S.tasks[t], o = 9 + 7 * t;
                bne[t] = span(bp[t], c0, c1);
                r[o] = d;
                r[o + 1] = (int32_t)in.cell_ptr[d][c0];
                r[o + 2] = (int32_t)cells(d, c0, c1);
                r[o + 3] = (int32_t)(bp[t] ? bp[t][c0] : 0);
                r[o + 4] = (int32_t)bne[t];
                if (d > 0 && bne[t] > 0) {          // boundary sources are staged only when read
                    r[o + 5] = (int32_t)in.cell_ptr[d - 1][c0];
                    r[o + 6] = (int32_t)cells(d - 1, c0, c1);
                    src += r[o + 6];
                    if (sh.variant == 0 || t == 0) src_lds += r[o + 6];
                }
            }
            if (!big) {
                max_src = std::max(max_src, sh.variant == 0 ? src : src_lds);
                max_item_lds = std::max(max_item_lds, sh.lds(sh.staged(n0, nc), src_lds));
                const int64_t b1 = pad4(une), b2 = pad4(b1 + bne[0]);
                r[23] = (int32_t)sh.first_coface_row(n0, nc);
                r[24] = (int32_t)sh.staged(n0, nc);
                r[25] = (int32_t)b1;
                r[26] = (int32_t)b2;
                r[27] = (int32_t)pad4(b2 + bne[1]);
            }
            recs.insert(recs.end(), r, r + kInts);
            weight.push_back((int64_t)r[11] + r[5]);       // (a big item is the heaviest of its set: it starts first)
            c0 = c1;
        }
        // heavy items first within the set: a workgroup with five row tiles should not start last
        const int64_t n = (int64_t)weight.size();
        order.resize(n);
        for (int64_t i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return weight[a] > weight[b]; });
        plan.set_start[s_] = (int32_t)(out.size() / kInts);
        for (int64_t i = 0; i < n; ++i) out.insert(out.end(), recs.begin() + order[i] * kInts, recs.begin() + (order[i] + 1) * kInts);
    }
    for (int s_ = n_sets; s_ <= CWN_LAYER_MAX_DIMS; ++s_) plan.set_start[s_] = 0;
    plan.n_items = (int64_t)(out.size() / kInts);
    plan.max_gemm_rows = (int32_t)std::max<int64_t>(max_rows, 16);
    plan.max_source_rows = (int32_t)max_src;
    plan.variant = sh.variant;
    plan.lds_bytes = sh.variant == 1 ? max_item_lds : 0;
    plan.n_big = n_big;
    for (int d = 0; d < CWN_LAYER_MAX_DIMS; ++d) {
        const bool on = d < in.n_dims;
        plan.cells_end[d] = on ? in.cell_ptr[d][C] : 0;
        plan.up_end[d] = on && in.has_up[d] && in.up_ptr[d] ? in.up_ptr[d][C] : 0;
        plan.b_end[d] = on && d > 0 && in.b_ptr[d] ? in.b_ptr[d][C] : 0;
    }
    return plan.n_items;
}

}  // namespace

extern "C" int64_t cwn_layer_items_build(const cwn_layer_sizes* in, int32_t F, int32_t* items, int64_t cap_items,
                                         cwn_layer_plan* plan) {
    if (in == nullptr || plan == nullptr || (F != 64 && F != 128) || in->n_dims < 1 || in->n_dims > CWN_LAYER_MAX_DIMS ||
        in->n_complexes < 0 || (cap_items > 0 && items == nullptr))
        return CWN_LAYER_ITEMS_BAD_ARG;
    if (in->n_complexes == 0) return 0;
    // every table is a prefix sum: starts at 0, never decreases, and its total fits the int32 fields of a record
    auto prefix_sum = [n = in->n_complexes](const int64_t* p) {
        if (p[0] != 0) return false;
        for (int64_t c = 0; c < n; ++c)
            if (p[c + 1] < p[c]) return false;
        return p[n] <= INT32_MAX;
    };
    for (int d = 0; d < in->n_dims; ++d) {
        if (in->cell_ptr[d] == nullptr || !prefix_sum(in->cell_ptr[d])) return CWN_LAYER_ITEMS_BAD_ARG;
        if (in->has_up[d] && (d + 1 >= in->n_dims || in->up_ptr[d] == nullptr)) return CWN_LAYER_ITEMS_BAD_ARG;
        if (in->up_ptr[d] != nullptr && !prefix_sum(in->up_ptr[d])) return CWN_LAYER_ITEMS_BAD_ARG;
        if (in->b_ptr[d] != nullptr && !prefix_sum(in->b_ptr[d])) return CWN_LAYER_ITEMS_BAD_ARG;
    }
    const int variant = plan->variant;
    if (variant != 0 && variant != 1) return CWN_LAYER_ITEMS_BAD_ARG;
    const Shape sh{F, cwn_layer_variant_round_rows(F, variant), variant, variant == 1 ? (int64_t)CWN_LAYER_W8_LDS_BYTES : (int64_t)160 * 1024,
                   variant == 1 ? (int64_t)CWN_LAYER_W8_HALF_ROWS(F) : (int64_t)0};
    const int64_t kLds = sh.kLds;
    if (sh.round_rows <= 0) return CWN_LAYER_ITEMS_BAD_ARG;
    // one launch = one LDS size: the planes for the LARGEST staged block of any item plus the sources of the item with
    // the most of them (different items, in general).  A few splits of the LDS between the two are tried -- row cap
    // from the top down, the source cap = what is left -- and the one with the fewest items wins
    const int64_t cap = variant == 1 ? CWN_LAYER_W8_GEMM_ROWS(F) : CWN_LAYER_GEMM_ROWS(F);
    const int64_t src_max = variant == 1 ? CWN_LAYER_W8_SOURCE_ROWS(F) : CWN_LAYER_SOURCE_ROWS(F), step = std::max<int64_t>(16, cap / 8);
    std::vector<int32_t> cur, best;
    int64_t floor_items = 0;
    {
        Set sets_[CWN_LAYER_MAX_DIMS];
        const int n_sets_ = make_sets(*in, sets_);
        const int64_t gmax_ = std::max<int64_t>(1, in->n_complexes / (variant == 1 ? 2 * kTargetItems : kTargetItems));
        floor_items = (in->skip == nullptr) ? n_sets_ * ((in->n_complexes + gmax_ - 1) / gmax_) : -1;
    }
    cwn_layer_plan pc = *plan, pb = *plan;
    bool have = false, too_big = false;
    for (int64_t row_cap = cap; row_cap >= step; row_cap -= step) {
        // variant 1: every item lays out its own rows, so rows and sources are coupled per ITEM (inside build_with) and
        // one pass with both caps at their maxima is the whole search
        const int64_t src_cap = variant == 1 ? src_max : std::min(src_max, (kLds - sh.lds(row_cap, -1)) / (F * 4) - 1);
        if (src_cap < 16) continue;
        const int64_t n = build_with(*in, sh, row_cap, src_cap, cur, pc);
        if (n < 0) { too_big = true; if (variant == 1) break; continue; }
        if (n == 0 || (variant == 0 && sh.lds(pc.max_gemm_rows, pc.max_source_rows) > kLds)) continue;
        if (variant == 1) { best.swap(cur); pb = pc; have = true; break; }
        if (!have || pc.n_big < pb.n_big || (pc.n_big == pb.n_big && n < pb.n_items)) {      // fewest streamed complexes first
            best.swap(cur);
            pb = pc;
            have = true;
        } else if (n > pb.n_items + pb.n_items / 8) {
            break;                               // getting worse: smaller row caps only split more
        }
        // no table has fewer items than one per gmax complexes and set: the search is over (the usual case at small
        // batches: one complex per item whatever the split -- six more greedy passes were half of the 88 us a table cost)
        if (have && pb.n_big == 0 && pb.n_items == floor_items) break;
    }
    if (!have) return too_big ? CWN_LAYER_ITEMS_TOO_LARGE : 0;
    if (pb.n_items > cap_items) return CWN_LAYER_ITEMS_BAD_ARG;
    std::copy(best.begin(), best.end(), items);
    const int32_t* keep_items = plan->items;
    void* keep_cache = plan->csr_cache;
    *plan = pb;
    plan->items = keep_items;
    plan->csr_cache = keep_cache;
    return pb.n_items;
}

// ---- the OWNER form of the backward launch (include/cwn_hip.h: cwn_layer_bwd_own_f32) ------------------------------------
// Same sets as the forward (a top dimension without upper adjacency rides with the one below), one greedy cut per set
// under the limits of the kernel; the LDS of an item follows from its record (cwn_layer_bwd_own.h).
extern "C" int64_t cwn_layer_bwd_items_build(const cwn_layer_sizes* in, int32_t F, int32_t* items, int64_t cap_items,
                                             cwn_layer_bwd_plan* plan) {
    namespace bo = cwn_bwd_own;
    if (in == nullptr || plan == nullptr || (F != 64 && F != 128) || in->n_dims < 1 || in->n_dims > CWN_LAYER_MAX_DIMS ||
        in->n_complexes < 0 || (cap_items > 0 && items == nullptr))
        return CWN_LAYER_ITEMS_BAD_ARG;
    const int64_t C = in->n_complexes;
    if (C == 0) return 0;
    auto prefix_sum = [C](const int64_t* p) {
        if (p[0] != 0) return false;
        for (int64_t c = 0; c < C; ++c)
            if (p[c + 1] < p[c]) return false;
        return p[C] <= INT32_MAX;
    };
    for (int d = 0; d < in->n_dims; ++d) {
        if (in->cell_ptr[d] == nullptr || !prefix_sum(in->cell_ptr[d])) return CWN_LAYER_ITEMS_BAD_ARG;
        if (in->has_up[d] && (d + 1 >= in->n_dims || in->up_ptr[d] == nullptr)) return CWN_LAYER_ITEMS_BAD_ARG;
        if (in->up_ptr[d] != nullptr && !prefix_sum(in->up_ptr[d])) return CWN_LAYER_ITEMS_BAD_ARG;
        if (in->b_ptr[d] != nullptr && !prefix_sum(in->b_ptr[d])) return CWN_LAYER_ITEMS_BAD_ARG;
    }
    Set sets[CWN_LAYER_MAX_DIMS];
    const int n_sets = make_sets(*in, sets);
    const int64_t gmax = std::max<int64_t>(1, C / kTargetItems);
    constexpr int kI = CWN_LAYER_BWD_ITEM_INTS;
    std::vector<int32_t> out, recs;
    std::vector<int64_t> weight, order;
    int64_t max_lds = 0;
    auto span = [](const int64_t* p, int64_t a, int64_t b) { return p ? p[b] - p[a] : (int64_t)0; };
    for (int s_ = 0; s_ < n_sets; ++s_) {
        const int d = sets[s_].tasks[0];
        const bool top = sets[s_].n_tasks == 2, pa = in->has_up[d] != 0, pb = d > 0 && in->has_up[d - 1] != 0;
        const bool above = d + 1 < in->n_dims;
        const int64_t* upa = pa ? in->up_ptr[d] : nullptr;
        const int64_t* upb = pb ? in->up_ptr[d - 1] : nullptr;
        const int64_t* bnd = above ? in->b_ptr[d + 1] : nullptr;
        const int flags = (pa ? bo::F_PA : 0) | (pb ? bo::F_PB : 0) | (top ? bo::F_TOP : 0);
        auto cells = [&](int dd, int64_t a, int64_t b) { return in->cell_ptr[dd][b] - in->cell_ptr[dd][a]; };
        // what an item over complexes [a, b) needs; false: beyond a limit
        auto fits = [&](int64_t a, int64_t b, bo::Layout* Lout) {
            const int64_t n_o = cells(d, a, b), n_a = above ? cells(d + 1, a, b) : 0, n_b = pb ? cells(d - 1, a, b) : 0;
            const int64_t ea = span(upa, a, b), eb = span(upb, a, b), bd = span(bnd, a, b);
            if (n_o > bo::own_rows_cap(F) || (top && n_a > bo::top_rows_cap(F))) return false;
            if (ea > CWN_LAYER_MAX_ENTRIES || eb > CWN_LAYER_MAX_ENTRIES || bd > CWN_LAYER_MAX_ENTRIES) return false;
            if (n_a > 4096 || n_b > 4096) return false;                       // (keeps the layout arithmetic far inside int32)
            const bool need_a = pa || top || bd > 0;
            const bo::Layout L = bo::layout(F, flags, (int)n_o, need_a ? (int)n_a : 0, (int)n_b, (int)ea, (int)eb, (int)bd);
            if (Lout != nullptr) *Lout = L;
            return L.total <= bo::kLdsCap;
        };
        recs.clear();
        weight.clear();
        for (int64_t c0 = 0; c0 < C;) {
            int64_t c1 = c0;
            while (c1 < C && c1 - c0 < gmax && fits(c0, c1 + 1, nullptr)) ++c1;
            if (c1 == c0) return CWN_LAYER_ITEMS_TOO_LARGE;                   // not even this one complex fits a workgroup
            bo::Layout L;
            fits(c0, c1, &L);
            const int64_t n_o = cells(d, c0, c1), n_a = above ? cells(d + 1, c0, c1) : 0;
            const int64_t ea = span(upa, c0, c1), eb = span(upb, c0, c1), bd = span(bnd, c0, c1);
            if (n_o == 0) {
                // cells of d+1 (or entries) without cells of d: not a cell complex
                if ((top && n_a > 0) || ea > 0 || eb > 0 || bd > 0) return CWN_LAYER_ITEMS_TOO_LARGE;
                c0 = c1;
                continue;                                                     // nothing to write
            }
            const bool need_a = pa || top || bd > 0;
            int32_t r[kI] = {0};
            r[bo::R_FLAGS] = flags | (s_ << 8);
            r[bo::R_DIM] = d;
            r[bo::R_OWN_R0] = (int32_t)in->cell_ptr[d][c0];
            r[bo::R_OWN_N] = (int32_t)n_o;
            if (need_a) {
                r[bo::R_ABOVE_R0] = (int32_t)in->cell_ptr[d + 1][c0];
                r[bo::R_ABOVE_N] = (int32_t)n_a;
            }
            if (pb) {
                r[bo::R_BELOW_R0] = (int32_t)in->cell_ptr[d - 1][c0];
                r[bo::R_BELOW_N] = (int32_t)cells(d - 1, c0, c1);
                r[bo::R_UPB_E0] = (int32_t)upb[c0];
                r[bo::R_UPB_NE] = (int32_t)eb;
            }
            if (pa) {
                r[bo::R_UPA_E0] = (int32_t)upa[c0];
                r[bo::R_UPA_NE] = (int32_t)ea;
            }
            if (bd > 0) {
                r[bo::R_BND_E0] = (int32_t)bnd[c0];
                r[bo::R_BND_NE] = (int32_t)bd;
            }
            r[bo::R_LDS_BYTES] = L.total;
            max_lds = std::max<int64_t>(max_lds, L.total);
            recs.insert(recs.end(), r, r + kI);
            weight.push_back(n_o + n_a + ea + eb);
            c0 = c1;
        }
        const int64_t n = (int64_t)weight.size();
        order.resize(n);
        for (int64_t i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return weight[a] > weight[b]; });
        for (int64_t i = 0; i < n; ++i) out.insert(out.end(), recs.begin() + order[i] * kI, recs.begin() + (order[i] + 1) * kI);
    }
    const int64_t n_items = (int64_t)(out.size() / kI);
    if (n_items > cap_items) return CWN_LAYER_ITEMS_BAD_ARG;
    std::copy(out.begin(), out.end(), items);
    plan->n_items = n_items;
    plan->lds_bytes = max_lds;
    for (int d = 0; d < CWN_LAYER_MAX_DIMS; ++d) {
        const bool on = d < in->n_dims;
        plan->cells_end[d] = on ? in->cell_ptr[d][C] : 0;
        plan->up_end[d] = on && in->has_up[d] && in->up_ptr[d] ? in->up_ptr[d][C] : 0;
        plan->b_end[d] = on && d > 0 && in->b_ptr[d] ? in->b_ptr[d][C] : 0;
    }
    return n_items;
}
