// cwn_lift.cpp -- graph -> 2-complex lifting on the HOST (integer work that feeds the hot path):
//   * ring lift   (data/utils.py:400-498 compute_ring_2complex, rings found by graph-tool's induced
//                  subgraph isomorphism, :300-330): 2-cells are the chordless cycles with
//                  3..max_k vertices;
//   * clique lift (data/utils.py:224-272 compute_clique_complex_with_gudhi, expansion_dim 2):
//                  2-cells are the triangles.
// and the adjacency structures of build_adj (:103-138): upper / lower adjacency index pairs with
// their shared (co)boundary cell, boundary indices.  The reference does this through graph-tool and
// gudhi (C++ libraries that are not in this image); here it is plain C++ behind the C ABI, with the
// cell and entry ORDER of the reference (cells sorted lexicographically by vertex tuple; pairs in
// itertools.combinations order, both directions interleaved), pinned by the expected tensors of
// data/test_utils.py through tests/test_lifting.py.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <map>
#include <thread>
#include <vector>
#include "../../include/cwn_hip.h"

namespace {

using Vec = std::vector<int64_t>;

struct Pairs {            // an adjacency in reference layout: index [2, L] (row-major) + shared [L]
    Vec row0, row1, shared;
    void add_group(int64_t gid, const Vec& members) {
        for (size_t a = 0; a < members.size(); ++a)
            for (size_t b = a + 1; b < members.size(); ++b) {
                row0.push_back(members[a]); row1.push_back(members[b]); shared.push_back(gid);
                row0.push_back(members[b]); row1.push_back(members[a]); shared.push_back(gid);
            }
    }
};

}  // namespace

struct cwn_lift_s {
    int64_t n = 0;
    std::vector<std::pair<int64_t, int64_t>> edges;      // sorted (u < v)
    std::vector<Vec> cells2;                              // vertex tuples of the 2-cells
    std::vector<Vec> cells2_edges;                        // boundary edge ids of the 2-cells
    Pairs up0, up1, down1, down2;
    Vec out[CWN_LIFT_N_ARRAYS];
};

namespace {

int64_t edge_id(const cwn_lift_s& L, int64_t u, int64_t v) {
    if (u > v) std::swap(u, v);
    auto it = std::lower_bound(L.edges.begin(), L.edges.end(), std::make_pair(u, v));
    return it - L.edges.begin();
}

// chordless cycles with 3..max_k vertices: every cycle is grown from its smallest vertex s along
// paths of larger vertices; a candidate w may touch the path only at its last vertex (and at s,
// which closes the cycle).  Keyed by sorted vertex set (a chordless cycle is determined by it).
struct CycleFinder {
    const std::vector<Vec>& adj;
    int max_k;
    std::vector<char> in_path;
    Vec path;
    std::map<Vec, Vec> found;        // sorted vertex set -> vertices in cyclic order

    CycleFinder(const std::vector<Vec>& a, int k) : adj(a), max_k(k), in_path(a.size(), 0) {}

    void extend() {
        const int64_t s = path.front(), last = path.back();
        for (int64_t w : adj[last]) {
            if (w <= s || in_path[w]) continue;
            bool closing = false, chord = false;
            for (int64_t t : adj[w]) {
                if (t == s) closing = true;
                else if (in_path[t] && t != last) { chord = true; break; }
            }
            if (chord) continue;
            if (closing) {           // path has >= 2 vertices, so the cycle has >= 3
                Vec cyc = path;
                cyc.push_back(w);
                Vec key = cyc;
                std::sort(key.begin(), key.end());
                if (!found.count(key)) {
                    if (cyc[1] > cyc.back()) std::reverse(cyc.begin() + 1, cyc.end());
                    found.emplace(std::move(key), std::move(cyc));
                }
                continue;            // a neighbour of s cannot be interior to a longer chordless cycle
            }
            if ((int)path.size() + 1 < max_k) {
                path.push_back(w);
                in_path[w] = 1;
                extend();
                in_path[w] = 0;
                path.pop_back();
            }
        }
    }

    void run() {
        const int64_t n = (int64_t)adj.size();
        for (int64_t s = 0; s < n; ++s)
            for (int64_t w : adj[s]) {
                if (w <= s) continue;
                path = {s, w};
                in_path[s] = in_path[w] = 1;
                extend();
                in_path[s] = in_path[w] = 0;
            }
    }
};

void lower_pairs(Pairs& P, const std::vector<Vec>& members_of_cell, int64_t n_lower) {
    std::vector<Vec> cof((size_t)n_lower);
    for (size_t cid = 0; cid < members_of_cell.size(); ++cid)
        for (int64_t m : members_of_cell[cid]) cof[(size_t)m].push_back((int64_t)cid);
    for (int64_t g = 0; g < n_lower; ++g) P.add_group(g, cof[(size_t)g]);
}

Vec index2(const Pairs& P) {          // [2, L] row-major
    Vec o;
    o.reserve(2 * P.row0.size());
    o.insert(o.end(), P.row0.begin(), P.row0.end());
    o.insert(o.end(), P.row1.begin(), P.row1.end());
    return o;
}

}  // namespace

extern "C" cwn_lift_t* cwn_lift_create(int kind, int64_t n, const int64_t* edges, int64_t n_edges,
                                       int max_k, int include_down) {
    if (n < 0 || n_edges < 0 || (n_edges > 0 && edges == nullptr)) return nullptr;
    if (kind != CWN_LIFT_RING && kind != CWN_LIFT_CLIQUE) return nullptr;
    cwn_lift_s* L = new cwn_lift_s();
    L->n = n;
    for (int64_t e = 0; e < n_edges; ++e) {
        int64_t u = edges[2 * e], v = edges[2 * e + 1];
        if (u < 0 || v < 0 || u >= n || v >= n || u == v) { delete L; return nullptr; }
        if (u > v) std::swap(u, v);
        L->edges.emplace_back(u, v);
    }
    std::sort(L->edges.begin(), L->edges.end());
    L->edges.erase(std::unique(L->edges.begin(), L->edges.end()), L->edges.end());
    const int64_t E = (int64_t)L->edges.size();
    std::vector<Vec> adj((size_t)n);
    for (auto& e : L->edges) { adj[(size_t)e.first].push_back(e.second); adj[(size_t)e.second].push_back(e.first); }
    for (auto& a : adj) std::sort(a.begin(), a.end());

    if (kind == CWN_LIFT_RING) {
        if (max_k >= 3) {
            CycleFinder F(adj, max_k);
            F.run();
            for (auto& kv : F.found) {           // std::map: sorted by vertex set = reference order
                const Vec& cyc = kv.second;
                Vec es;
                for (size_t i = 0; i < cyc.size(); ++i) es.push_back(edge_id(*L, cyc[i], cyc[(i + 1) % cyc.size()]));
                std::sort(es.begin(), es.end());  // get_ring_boundaries (:355-367): sorted edge tuples
                L->cells2.push_back(cyc);
                L->cells2_edges.push_back(std::move(es));
            }
        }
    } else {
        for (auto& e : L->edges) {               // triangles (u < v < w), lexicographic
            const Vec &a = adj[(size_t)e.first], &b = adj[(size_t)e.second];
            size_t i = 0, j = 0;
            while (i < a.size() && j < b.size()) {
                if (a[i] < b[j]) ++i;
                else if (a[i] > b[j]) ++j;
                else {
                    if (a[i] > e.second) L->cells2.push_back({e.first, e.second, a[i]});
                    ++i; ++j;
                }
            }
        }
        std::sort(L->cells2.begin(), L->cells2.end());
        for (auto& t : L->cells2)
            L->cells2_edges.push_back({edge_id(*L, t[0], t[1]), edge_id(*L, t[0], t[2]), edge_id(*L, t[1], t[2])});
    }

    // adjacencies (build_adj): vertices share an edge; edges share a 2-cell; lower: share a boundary
    for (int64_t e = 0; e < E; ++e) L->up0.add_group(e, {L->edges[(size_t)e].first, L->edges[(size_t)e].second});
    for (size_t c = 0; c < L->cells2_edges.size(); ++c) L->up1.add_group((int64_t)c, L->cells2_edges[c]);
    if (include_down) {
        std::vector<Vec> ev((size_t)E);
        for (int64_t e = 0; e < E; ++e) ev[(size_t)e] = {L->edges[(size_t)e].first, L->edges[(size_t)e].second};
        lower_pairs(L->down1, ev, n);
        lower_pairs(L->down2, L->cells2_edges, E);
    }

    Vec* o = L->out;
    for (auto& e : L->edges) { o[CWN_LIFT_EDGES].push_back(e.first); o[CWN_LIFT_EDGES].push_back(e.second); }
    o[CWN_LIFT_CELLS2_PTR].push_back(0);
    for (auto& c : L->cells2) {
        o[CWN_LIFT_CELLS2_VERTS].insert(o[CWN_LIFT_CELLS2_VERTS].end(), c.begin(), c.end());
        o[CWN_LIFT_CELLS2_PTR].push_back((int64_t)o[CWN_LIFT_CELLS2_VERTS].size());
    }
    o[CWN_LIFT_UP0] = index2(L->up0);       o[CWN_LIFT_COB0] = L->up0.shared;
    o[CWN_LIFT_UP1] = index2(L->up1);       o[CWN_LIFT_COB1] = L->up1.shared;
    o[CWN_LIFT_DOWN1] = index2(L->down1);   o[CWN_LIFT_BND1] = L->down1.shared;
    o[CWN_LIFT_DOWN2] = index2(L->down2);   o[CWN_LIFT_BND2] = L->down2.shared;
    {   // boundary_index of the edges: [[u0, v0, u1, v1, ...], [0, 0, 1, 1, ...]]
        Vec r0, r1;
        for (int64_t e = 0; e < E; ++e) {
            r0.push_back(L->edges[(size_t)e].first); r0.push_back(L->edges[(size_t)e].second);
            r1.push_back(e); r1.push_back(e);
        }
        r0.insert(r0.end(), r1.begin(), r1.end());
        o[CWN_LIFT_BINDEX1] = std::move(r0);
    }
    {   // boundary_index of the 2-cells
        Vec r0, r1;
        for (size_t c = 0; c < L->cells2_edges.size(); ++c)
            for (int64_t e : L->cells2_edges[c]) { r0.push_back(e); r1.push_back((int64_t)c); }
        r0.insert(r0.end(), r1.begin(), r1.end());
        o[CWN_LIFT_BINDEX2] = std::move(r0);
    }
    return L;
}

extern "C" int64_t cwn_lift_size(const cwn_lift_t* L, int which) {
    if (L == nullptr || which < 0 || which >= CWN_LIFT_N_ARRAYS) return -1;
    return (int64_t)L->out[which].size();
}

extern "C" int cwn_lift_copy(const cwn_lift_t* L, int which, int64_t* out) {
    if (L == nullptr || which < 0 || which >= CWN_LIFT_N_ARRAYS) return CWN_ERR_BAD_ARG;
    const Vec& v = L->out[which];
    if (!v.empty()) {
        if (out == nullptr) return CWN_ERR_BAD_ARG;
        std::memcpy(out, v.data(), v.size() * sizeof(int64_t));
    }
    return CWN_OK;
}

extern "C" void cwn_lift_destroy(cwn_lift_t* L) { delete L; }

// ---- a whole dataset at once ------------------------------------------------------------------------------
// The reference lifts a dataset graph by graph under joblib (data/utils.py:501-560 convert_graph_dataset_with_rings,
// n_jobs processes); a Python loop over cwn_lift_create costs ~115 us per ZINC-sized molecule, of which the
// lifting is ~15.  Here the graphs of a dataset are lifted by host threads and handed back CONCATENATED per array,
// in the layout the HBM-resident packed dataset wants (cwn_amd/packed.py): no per-complex objects in between.
struct cwn_lift_set_s {
    std::vector<cwn_lift_s*> lifts;
    ~cwn_lift_set_s() { for (auto* l : lifts) delete l; }
};

namespace {
inline bool two_rows(int which) {
    return which == CWN_LIFT_UP0 || which == CWN_LIFT_UP1 || which == CWN_LIFT_DOWN1 || which == CWN_LIFT_DOWN2 ||
           which == CWN_LIFT_BINDEX1 || which == CWN_LIFT_BINDEX2;
}
}  // namespace

extern "C" cwn_lift_set_t* cwn_lift_many(int kind, int64_t n_graphs, const int64_t* n_vertices, const int64_t* edge_ptr,
                                         const int64_t* edges, int max_k, int include_down, int n_threads) {
    if (n_graphs < 0 || (n_graphs > 0 && (n_vertices == nullptr || edge_ptr == nullptr))) return nullptr;
    if (kind != CWN_LIFT_RING && kind != CWN_LIFT_CLIQUE) return nullptr;
    for (int64_t g = 0; g < n_graphs; ++g)
        if (edge_ptr[g + 1] < edge_ptr[g] || edge_ptr[0] != 0) return nullptr;
    if (n_graphs > 0 && edge_ptr[n_graphs] > 0 && edges == nullptr) return nullptr;
    auto* S = new cwn_lift_set_s();
    S->lifts.assign((size_t)n_graphs, nullptr);
    std::atomic<int64_t> next{0};
    std::atomic<bool> bad{false};
    auto work = [&] {
        for (;;) {
            const int64_t g0 = next.fetch_add(64);          // chunks: molecules are tiny
            if (g0 >= n_graphs || bad.load()) return;
            for (int64_t g = g0; g < std::min(g0 + 64, n_graphs); ++g) {
                cwn_lift_s* l = cwn_lift_create(kind, n_vertices[g], edges + 2 * edge_ptr[g], edge_ptr[g + 1] - edge_ptr[g],
                                                max_k, include_down);
                if (l == nullptr) { bad.store(true); return; }
                S->lifts[(size_t)g] = l;
            }
        }
    };
    int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    T = (int)std::max<int64_t>(1, std::min<int64_t>(T, (n_graphs + 63) / 64));
    if (T == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    if (bad.load()) { delete S; return nullptr; }
    return S;
}

extern "C" int64_t cwn_lift_many_count(const cwn_lift_set_t* S) { return S ? (int64_t)S->lifts.size() : -1; }

// per graph: the columns L of a [2, L] array, the elements of any other
extern "C" int cwn_lift_many_lengths(const cwn_lift_set_t* S, int which, int64_t* out) {
    if (S == nullptr || which < 0 || which >= CWN_LIFT_N_ARRAYS || (out == nullptr && !S->lifts.empty())) return CWN_ERR_BAD_ARG;
    for (size_t g = 0; g < S->lifts.size(); ++g) {
        const int64_t n = (int64_t)S->lifts[g]->out[which].size();
        out[g] = two_rows(which) ? n / 2 : n;
    }
    return CWN_OK;
}

// every graph's array `which` in graph order; a [2, L] array comes out as ONE [2, sum L] array (row 0 of every graph,
// then row 1 of every graph): torch.cat(per-graph arrays, dim=-1).  Vertex / cell ids stay local to their graph.
extern "C" int cwn_lift_many_copy(const cwn_lift_set_t* S, int which, int64_t* out) {
    if (S == nullptr || which < 0 || which >= CWN_LIFT_N_ARRAYS) return CWN_ERR_BAD_ARG;
    int64_t total = 0;
    for (auto* l : S->lifts) total += (int64_t)l->out[which].size();
    if (total == 0) return CWN_OK;
    if (out == nullptr) return CWN_ERR_BAD_ARG;
    if (!two_rows(which)) {
        for (auto* l : S->lifts) {
            const Vec& v = l->out[which];
            if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(int64_t));
            out += v.size();
        }
        return CWN_OK;
    }
    int64_t* r0 = out;
    int64_t* r1 = out + total / 2;
    for (auto* l : S->lifts) {
        const Vec& v = l->out[which];
        const size_t L = v.size() / 2;
        if (L) {
            std::memcpy(r0, v.data(), L * sizeof(int64_t));
            std::memcpy(r1, v.data() + L, L * sizeof(int64_t));
        }
        r0 += L;
        r1 += L;
    }
    return CWN_OK;
}

extern "C" void cwn_lift_many_destroy(cwn_lift_set_t* S) { delete S; }
