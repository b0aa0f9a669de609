// cwn_csr.hip -- COO (int64, as delivered) -> destination-sorted int32 CSR, gfx950.
//
// One fixed launch sequence builds up to CWN_MAX_DESCS structures at once (all adjacencies of a
// batched complex), so the cost per batch is 5-7 launches regardless of how many index tensors
// there are:
//   1. hipMemsetAsync          zero the per-destination counters of every descriptor
//   2. count_kernel            cnt[key[e]]++ (returned value = arrival slot inside the row)
//   3. scan (1 or 3 kernels)   rowptr = exclusive scan of cnt
//   4. place_kernel            tmp[rowptr[key] + slot] = e           (row-grouped, unordered)
//   5. emit_kernel             rank every entry inside its row by ORIGINAL entry id (counting
//                              rank, O(sum deg^2) but coalesced/broadcast reads) and write
//                              perm / col / aux_out in stable order
// Integer work, HBM/L2-bound; no LDS tiling is needed (rows are short, reads are broadcast).
#include <hip/hip_runtime.h>
#include "../../include/cwn_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kScanThreads = 1024;
constexpr int kScanTile = 4096;        // counts per block in the multi-block scan
constexpr int64_t kSingleScanMax = 1 << 16;

struct CsrBatch {
    cwn_csr_desc d[CWN_MAX_DESCS];
    int32_t* cnt[CWN_MAX_DESCS];        // [n_dst] counters (workspace)
    int32_t* slot[CWN_MAX_DESCS];       // [E] arrival slot, later reused as tmp (row-grouped ids)
    int32_t* tmp[CWN_MAX_DESCS];        // [E]
    int32_t* tile_sums[CWN_MAX_DESCS];  // [tiles] multi-block scan only
    int64_t blk_start[CWN_MAX_DESCS + 1];  // block prefix for the entry-parallel kernels
    int64_t tile_start[CWN_MAX_DESCS + 1]; // block prefix for the tile-parallel scan kernels
    int n;
};

__device__ __forceinline__ int find_desc(const int64_t* start, int n, int64_t b) {
    int d = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_DESCS; ++i)
        if (i < n && b >= start[i]) d = i;
    return d;
}

__global__ __launch_bounds__(kThreads) void count_kernel(CsrBatch B, int32_t* err) {
    const int di = find_desc(B.blk_start, B.n, blockIdx.x);
    const cwn_csr_desc& D = B.d[di];
    const int64_t e = (int64_t)(blockIdx.x - B.blk_start[di]) * kThreads + threadIdx.x;
    if (e >= D.n_entries) return;
    const int64_t k = D.key[e];
    const int64_t v = D.val[e];
    int bad = 0;
    if (k < 0 || k >= D.n_dst) bad |= 1;
    if (v < 0 || v >= D.n_val) bad |= 2;
    if (D.aux != nullptr) {
        const int64_t a = D.aux[e];
        if (a < 0 || a >= D.n_aux) bad |= 4;
    }
    if (bad) {
        atomicOr(err, bad);
        B.slot[di][e] = -1;
        return;
    }
    B.slot[di][e] = atomicAdd(&B.cnt[di][k], 1);
}

// ---- exclusive scan, single block per descriptor (n_dst <= kSingleScanMax) -----------------
__device__ __forceinline__ int block_exclusive_scan(int v, int* total, int* lds /* >= 32 ints */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wid] = x;
    __syncthreads();
    const int nw = blockDim.x >> 6;
    if (wid == 0) {
        int s = lane < nw ? lds[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            int y = __shfl_up(s, o, 64);
            if (lane >= o) s += y;
        }
        if (lane < nw) lds[16 + lane] = s;  // inclusive wave sums
    }
    __syncthreads();
    const int wave_off = wid == 0 ? 0 : lds[16 + wid - 1];
    *total = lds[16 + nw - 1];
    __syncthreads();
    return wave_off + x - v;
}

// general path: every long row goes to sub-list 0 (counter zeroed by an earlier kernel of the
// same call); the single-launch path keeps one sub-list per workgroup instead
__device__ __forceinline__ void note_long_row(const cwn_csr_desc& D, int64_t row, int count) {
    if (count > CWN_LONG_ROW && D.long_rows != nullptr && D.n_long != nullptr)
        D.long_rows[atomicAdd(D.n_long, 1)] = (int32_t)row;
}

__device__ __forceinline__ void zero_long_counters(const cwn_csr_desc& D) {
    if (D.n_long != nullptr && threadIdx.x < CWN_LONG_PARTS) D.n_long[threadIdx.x] = 0;
}

__global__ __launch_bounds__(kScanThreads) void scan_single_kernel(CsrBatch B) {
    __shared__ int lds[32];
    const int di = blockIdx.x;
    const cwn_csr_desc& D = B.d[di];
    const int32_t* cnt = B.cnt[di];
    int32_t* rowptr = D.rowptr;
    const int64_t n = D.n_dst;
    int running = 0;
    zero_long_counters(D);
    __syncthreads();
    for (int64_t base = 0; base < n; base += kScanThreads) {
        const int64_t i = base + threadIdx.x;
        const int v = i < n ? cnt[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total, lds);
        if (i < n) {
            rowptr[i] = running + ex;
            note_long_row(D, i, v);
        }
        running += total;
    }
    if (threadIdx.x == 0) rowptr[n] = running;
}

// ---- exclusive scan, multi block (tile sums -> scan of sums -> tile scan) ------------------
__global__ __launch_bounds__(kScanThreads) void tile_sum_kernel(CsrBatch B) {
    __shared__ int lds[32];
    const int di = find_desc(B.tile_start, B.n, blockIdx.x);
    const int64_t tile = blockIdx.x - B.tile_start[di];
    const int32_t* cnt = B.cnt[di];
    const int64_t n = B.d[di].n_dst;
    int v = 0;
    for (int k = 0; k < kScanTile / kScanThreads; ++k) {
        const int64_t i = tile * kScanTile + k * kScanThreads + threadIdx.x;
        if (i < n) v += cnt[i];
    }
    int total;
    block_exclusive_scan(v, &total, lds);
    if (threadIdx.x == 0) B.tile_sums[di][tile] = total;
}

__global__ __launch_bounds__(kScanThreads) void scan_tile_sums_kernel(CsrBatch B) {
    __shared__ int lds[32];
    const int di = blockIdx.x;
    const int64_t tiles = B.tile_start[di + 1] - B.tile_start[di];
    int32_t* ts = B.tile_sums[di];
    int running = 0;
    zero_long_counters(B.d[di]);   // tile_scan_kernel (next launch) appends the long rows
    for (int64_t base = 0; base < tiles; base += kScanThreads) {
        const int64_t i = base + threadIdx.x;
        const int v = i < tiles ? ts[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total, lds);
        if (i < tiles) ts[i] = running + ex;
        running += total;
    }
    if (threadIdx.x == 0) B.d[di].rowptr[B.d[di].n_dst] = running;
}

__global__ __launch_bounds__(kScanThreads) void tile_scan_kernel(CsrBatch B) {
    __shared__ int lds[32];
    const int di = find_desc(B.tile_start, B.n, blockIdx.x);
    const int64_t tile = blockIdx.x - B.tile_start[di];
    const int32_t* cnt = B.cnt[di];
    int32_t* rowptr = B.d[di].rowptr;
    const int64_t n = B.d[di].n_dst;
    int running = B.tile_sums[di][tile];
    for (int k = 0; k < kScanTile / kScanThreads; ++k) {
        const int64_t i = tile * kScanTile + k * kScanThreads + threadIdx.x;
        const int v = i < n ? cnt[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total, lds);
        if (i < n) {
            rowptr[i] = running + ex;
            note_long_row(B.d[di], i, v);
        }
        running += total;
    }
}

__global__ __launch_bounds__(kThreads) void place_kernel(CsrBatch B) {
    const int di = find_desc(B.blk_start, B.n, blockIdx.x);
    const cwn_csr_desc& D = B.d[di];
    const int64_t e = (int64_t)(blockIdx.x - B.blk_start[di]) * kThreads + threadIdx.x;
    if (e >= D.n_entries) return;
    const int s = B.slot[di][e];
    if (s < 0) return;
    B.tmp[di][D.rowptr[D.key[e]] + s] = (int32_t)e;
}

__global__ __launch_bounds__(kThreads) void emit_kernel(CsrBatch B) {
    const int di = find_desc(B.blk_start, B.n, blockIdx.x);
    const cwn_csr_desc& D = B.d[di];
    const int64_t p = (int64_t)(blockIdx.x - B.blk_start[di]) * kThreads + threadIdx.x;
    // valid positions are [0, rowptr[n_dst]); dropped (out-of-range) entries shorten the tail
    if (p >= D.n_entries || p >= D.rowptr[D.n_dst]) return;
    const int32_t* tmp = B.tmp[di];
    const int32_t e = tmp[p];
    const int64_t r = D.key[e];
    const int s = D.rowptr[r], t = D.rowptr[r + 1];
    int rank = 0;
    for (int q = s; q < t; ++q) rank += (tmp[q] < e) ? 1 : 0;
    const int P = s + rank;
    D.perm[P] = e;
    D.col[P] = (int32_t)D.val[e];
    if (D.aux_out != nullptr) D.aux_out[P] = (int32_t)D.aux[e];
}

// ---- small path: ONE launch, everything in LDS ---------------------------------------------------
// Used when the per-workgroup LDS need (see small_lds_ints) fits 150 KiB for every descriptor of
// the call (a ZINC batch of 128 has E <= 8.2e3, n_dst <= 3.4e3 per adjacency).  At this size the
// general path is six dependent launches; here the phases are separated by workgroup barriers.
// One workgroup can only pull ~50 GB/s, and an adjacency is ~0.4 MB of index traffic, so each
// descriptor is split into up to kMaxParts ROW RANGES handled by independent workgroups: a part
// scans every key (to count the entries of smaller rows -- its base offset -- and to pick its own
// entries) but touches val / aux and the outputs only for its own rows.  No inter-workgroup
// communication is needed.
constexpr int kSmallThreads = 1024;
constexpr size_t kSmallLdsBytes = 150 * 1024;
constexpr int kMaxParts = CWN_LONG_PARTS;

struct SmallBatch {
    cwn_csr_desc d[CWN_MAX_DESCS];
    int32_t part_start[CWN_MAX_DESCS + 1];   // first workgroup of each descriptor
    int32_t n;
};

inline int small_parts(int64_t n_entries, int64_t n_dst) {
    int64_t p = (n_entries + 1535) / 1536;
    if (p > kMaxParts) p = kMaxParts;
    if (p > n_dst) p = n_dst;
    return p < 1 ? 1 : (int)p;
}

inline size_t small_lds_ints(int64_t n_entries, int64_t n_dst) {
    const int parts = small_parts(n_entries, n_dst);
    const int64_t rows = (n_dst + parts - 1) / parts;
    return (size_t)(rows + 1 + 4 * n_entries + 8);       // cnt | ent | slot | byrow | lrow | misc
}

__global__ __launch_bounds__(kSmallThreads) void csr_small_kernel(SmallBatch B, int32_t* err) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    __shared__ int scan_tmp[32];
    constexpr int U = 8;   // independent global loads in flight per thread and phase
    int di = 0;
#pragma unroll
    for (int i = 1; i < CWN_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.part_start[i]) di = i;
    const cwn_csr_desc& D = B.d[di];
    const int parts = B.part_start[di + 1] - B.part_start[di];
    const int part = blockIdx.x - B.part_start[di];
    const int n = (int)D.n_dst, E = (int)D.n_entries;
    const int rows_per = (n + parts - 1) / parts;
    const int lo = min(part * rows_per, n), hi = min(lo + rows_per, n);
    const int rows = hi - lo;
    int32_t* cnt = lds;                     // [rows + 1] counters, then exclusive row starts
    int32_t* ent = cnt + (rows_per + 1);    // [E] original id of the part's j-th entry
    int32_t* slot = ent + E;                // [E] arrival slot of that entry inside its row
    int32_t* byrow = slot + E;              // [E] local entry ids grouped by row
    int32_t* lrow = byrow + E;              // [E] local row (key - lo) of that entry
    int32_t* misc = lrow + E;               // [0] entries of smaller rows, [1] entries of this part,
                                            // [2] long rows of this part
    const bool has_aux = D.aux != nullptr;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i <= rows; i += kSmallThreads) cnt[i] = 0;
    if (threadIdx.x < 3) misc[threadIdx.x] = 0;
    __syncthreads();
    // phase 1: every key once (U independent loads per thread); wave-aggregated bookkeeping
    int below_local = 0;
    for (int base = 0; base < E; base += kSmallThreads * U) {
        int64_t k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * kSmallThreads + threadIdx.x;
            k[u] = D.key[e < E ? e : E - 1];          // clamped: unconditional loads
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * kSmallThreads + threadIdx.x;
            const bool live = e < E;
            const bool bad = live && (k[u] < 0 || k[u] >= D.n_dst);
            if (bad && part == 0) atomicOr(err, 1);
            below_local += (live && !bad && k[u] < lo) ? 1 : 0;
            const bool mine = live && !bad && k[u] >= lo && k[u] < hi;
            const unsigned long long m = __ballot(mine);
            int wbase = 0;
            if (lane == 0 && m) wbase = atomicAdd(&misc[1], __popcll(m));
            wbase = __shfl(wbase, 0, 64);
            if (mine) {
                const int li = wbase + __popcll(m & ((1ull << lane) - 1ull));
                ent[li] = e;
                lrow[li] = (int)k[u] - lo;
                slot[li] = atomicAdd(&cnt[(int)k[u] - lo], 1);
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) below_local += __shfl_xor(below_local, o, 64);
    if (lane == 0 && below_local) atomicAdd(&misc[0], below_local);
    __syncthreads();
    const int gbase = misc[0], mine_total = misc[1];
    // phase 2: exclusive scan of the part's counters in place; global row pointers
    int running = 0;
    for (int base = 0; base < rows; base += kSmallThreads) {
        const int i = base + threadIdx.x;
        const int c = i < rows ? cnt[i] : 0;
        int total;
        const int ex = block_exclusive_scan(c, &total, scan_tmp);
        if (i < rows) {
            cnt[i] = running + ex;
            D.rowptr[lo + i] = gbase + running + ex;
            if (c > CWN_LONG_ROW && D.long_rows != nullptr)   // this part's own sub-list
                D.long_rows[(int64_t)part * (E / CWN_LONG_ROW + 1) + atomicAdd(&misc[2], 1)] = lo + i;
        }
        running += total;
    }
    if (threadIdx.x == 0) {
        cnt[rows] = running;
        if (part == parts - 1) D.rowptr[n] = gbase + running;
    }
    __syncthreads();
    if (D.n_long != nullptr) {
        if (threadIdx.x == 0) D.n_long[part] = misc[2];
        if (part == 0 && threadIdx.x >= parts && threadIdx.x < CWN_LONG_PARTS) D.n_long[threadIdx.x] = 0;
    }
    // phase 3: group the part's entries by row (LDS only)
    for (int li = threadIdx.x; li < mine_total; li += kSmallThreads) byrow[cnt[lrow[li]] + slot[li]] = li;
    __syncthreads();
    // phase 4: stable rank inside the row by ORIGINAL entry id (LDS only), then the gathered outputs
    for (int base = 0; base < mine_total; base += kSmallThreads * U) {
        int e[U], P[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int li = base + u * kSmallThreads + threadIdx.x;
            e[u] = -1;
            P[u] = 0;
            if (li < mine_total) {
                const int ee = ent[li];
                const int r = lrow[li];
                const int s = cnt[r], t = cnt[r + 1];
                int rank = 0;
                for (int q = s; q < t; ++q) rank += (ent[byrow[q]] < ee) ? 1 : 0;
                e[u] = ee;
                P[u] = gbase + s + rank;
            }
        }
        int64_t v[U], a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ec = e[u] >= 0 ? e[u] : 0;
            v[u] = D.val[ec];
            a[u] = has_aux ? D.aux[ec] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e[u] < 0) continue;
            int bad = 0;
            if (v[u] < 0 || v[u] >= D.n_val) bad |= 2;
            if (has_aux && (a[u] < 0 || a[u] >= D.n_aux)) bad |= 4;
            if (bad) atomicOr(err, bad);
            // out-of-range values are reported AND clamped, so that a consumer that runs before the
            // host has looked at the error word (stream capture, overlap mode) can never fault
            const int64_t vc = v[u] < 0 ? 0 : (v[u] >= D.n_val ? D.n_val - 1 : v[u]);
            const int64_t ac = a[u] < 0 ? 0 : (a[u] >= D.n_aux ? D.n_aux - 1 : a[u]);
            D.perm[P[u]] = e[u];
            D.col[P[u]] = (int32_t)vc;
            if (D.aux_out != nullptr) D.aux_out[P[u]] = (int32_t)ac;
        }
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
    size_t cnt_off[CWN_MAX_DESCS], slot_off[CWN_MAX_DESCS], tmp_off[CWN_MAX_DESCS],
        tiles_off[CWN_MAX_DESCS];
    size_t cnt_total;  // the leading region that must be zeroed
    size_t total;
};

WsLayout layout(const cwn_csr_desc* d, int n) {
    WsLayout L{};
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        L.cnt_off[i] = off;
        off += align_up((size_t)(d[i].n_dst + 1) * 4, 256);
    }
    L.cnt_total = off;
    for (int i = 0; i < n; ++i) {
        L.slot_off[i] = off;
        off += align_up((size_t)(d[i].n_entries + 1) * 4, 256);
        L.tmp_off[i] = off;
        off += align_up((size_t)(d[i].n_entries + 1) * 4, 256);
        L.tiles_off[i] = off;
        off += align_up(((size_t)d[i].n_dst / kScanTile + 2) * 4, 256);
    }
    L.total = off;
    return L;
}

}  // namespace

extern "C" size_t cwn_csr_workspace_bytes(const cwn_csr_desc* descs, int n) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_DESCS) return 0;
    return layout(descs, n).total;
}

extern "C" int cwn_csr_build(const cwn_csr_desc* descs, int n, void* workspace, size_t ws_bytes,
                             int32_t* err_flag, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_DESCS || err_flag == nullptr) return CWN_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int64_t max_dst = 0;
    for (int i = 0; i < n; ++i) {
        const cwn_csr_desc& D = descs[i];
        if (D.n_entries < 0 || D.n_dst < 0 || D.rowptr == nullptr) return CWN_ERR_BAD_ARG;
        if (D.n_entries > 0 && (D.key == nullptr || D.val == nullptr || D.col == nullptr ||
                                D.perm == nullptr))
            return CWN_ERR_BAD_ARG;
        if (D.aux != nullptr && D.aux_out == nullptr) return CWN_ERR_BAD_ARG;
        if (D.n_entries >= INT32_MAX || D.n_dst >= INT32_MAX || D.n_val >= INT32_MAX ||
            D.n_aux >= INT32_MAX)
            return CWN_ERR_TOO_LARGE;
        if (D.n_dst > max_dst) max_dst = D.n_dst;
    }
    // small path: no workspace, one launch
    size_t small_bytes = 0;
    for (int i = 0; i < n; ++i) {
        const size_t need = small_lds_ints(descs[i].n_entries, descs[i].n_dst) * 4;
        if (need > small_bytes) small_bytes = need;
    }
    if (small_bytes <= kSmallLdsBytes) {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)csr_small_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kSmallLdsBytes) != hipSuccess)
                return CWN_ERR_LAUNCH;
            attr_set = true;
        }
        SmallBatch S{};
        S.n = n;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            S.d[i] = descs[i];
            S.part_start[i] = blocks;
            blocks += small_parts(descs[i].n_entries, descs[i].n_dst);
        }
        for (int i = n; i <= CWN_MAX_DESCS; ++i) S.part_start[i] = blocks;
        csr_small_kernel<<<dim3(blocks), dim3(kSmallThreads), align_up(small_bytes, 16), stream>>>(S, err_flag);
        return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
    }
    const WsLayout L = layout(descs, n);
    if (workspace == nullptr || ws_bytes < L.total) return CWN_ERR_WORKSPACE;

    CsrBatch B{};
    B.n = n;
    char* ws = (char*)workspace;
    int64_t blocks = 0, tiles = 0;
    for (int i = 0; i < n; ++i) {
        B.d[i] = descs[i];
        B.cnt[i] = (int32_t*)(ws + L.cnt_off[i]);
        B.slot[i] = (int32_t*)(ws + L.slot_off[i]);
        B.tmp[i] = (int32_t*)(ws + L.tmp_off[i]);
        B.tile_sums[i] = (int32_t*)(ws + L.tiles_off[i]);
        B.blk_start[i] = blocks;
        B.tile_start[i] = tiles;
        blocks += (descs[i].n_entries + kThreads - 1) / kThreads;
        tiles += (descs[i].n_dst + kScanTile - 1) / kScanTile;
    }
    for (int i = n; i <= CWN_MAX_DESCS; ++i) {
        B.blk_start[i] = blocks;
        B.tile_start[i] = tiles;
    }
    if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;

    if (hipMemsetAsync(ws, 0, L.cnt_total, stream) != hipSuccess) return CWN_ERR_LAUNCH;
    if (blocks > 0) count_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B, err_flag);
    if (max_dst <= kSingleScanMax) {
        scan_single_kernel<<<dim3(n), dim3(kScanThreads), 0, stream>>>(B);
    } else {
        tile_sum_kernel<<<dim3((unsigned)tiles), dim3(kScanThreads), 0, stream>>>(B);
        scan_tile_sums_kernel<<<dim3(n), dim3(kScanThreads), 0, stream>>>(B);
        tile_scan_kernel<<<dim3((unsigned)tiles), dim3(kScanThreads), 0, stream>>>(B);
    }
    if (blocks > 0) {
        place_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
        emit_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
    }
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
