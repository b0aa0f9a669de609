// cwn_csr.hip -- COO (int64, as delivered) -> destination-sorted int32 CSR, gfx950.
//
// One fixed launch sequence builds up to CWN_CSR_MAX_DESCS structures at once (all adjacencies of a
// batched complex), so the cost per batch is 5-6 launches regardless of how many index tensors
// there are (inputs that fit LDS take the ONE-launch path further down instead):
//   1. zero_words_kernel       zero the per-destination counters of every descriptor (a kernel, not a memset node: see there)
//   2. count_kernel            cnt[key[e]] += ... (one atomic per distinct key per wavefront;
//                              returned value = arrival slot inside the row)
//   3. scan (1 or 2 kernels)   rowptr = exclusive scan of cnt; rows longer than CWN_LONG_ROW
//                              are appended to the long-row list on the way
//   4. place_kernel            tmp[rowptr[key] + slot] = e, rows[..] = key  (row-grouped, unordered)
//   5. emit_kernel             rank every entry inside its row by ORIGINAL entry id (counting
//                              rank, O(sum deg^2) but broadcast reads, 8 in flight) and write
//                              perm / col / aux_out in stable order
// Integer work, latency- rather than bandwidth-bound at REDDIT-like sizes (a few MB): what
// matters is the number of DEPENDENT memory round trips (~0.9 us each) per kernel and keeping
// every wave access coalesced; see the notes at ScanChunk, count_kernel and emit_kernel.
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "../../include/cwn_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kScanThreads = 1024;
constexpr int kScanTile = 4096;        // counts per block in the multi-block scan
constexpr int64_t kSingleScanMax = 1 << 14;   // two trips of the single-block scan; above: tiled

struct CsrBatch {
    cwn_csr_desc d[CWN_CSR_MAX_DESCS];
    int32_t* cnt[CWN_CSR_MAX_DESCS];        // [n_dst] counters (workspace)
    int32_t* slot[CWN_CSR_MAX_DESCS];       // [E] arrival slot, later reused as tmp (row-grouped ids)
    int32_t* tmp[CWN_CSR_MAX_DESCS];        // [E]
    int32_t* rows[CWN_CSR_MAX_DESCS];       // [E] destination row of tmp[p] (saves emit a dependent load)
    int32_t* tile_sums[CWN_CSR_MAX_DESCS];  // [tiles] multi-block scan only
    int64_t blk_start[CWN_CSR_MAX_DESCS + 1];  // block prefix for the entry-parallel kernels
    int64_t tile_start[CWN_CSR_MAX_DESCS + 1]; // block prefix for the tile-parallel scan kernels
    int n;
    int dbg;   // timing experiments (CWN_CSR_DBG): 1 no rowptr stores, 2 no long-row notes, 4 no loads
};

// entries that exist: the host count, or -- a static buffer -- what the device says, never more than the capacity
__device__ __forceinline__ int64_t live_entries(const cwn_csr_desc& D) {
    if (D.e_dev == nullptr) return D.n_entries;
    const int64_t e = *D.e_dev;
    return e < 0 ? 0 : (e < D.n_entries ? e : D.n_entries);
}

__device__ __forceinline__ int find_desc(const int64_t* start, int n, int64_t b) {
    int d = 0;
#pragma unroll
    for (int i = 1; i < CWN_CSR_MAX_DESCS; ++i)
        if (i < n && b >= start[i]) d = i;
    return d;
}

// One atomic per DISTINCT key per wavefront: the lanes that hold the same key elect a leader
// (lowest lane), which adds the group's size; members take consecutive slots after the returned
// base.  A hub row (REDDIT-like: one key in half of the lanes of many consecutive waves) then
// costs 1/32 of the same-address atomics, which is what the kernel's time was going to.
__global__ __launch_bounds__(kThreads) void count_kernel(CsrBatch B, int32_t* err) {
    const int di = find_desc(B.blk_start, B.n, blockIdx.x);
    const cwn_csr_desc& D = B.d[di];
    const int64_t e = (int64_t)(blockIdx.x - B.blk_start[di]) * kThreads + threadIdx.x;
    const bool in = e < live_entries(D);
    int64_t k = -1;
    int bad = 0;
    if (in) {
        k = D.key[e];
        const int64_t v = D.val[e];
        if (k < 0 || k >= D.n_dst) bad |= 1;
        if (v < 0 || v >= D.n_val) bad |= 2;
        if (D.aux != nullptr) {
            const int64_t a = D.aux[e];
            if (a < 0 || a >= D.n_aux) bad |= 4;
        }
        // Both build paths treat bad indices alike: every one is REPORTED; an entry whose
        // destination is out of range cannot be placed and is dropped; a source / shared index out of
        // range is clamped when it is emitted, so a consumer that runs before the host has looked at the
        // error word (stream capture, overlap mode) can never fault.
        if (bad) atomicOr(err, bad);
        if (bad & 1) B.slot[di][e] = -1;
    }
    const int lane = threadIdx.x & 63;
    const int key32 = (in && !(bad & 1)) ? (int)k : -1;
    // 1. group the lanes by key (ALU only, <= 64 wave-uniform trips)
    unsigned long long todo = __ballot(key32 >= 0);
    int my_leader = lane, my_rank = 0, my_size = 1;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int cand = __shfl(key32, leader, 64);
        const unsigned long long same = __ballot(key32 == cand) & todo;
        if (key32 == cand) {
            my_leader = leader;
            my_rank = __popcll(same & ((1ull << lane) - 1ull));
            my_size = __popcll(same);
        }
        todo &= ~same;
    }
    // 2. every leader's atomic is in flight at once; 3. members read their leader's base
    int base = 0;
    if (key32 >= 0 && lane == my_leader) base = atomicAdd(&B.cnt[di][key32], my_size);
    base = __shfl(base, my_leader, 64);
    const int slot = base + my_rank;
    if (key32 >= 0) B.slot[di][e] = slot;
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
// same call); the single-launch path keeps one sub-list per workgroup instead.  Called by ALL
// lanes of a wave (`valid` masks the tail): one atomic per wave that holds any long row -- a
// returning same-address atomic per long row (each in its own divergent branch) cost ~0.3 us
// apiece and made this the longest kernel of the REDDIT-like build.
__device__ __forceinline__ void note_long_row(const cwn_csr_desc& D, int64_t row, int count, bool valid) {
    const bool is_long = valid && count > CWN_LONG_ROW && D.long_rows != nullptr && D.n_long != nullptr;
    const unsigned long long m = __ballot(is_long);
    if (m == 0) return;                               // wave-uniform
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(D.n_long, __popcll(m));
    base = __shfl(base, leader, 64);
    if (is_long) D.long_rows[base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)row;
}

__device__ __forceinline__ void zero_long_counters(const cwn_csr_desc& D) {
    if (D.n_long != nullptr && threadIdx.x < CWN_LONG_PARTS) D.n_long[threadIdx.x] = 0;
}

// Scan of one block-sized chunk, shared by the single-block and the tiled scan.
//  * a dependent global load costs ~0.9 us here (the counters were produced by device-scope
//    atomics, so they come from the memory side, not from this XCD's L2): all of a thread's ITEMS
//    loads are issued before the first use (clamped index, no guarding branch -- a branch per
//    load makes hipcc wait for each one);
//  * every global load / store instruction is a coalesced 256-B wave access (lane l of wave w
//    touches element w * 64 * ITEMS + u * 64 + l).  Giving each THREAD consecutive counters in
//    global memory instead makes every instruction touch 64 different lines, the 16 waves thrash
//    the CU's L1 and one CU re-fetches 32x the data: measured 45 us for a 43 k-row descriptor.
//  * the scan itself wants consecutive elements per thread (a serial prefix in registers, ONE
//    cross-lane scan per chunk; ITEMS chained ds_bpermute scans measured 22 us of pure ALU), so
//    each wave transposes its 64 * ITEMS elements through a private LDS slice, rows padded to
//    S words so that both access patterns are bank-conflict free.  No barrier: LDS operations
//    of one wave execute in order.
// Returns the chunk total; writes rowptr[i] = offset + exclusive prefix.
template <int ITEMS>
struct ScanChunk {
    static constexpr int S = ITEMS > 4 ? ITEMS + 4 : ITEMS;   // padded words per thread
    static constexpr int kLdsInts = kScanThreads * S;
    int v[ITEMS];

    __device__ __forceinline__ void load(const cwn_csr_desc& D, const int32_t* __restrict__ cnt,
                                         int64_t first, int dbg) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const int64_t n = D.n_dst;
        const int64_t w0 = first + (int64_t)wid * (64 * ITEMS) + lane;
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const int64_t i = w0 + u * 64;
            v[u] = (dbg & 4) ? 1 : cnt[i < n ? i : (n > 0 ? n - 1 : 0)];
        }
    }

    __device__ __forceinline__ int finish(const cwn_csr_desc& D, int64_t first, int offset, int dbg,
                                          int* lds, int* xpose) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const int64_t n = D.n_dst;
        const int64_t w0 = first + (int64_t)wid * (64 * ITEMS) + lane;
        int* my = xpose + wid * (64 * S);
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            v[u] = w0 + u * 64 < n ? v[u] : 0;
            const int e = u * 64 + lane;
            my[(e / ITEMS) * S + e % ITEMS] = v[u];
        }
        int c[ITEMS];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            c[k] = my[lane * S + k];
            mine += c[k];
        }
        int total;
        int p = offset + block_exclusive_scan(mine, &total, lds);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            my[lane * S + k] = p;
            p += c[k];
        }
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const int64_t i = w0 + u * 64;
            const int e = u * 64 + lane;
            const int pre = my[(e / ITEMS) * S + e % ITEMS];
            if (i < n && !(dbg & 1)) D.rowptr[i] = pre;
            if (!(dbg & 2)) note_long_row(D, i, v[u], i < n);
        }
        return total;
    }
};

constexpr int kScanItems = 8;   // single-block scan: 8 k rows per trip

__global__ __launch_bounds__(kScanThreads) void scan_single_kernel(CsrBatch B) {
    __shared__ int lds[32];
    __shared__ int xpose[ScanChunk<kScanItems>::kLdsInts];
    const int di = blockIdx.x;
    const cwn_csr_desc& D = B.d[di];
    const int64_t n = D.n_dst;
    constexpr int64_t kStep = (int64_t)kScanThreads * kScanItems;
    int running = 0;
    zero_long_counters(D);
    __syncthreads();
    ScanChunk<kScanItems> cur, nxt;
    if (n > 0) cur.load(D, B.cnt[di], 0, B.dbg);
    for (int64_t base = 0; base < n; base += kStep) {
        if (base + kStep < n) nxt.load(D, B.cnt[di], base + kStep, B.dbg);   // in flight during this trip
        running += cur.finish(D, base, running, B.dbg, lds, xpose);
        cur = nxt;
    }
    if (threadIdx.x == 0) D.rowptr[n] = running;
}

// ---- exclusive scan, multi block (tile sums -> scan of sums -> tile scan) ------------------
__global__ __launch_bounds__(kScanThreads) void tile_sum_kernel(CsrBatch B) {
    __shared__ int lds[32];
    const int di = find_desc(B.tile_start, B.n, blockIdx.x);
    const int64_t tile = blockIdx.x - B.tile_start[di];
    const int32_t* cnt = B.cnt[di];
    const int64_t n = B.d[di].n_dst;
    int v = 0;
    int w[kScanTile / kScanThreads];
#pragma unroll
    for (int k = 0; k < kScanTile / kScanThreads; ++k) {   // all loads in flight, then the selects
        const int64_t i = tile * kScanTile + k * kScanThreads + threadIdx.x;
        w[k] = cnt[i < n ? i : (n > 0 ? n - 1 : 0)];
    }
#pragma unroll
    for (int k = 0; k < kScanTile / kScanThreads; ++k) {
        const int64_t i = tile * kScanTile + k * kScanThreads + threadIdx.x;
        v += i < n ? w[k] : 0;
    }
    int total;
    block_exclusive_scan(v, &total, lds);
    if (threadIdx.x == 0) B.tile_sums[di][tile] = total;
    if (tile == 0) zero_long_counters(B.d[di]);   // tile_scan_kernel (next launch) appends the long rows
}

// Every tile adds up the sums of the tiles before it itself (<= a few hundred values, loaded
// while its own counters are in flight): no separate scan-of-sums launch between the two.
__global__ __launch_bounds__(kScanThreads) void tile_scan_kernel(CsrBatch B) {
    __shared__ int lds[32];
    using Chunk = ScanChunk<kScanTile / kScanThreads>;
    __shared__ int xpose[Chunk::kLdsInts];
    const int di = find_desc(B.tile_start, B.n, blockIdx.x);
    const int64_t tile = blockIdx.x - B.tile_start[di];
    const int64_t tiles = B.tile_start[di + 1] - B.tile_start[di];
    Chunk c;
    c.load(B.d[di], B.cnt[di], tile * kScanTile, B.dbg);
    int before = 0;
    for (int64_t t = threadIdx.x; t < tile; t += kScanThreads) before += B.tile_sums[di][t];
    int offset;
    block_exclusive_scan(before, &offset, lds);
    const int total = c.finish(B.d[di], tile * kScanTile, offset, B.dbg, lds, xpose);
    if (tile == tiles - 1 && threadIdx.x == 0) B.d[di].rowptr[B.d[di].n_dst] = offset + total;
}

__global__ __launch_bounds__(kThreads) void place_kernel(CsrBatch B) {
    const int di = find_desc(B.blk_start, B.n, blockIdx.x);
    const cwn_csr_desc& D = B.d[di];
    const int64_t e = (int64_t)(blockIdx.x - B.blk_start[di]) * kThreads + threadIdx.x;
    if (e >= live_entries(D)) return;
    const int s = B.slot[di][e];
    if (s < 0) return;
    const int64_t r = D.key[e];
    const int pos = D.rowptr[r] + s;
    B.tmp[di][pos] = (int32_t)e;
    B.rows[di][pos] = (int32_t)r;
}

__global__ __launch_bounds__(kThreads) void emit_kernel(CsrBatch B) {
    const int di = find_desc(B.blk_start, B.n, blockIdx.x);
    const cwn_csr_desc& D = B.d[di];
    const int64_t p = (int64_t)(blockIdx.x - B.blk_start[di]) * kThreads + threadIdx.x;
    // valid positions are [0, rowptr[n_dst]); dropped (out-of-range) entries shorten the tail
    if (p >= D.n_entries || p >= D.rowptr[D.n_dst]) return;
    // dependent hops (each ~0.9 us): {tmp, rows} -> {rowptr x 2, val, aux} -> rank loop -> stores
    const int32_t* tmp = B.tmp[di];
    const int32_t e = tmp[p];
    const int64_t r = B.rows[di][p];
    const int s = D.rowptr[r], t = D.rowptr[r + 1];
    const int64_t v64 = D.val[e], a64 = D.aux_out != nullptr ? D.aux[e] : 0;
    const int32_t my_val = (int32_t)(v64 < 0 ? 0 : (v64 >= D.n_val ? D.n_val - 1 : v64));      // reported in count_kernel
    const int32_t my_aux = (int32_t)(a64 < 0 ? 0 : (a64 >= D.n_aux ? D.n_aux - 1 : a64));
    // stable rank inside the row by ORIGINAL entry id.  Eight independent loads per trip: the
    // lanes of a row read the same addresses (broadcast), so a 300-entry hub row is ~40 round
    // trips to L2, not 300.
    int rank = 0;
    int q = s;
    for (; q + 8 <= t; q += 8) {
        int w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = tmp[q + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (w[u] < e) ? 1 : 0;
    }
    for (; q < t; ++q) rank += (tmp[q] < e) ? 1 : 0;
    const int P = s + rank;
    D.perm[P] = e;
    D.col[P] = my_val;
    if (D.aux_out != nullptr) D.aux_out[P] = my_aux;
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
    cwn_csr_desc d[CWN_CSR_MAX_DESCS];
    int32_t part_start[CWN_CSR_MAX_DESCS + 1];   // first workgroup of each descriptor
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
    for (int i = 1; i < CWN_CSR_MAX_DESCS; ++i)
        if (i < B.n && (int)blockIdx.x >= B.part_start[i]) di = i;
    const cwn_csr_desc& D = B.d[di];
    const int parts = B.part_start[di + 1] - B.part_start[di];
    const int part = blockIdx.x - B.part_start[di];
    const int n = (int)D.n_dst, Ecap = (int)D.n_entries, E = (int)live_entries(D);      // (LDS is laid out for the capacity)
    const int rows_per = (n + parts - 1) / parts;
    const int lo = min(part * rows_per, n), hi = min(lo + rows_per, n);
    const int rows = hi - lo;
    int32_t* cnt = lds;                     // [rows + 1] counters, then exclusive row starts
    int32_t* ent = cnt + (rows_per + 1);    // [E] original id of the part's j-th entry
    int32_t* slot = ent + Ecap;             // [E] arrival slot of that entry inside its row
    int32_t* byrow = slot + Ecap;           // [E] local entry ids grouped by row
    int32_t* lrow = byrow + Ecap;           // [E] local row (key - lo) of that entry
    int32_t* misc = lrow + Ecap;            // [0] entries of smaller rows, [1] entries of this part,
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
            k[u] = D.key[e < E ? e : (E > 0 ? E - 1 : 0)];          // clamped: unconditional loads
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
                D.long_rows[(int64_t)part * (Ecap / CWN_LONG_ROW + 1) + atomicAdd(&misc[2], 1)] = lo + i;
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

// the counters of a build, zeroed by a kernel: a memset node inside a captured graph did not reliably run again on replay
// (ROCm 7.2: the second replay of a graph holding this build counted on top of the first's counters and wrote past the arrays)
__global__ __launch_bounds__(256) void zero_words_kernel(uint4* __restrict__ p, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = z;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
    size_t cnt_off[CWN_CSR_MAX_DESCS], slot_off[CWN_CSR_MAX_DESCS], tmp_off[CWN_CSR_MAX_DESCS],
        rows_off[CWN_CSR_MAX_DESCS], tiles_off[CWN_CSR_MAX_DESCS];
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
        L.rows_off[i] = off;
        off += align_up((size_t)(d[i].n_entries + 1) * 4, 256);
        L.tiles_off[i] = off;
        off += align_up(((size_t)d[i].n_dst / kScanTile + 2) * 4, 256);
    }
    L.total = off;
    return L;
}

}  // namespace

extern "C" size_t cwn_csr_workspace_bytes(const cwn_csr_desc* descs, int n) {
    if (descs == nullptr || n <= 0 || n > CWN_CSR_MAX_DESCS) return 0;
    return layout(descs, n).total;
}

extern "C" int cwn_csr_build(const cwn_csr_desc* descs, int n, void* workspace, size_t ws_bytes,
                             int32_t* err_flag, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_CSR_MAX_DESCS || err_flag == nullptr) return CWN_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int64_t max_dst = 0;
    for (int i = 0; i < n; ++i) {
        const cwn_csr_desc& D = descs[i];
        if (D.n_entries < 0 || D.n_dst < 0 || D.rowptr == nullptr) return CWN_ERR_BAD_ARG;
        if (D.n_entries > 0 && (D.key == nullptr || D.val == nullptr || D.col == nullptr ||
                                D.perm == nullptr))
            return CWN_ERR_BAD_ARG;
        if (D.aux != nullptr && D.aux_out == nullptr) return CWN_ERR_BAD_ARG;
        // entries that name rows of an EMPTY source (or shared-cell) matrix cannot be clamped into it: refused here,
        // so that "reported and clamped: no consumer can fault" holds for every plan this call produces
        if (D.n_entries > 0 && (D.n_val <= 0 || (D.aux != nullptr && D.n_aux <= 0))) return CWN_ERR_BAD_ARG;
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
        static std::once_flag attr_once;        // thread-safe, once per process (the library keeps no other state)
        static hipError_t attr_err = hipSuccess;
        std::call_once(attr_once, [] {
            attr_err = hipFuncSetAttribute((const void*)csr_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kSmallLdsBytes);
        });
        if (attr_err != hipSuccess) return CWN_ERR_LAUNCH;
        SmallBatch S{};
        S.n = n;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            S.d[i] = descs[i];
            S.part_start[i] = blocks;
            blocks += small_parts(descs[i].n_entries, descs[i].n_dst);
        }
        for (int i = n; i <= CWN_CSR_MAX_DESCS; ++i) S.part_start[i] = blocks;
        csr_small_kernel<<<dim3(blocks), dim3(kSmallThreads), align_up(small_bytes, 16), stream>>>(S, err_flag);
        return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
    }
    const WsLayout L = layout(descs, n);
    if (workspace == nullptr || ws_bytes < L.total) return CWN_ERR_WORKSPACE;

    CsrBatch B{};
    B.n = n;
    static const int dbg = getenv("CWN_CSR_DBG") ? atoi(getenv("CWN_CSR_DBG")) : 0;
    B.dbg = dbg;
    char* ws = (char*)workspace;
    int64_t blocks = 0, tiles = 0;
    for (int i = 0; i < n; ++i) {
        B.d[i] = descs[i];
        B.cnt[i] = (int32_t*)(ws + L.cnt_off[i]);
        B.slot[i] = (int32_t*)(ws + L.slot_off[i]);
        B.tmp[i] = (int32_t*)(ws + L.tmp_off[i]);
        B.rows[i] = (int32_t*)(ws + L.rows_off[i]);
        B.tile_sums[i] = (int32_t*)(ws + L.tiles_off[i]);
        B.blk_start[i] = blocks;
        B.tile_start[i] = tiles;
        blocks += (descs[i].n_entries + kThreads - 1) / kThreads;
        tiles += descs[i].n_dst > 0 ? (descs[i].n_dst + kScanTile - 1) / kScanTile : 1;   // >= 1: rowptr[0]
    }
    for (int i = n; i <= CWN_CSR_MAX_DESCS; ++i) {
        B.blk_start[i] = blocks;
        B.tile_start[i] = tiles;
    }
    if (blocks >= INT32_MAX) return CWN_ERR_TOO_LARGE;

    if (((uintptr_t)ws & 15u) != 0) return CWN_ERR_ALIGN;
    {
        const int64_t n16 = (int64_t)(L.cnt_total / 16);                  // (cnt_total is a multiple of 256)
        int64_t zb = (n16 + 255) / 256;
        zb = zb < 1 ? 1 : (zb > 1024 ? 1024 : zb);
        zero_words_kernel<<<dim3((unsigned)zb), dim3(256), 0, stream>>>((uint4*)ws, n16);
    }
    if (blocks > 0) count_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B, err_flag);
    static const char* force = getenv("CWN_CSR_SCAN");   // timing experiments: "tiled" / "single"
    const bool single = force != nullptr ? force[0] == 's' : max_dst <= kSingleScanMax;
    if (single) {
        scan_single_kernel<<<dim3(n), dim3(kScanThreads), 0, stream>>>(B);
    } else {
        tile_sum_kernel<<<dim3((unsigned)tiles), dim3(kScanThreads), 0, stream>>>(B);
        tile_scan_kernel<<<dim3((unsigned)tiles), dim3(kScanThreads), 0, stream>>>(B);
    }
    if (blocks > 0) {
        place_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
        emit_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(B);
    }
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

// ---- the long-row lists of structures that were not built here (include/cwn_hip.h: cwn_csr_long_rows) -----------------------
namespace {
struct LongBatch { cwn_long_rows_desc d[CWN_CSR_MAX_DESCS]; };

__global__ __launch_bounds__(1024) void long_rows_kernel(LongBatch B) {
    __shared__ int count;
    const cwn_long_rows_desc D = B.d[blockIdx.x];          // (a small struct, a launch-constant index per workgroup)
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    int64_t n = D.n_rows;
    if (D.m_dev != nullptr) {
        const int64_t m = *D.m_dev;
        n = m < 0 ? 0 : (m < n ? m : n);
    }
    for (int64_t r = threadIdx.x; r < n; r += blockDim.x) {
        if (D.rowptr[r + 1] - D.rowptr[r] > CWN_LONG_ROW) {
            const int p = atomicAdd(&count, 1);
            if (p < D.long_cap) D.long_rows[p] = (int32_t)r;
        }
    }
    __syncthreads();
    if (threadIdx.x < CWN_LONG_PARTS)
        D.n_long[threadIdx.x] = threadIdx.x == 0 ? (int32_t)(count < D.long_cap ? count : D.long_cap) : 0;
}
}  // namespace

extern "C" int cwn_csr_long_rows(const cwn_long_rows_desc* descs, int n, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_CSR_MAX_DESCS) return CWN_ERR_BAD_ARG;
    LongBatch B{};
    for (int i = 0; i < n; ++i) {
        const cwn_long_rows_desc& D = descs[i];
        if (D.rowptr == nullptr || D.long_rows == nullptr || D.n_long == nullptr || D.n_rows < 0 || D.long_cap < 1) return CWN_ERR_BAD_ARG;
        if (D.n_rows >= INT32_MAX) return CWN_ERR_TOO_LARGE;
        B.d[i] = D;
    }
    long_rows_kernel<<<dim3((unsigned)n), dim3(1024), 0, (hipStream_t)stream_>>>(B);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
