// cwn_collate.hip -- device-side batching of complexes resident in HBM.
//
// One launch fills every array of a ComplexBatch: workgroup (descriptor d, segment s) copies the
// s-th selected complex's slice of array d to its place in the batched array, adding that
// complex's running cell offset to index values.  Byte / integer work, HBM-bound; segments are
// tens to hundreds of elements, so a WAVE per (array, complex) with coalesced element-wise accesses is the
// natural grain -- four to a workgroup: a static batch of 8 slots x 512 complexes x ~30 arrays was 123 k
// one-wave workgroups, 38 us of workgroup launches.
#include <hip/hip_runtime.h>
#include "../../include/cwn_hip.h"

namespace {

constexpr int kThreads = 256;        // four segments per workgroup, a wave each
constexpr int kSegPerWg = kThreads / 64;

struct CollateBatch {
    cwn_collate_desc d[CWN_MAX_COLLATE_DESCS];
    int64_t dst_slot_bytes[CWN_MAX_COLLATE_DESCS];   // cwn_collate_slots: bytes between the output arrays of consecutive slots
    int64_t table_slot_stride;                       // ... and int64 elements between their tables
    int64_t* cursor;                                 // or NULL: advanced by the number of slots (one thread of the launch)
    int32_t n;
};

// One segment, `step` lanes on it (lane `l` of them): four elements requested before the first is stored (dst and src never
// overlap: a batched array and the packed dataset).
template <typename T>
__device__ __forceinline__ void copy_seg(T* __restrict__ dst, const T* __restrict__ src, int64_t len, T a, int l, int step) {
    for (int64_t q = l; q < len; q += 4 * step) {
        T v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = q + u * step < len ? src[q + u * step] : T(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (q + u * step < len) dst[q + u * step] = v[u] + a;
    }
}
template <typename T>
__device__ __forceinline__ void fill_seg(T* __restrict__ dst, int64_t len, T v, int l, int step) {
    for (int64_t q = l; q < len; q += step) dst[q] = v;
}

// SPW segments per wave.  1: a wave per (array, complex).  4 (batches of many small complexes: a molecule's arrays are tens
// of elements, a 64-lane pass leaves most lanes idle and the launch is ~10^5 waves of three dependent round trips each):
// every 16 lanes read the table entries of their own segment in ONE round trip, then either copy it themselves (all four
// short) or the whole wave takes the four in turn (any of them long), the entries passed by lane shuffles.
template <int SPW>
__global__ __launch_bounds__(kThreads) void collate_kernel(CollateBatch B, int64_t n_seg) {
    constexpr int GL = 64 / SPW;                     // lanes per segment
    const int di = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int sub = lane / GL, sl = lane % GL;
    const int64_t s = ((int64_t)blockIdx.x * kSegPerWg + (threadIdx.x >> 6)) * SPW + sub;
    const int64_t slot = blockIdx.z;
    if (B.cursor != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        *B.cursor += (int64_t)gridDim.z;             // (the tables of this launch were cut by an EARLIER launch: no reader left)
    if (s - sub >= n_seg) return;                    // (a whole wave)
    const cwn_collate_desc& D = B.d[di];
    const int64_t tab_off = slot * B.table_slot_stride;
    const bool live = s < n_seg;
    const int64_t d0 = live ? D.dst_start[tab_off + s] : 0;
    int64_t len = live ? D.dst_start[tab_off + s + 1] - d0 : 0;
    if (len < 0) len = 0;
    const int64_t s0 = (D.op == CWN_COLLATE_SEGID64 || !live) ? 0 : D.src_start[tab_off + s];
    int64_t add[2] = {0, 0};
    if (D.add != nullptr && live) {
        add[0] = D.add[tab_off + s];
        if (D.n_rows > 1) add[1] = D.add[tab_off + n_seg + s];
    }
    char* const dst_base = (char*)D.dst + slot * B.dst_slot_bytes[di];
    bool own = true;                                 // every group copies its own segment
    if (SPW > 1) own = __all(len <= 8 * GL);
    for (int g = 0; g < (own ? 1 : SPW); ++g) {
        // (own: this lane's group and entries; otherwise the wave on the entries of group g)
        const int64_t d0_ = own ? d0 : __shfl(d0, g * GL, 64), len_ = own ? len : __shfl(len, g * GL, 64);
        const int64_t s0_ = own ? s0 : __shfl(s0, g * GL, 64), seg_ = own ? s : __shfl(s, g * GL, 64);
        const int l = own ? sl : lane, step = own ? GL : 64;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r >= D.n_rows) break;
            const int64_t a_ = own ? add[r] : __shfl(add[r], g * GL, 64);
            if (len_ <= 0) continue;
            if (D.op == CWN_COLLATE_COPY32 || D.op == CWN_COLLATE_ADD32) {
                copy_seg<int32_t>((int32_t*)dst_base + r * D.dst_row_stride + d0_, (const int32_t*)D.src + r * D.src_row_stride + s0_,
                                  len_, D.op == CWN_COLLATE_ADD32 ? (int32_t)a_ : 0, l, step);
            } else if (D.op == CWN_COLLATE_SEGID64) {
                fill_seg<int64_t>((int64_t*)dst_base + r * D.dst_row_stride + d0_, len_, seg_, l, step);
            } else {
                copy_seg<int64_t>((int64_t*)dst_base + r * D.dst_row_stride + d0_, (const int64_t*)D.src + r * D.src_row_stride + s0_,
                                  len_, D.op == CWN_COLLATE_ADD64 ? a_ : 0, l, step);
            }
        }
    }
}

// ---- the segment tables of a batch, on the device (include/cwn_hip.h: cwn_collate_tables) ------------------------------------
// One workgroup of 16 waves; wave w takes the columns w, w + 16, ... of the 3 D + K scanned ones (cells / below / above of
// every dimension, the length of every key): per column an inclusive scan over the B complexes of the batch, 64 at a time
// (a gather of the complexes' metadata rows, a wave scan, a running carry), written out in the forms the collate launch
// and the item-table launches read.  Integer work on a few kilobytes: its cost is two dependent memory round trips per 64
// complexes and column, all columns in parallel.
constexpr int kTabThreads = 1024;

__global__ __launch_bounds__(kTabThreads) void collate_tables_kernel(const int64_t* __restrict__ meta, int64_t num, int D, int K,
                                                                     const int64_t* __restrict__ idx_all, int64_t B, int64_t n_batches,
                                                                     const int64_t* __restrict__ cursor, int64_t slot_stride,
                                                                     int64_t* __restrict__ tab_all, int32_t* __restrict__ err) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = 3 * D + 3 * K, ncol = 3 * D + K;
    // workgroup j of the launch cuts the tables of slot j = batch (*cursor + j) of the buffer; past the buffer: an empty batch
    // (and the sticky bit: a replay too many)
    const int64_t cur = (cursor != nullptr ? *cursor : 0) + (int64_t)blockIdx.x;
    const bool past = cur < 0 || cur >= n_batches;
    const int64_t* idx = idx_all + (past ? 0 : cur) * B;
    int64_t* const tab = tab_all + (int64_t)blockIdx.x * slot_stride;
    if (past) num = 0;                                  // every entry counts as absent ...
    const int64_t o_src = (int64_t)K * (B + 1), o_off = o_src + (int64_t)K * B, o_seg = o_off + (int64_t)D * 5 * B;
    const int64_t o_sizes = o_seg + (int64_t)D * (B + 1);
    bool bad = past;                                    // ... and is reported
    // columns: wave w of workgroup (slot, y) takes w + 16 y, + 16 gridDim.y, ...; the chunks of a column (64 complexes each) are
    // REQUESTED kTabChunks at a time -- index, then both metadata words -- before the first scan: two dependent round trips per
    // 512 complexes instead of two per 64
    constexpr int kTabChunks = 8;
    for (int col = wave + (kTabThreads / 64) * (int)blockIdx.y; col < ncol; col += (kTabThreads / 64) * (int)gridDim.y) {
        int64_t carry = 0;
        const bool is_key = col >= 3 * D;
        const int k = col - 3 * D, d = col / 3, which = col % 3;
        for (int64_t sb0 = 0; sb0 < B; sb0 += 64 * kTabChunks) {
        int64_t cc[kTabChunks], vv[kTabChunks], st[kTabChunks];
#pragma unroll
        for (int u = 0; u < kTabChunks; ++u) {
            const int64_t s = sb0 + 64 * u + lane;
            cc[u] = s < B ? idx[s] : -1;
        }
#pragma unroll
        for (int u = 0; u < kTabChunks; ++u) {
            if (cc[u] >= num) { bad = bad || num > 0; cc[u] = -1; }
            vv[u] = cc[u] >= 0 ? meta[cc[u] * W + col] : 0;
            st[u] = (is_key && cc[u] >= 0) ? meta[cc[u] * W + col + K] : 0;
        }
#pragma unroll
        for (int u = 0; u < kTabChunks; ++u) {
            const int64_t s0 = sb0 + 64 * u;
            if (s0 >= B) break;
            const int64_t s = s0 + lane;
            const int64_t v = vv[u], start = st[u];
            int64_t x = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int64_t y = __shfl_up(x, o, 64);
                if (lane >= o) x += y;
            }
            const int64_t incl = carry + x, excl = incl - v;
            if (s < B) {
                if (is_key) {
                    tab[(int64_t)k * (B + 1) + s + 1] = incl;
                    tab[o_src + (int64_t)k * B + s] = start;
                } else {
                    int64_t* off = tab + o_off + (int64_t)d * 5 * B;
                    if (which == 0) {                   // (here, here, below, here, above)
                        off[0 * B + s] = excl;
                        off[1 * B + s] = excl;
                        off[3 * B + s] = excl;
                        tab[o_seg + (int64_t)d * (B + 1) + s] = excl;
                    } else if (which == 1) {
                        off[2 * B + s] = excl;
                    } else {
                        off[4 * B + s] = excl;
                    }
                }
            }
            carry += __shfl(x, 63, 64);
        }
        }
        if (lane == 0) {
            if (is_key) {
                tab[(int64_t)k * (B + 1)] = 0;
                tab[o_sizes + 8 + k] = carry;
            } else if (which == 0) {
                tab[o_seg + (int64_t)d * (B + 1) + B] = carry;
                if (d < 3) tab[o_sizes + d] = carry;
            }
        }
    }
    if (wave == 0 && blockIdx.y == 0) {                 // complexes in the batch; the unused size slots
        int64_t n = 0;
        for (int64_t s0 = 0; s0 < B; s0 += 64) {
            const int64_t s = s0 + lane;
            const int64_t c = s < B ? idx[s] : -1;
            n += __popcll(__ballot(c >= 0 && c < num));
        }
        if (lane == 0) tab[o_sizes + 3] = n;
        if (lane >= 4 && lane < 8) tab[o_sizes + lane] = 0;
        if (lane < 3 && lane >= D) tab[o_sizes + lane] = 0;
    }
    if (bad) atomicOr(err, 2);
}

// ---- a batch beyond the capacity of its buffers becomes an EMPTY batch (include/cwn_hip.h: cwn_collate_guard) ---------------------
// One workgroup per slot: the totals cwn_collate_tables scanned (cells per dimension = seg[d][B], length per key = sizes[8 + k])
// against the capacities of the arrays the collate launch is about to write.  Over any of them: the slot's whole table is zeroed
// (= the tables of a batch without complexes: every segment empty, every row count 0 -- its step changes nothing) and the sticky
// bit is set.  Without this a batch heavier than the statistical capacity of a static batch wrote past its buffers inside a
// replayed graph.
__global__ __launch_bounds__(256) void collate_guard_kernel(int64_t* __restrict__ tab_all, int64_t slot_stride, int D, int K, int64_t B,
                                                            int64_t n_tab, const int64_t* __restrict__ caps, int32_t* __restrict__ err) {
    int64_t* const tab = tab_all + (int64_t)blockIdx.x * slot_stride;
    const int64_t o_src = (int64_t)K * (B + 1), o_off = o_src + (int64_t)K * B, o_seg = o_off + (int64_t)D * 5 * B;
    const int64_t o_sizes = o_seg + (int64_t)D * (B + 1);
    int over = 0;
    for (int c = threadIdx.x; c < D + K; c += blockDim.x) {
        const int64_t total = c < D ? tab[o_seg + (int64_t)c * (B + 1) + B] : tab[o_sizes + 8 + (c - D)];
        over |= total > caps[c];
    }
    if (!__syncthreads_or(over)) return;
    for (int64_t q = threadIdx.x; q < n_tab; q += blockDim.x) tab[q] = 0;
    if (threadIdx.x == 0) atomicOr(err, CWN_ERR_BIT_CAPACITY);
}

}  // namespace

extern "C" int cwn_collate_guard(int64_t* tables, int32_t D, int32_t K, int64_t B, int32_t n_slots, int64_t slot_stride,
                                 const int64_t* caps, int32_t* err_flag, cwn_stream_t stream_) {
    if (tables == nullptr || caps == nullptr || err_flag == nullptr || D < 1 || D > 8 || K < 0 || B < 1 || n_slots < 1 || n_slots > 1024)
        return CWN_ERR_BAD_ARG;
    const int64_t n_tab = (int64_t)cwn_collate_tables_len(D, K, B);
    if (n_slots > 1 && slot_stride < n_tab) return CWN_ERR_BAD_ARG;
    collate_guard_kernel<<<dim3(n_slots), dim3(256), 0, (hipStream_t)stream_>>>(tables, slot_stride, D, K, B, n_tab, caps, err_flag);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" size_t cwn_collate_tables_len(int32_t D, int32_t K, int64_t B) {
    if (D < 1 || K < 0 || B < 1) return 0;
    return (size_t)((int64_t)K * (B + 1) + (int64_t)K * B + (int64_t)D * 5 * B + (int64_t)D * (B + 1) + 8 + K);
}

extern "C" int cwn_collate_tables(const int64_t* meta, int64_t num, int32_t D, int32_t K, const int64_t* idx, int64_t B,
                                  int64_t n_batches, const int64_t* cursor, int32_t n_slots, int64_t slot_stride, int64_t* tables,
                                  int32_t* err_flag, cwn_stream_t stream_) {
    if (meta == nullptr || idx == nullptr || tables == nullptr || err_flag == nullptr || num < 0 || D < 1 || D > 8 || K < 0 || B < 1 ||
        n_batches < 1 || n_slots < 1 || n_slots > 1024)
        return CWN_ERR_BAD_ARG;
    if (n_slots > 1 && slot_stride < (int64_t)cwn_collate_tables_len(D, K, B)) return CWN_ERR_BAD_ARG;
    if (B >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    const int ncol = 3 * D + K, per = kTabThreads / 64;
    collate_tables_kernel<<<dim3(n_slots, (unsigned)((ncol + per - 1) / per)), dim3(kTabThreads), 0, (hipStream_t)stream_>>>(
        meta, num, D, K, idx, B, n_batches, cursor, slot_stride, tables, err_flag);
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}

extern "C" int cwn_collate(const cwn_collate_desc* descs, int n, int64_t n_seg, cwn_stream_t stream_) {
    return cwn_collate_slots(descs, n, n_seg, 1, 0, nullptr, nullptr, stream_);
}

extern "C" int cwn_collate_slots(const cwn_collate_desc* descs, int n, int64_t n_seg, int32_t n_slots, int64_t table_slot_stride,
                                 const int64_t* dst_slot_bytes, int64_t* cursor, cwn_stream_t stream_) {
    if (descs == nullptr || n <= 0 || n > CWN_MAX_COLLATE_DESCS || n_seg < 0 || n_slots < 1 || n_slots > 1024) return CWN_ERR_BAD_ARG;
    if (n_slots > 1 && (dst_slot_bytes == nullptr || table_slot_stride <= 0)) return CWN_ERR_BAD_ARG;
    if (n_seg == 0) return CWN_OK;
    if (n_seg >= INT32_MAX) return CWN_ERR_TOO_LARGE;
    CollateBatch B{};
    B.n = n;
    B.table_slot_stride = n_slots > 1 ? table_slot_stride : 0;
    B.cursor = cursor;
    for (int i = 0; i < n; ++i) B.dst_slot_bytes[i] = (n_slots > 1 && dst_slot_bytes != nullptr) ? dst_slot_bytes[i] : 0;
    for (int i = 0; i < n; ++i) {
        const cwn_collate_desc& D = descs[i];
        if (D.dst == nullptr || D.dst_start == nullptr || D.n_rows < 1 || D.n_rows > 2) return CWN_ERR_BAD_ARG;
        if (D.op < CWN_COLLATE_COPY32 || D.op > CWN_COLLATE_ADD32) return CWN_ERR_BAD_ARG;
        if (D.op != CWN_COLLATE_SEGID64 && (D.src == nullptr || D.src_start == nullptr)) return CWN_ERR_BAD_ARG;
        if ((D.op == CWN_COLLATE_ADD64 || D.op == CWN_COLLATE_ADD32) && D.add == nullptr) return CWN_ERR_BAD_ARG;
        B.d[i] = D;
    }
    // four segments per wave where that still leaves the chip more waves than it holds (256 CUs x 32)
    const int64_t waves1 = n_seg * n * n_slots;
    if (waves1 >= 4 * 8192 && n_seg >= 64) {
        constexpr int per = kSegPerWg * 4;
        collate_kernel<4><<<dim3((unsigned)((n_seg + per - 1) / per), (unsigned)n, (unsigned)n_slots), dim3(kThreads), 0,
                            (hipStream_t)stream_>>>(B, n_seg);
    } else {
        collate_kernel<1><<<dim3((unsigned)((n_seg + kSegPerWg - 1) / kSegPerWg), (unsigned)n, (unsigned)n_slots), dim3(kThreads), 0,
                            (hipStream_t)stream_>>>(B, n_seg);
    }
    return hipGetLastError() == hipSuccess ? CWN_OK : CWN_ERR_LAUNCH;
}
