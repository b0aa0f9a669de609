// cwn_layer_bwd_own.h -- record fields and LDS layout of the OWNER form of the blocked backward launch
// (cwn_layer_bwd_own_f32): shared by the kernel (cwn_layer_bwd_own.hip) and the host table builder
// (cwn_blockplan.cpp: cwn_layer_bwd_items_build), so that both derive one item's layout from its record the same way.
// Record layout: include/cwn_hip.h.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CWN_BWD_HD __host__ __device__ __forceinline__
#else
#define CWN_BWD_HD inline
#endif

namespace cwn_bwd_own {

enum {
    R_FLAGS = 0, R_DIM, R_OWN_R0, R_OWN_N, R_ABOVE_R0, R_ABOVE_N, R_BELOW_R0, R_BELOW_N,
    R_UPA_E0, R_UPA_NE, R_UPB_E0, R_UPB_NE, R_BND_E0, R_BND_NE, R_LDS_BYTES
};
enum { F_PA = 1, F_PB = 2, F_TOP = 4 };

constexpr int kThreads = 1024;
// what an item's layout may take: the CU's 160 KiB less the launch's scratch for the BatchNorm sums (cwn_layer_bwd_dim.out_bn: [4][F] floats)
constexpr int kLdsCap = 160 * 1024 - 2048;

CWN_BWD_HD int pad16i(int n) { return (n + 15) & ~15; }
CWN_BWD_HD int pad4i(int n) { return (n + 3) & ~3; }

// rows a lane group may own: own rows = kMaxOwn rounds of the workgroup, top rows = one round
CWN_BWD_HD int rows_per_round(int F) { return kThreads / (F / 4); }
CWN_BWD_HD int own_rows_cap(int F) { return F == 64 ? 256 : 96; }          // = CWN_LAYER_GEMM_ROWS(F)
CWN_BWD_HD int top_rows_cap(int F) { return rows_per_round(F); }

// One item's LDS, all derived from the record:
//   region 0: the staged fp32 rows (stride F + 4) while the entries are walked, the bf16 planes of the gradients of
//             the products (stride F + 8 halves, three planes) afterwards
//   O:        fp32 [RO + RT][F + 4]: self terms + boundary transposes, then + the products: the rows of dx
//   entries:  int32, structure of arrays: the keys the rows are matched on (local row numbers) and, per entry, the LDS
//             offsets of the staged rows its term reads (6 ints per entry of up_index_d, 3 of up_index_{d-1}, 2 of b_index_{d+1})
//   P (TOP):  fp32 [rows_per_round][F + 4]: one partial sum per lane group of the TOP rows' gradients (a ring of six is
//             the coface of thirty entries: its row is gathered by several lane groups, each over a slice of the entries)
struct Layout {
    // staged blocks, first row in region 0 (meaningful only when the block is present)
    int y1o, guo, y2a, y2o, y1b, gub, gba, s_rows;
    // plane blocks (rows), padded to 16
    int pl_a, pl_b, pl_c, pl_rows;
    int RO, RT;                      // rows of O: own (padded to 16), top (padded to 16, 0 without TOP)
    int ea4, eb4, bd4;               // entries of the three lists padded to 4
    int o_off, ent_off, p_off, total; // byte offsets of O, of the entries, of P; bytes in all
};

CWN_BWD_HD Layout layout(int F, int flags, int n_o, int n_a, int n_b, int ne_a, int ne_b, int ne_bd) {
    Layout L;
    const bool pa = (flags & F_PA) != 0, pb = (flags & F_PB) != 0, top = (flags & F_TOP) != 0;
    int r = 0;
    L.y1o = r; r += pa ? n_o : 0;
    L.guo = r; r += pa ? n_o : 0;
    L.y2a = r; r += pa ? n_a : 0;
    L.y2o = r; r += pb ? n_o : 0;
    L.y1b = r; r += pb ? n_b : 0;
    L.gub = r; r += pb ? n_b : 0;
    L.gba = r; r += ne_bd > 0 ? n_a : 0;
    L.s_rows = r;
    L.RO = pad16i(n_o);
    L.RT = top ? pad16i(n_a) : 0;
    int p = 0;
    L.pl_a = p; p += pa ? L.RO : 0;
    L.pl_b = p; p += pb ? L.RO : 0;
    L.pl_c = p; p += L.RT;
    L.pl_rows = p;
    const int s_bytes = L.s_rows * (F + 4) * 4, p_bytes = 3 * L.pl_rows * (F + 8) * 2;
    const int reg0 = ((s_bytes > p_bytes ? s_bytes : p_bytes) + 15) & ~15;
    L.o_off = reg0;
    L.ent_off = reg0 + (L.RO + L.RT) * (F + 4) * 4;
    L.ea4 = pa ? pad4i(ne_a) : 0;
    L.eb4 = pb ? pad4i(ne_b) : 0;
    L.bd4 = pad4i(ne_bd);
    L.p_off = (L.ent_off + (6 * L.ea4 + 3 * L.eb4 + 2 * L.bd4) * 4 + 15) & ~15;
    L.total = L.p_off + (top && n_a > 0 ? rows_per_round(F) * (F + 4) * 4 : 0);
    return L;
}

}  // namespace cwn_bwd_own
