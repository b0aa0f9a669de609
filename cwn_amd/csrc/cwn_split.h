// cwn_split.h -- the exact three-way bf16 split shared by the kernels that run fp32 products on the
// bf16 matrix pipe (cwn_gemm_split.hip, cwn_layer.hip).
//
//     x = hi + mid + lo        hi = bf16_rne(x),  mid = bf16_rne(x - hi),  lo = x - hi - mid
//
// Round-to-nearest-even at every step (v_cvt_pk_bf16_f32, two elements per instruction): every
// subtraction is exact in fp32 (Sterbenz), |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|, and what is left
// after two steps has at most 8 significant bits, so `lo` is exact and the three pieces add up to x
// bit for bit.  A product x*w is then nine bf16 products; the kernels keep six
//     hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi
// and drop mid*lo, lo*mid, lo*lo: <= 2^-26 |x||w| together, below the rounding of the fp32
// accumulation itself.  (Round 1 split by TRUNCATION: pieces of the operand's own sign, dropped terms
// up to 2^-22 |x||w| and all of one sign -- a bias; the rounding split has neither property.)
// Limits, stated where the kernels are documented: a non-finite x gives NaN pieces (inf - inf), and
// |x| within one bf16 ulp of FLT_MAX rounds hi to inf; both give NaN where an fmaf chain gives
// inf/NaN/large.  The exact fp32-MFMA kernels serve callers that need those cases.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cwn {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// two fp32 -> packed bf16 pair (element 0 in the low half), round to nearest even
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

struct Split2 { uint32_t h, m, l; };      // packed pairs: element 0 low half, element 1 high half

__device__ __forceinline__ Split2 split3_pair(float a, float b) {
    Split2 s;
    s.h = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(s.h << 16), rb = b - __uint_as_float(s.h & 0xFFFF0000u);
    s.m = cvt_pk_bf16(ra, rb);
    const float qa = ra - __uint_as_float(s.m << 16), qb = rb - __uint_as_float(s.m & 0xFFFF0000u);
    s.l = cvt_pk_bf16(qa, qb);
    return s;
}

// 4 consecutive fp32 -> 4 bf16 of each plane
__device__ __forceinline__ void split4(const float4& v, uint2& ph, uint2& pm, uint2& pl) {
    const Split2 a = split3_pair(v.x, v.y), b = split3_pair(v.z, v.w);
    ph = make_uint2(a.h, b.h);
    pm = make_uint2(a.m, b.m);
    pl = make_uint2(a.l, b.l);
}

// 8 consecutive fp32 -> one MFMA operand fragment (8 bf16) of each plane
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& ph, uint4& pm, uint4& pl) {
    uint2 h0, m0, l0, h1, m1, l1;
    split4(a, h0, m0, l0);
    split4(b, h1, m1, l1);
    ph = make_uint4(h0.x, h0.y, h1.x, h1.y);
    pm = make_uint4(m0.x, m0.y, m1.x, m1.y);
    pl = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

typedef __bf16 frag_ab __attribute__((ext_vector_type(8)));
typedef float frag_cd __attribute__((ext_vector_type(4)));

__device__ __forceinline__ frag_ab as_frag(const uint4& v) { return __builtin_bit_cast(frag_ab, v); }

// One 16x16 output tile += (W fragment) x (X fragment) over 32 k-values, six bf16 MFMAs, smallest
// terms first.  Operand roles are swapped (A = W rows = output columns, B = X rows), so lane l ends
// up with output columns (l >> 4) * 4 .. +3 of X row (l & 15).  Every kernel that wants results
// bit-identical to another one must go through this function (same terms, same order).
__device__ __forceinline__ frag_cd mfma_split6(const uint4& wh, const uint4& wm, const uint4& wl,
                                               const uint4& xh, const uint4& xm, const uint4& xl, frag_cd c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wl), as_frag(xh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wh), as_frag(xl), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wm), as_frag(xm), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wm), as_frag(xh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wh), as_frag(xm), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(wh), as_frag(xh), c, 0, 0, 0);
    return c;
}

}  // namespace cwn
