// cwn_dropout.h -- F.dropout of the callers (mp/molec_models.py:104-106, 298-300, 345-346; mp/models.py) without a mask tensor:
// the keep decision of element e of a dropout application is a pure function of (seed, step, site, e), so the forward's
// epilogue and the backward's prologue each derive it where the value already sits in registers.
//
//   r = Philox4x32-10( counter = (e / 4, site, step_lo, step_hi), key = (seed_lo, seed_hi) )[e % 4]
//   keep(e)  <=>  r >= floor(p * 2^32)             multiplier = keep ? 1 / (1 - p) : 0
//
// seed / step live in device memory (cwn_dropout.state): a captured training step advances `step` in its opening launch
// (cwn_step_begin), so every replay of one hipGraph draws fresh masks; `site` tells the applications of one step apart (a host
// counter baked into the launch).  Philox4x32-10 as published (Salmon et al., SC'11: multipliers 0xD2511F53 / 0xCD9E8D57,
// Weyl constants 0x9E3779B9 / 0xBB67AE85) -- the generator torch's own dropout uses; the streams are not torch's (torch keys
// on its generator's offset; nothing in the reference pins a mask).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/cwn_hip.h"

namespace cwn {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

struct Dropout {
    uint2 key;
    uint32_t site, step_lo, step_hi, thresh;
    float scale;
    bool on;

    __device__ __forceinline__ void init(const cwn_dropout& d) {
        on = d.state != nullptr && d.p > 0.f;
        key = make_uint2(0u, 0u);
        site = d.site;
        step_lo = step_hi = 0u;
        thresh = 0u;
        scale = 1.f;
        if (on) {
            const uint64_t seed = (uint64_t)d.state[0], step = (uint64_t)d.state[1];
            key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
            step_lo = (uint32_t)step;
            step_hi = (uint32_t)(step >> 32);
            const double t = (double)d.p * 4294967296.0;
            thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
            scale = 1.0f / (1.0f - d.p);
        }
    }
    // multipliers of elements 4 q .. 4 q + 3
    __device__ __forceinline__ float4 mul4(uint32_t q) const {
        if (!on) return make_float4(1.f, 1.f, 1.f, 1.f);
        const uint4 r = philox4x32_10(make_uint4(q, site, step_lo, step_hi), key);
        return make_float4(r.x >= thresh ? scale : 0.f, r.y >= thresh ? scale : 0.f, r.z >= thresh ? scale : 0.f,
                           r.w >= thresh ? scale : 0.f);
    }
    // multiplier of element e
    __device__ __forceinline__ float mul1(uint64_t e) const {
        if (!on) return 1.f;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)(e >> 2), site, step_lo, step_hi), key);
        const uint32_t j = (uint32_t)e & 3u;
        const uint32_t v = j == 0 ? r.x : (j == 1 ? r.y : (j == 2 ? r.z : r.w));
        return v >= thresh ? scale : 0.f;
    }
};

}  // namespace cwn
