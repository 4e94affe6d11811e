// philox.hpp - Philox4x32-10 counter-based generator (Salmon et al., SC'11;
// the published Random123 algorithm and constants), used for the synthetic
// action slabs and for goal sampling (keyed by (seed, global
// env id) so results do not depend on how the batch is sharded over GPUs).
//
// The reference draws goals and test actions from numpy's global generator
// (roboy_env.py:114-115, simulation_client.py:46-47); only "random, distinct,
// inside the box" is asserted (test_simulation_client.py:47-51), so the
// stream itself is build-defined.  oracle/philox_np.py restates it in numpy
// and is checked bit-for-bit, including against the Random123 known-answer
// vectors.
#pragma once
#include "rtc_compat.hpp"

#if defined(__HIPCC__)
#define RB_PHILOX_HD __host__ __device__ __forceinline__
#else
#define RB_PHILOX_HD inline
#endif

namespace rb {

enum { STREAM_ACTIONS = 0, STREAM_GOALS = 1 };

struct Philox4 { uint32_t v[4]; };

RB_PHILOX_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = uint64_t(0xD2511F53u) * c0;
        const uint64_t p1 = uint64_t(0xCD9E8D57u) * c2;
        const uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = uint32_t(p1);
        const uint32_t n2 = uint32_t(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = uint32_t(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 out;
    out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

// counter layout: (env id low, env id high, index, stream << 8 | block)
RB_PHILOX_HD Philox4 philox_draw(uint64_t seed, uint64_t env_id, uint32_t index,
                                 uint32_t stream, uint32_t block) {
    return philox4x32_10(uint32_t(env_id), uint32_t(env_id >> 32), index, (stream << 8) | block,
                         uint32_t(seed), uint32_t(seed >> 32));
}

// 24 random bits -> [0, 1): exact in fp32
RB_PHILOX_HD float u01(uint32_t u) { return float(u >> 8) * (1.0f / 16777216.0f); }
// -> [-1, 1): 2x - 1 is exact in fp32 for x a multiple of 2^-24
RB_PHILOX_HD float usym(uint32_t u) { return 2.0f * u01(u) - 1.0f; }

}  // namespace rb
