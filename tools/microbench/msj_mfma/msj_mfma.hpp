// msj_mfma.hpp - env-per-lane kernels of the ball-joint class with the tendon ROUTING on the matrix cores.
//
// The env-per-lane step is bound by VALU issue, and 18 of the ~47 vector instructions a
// tendon costs only route it: a = R^T A, d2 = |B - a|^2, m = a x B.  All 32 of those numbers (8 tendons x
// (d2, m)) are LINEAR in the 9 entries of the env's rotation matrix with constant coefficients, i.e. one
// [32 x 10] . [10 x 64 envs] product per wave and acceleration evaluation (msj_build.hpp: msj_geom_table).
// gfx950 has an exact-f32 matrix instruction (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, bit for bit)
// that runs beside the VALU, so the routing leaves the vector pipe: per wave and evaluation 10 MFMAs
// (2 column tiles of 32 envs x 5 k-pairs, 640 matrix-pipe cycles that other waves' vector work hides) plus 21
// half-wave swaps replace 144 vector instructions.
//
// Operand layouts (cdna guide §3): A: lane l holds [row l & 31][k = l >> 5]; B: lane l holds [k = l >> 5]
// [column l & 31]; D: lane l, register r holds [row (r & 3) + 8 (r >> 2) + 4 (l >> 5)][column l & 31].
// Columns are envs.  With one env per lane, registers X = x_{2p} and Y = x_{2p+1} (all 64 envs each) become
// the B operands of column tile 0 (envs 0-31) and tile 1 (envs 32-63) by ONE v_permlane32_swap (X.hi <-> Y.lo),
// and the same swap on the result registers (D0[r].hi <-> D1[r].lo) brings every output row back to "one env
// per lane": D0'[r] = row 8 (r >> 2) + (r & 3) of all 64 envs, D1'[r] = that row + 4.  Row 4k + i is tendon k's
// (d2, m_x, m_y, m_z)[i], so tendon k reads registers 4 (k >> 1) .. + 3 of D0' (k even) or D1' (k odd).
//
// Every lane of a wave takes part in an MFMA, so lanes past the batch end run the arithmetic on the last env
// and only skip the stores.
#pragma once
#include "msj_kernels.hpp"

namespace rbk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// a.hi <-> b.lo (lanes 32-63 of a with lanes 0-31 of b)
__device__ __forceinline__ void half_swap(float &a, float &b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// acceleration functor (msj_math.hpp: integrate()) with the routing on the matrix cores
struct AccelMfma {
    using M = rb::MsjModel<float, NT8>;
    const Const8 &c;
    const float *u;        // activation offsets, registers
    const float *ga;       // this lane's 5 elements of the routing table (A operands), registers
    __device__ __forceinline__ void operator()(const float q[3], const float qd[3], float qdd[3]) const {
        const M::Frame f = M::frame(q, qd);
        float x[10] = {f.r00, f.r01, f.r02, f.r10, f.r11, f.r12, f.r20, f.r21, f.r22, 1.0f};
#pragma unroll
        for (int p = 0; p < 5; ++p) half_swap(x[2 * p], x[2 * p + 1]);
        f32x16 d0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d1 = d0;
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[p], x[2 * p], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[p], x[2 * p + 1], d1, 0, 0, 0);
        }
        float e0[16], e1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { e0[r] = d0[r]; e1[r] = d1[r]; half_swap(e0[r], e1[r]); }
        float tx = 0.0f, ty = 0.0f, tz = 0.0f;
#pragma unroll
        for (int k = 0; k < NT8; ++k) {
            const float *e = (k & 1) ? e1 : e0;
            const int b = 4 * (k >> 1);
            M::tendon_force(c, f, c.ten[k], u[k], e[b], e[b + 1], e[b + 2], e[b + 3], tx, ty, tz);
        }
        M::rigid_body(c, f, qd, tx, ty, tz, qdd);
    }
};

// msj_step_env_per_lane (msj_kernels.hpp) with AccelMfma; `geom`: msj_geom_table, [5][64] floats
template <int INTEG, int BLOCK, bool BK = false>
__global__ void __launch_bounds__(BLOCK, 4)      // 4 waves per SIMD: <= 128 registers, MFMA results in VGPRs
msj_step_mfma(const Const8 c_arg, const float *__restrict__ geom, float *__restrict__ q, float *__restrict__ qd,
              uint32_t *__restrict__ feas, const float *__restrict__ act, const Scale8 us, long n) {
    const Const8 &c = robot_consts<BK>(c_arg);
    const long i0 = long(blockIdx.x) * BLOCK + threadIdx.x;
    const bool live = i0 < n;
    const long i = live ? i0 : n - 1;
    float ga[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) ga[p] = geom[p * 64 + (threadIdx.x & 63)];
    float qq[3], vv[3], u[NT8];
    const float4 a0 = reinterpret_cast<const float4 *>(act)[2 * i];
    const float4 a1 = reinterpret_cast<const float4 *>(act)[2 * i + 1];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + i]; vv[j] = qd[j * n + i]; }
    const float a[NT8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int k = 0; k < NT8; ++k) u[k] = a[k] * us.v[k];
    const bool ok = rb::MsjModel<float, NT8>::template integrate<INTEG>(c, qq, vv, AccelMfma{c, u, ga});
    if (live) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { q[j * n + i] = qq[j]; qd[j * n + i] = vv[j]; }
        feas[i] = ok ? 1u : 0u;
    }
}

}  // namespace rbk
