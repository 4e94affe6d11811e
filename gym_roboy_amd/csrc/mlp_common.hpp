// mlp_common.hpp - shared by the policy kernels (mlp_policy.hip: rollout step, mlp_train.hip: PPO gradient):
// operand layout of the packed parameter blob, MFMA / half-wave helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/roboy_policy.h"

namespace rpd {

extern thread_local std::string g_err;
int fail(int code, const std::string &msg);

// The rp_* entry points take no device argument: the device of a call is the one its first device pointer lives on.
// DeviceScope looks it up and makes it current for the duration of the call, then puts the caller's current device back
// (one process per GPU never notices; a process that drives several GPUs gets its launches, its CU count and its LDS
// opt-ins on the right one without torch's current device changing behind its back).  rc != 0: the call fails with it.
struct DeviceScope {
    int rc = 0, dev = 0, prev = -1;
    explicit DeviceScope(const void *d_ptr);
    ~DeviceScope();
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
int cu_count(int dev);
// Dynamic LDS above 64 KB has to be granted per kernel with hipFuncSetAttribute - per DEVICE: the grant is remembered per
// (kernel id, device), under a mutex.  lds_grant_needed() is the bookkeeping alone (host only; tests/test_policy_abi.py
// drives it through rp_debug_lds_grant_needed).
bool lds_grant_needed(int kernel_id, int dev, size_t lds);
int grant_lds(const void *kernel, int kernel_id, int dev, size_t lds);

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int H = RP_HIDDEN, HT = H / 32;            // hidden units, row tiles of a hidden layer
constexpr int STREAM_POLICY = 2;                     // Philox stream of the exploration noise (0: actions, 1: goals)

// unit of D register r within a 32-row tile, lower half-wave (upper: + 4)
constexpr int unit_of(int r) { return (r & 3) + 8 * (r >> 2); }

struct Layout {                  // offsets (floats) into the packed blob
    int k1s;                     // K-steps of layer 1: ceil((obs_dim + 1) / 2)
    int ot_pi;                   // row tiles of the action mean: ceil(act_dim / 32)
    int o_l1;                    // [4 row tiles: pi 0, pi 1, vf 0, vf 1][k1s][64]
    int o_l2[2], o_b2[2];        // per net: [HT out][HT in][16][64], bias [HT][64]
    int o_l3[2], o_b3[2];        // per net: [ot][HT][16][64], bias [ot][64]
    int o_logstd;                // [64]
    int total;                   // floats of the rollout blob (rp_pack)
    // the PPO-gradient blob (rp_pack_train) continues with the TRANSPOSED weights as A operands:
    int k3s[2];                  // per net: K-steps over the outputs, ceil(n_out / 2)
    int o_l3t[2];                // per net: W3^T  [HT unit tiles][k3s][64]: row = unit, k = output 2 s + (l >> 5)
    int o_l2t[2];                // per net: W2^T  [HT in tiles][HT out tiles][16][64]: row = in unit, k = out unit U(r) + 4 (l >> 5)
    int total_train;
};
__host__ __device__ inline Layout layout_of(int obs_dim, int act_dim) {
    Layout L;
    L.k1s = (obs_dim + 2) / 2;
    L.ot_pi = (act_dim + 31) / 32;
    int o = 0;
    L.o_l1 = o; o += 4 * L.k1s * 64;
    for (int n = 0; n < 2; ++n) {
        L.o_l2[n] = o; o += HT * HT * 16 * 64;
        L.o_b2[n] = o; o += HT * 64;
        const int ot = n == 0 ? L.ot_pi : 1;
        L.o_l3[n] = o; o += ot * HT * 16 * 64;
        L.o_b3[n] = o; o += ot * 64;
    }
    L.o_logstd = o; o += 64;
    L.total = (o + 3) & ~3;
    o = L.total;
    for (int n = 0; n < 2; ++n) {
        L.k3s[n] = ((n == 0 ? act_dim : 1) + 1) / 2;
        L.o_l3t[n] = o; o += HT * L.k3s[n] * 64;
        L.o_l2t[n] = o; o += HT * HT * 16 * 64;
    }
    L.total_train = (o + 3) & ~3;
    return L;
}

__device__ __forceinline__ void half_swap(float &a, float &b) {      // a.hi <-> b.lo
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float tanh_fast(float x) {                 // 1 - 2 / (exp(2x) + 1): v_exp_f32 + v_rcp_f32
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ void tanh_tile(f32x16 &d) {
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = tanh_fast(d[r]);
}
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }


// acc0 / acc1 += A_k x b0[k] / b1[k] for k = 0 .. N-1, A operands from LDS at w[k * 64] (this lane's element), with the
// operand of step k + 2 requested before the MFMAs of step k are issued: at one wave per SIMD nothing else hides the
// LDS latency, and without the explicit distance the compiler reloads into the register it has just consumed and
// waits for it (25-30 % of the matrix pipe idle in these loops).
template <int N, typename B0, typename B1>
__device__ __forceinline__ void mfma_stream(const float *w, const B0 &b0, const B1 &b1, f32x16 &acc0, f32x16 &acc1) {
    float a0 = w[0], a1 = N > 1 ? w[64] : 0.0f;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float a2 = k + 2 < N ? w[(k + 2) * 64] : 0.0f;
        acc0 = mfma(a0, b0(k), acc0);
        acc1 = mfma(a0, b1(k), acc1);
        a0 = a1; a1 = a2;
        // keep that order through the machine scheduler: one LDS read, then the step's two MFMAs
        if (k + 2 < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
}

// offsets of one net's parameters inside its gradient vector (torch layout)
struct GOff { int w1, b1, w2, b2, w3, b3, ls, loss; };
__host__ __device__ inline GOff goff_of(int obs_dim, int n_out) {
    GOff g;
    int o = 0;
    g.w1 = o; o += H * obs_dim;
    g.b1 = o; o += H;
    g.w2 = o; o += H * H;
    g.b2 = o; o += H;
    g.w3 = o; o += n_out * H;
    g.b3 = o; o += n_out;
    g.ls = o; o += n_out;          // log-std gradient (action net only; zero for the value net)
    g.loss = o; o += 4;            // [sum of the per-sample loss terms / B, ...]
    return g;
}
__host__ __device__ inline int gstride_of(int obs_dim, int act_dim) {
    const int a = goff_of(obs_dim, act_dim).loss + 4;
    return (a + 3) & ~3;
}

}  // namespace rpd
