// mlp_policy.hip - the PPO consumer's policy step as one kernel on the matrix cores (include/roboy_policy.h).
//
//   obs [n][obs_dim]  ->  two MLPs (obs -> 64 tanh -> 64 tanh -> out; out = act_dim for the action mean, 1 for the
//   value; gym_roboy_amd/ppo.py: MlpPolicy)  ->  action = mean + exp(log_std) * eps, its log-probability, the value
//
// Mapping.  A wave owns a tile of 64 samples = two column tiles of 32 for v_mfma_f32_32x32x2_f32 (f32 in, f32
// accumulate: bit for bit a k-ordered fmaf chain, so the result is the fp32 network, not a reduced-precision one).
// Samples are the COLUMNS of every product (lane & 31), units the rows:
//   * layer 1: B operand = obs read straight from HBM in operand order (lane l of K-step s reads element
//     2s + (l >> 5) of sample l & 31; column obs_dim is the constant 1 that carries the bias);
//   * a layer's result registers ARE the next layer's B operands: D register r of a 32x32 tile holds, for the
//     lane's sample, unit U(r) = (r & 3) + 8 (r >> 2) in lanes 0-31 and unit U(r) + 4 in lanes 32-63 - exactly a
//     B operand whose K pair is (U(r), U(r) + 4).  The weights (A operands) are packed in that k order by
//     rp_pack(), so between the layers the data never move: tanh is applied to the accumulators in place;
//   * the output tile of the last layer is brought back to "one sample per lane" by v_permlane32_swap of the two
//     column tiles' registers (lanes 32-63 of tile 0 <-> lanes 0-31 of tile 1), and every lane finishes its own
//     sample: Philox + Box-Muller noise, action, log-probability, stores.
// The A operands (weights, 56 KB for MsjRobot's 9 -> 8 policy) are staged into LDS once per workgroup and read
// one ds_read_b32 per MFMA pair (requested two steps ahead: mlp_common.hpp); waves are persistent over tiles.  f32 MFMA
// does not overlap f32 VALU work on gfx950 (tools/microbench/mfma_valu_overlap.hip), so the tanh passes add to the
// 436 MFMAs per tile; the two nets run one after the other to stay within 256 registers = two waves per SIMD, which
// is what keeps the quarter-rate instructions of tanh issuing (85 -> 78 us at 262 144 samples).
#include <hip/hip_runtime.h>

#include <mutex>

#include <cmath>
#include <cstring>
#include <string>

#include "../../include/roboy_policy.h"
#include "mlp_common.hpp"
#include "philox.hpp"

namespace rpd {
thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }

DeviceScope::DeviceScope(const void *d_ptr) {
    hipPointerAttribute_t at;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); rc = fail(RP_EHIP, "no HIP device"); return; }
    dev = cur;
    if (d_ptr && hipPointerGetAttributes(&at, d_ptr) == hipSuccess && at.type == hipMemoryTypeDevice) dev = at.device;
    else (void)hipGetLastError();                          // (a pointer the runtime does not know leaves an error behind)
    if (dev != cur) {
        const hipError_t e = hipSetDevice(dev);
        if (e != hipSuccess) { rc = fail(RP_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); return; }
        prev = cur;                                        // restored when the call returns
    }
}
DeviceScope::~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }

int cu_count(int dev) {
    int n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
}

namespace {
constexpr int GRANT_KERNELS = 8, GRANT_DEVICES = 64;
std::mutex g_grant_mutex;
size_t g_granted[GRANT_KERNELS][GRANT_DEVICES];            // bytes granted so far (0: the default 64 KB)
}
static bool tracked(int kernel_id, int dev) { return kernel_id >= 0 && kernel_id < GRANT_KERNELS && dev >= 0 && dev < GRANT_DEVICES; }
// the bookkeeping alone (caller holds the mutex or is the single-threaded test hook): does (kernel, device) still need a grant of `lds`?
static bool needs_grant_locked(int kernel_id, int dev, size_t lds) {
    if (lds <= 64 * 1024) return false;
    if (!tracked(kernel_id, dev)) return true;                                                           // not tracked: always ask
    return g_granted[kernel_id][dev] < lds;
}
bool lds_grant_needed(int kernel_id, int dev, size_t lds) {
    std::lock_guard<std::mutex> lock(g_grant_mutex);
    const bool need = needs_grant_locked(kernel_id, dev, lds);
    if (need && tracked(kernel_id, dev)) g_granted[kernel_id][dev] = lds;      // (the test hook plays the successful grant)
    return need;
}
int grant_lds(const void *kernel, int kernel_id, int dev, size_t lds) {
    // the mutex is held across hipFuncSetAttribute and the grant is recorded only once it has succeeded: a second thread
    // cannot see "granted" and launch with more than 64 KB before the attribute is set
    std::lock_guard<std::mutex> lock(g_grant_mutex);
    if (!needs_grant_locked(kernel_id, dev, lds)) return RP_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return fail(RP_EHIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    if (tracked(kernel_id, dev)) g_granted[kernel_id][dev] = lds;
    return RP_OK;
}
}  // namespace rpd

namespace {
using namespace rpd;

// one net on a 64-sample tile: x0 / x1: the observation rows of this lane's samples in column tile 0 / 1;
// y[q][t]: output row tile q (only q < ot computed), column tile t, brought back to one sample per lane.
// The two nets run one after the other so that a wave stays within 256 registers (two waves per SIMD: at one wave
// per SIMD the tanh passes - 2 quarter-rate instructions per element - issue at less than half their rate).
__device__ __forceinline__ void net_forward(const float *lds, const Layout &L, int net, int ot, int lane, float onehot,
                                            const float *x0, const float *x1, int obs_dim, f32x16 (&y)[2][2]) {
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int half = lane >> 5;
    f32x16 h1[HT][2];
#pragma unroll
    for (int m = 0; m < HT; ++m) { h1[m][0] = zero; h1[m][1] = zero; }
    // layer 1: K loop over the observation, B operands straight from HBM (column obs_dim = 1 carries the bias)
    const float *w1 = lds + L.o_l1 + (net * HT) * L.k1s * 64 + lane;
    for (int s = 0; s < L.k1s; ++s) {
        const int k = 2 * s + half;
        const float b0 = k < obs_dim ? x0[k] : (k == obs_dim ? 1.0f : 0.0f);
        const float b1 = k < obs_dim ? x1[k] : (k == obs_dim ? 1.0f : 0.0f);
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            const float a = w1[(m * L.k1s + s) * 64];
            h1[m][0] = mfma(a, b0, h1[m][0]);
            h1[m][1] = mfma(a, b1, h1[m][1]);
        }
    }
#pragma unroll
    for (int m = 0; m < HT; ++m) { tanh_tile(h1[m][0]); tanh_tile(h1[m][1]); }
    f32x16 h2[HT][2];
#pragma unroll
    for (int o = 0; o < HT; ++o) {
        h2[o][0] = zero; h2[o][1] = zero;
        const float *w = lds + L.o_l2[net] + o * (HT * 16 * 64) + lane;
        mfma_stream<HT * 16>(w, [&](int k) { return h1[k >> 4][0][k & 15]; }, [&](int k) { return h1[k >> 4][1][k & 15]; },
                             h2[o][0], h2[o][1]);
        const float b = lds[L.o_b2[net] + o * 64 + lane];      // bias: K pair (1, 0)
        h2[o][0] = mfma(b, onehot, h2[o][0]);
        h2[o][1] = mfma(b, onehot, h2[o][1]);
        tanh_tile(h2[o][0]); tanh_tile(h2[o][1]);
    }
    if (net == 1) {
        // the value net's ONE output row on the VALU (as a 32-row MFMA tile it would be 31/32 padding: 66 of the 436
        // MFMAs per tile): every lane holds 32 of the 64 h2 units of its column's sample; the half swap adds the other
        // half's partial sum and leaves one sample per lane, as the MFMA path does.  y[0][0][0] = the value.
        float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
        for (int m = 0; m < HT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float wv = lds[L.o_l3[net] + (m * 16 + r) * 64 + 32 * half];     // row 0 of the A operand: lane 0 / 32
                p0 += wv * h2[m][0][r];
                p1 += wv * h2[m][1][r];
            }
        half_swap(p0, p1);
        y[0][0] = zero; y[0][1] = zero; y[1][0] = zero; y[1][1] = zero;
        y[0][0][0] = p0 + p1 + lds[L.o_b3[net]];
        return;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        y[q][0] = zero; y[q][1] = zero;
        if (q < ot) {
            const float *w = lds + L.o_l3[net] + q * (HT * 16 * 64) + lane;
            mfma_stream<HT * 16>(w, [&](int k) { return h2[k >> 4][0][k & 15]; }, [&](int k) { return h2[k >> 4][1][k & 15]; },
                                 y[q][0], y[q][1]);
            const float b = lds[L.o_b3[net] + q * 64 + lane];
            y[q][0] = mfma(b, onehot, y[q][0]);
            y[q][1] = mfma(b, onehot, y[q][1]);
            // back to one sample per lane: afterwards y[q][0][r] is output row 32 q + U(r), y[q][1][r] row 32 q + U(r) + 4
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float lo = y[q][0][r], hi = y[q][1][r];
                half_swap(lo, hi);
                y[q][0][r] = lo; y[q][1][r] = hi;
            }
        }
    }
}

__global__ void __launch_bounds__(256, 2)
mlp_act_kernel(const float *__restrict__ packed, const float *__restrict__ obs, float *__restrict__ act,
               float *__restrict__ logp, float *__restrict__ value, float *__restrict__ mean_out, long n, int obs_dim,
               int act_dim, uint64_t seed, uint64_t sample_offset, uint32_t step, const uint32_t *__restrict__ step_base,
               int deterministic) {
    extern __shared__ float4 lds4[];
    if (step_base) step += *step_base;
    float *lds = reinterpret_cast<float *>(lds4);
    const Layout L = layout_of(obs_dim, act_dim);
    for (int k = threadIdx.x; k < L.total / 4; k += blockDim.x) lds4[k] = reinterpret_cast<const float4 *>(packed)[k];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const float onehot = half ? 0.0f : 1.0f;
    const long n_tiles = (n + 63) / 64;
    // work item = (tile, net): the two nets of a tile share nothing but the observation, so they go to different
    // waves - twice the items, half the dependent MFMA chain per item (what a small batch waits for: 4 096 samples are
    // 64 tiles for 1 024 SIMDs)
    for (long item = long(blockIdx.x) * nw + wave; item < 2 * n_tiles; item += long(gridDim.x) * nw) {
        const long tile = item >> 1;
        long s0 = tile * 64 + col, s1 = s0 + 32;                    // this lane's samples in column tile 0 / 1
        s0 = s0 < n ? s0 : n - 1; s1 = s1 < n ? s1 : n - 1;         // past the end: shadow the last sample
        const float *x0 = obs + s0 * obs_dim, *x1 = obs + s1 * obs_dim;
        const long i = tile * 64 + lane;
        const bool live = i < n;
        f32x16 ypi[2][2];
        if (item & 1) {                                              // the value net
            net_forward(lds, L, 1, 1, lane, onehot, x0, x1, obs_dim, ypi);
            if (live) value[i] = ypi[0][0][0];                       // row 0 of its output tile
            continue;
        }
        net_forward(lds, L, 0, L.ot_pi, lane, onehot, x0, x1, obs_dim, ypi);    // the action-mean net
        // ---- epilogue: this lane's sample ----
        float lp = -0.91893853320467274f * float(act_dim);          // -1/2 log(2 pi) per dimension
        float *arow = act + (live ? i : 0) * act_dim, *mrow = mean_out ? mean_out + (live ? i : 0) * act_dim : nullptr;
        const uint64_t gid = sample_offset + uint64_t(i);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    // rows 32 q + 8 g + 4 hh + (0..3): one Philox block gives their four normals
                    const int j0 = 32 * q + 8 * g + 4 * hh;
                    if (j0 < act_dim) {
                        float eps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (!deterministic) {
                            const rb::Philox4 u = rb::philox_draw(seed, gid, step, STREAM_POLICY, uint32_t(j0 >> 2));
#pragma unroll
                            for (int pr = 0; pr < 2; ++pr) {           // Box-Muller on (u1 in (0,1], u2 in [0,1))
                                const float u1 = float((u.v[2 * pr] >> 8) + 1u) * (1.0f / 16777216.0f);
                                const float u2 = rb::u01(u.v[2 * pr + 1]);
                                const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1));
                                float sn, cs;
                                __sincosf(6.2831853071795865f * u2, &sn, &cs);
                                eps[2 * pr] = rad * cs; eps[2 * pr + 1] = rad * sn;
                            }
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int j = j0 + c;
                            if (j < act_dim) {
                                const float mu = hh ? ypi[q][1][4 * g + c] : ypi[q][0][4 * g + c];
                                const float ls = lds[L.o_logstd + j];
                                const float a = mu + __expf(ls) * eps[c];
                                lp -= 0.5f * eps[c] * eps[c] + ls;
                                if (live) { arow[j] = a; if (mrow) mrow[j] = mu; }
                            }
                        }
                    }
                }
        if (live) logp[i] = lp;
    }
}

}  // namespace

extern "C" {

int rp_abi_version(void) { return RP_ABI_VERSION; }
int rp_debug_lds_grant_needed(int kernel_id, int dev, int64_t lds_bytes) { return lds_grant_needed(kernel_id, dev, size_t(lds_bytes)) ? 1 : 0; }
const char *rp_last_error(void) { return g_err.c_str(); }

int64_t rp_packed_floats(int obs_dim, int act_dim) {
    if (obs_dim < 1 || obs_dim > RP_MAX_OBS || act_dim < 1 || act_dim > RP_MAX_ACT)
        return fail(RP_EUNSUPPORTED, "supported: 1 <= obs_dim <= 95, 1 <= act_dim <= 64");
    return layout_of(obs_dim, act_dim).total;
}

int rp_pack(const rp_mlp_params *p, int obs_dim, int act_dim, float *out) {
    if (!p || !out) return fail(RP_EINVAL, "null argument");
    if (rp_packed_floats(obs_dim, act_dim) < 0) return RP_EUNSUPPORTED;
    const Layout L = layout_of(obs_dim, act_dim);
    std::memset(out, 0, sizeof(float) * size_t(L.total));
    const float *w1[2] = {p->pi_w1, p->vf_w1}, *b1[2] = {p->pi_b1, p->vf_b1}, *w2[2] = {p->pi_w2, p->vf_w2};
    const float *b2[2] = {p->pi_b2, p->vf_b2}, *w3[2] = {p->pi_w3, p->vf_w3}, *b3[2] = {p->pi_b3, p->vf_b3};
    for (int net = 0; net < 2; ++net) {
        const int n_out = net == 0 ? act_dim : 1, ot = net == 0 ? L.ot_pi : 1;
        for (int m = 0; m < HT; ++m)                                  // layer 1: A[row][k], k = column of [W1 | b1 | 0]
            for (int s = 0; s < L.k1s; ++s)
                for (int l = 0; l < 64; ++l) {
                    const int row = 32 * m + (l & 31), k = 2 * s + (l >> 5);
                    const float v = k < obs_dim ? w1[net][row * obs_dim + k] : (k == obs_dim ? b1[net][row] : 0.0f);
                    out[L.o_l1 + ((2 * net + m) * L.k1s + s) * 64 + l] = v;
                }
        for (int o = 0; o < HT; ++o) {                                // layer 2: K pair of register r = (U(r), U(r) + 4) of row tile m
            for (int m = 0; m < HT; ++m)
                for (int r = 0; r < 16; ++r)
                    for (int l = 0; l < 64; ++l) {
                        const int row = 32 * o + (l & 31), in = 32 * m + unit_of(r) + 4 * (l >> 5);
                        out[L.o_l2[net] + ((o * HT + m) * 16 + r) * 64 + l] = w2[net][row * H + in];
                    }
            for (int l = 0; l < 32; ++l) out[L.o_b2[net] + o * 64 + l] = b2[net][32 * o + l];
        }
        for (int q = 0; q < ot; ++q) {                                // output layer, rows past n_out are zero
            for (int m = 0; m < HT; ++m)
                for (int r = 0; r < 16; ++r)
                    for (int l = 0; l < 64; ++l) {
                        const int row = 32 * q + (l & 31), in = 32 * m + unit_of(r) + 4 * (l >> 5);
                        out[L.o_l3[net] + ((q * HT + m) * 16 + r) * 64 + l] = row < n_out ? w3[net][row * H + in] : 0.0f;
                    }
            for (int l = 0; l < 32; ++l) out[L.o_b3[net] + q * 64 + l] = 32 * q + l < n_out ? b3[net][32 * q + l] : 0.0f;
        }
    }
    for (int j = 0; j < act_dim; ++j) out[L.o_logstd + j] = p->log_std[j];
    return RP_OK;
}

int rp_act_dev(const float *d_packed, const float *d_obs, float *d_act, float *d_logp, float *d_value, float *d_mean,
               int64_t n, int obs_dim, int act_dim, uint64_t seed, uint64_t sample_offset, uint32_t step,
               const uint32_t *d_step_base, int deterministic, void *stream) {
    if (!d_packed || !d_obs || !d_act || !d_logp || !d_value) return fail(RP_EINVAL, "null argument");
    if (n < 1) return fail(RP_EINVAL, "n must be >= 1");
    if (rp_packed_floats(obs_dim, act_dim) < 0) return RP_EUNSUPPORTED;
    const Layout L = layout_of(obs_dim, act_dim);
    const size_t lds = sizeof(float) * size_t(L.total);
    if (lds > 160 * 1024) return fail(RP_EUNSUPPORTED, "policy too large for the LDS-resident form");
    hipError_t e = hipSuccess;
    int dev = 0;
    DeviceScope scope(d_packed); if (scope.rc) return scope.rc; dev = scope.dev;          // the blob's device is the device of the call
    if (int rc = grant_lds(reinterpret_cast<const void *>(&mlp_act_kernel), 0, dev, lds)) return rc;   // per kernel and device
    const int n_cu = cu_count(dev);
    const long n_tiles = (n + 63) / 64;
    // 4 waves per workgroup (one per SIMD), two workgroups per CU, persistent over tiles; 6-wave workgroups spread
    // unevenly over the 4 SIMDs and were slower (107 against 78 us at 262 144 samples)
    long blocks = (2 * n_tiles + 3) / 4;             // items = (tile, net)
    if (blocks > 2 * n_cu) blocks = 2 * n_cu;
    hipLaunchKernelGGL(mlp_act_kernel, dim3(unsigned(blocks)), dim3(256), lds, static_cast<hipStream_t>(stream), d_packed,
                       d_obs, d_act, d_logp, d_value, d_mean, long(n), obs_dim, act_dim, seed, sample_offset, step,
                       d_step_base, deterministic);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(RP_EHIP, std::string("mlp_act_kernel: ") + hipGetErrorString(e));
    return RP_OK;
}

}  // extern "C"
