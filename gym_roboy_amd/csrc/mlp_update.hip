// mlp_update.hip - what surrounds the minibatch gradient in a PPO update (gym_roboy_amd/ppo.py: update), as kernels of
// their own instead of ~10 torch launches per minibatch (include/roboy_policy.h):
//   rp_perm_dev        the epoch's sample order: a keyed bijection of [0, n) evaluated per element (a cycle-walking
//                      Feistel network on the next even power of two), instead of torch.randperm's sort of n keys
//   rp_adv_stats_dev   mean and 1 / (std + 1e-8) of the minibatch's advantages (gathered through the index), which
//                      rp_ppo_grad_dev then applies per sample: no gathered / normalised copy is materialised
//   rp_clip_adam_dev   clip_grad_norm_ + Adam.step over the flat gradient vector of rp_ppo_grad_dev and a parameter
//                      buffer of the same layout: one workgroup, two passes over ~10^4 floats
// The reference's consumer is stable_baselines' PPO2 (train_parallel.py:28-31); torch's optimiser is the statement these
// kernels are tested against (tests/test_policy_gpu.py).
#include <hip/hip_runtime.h>

#include "../../include/roboy_policy.h"
#include "mlp_common.hpp"

namespace {
using namespace rpd;

// ---- keyed permutation of [0, n) ----
__host__ __device__ inline uint32_t mix32(uint32_t v) {
    v *= 0x9E3779B1u; v ^= v >> 15; v *= 0x85EBCA77u; v ^= v >> 13; v *= 0xC2B2AE3Du; v ^= v >> 16;
    return v;
}
struct PermKey { uint32_t k[4]; int half; uint32_t mask; };
__host__ __device__ inline PermKey perm_key(uint64_t key, long long n) {
    PermKey p;
    int bits = 2;
    while ((1ll << bits) < n) bits += 2;                  // an even number of bits: two equal halves
    p.half = bits / 2;
    p.mask = (1u << p.half) - 1u;
    const uint32_t lo = uint32_t(key), hi = uint32_t(key >> 32);
    for (int r = 0; r < 4; ++r) p.k[r] = mix32(lo + 0x9E3779B9u * uint32_t(r + 1)) ^ mix32(hi + 0x7F4A7C15u * uint32_t(r + 1));
    return p;
}
__host__ __device__ inline long long perm_at(const PermKey &p, long long n, long long i) {
    unsigned long long x = (unsigned long long)i;
    do {                                                   // cycle walking: the network permutes [0, 4^half) and i < n
        uint32_t L = uint32_t(x >> p.half), R = uint32_t(x) & p.mask;
        for (int r = 0; r < 4; ++r) {
            const uint32_t F = mix32(R ^ p.k[r]) & p.mask;
            const uint32_t t = L ^ F;
            L = R; R = t;
        }
        x = ((unsigned long long)L << p.half) | R;
    } while ((long long)x >= n);
    return (long long)x;
}
__global__ void perm_kernel(uint64_t key, long long n, long long first, long long count, long long *out) {
    const PermKey p = perm_key(key, n);
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (long long)gridDim.x * blockDim.x)
        out[j] = perm_at(p, n, first + j);
}

// ---- advantage statistics of a minibatch ----
constexpr int STAT_BLOCKS = 256;
__global__ void __launch_bounds__(256)
adv_stats_kernel(const float *__restrict__ adv, const long long *__restrict__ index, long long B, double *scratch, float *stats) {
    __shared__ double sh[2][4];
    __shared__ bool last;
    double s = 0.0, ss = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < B; i += (long long)gridDim.x * 256) {
        const double a = double(adv[index ? index[i] : i]);
        s += a; ss += a * a;
    }
    auto block_sum = [&]() {
        for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); ss += __shfl_xor(ss, off, 64); }
        if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = ss; }
        __syncthreads();
    };
    block_sum();
    unsigned int *ticket = reinterpret_cast<unsigned int *>(scratch + 2 * STAT_BLOCKS);
    if (threadIdx.x == 0) {
        scratch[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        scratch[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        // hand-off as MI355X_MICROARCH.md prescribes: stores -> agent-scope release -> vmcnt(0) -> ticket; the last block acquires
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last) return;
    s = threadIdx.x < gridDim.x ? __hip_atomic_load(scratch + 2 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    ss = threadIdx.x < gridDim.x ? __hip_atomic_load(scratch + 2 * threadIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
    block_sum();
    if (threadIdx.x == 0) {
        const double sum = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3], sq = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        const double mean = sum / double(B);
        // torch.Tensor.std(): the unbiased estimate
        const double var = B > 1 ? (sq - sum * mean) / double(B - 1) : 0.0;
        stats[0] = float(mean);
        stats[1] = float(1.0 / (sqrt(var > 0.0 ? var : 0.0) + 1e-8));
        *ticket = 0u;                                      // ready for the next launch (same stream: ordered behind this one)
    }
}

// ---- gradient clipping by global norm + Adam, one workgroup ----
struct AdamArgs {
    float *p, *m, *v;
    const float *g;
    int n;
    float lr, beta1, beta2, eps, bc1, bc2, max_norm, gscale;
    int ls_off, ls_len;          // the log-std range: its entropy bonus -ent_coef is added to the gradient here
    float ent_coef;
    int pi_end, vf_begin, vf_end; // the parameters' slots: [0, pi_end) and [vf_begin, vf_end); the rest of the vector (loss terms,
                                 // the value net's unused log-std slot, padding) is neither counted in the norm nor updated
};
__global__ void __launch_bounds__(1024)
clip_adam_kernel(const AdamArgs a) {
    __shared__ float sh[16];
    auto grad = [&](int i) {
        if (!(i < a.pi_end || (i >= a.vf_begin && i < a.vf_end))) return 0.0f;
        float g = a.g[i] * a.gscale;
        if (i >= a.ls_off && i < a.ls_off + a.ls_len) g -= a.ent_coef;
        return g;
    };
    float ss = 0.0f;
    for (int i = threadIdx.x; i < a.n; i += 1024) { const float g = grad(i); ss += g * g; }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += sh[w];
    // torch.nn.utils.clip_grad_norm_: coefficient max_norm / (norm + 1e-6), clamped to 1
    const float coef = fminf(a.max_norm / (sqrtf(tot) + 1e-6f), 1.0f);
    const float step = a.lr / a.bc1, isb2 = 1.0f / sqrtf(a.bc2);
    for (int i = threadIdx.x; i < a.n; i += 1024) {
        if (!(i < a.pi_end || (i >= a.vf_begin && i < a.vf_end))) continue;
        const float g = grad(i) * coef;
        const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
        a.m[i] = m; a.v[i] = v;
        a.p[i] -= step * m / (sqrtf(v) * isb2 + a.eps);     // torch.optim.Adam: denom = sqrt(v) / sqrt(bc2) + eps
    }
}

}  // namespace

extern "C" {

int rp_perm_host(uint64_t key, int64_t n, int64_t first, int64_t count, int64_t *out) {
    if (!out || n < 1 || first < 0 || count < 0 || first + count > n) return fail(RP_EINVAL, "need 0 <= first, first + count <= n");
    const PermKey p = perm_key(key, n);
    for (int64_t j = 0; j < count; ++j) out[j] = perm_at(p, n, first + j);
    return RP_OK;
}

int rp_perm_dev(uint64_t key, int64_t n, int64_t first, int64_t count, int64_t *d_out, void *stream) {
    if (!d_out || n < 1 || first < 0 || count < 0 || first + count > n) return fail(RP_EINVAL, "need 0 <= first, first + count <= n");
    int dev = 0;
    DeviceScope scope(d_out); if (scope.rc) return scope.rc; dev = scope.dev;
    if (count == 0) return RP_OK;
    long blocks = (count + 255) / 256;
    const long cap = 8l * cu_count(dev);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(perm_kernel, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), key, (long long)n,
                       (long long)first, (long long)count, reinterpret_cast<long long *>(d_out));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RP_EHIP, std::string("perm_kernel: ") + hipGetErrorString(e));
    return RP_OK;
}

int64_t rp_adv_stats_scratch_doubles(void) { return 2 * STAT_BLOCKS + 1; }

int rp_adv_stats_dev(const float *d_adv, const int64_t *d_index, int64_t batch, float *d_stats2, double *d_scratch, void *stream) {
    if (!d_adv || !d_stats2 || !d_scratch) return fail(RP_EINVAL, "null argument");
    if (batch < 1) return fail(RP_EINVAL, "batch must be >= 1");
    DeviceScope scope(d_adv); if (scope.rc) return scope.rc;
    long blocks = (batch + 2047) / 2048;                   // at least eight samples per thread
    if (blocks > STAT_BLOCKS) blocks = STAT_BLOCKS;
    hipLaunchKernelGGL(adv_stats_kernel, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), d_adv,
                       reinterpret_cast<const long long *>(d_index), (long long)batch, d_scratch, d_stats2);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RP_EHIP, std::string("adv_stats_kernel: ") + hipGetErrorString(e));
    return RP_OK;
}

int rp_clip_adam_dev(float *d_params, const float *d_grad, float *d_m, float *d_v, int obs_dim, int act_dim, float lr,
                     float beta1, float beta2, float eps, int64_t step, float max_grad_norm, float grad_scale, float ent_coef,
                     void *stream) {
    if (!d_params || !d_grad || !d_m || !d_v) return fail(RP_EINVAL, "null argument");
    if (step < 1) return fail(RP_EINVAL, "step counts from 1");
    const int64_t n = rp_grad_floats(obs_dim, act_dim);
    if (n < 0) return RP_EUNSUPPORTED;
    DeviceScope scope(d_params); if (scope.rc) return scope.rc;
    AdamArgs a;
    a.p = d_params; a.g = d_grad; a.m = d_m; a.v = d_v; a.n = int(n);
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.bc1 = float(1.0 - pow(double(beta1), double(step))); a.bc2 = float(1.0 - pow(double(beta2), double(step)));
    a.max_norm = max_grad_norm; a.gscale = grad_scale; a.ent_coef = ent_coef;
    // layout of the gradient vector (mlp_common.hpp: goff_of): per net w1, b1, w2, b2, w3, b3, log_std, loss (4)
    const GOff pi = goff_of(obs_dim, act_dim), vf = goff_of(obs_dim, 1);
    const int gs = gstride_of(obs_dim, act_dim);
    a.ls_off = pi.ls; a.ls_len = act_dim;
    a.pi_end = pi.loss; a.vf_begin = gs; a.vf_end = gs + vf.ls;
    hipLaunchKernelGGL(clip_adam_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RP_EHIP, std::string("clip_adam_kernel: ") + hipGetErrorString(e));
    return RP_OK;
}

}  // extern "C"
