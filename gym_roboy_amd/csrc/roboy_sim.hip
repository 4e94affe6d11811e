// roboy_sim.hip - MI355X (gfx950) implementation of include/roboy_sim.h.
//
// Replaces the per-step ROS round-trip of the reference's RosSimulationClient
// (gym_roboy/envs/simulations/ros_simulation_client.py:32-81) by kernels that
// advance N independent environments in lock-step.  State is struct-of-arrays
// in HBM, one env per lane (or one tendon per lane for small batches), robot
// constants are wave-uniform and arrive through the kernarg (scalar loads ->
// SGPRs), no LDS is needed in the env-per-lane form.  DESIGN.md §4-§5.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/roboy_sim.h"
#include "msj_build.hpp"
#include "msj_math.hpp"
#include "philox.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define RB_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(RB_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

constexpr int NT8 = 8;
using Const8 = rb::MsjConst<float, NT8>;

// ------------------------------------------------------------------ kernels

// One env per lane.  Loads: q, qd planes (dword per lane, 256 B contiguous per
// wave and plane) and the env's 32-byte action record (two dwordx4).  Stores:
// q', qd' planes and the feasibility word.  84 algorithmic bytes per env step.
template <int INTEG, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
msj_step_env_per_lane(const Const8 c, float *__restrict__ q, float *__restrict__ qd,
                      uint32_t *__restrict__ feas, const float *__restrict__ act,
                      float act_scale, long n) {
    const long i = long(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    float qq[3], vv[3], sp[NT8];
    const float4 a0 = reinterpret_cast<const float4 *>(act)[2 * i];
    const float4 a1 = reinterpret_cast<const float4 *>(act)[2 * i + 1];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + i]; vv[j] = qd[j * n + i]; }
    sp[0] = a0.x * act_scale; sp[1] = a0.y * act_scale; sp[2] = a0.z * act_scale; sp[3] = a0.w * act_scale;
    sp[4] = a1.x * act_scale; sp[5] = a1.y * act_scale; sp[6] = a1.z * act_scale; sp[7] = a1.w * act_scale;
    const bool ok = rb::MsjModel<float, NT8>::template step<INTEG>(c, qq, vv, sp);
#pragma unroll
    for (int j = 0; j < 3; ++j) { q[j * n + i] = qq[j]; qd[j * n + i] = vv[j]; }
    feas[i] = ok ? 1u : 0u;
}

__global__ void reset_kernel(float *q, float *qd, uint32_t *feas, const uint8_t *mask, int n_q, long n) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    for (int j = 0; j < n_q; ++j) { q[j * n + i] = 0.0f; qd[j * n + i] = 0.0f; }
    feas[i] = 1u;
}

// SoA planes -> row-major [n][n_q] staging (host I/O only)
__global__ void pack_rows_kernel(const float *planes, float *rows, int n_q, long n) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < n_q; ++j) rows[i * n_q + j] = planes[j * n + i];
}
__global__ void unpack_rows_kernel(const float *rows, float *planes, int n_q, long n) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < n_q; ++j) planes[j * n + i] = rows[i * n_q + j];
}
__global__ void feas_to_u8_kernel(const uint32_t *f, uint8_t *o, long n) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) o[i] = f[i] ? 1 : 0;
}
__global__ void feas_from_u8_kernel(const uint8_t *f, uint32_t *o, long n) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) o[i] = (f == nullptr || f[i]) ? 1u : 0u;
}

// synthetic actions: env i, step t -> n_t uniforms in [-1, 1), row-major
__global__ void fill_actions_kernel(float *act, int n_t, long n, uint64_t seed, uint64_t env0, uint32_t step) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int b = 0; 4 * b < n_t; ++b) {
        const rb::Philox4 r = rb::philox_draw(seed, env0 + uint64_t(i), step, rb::STREAM_ACTIONS, uint32_t(b));
        for (int e = 0; e < 4 && 4 * b + e < n_t; ++e) act[i * n_t + 4 * b + e] = rb::usym(r.v[e]);
    }
}

// goal = lo + (hi - lo) * u, both operations rounded separately in fp32 so the
// numpy restatement reproduces it bit for bit
__device__ __forceinline__ float goal_value(float lo, float hi, uint32_t u) {
    return __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), rb::u01(u)));
}
struct GoalBox { float lo[32]; float hi[32]; };
__global__ void sample_goals_kernel(float *goal, uint32_t *count, const uint8_t *mask, GoalBox box,
                                    int n_q, long n, uint64_t seed, uint64_t env0, int rows) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    const uint32_t draw = count[i];
    count[i] = draw + 1u;
    for (int b = 0; 4 * b < n_q; ++b) {
        const rb::Philox4 r = rb::philox_draw(seed, env0 + uint64_t(i), draw, rb::STREAM_GOALS, uint32_t(b));
        for (int e = 0; e < 4 && 4 * b + e < n_q; ++e) {
            const int j = 4 * b + e;
            const float g = goal_value(box.lo[j], box.hi[j], r.v[e]);
            if (rows) goal[i * n_q + j] = g; else goal[j * n + i] = g;
        }
    }
}

inline unsigned blocks_for(long n, int block) { return unsigned((n + block - 1) / block); }

}  // namespace

// --------------------------------------------------------------------- handle
struct rb_sim {
    int device = 0;
    long n = 0;
    int n_q = 0, n_t = 0;
    int integrator = 0, nsub = 1, kernel = RB_KERNEL_ENV_PER_LANE;
    double step_size = 0.1;
    uint64_t seed = 0;
    int64_t env0 = 0;
    Const8 c8;
    GoalBox box;
    hipStream_t own_stream = nullptr, stream = nullptr;
    float *d_q = nullptr, *d_qd = nullptr;
    uint32_t *d_feas = nullptr, *d_goal_count = nullptr;
    // host I/O staging
    float *d_rows = nullptr;   // [n][max(n_q, n_t)]
    uint8_t *d_u8 = nullptr;   // [n]
    // rollout graph cache
    struct GraphKey {
        const float *ring_ptr; int ring; int chunk; float scale; int kernel;
        bool operator<(const GraphKey &o) const {
            return std::memcmp(this, &o, sizeof(GraphKey)) < 0;
        }
    };
    std::map<GraphKey, hipGraphExec_t> graphs;
};

namespace {

int launch_step(rb_sim *s, const float *d_act, float act_scale) {
    const long n = s->n;
    if (n <= 65536) {   // few waves: spread one wave per workgroup over the CUs
        constexpr int B = 64;
        if (s->integrator == RB_EULER)
            hipLaunchKernelGGL((msj_step_env_per_lane<0, B>), dim3(blocks_for(n, B)), dim3(B), 0, s->stream,
                               s->c8, s->d_q, s->d_qd, s->d_feas, d_act, act_scale, n);
        else
            hipLaunchKernelGGL((msj_step_env_per_lane<1, B>), dim3(blocks_for(n, B)), dim3(B), 0, s->stream,
                               s->c8, s->d_q, s->d_qd, s->d_feas, d_act, act_scale, n);
    } else {
        constexpr int B = 256;
        if (s->integrator == RB_EULER)
            hipLaunchKernelGGL((msj_step_env_per_lane<0, B>), dim3(blocks_for(n, B)), dim3(B), 0, s->stream,
                               s->c8, s->d_q, s->d_qd, s->d_feas, d_act, act_scale, n);
        else
            hipLaunchKernelGGL((msj_step_env_per_lane<1, B>), dim3(blocks_for(n, B)), dim3(B), 0, s->stream,
                               s->c8, s->d_q, s->d_qd, s->d_feas, d_act, act_scale, n);
    }
    RB_HIP(hipGetLastError());
    return RB_OK;
}

int check(const rb_sim *s) {
    if (!s) return fail(RB_EINVAL, "null simulation handle");
    return RB_OK;
}

int read_state_host(rb_sim *s, float *q, float *qd, uint8_t *feasible) {
    const long n = s->n;
    const unsigned g = blocks_for(n, 256);
    if (q) {
        hipLaunchKernelGGL(pack_rows_kernel, dim3(g), dim3(256), 0, s->stream, s->d_q, s->d_rows, s->n_q, n);
        RB_HIP(hipMemcpyAsync(q, s->d_rows, sizeof(float) * n * s->n_q, hipMemcpyDeviceToHost, s->stream));
        RB_HIP(hipStreamSynchronize(s->stream));
    }
    if (qd) {
        hipLaunchKernelGGL(pack_rows_kernel, dim3(g), dim3(256), 0, s->stream, s->d_qd, s->d_rows, s->n_q, n);
        RB_HIP(hipMemcpyAsync(qd, s->d_rows, sizeof(float) * n * s->n_q, hipMemcpyDeviceToHost, s->stream));
        RB_HIP(hipStreamSynchronize(s->stream));
    }
    if (feasible) {
        hipLaunchKernelGGL(feas_to_u8_kernel, dim3(g), dim3(256), 0, s->stream, s->d_feas, s->d_u8, n);
        RB_HIP(hipMemcpyAsync(feasible, s->d_u8, size_t(n), hipMemcpyDeviceToHost, s->stream));
        RB_HIP(hipStreamSynchronize(s->stream));
    }
    RB_HIP(hipGetLastError());
    return RB_OK;
}

}  // namespace

// ----------------------------------------------------------------------- C ABI
extern "C" {

const char *rb_last_error(void) { return g_err.c_str(); }
int rb_abi_version(void) { return RB_ABI_VERSION; }

int rb_device_count(int *count) {
    if (!count) return fail(RB_EINVAL, "count is null");
    *count = 0;
    RB_HIP(hipGetDeviceCount(count));
    return RB_OK;
}

int rb_create(const rb_robot_desc *robot, int64_t n_envs, int integrator, double step_size,
              int n_substeps, int device, uint64_t seed, int64_t env_id_offset, rb_sim **out) {
    if (!robot || !out) return fail(RB_EINVAL, "robot/out is null");
    *out = nullptr;
    if (n_envs < 1) return fail(RB_EINVAL, "n_envs must be >= 1");
    if (integrator != RB_EULER && integrator != RB_RK4) return fail(RB_EINVAL, "unknown integrator");
    if (!(step_size > 0.0) || n_substeps < 1) return fail(RB_EINVAL, "step_size must be > 0 and n_substeps >= 1");
    if (robot->n_q < 1 || robot->n_q > 32 || robot->n_t < 1) return fail(RB_EINVAL, "n_q must be in [1, 32], n_t >= 1");
    if (env_id_offset < 0) return fail(RB_EINVAL, "env_id_offset must be >= 0");

    rb_sim *s = new (std::nothrow) rb_sim();
    if (!s) return fail(RB_ENOMEM, "out of host memory");
    std::string why;
    int rc = rb::msj_build<float, NT8>(robot, step_size, n_substeps, &s->c8, why);
    if (rc != RB_OK) {
        delete s;
        return fail(rc, "no HIP kernel for this robot structure: " + why +
                            " (built so far: 3-DOF ball joint, one body, 8 tendons)");
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1) {
        delete s;
        return fail(RB_EHIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    }
    if (device < 0 || device >= count) { delete s; return fail(RB_EINVAL, "device index out of range"); }
    s->device = device; s->n = n_envs; s->n_q = robot->n_q; s->n_t = robot->n_t;
    s->integrator = integrator; s->nsub = n_substeps; s->step_size = step_size;
    s->seed = seed; s->env0 = env_id_offset;
    for (int j = 0; j < s->n_q; ++j) { s->box.lo[j] = float(robot->q_lo[j]); s->box.hi[j] = float(robot->q_hi[j]); }

#define RB_TRY(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            std::string m_ = std::string(#call) + ": " + hipGetErrorString(e_);           \
            rb_destroy(s);                                                                \
            return fail(e_ == hipErrorOutOfMemory ? RB_ENOMEM : RB_EHIP, m_);             \
        }                                                                                 \
    } while (0)
    RB_TRY(hipSetDevice(device));
    RB_TRY(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    s->stream = s->own_stream;
    const size_t plane = sizeof(float) * size_t(n_envs);
    RB_TRY(hipMalloc(&s->d_q, plane * s->n_q));
    RB_TRY(hipMalloc(&s->d_qd, plane * s->n_q));
    RB_TRY(hipMalloc(&s->d_feas, sizeof(uint32_t) * size_t(n_envs)));
    RB_TRY(hipMalloc(&s->d_goal_count, sizeof(uint32_t) * size_t(n_envs)));
    const int width = s->n_q > s->n_t ? s->n_q : s->n_t;
    RB_TRY(hipMalloc(&s->d_rows, plane * width));
    RB_TRY(hipMalloc(&s->d_u8, size_t(n_envs)));
    RB_TRY(hipMemsetAsync(s->d_goal_count, 0, sizeof(uint32_t) * size_t(n_envs), s->stream));
#undef RB_TRY
    *out = s;
    rc = rb_reset(s, nullptr);
    if (rc != RB_OK) { std::string m = g_err; rb_destroy(s); *out = nullptr; return fail(rc, m); }
    return RB_OK;
}

void rb_destroy(rb_sim *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->own_stream) (void)hipStreamSynchronize(s->own_stream);
    for (auto &kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
    (void)hipFree(s->d_q); (void)hipFree(s->d_qd); (void)hipFree(s->d_feas);
    (void)hipFree(s->d_goal_count); (void)hipFree(s->d_rows); (void)hipFree(s->d_u8);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
    delete s;
}

int rb_info(const rb_sim *s, rb_sim_info *info) {
    if (check(s) || !info) return fail(RB_EINVAL, "null argument");
    info->n_envs = s->n; info->n_q = s->n_q; info->n_t = s->n_t;
    info->integrator = s->integrator; info->n_substeps = s->nsub; info->kernel = s->kernel;
    info->device = s->device; info->step_size = s->step_size;
    info->bytes_per_env_step = 4 * (4 * int64_t(s->n_q) + s->n_t + 1);
    info->env_id_offset = s->env0;
    return RB_OK;
}

int rb_select_kernel(rb_sim *s, int kernel) {
    if (check(s)) return RB_EINVAL;
    if (kernel == RB_KERNEL_AUTO || kernel == RB_KERNEL_ENV_PER_LANE) { s->kernel = RB_KERNEL_ENV_PER_LANE; return RB_OK; }
    return fail(RB_EUNSUPPORTED, "kernel variant not built");
}

int rb_set_stream(rb_sim *s, void *hip_stream) {
    if (check(s)) return RB_EINVAL;
    RB_HIP(hipSetDevice(s->device));
    RB_HIP(hipStreamSynchronize(s->stream));
    s->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->own_stream;
    return RB_OK;
}

int rb_synchronize(rb_sim *s) {
    if (check(s)) return RB_EINVAL;
    RB_HIP(hipStreamSynchronize(s->stream));
    return RB_OK;
}

int rb_reset(rb_sim *s, const uint8_t *mask) {
    if (check(s)) return RB_EINVAL;
    RB_HIP(hipSetDevice(s->device));
    const uint8_t *d_mask = nullptr;
    if (mask) {
        RB_HIP(hipMemcpyAsync(s->d_u8, mask, size_t(s->n), hipMemcpyHostToDevice, s->stream));
        d_mask = s->d_u8;
    }
    hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(s->n, 256)), dim3(256), 0, s->stream,
                       s->d_q, s->d_qd, s->d_feas, d_mask, s->n_q, s->n);
    RB_HIP(hipGetLastError());
    RB_HIP(hipStreamSynchronize(s->stream));
    return RB_OK;
}

int rb_set_state(rb_sim *s, const float *q, const float *qd, const uint8_t *feasible) {
    if (check(s) || !q || !qd) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    const long n = s->n;
    const unsigned g = blocks_for(n, 256);
    RB_HIP(hipMemcpyAsync(s->d_rows, q, sizeof(float) * n * s->n_q, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(g), dim3(256), 0, s->stream, s->d_rows, s->d_q, s->n_q, n);
    RB_HIP(hipStreamSynchronize(s->stream));
    RB_HIP(hipMemcpyAsync(s->d_rows, qd, sizeof(float) * n * s->n_q, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(g), dim3(256), 0, s->stream, s->d_rows, s->d_qd, s->n_q, n);
    const uint8_t *d_f = nullptr;
    if (feasible) {
        RB_HIP(hipMemcpyAsync(s->d_u8, feasible, size_t(n), hipMemcpyHostToDevice, s->stream));
        d_f = s->d_u8;
    }
    hipLaunchKernelGGL(feas_from_u8_kernel, dim3(g), dim3(256), 0, s->stream, d_f, s->d_feas, n);
    RB_HIP(hipGetLastError());
    RB_HIP(hipStreamSynchronize(s->stream));
    return RB_OK;
}

int rb_read_state(rb_sim *s, float *q, float *qd, uint8_t *feasible) {
    if (check(s)) return RB_EINVAL;
    RB_HIP(hipSetDevice(s->device));
    return read_state_host(s, q, qd, feasible);
}

int rb_step(rb_sim *s, const float *act, float act_scale, float *q, float *qd, uint8_t *feasible) {
    if (check(s) || !act) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    RB_HIP(hipMemcpyAsync(s->d_rows, act, sizeof(float) * s->n * s->n_t, hipMemcpyHostToDevice, s->stream));
    int rc = launch_step(s, s->d_rows, act_scale);
    if (rc) return rc;
    RB_HIP(hipStreamSynchronize(s->stream));   // d_rows is reused by the read-back
    return read_state_host(s, q, qd, feasible);
}

int rb_sample_goals(rb_sim *s, const uint8_t *mask, float *goal_q) {
    if (check(s) || !goal_q) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    const uint8_t *d_mask = nullptr;
    if (mask) {
        RB_HIP(hipMemcpyAsync(s->d_u8, mask, size_t(s->n), hipMemcpyHostToDevice, s->stream));
        d_mask = s->d_u8;
    }
    if (mask) RB_HIP(hipMemsetAsync(s->d_rows, 0, sizeof(float) * s->n * s->n_q, s->stream));
    hipLaunchKernelGGL(sample_goals_kernel, dim3(blocks_for(s->n, 256)), dim3(256), 0, s->stream,
                       s->d_rows, s->d_goal_count, d_mask, s->box, s->n_q, s->n, s->seed, uint64_t(s->env0), 1);
    RB_HIP(hipGetLastError());
    RB_HIP(hipMemcpyAsync(goal_q, s->d_rows, sizeof(float) * s->n * s->n_q, hipMemcpyDeviceToHost, s->stream));
    RB_HIP(hipStreamSynchronize(s->stream));
    return RB_OK;
}

int rb_state_ptrs(rb_sim *s, float **d_q, float **d_qd, uint32_t **d_feasible) {
    if (check(s)) return RB_EINVAL;
    if (d_q) *d_q = s->d_q;
    if (d_qd) *d_qd = s->d_qd;
    if (d_feasible) *d_feasible = s->d_feas;
    return RB_OK;
}

int rb_step_dev(rb_sim *s, const float *d_act, float act_scale) {
    if (check(s) || !d_act) return fail(RB_EINVAL, "null argument");
    if (reinterpret_cast<uintptr_t>(d_act) % 16) return fail(RB_EINVAL, "action slab must be 16-byte aligned");
    return launch_step(s, d_act, act_scale);
}

int rb_rollout_dev(rb_sim *s, const float *d_ring, int ring, int n_steps, float act_scale, int use_graph) {
    if (check(s) || !d_ring) return fail(RB_EINVAL, "null argument");
    if (ring < 1 || n_steps < 0) return fail(RB_EINVAL, "ring must be >= 1 and n_steps >= 0");
    if (reinterpret_cast<uintptr_t>(d_ring) % 16) return fail(RB_EINVAL, "action ring must be 16-byte aligned");
    const size_t slab = size_t(s->n) * s->n_t;
    int t = 0;
    if (use_graph) {
        int chunk = ring;
        while (chunk < 32) chunk += ring;
        if (n_steps >= chunk) {
            rb_sim::GraphKey key;
            std::memset(&key, 0, sizeof(key));
            key.ring_ptr = d_ring; key.ring = ring; key.chunk = chunk; key.scale = act_scale; key.kernel = s->kernel;
            auto it = s->graphs.find(key);
            if (it == s->graphs.end()) {
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                RB_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
                int rc = RB_OK;
                for (int k = 0; k < chunk && rc == RB_OK; ++k) rc = launch_step(s, d_ring + size_t(k % ring) * slab, act_scale);
                hipError_t e = hipStreamEndCapture(s->stream, &graph);
                if (rc != RB_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                if (e != hipSuccess) return fail(RB_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
                e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (e != hipSuccess) return fail(RB_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
                it = s->graphs.emplace(key, exec).first;
            }
            for (; t + chunk <= n_steps; t += chunk) RB_HIP(hipGraphLaunch(it->second, s->stream));
        }
    }
    for (; t < n_steps; ++t) {
        int rc = launch_step(s, d_ring + size_t(t % ring) * slab, act_scale);
        if (rc) return rc;
    }
    return RB_OK;
}

int rb_fill_actions_dev(rb_sim *s, float *d_act, uint32_t step) {
    if (check(s) || !d_act) return fail(RB_EINVAL, "null argument");
    hipLaunchKernelGGL(fill_actions_kernel, dim3(blocks_for(s->n, 256)), dim3(256), 0, s->stream,
                       d_act, s->n_t, s->n, s->seed, uint64_t(s->env0), step);
    RB_HIP(hipGetLastError());
    return RB_OK;
}

int rb_sample_goals_dev(rb_sim *s, const uint8_t *d_mask, float *d_goal_q) {
    if (check(s) || !d_goal_q) return fail(RB_EINVAL, "null argument");
    hipLaunchKernelGGL(sample_goals_kernel, dim3(blocks_for(s->n, 256)), dim3(256), 0, s->stream,
                       d_goal_q, s->d_goal_count, d_mask, s->box, s->n_q, s->n, s->seed, uint64_t(s->env0), 0);
    RB_HIP(hipGetLastError());
    return RB_OK;
}

int rb_env_configure(rb_sim *, const rb_env_config *) { return fail(RB_EUNSUPPORTED, "fused env layer not built yet"); }
int rb_env_reset_dev(rb_sim *, float *) { return fail(RB_EUNSUPPORTED, "fused env layer not built yet"); }
int rb_env_step_dev(rb_sim *, const float *, float *, float *, uint32_t *) { return fail(RB_EUNSUPPORTED, "fused env layer not built yet"); }
int rb_env_stats(rb_sim *, double *, int) { return fail(RB_EUNSUPPORTED, "fused env layer not built yet"); }

int rb_malloc(rb_sim *s, int64_t bytes, void **d_ptr) {
    if (check(s) || !d_ptr || bytes < 0) return fail(RB_EINVAL, "bad argument");
    RB_HIP(hipSetDevice(s->device));
    RB_HIP(hipMalloc(d_ptr, size_t(bytes)));
    return RB_OK;
}
int rb_free(rb_sim *s, void *d_ptr) {
    if (check(s)) return RB_EINVAL;
    RB_HIP(hipFree(d_ptr));
    return RB_OK;
}
int rb_memcpy_h2d(rb_sim *s, void *d_dst, const void *h_src, int64_t bytes) {
    if (check(s) || !d_dst || !h_src) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipMemcpyAsync(d_dst, h_src, size_t(bytes), hipMemcpyHostToDevice, s->stream));
    RB_HIP(hipStreamSynchronize(s->stream));
    return RB_OK;
}
int rb_memcpy_d2h(rb_sim *s, void *h_dst, const void *d_src, int64_t bytes) {
    if (check(s) || !h_dst || !d_src) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipMemcpyAsync(h_dst, d_src, size_t(bytes), hipMemcpyDeviceToHost, s->stream));
    RB_HIP(hipStreamSynchronize(s->stream));
    return RB_OK;
}

}  // extern "C"
