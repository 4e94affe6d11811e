// env_common.hpp - definitions shared by the fused env-layer kernels of both
// robot classes (msj_kernels.hpp: ball-joint robots; tree_aba.hpp: joint trees).
#pragma once
#include "rtc_compat.hpp"
#include "philox.hpp"

namespace rbe {

// a * b + c with two roundings.  hipcc contracts a*b+c into one fma by default
// (-ffp-contract=fast) and HIP's __fmul_rn/__fadd_rn are plain operators, so
// the contraction has to be switched off for this statement.
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
    const float p = a * b;
    return p + c;
}
__device__ __forceinline__ float goal_value(float lo, float hi, uint32_t u) {
    return mul_then_add(hi - lo, rb::u01(u), lo);
}
// Per-env episode statistics, touched when an episode ends: sums of returns in fp64, counts as integers, each slot by its own env's
// lane only (so the sums are reproducible) - as atomics that RETURN NOTHING: `x[me] += v` is a load the lane has to wait for before it
// can store, i.e. a full memory latency on the serial tail of a kernel whose waves are alone on their SIMDs (fused env step of the
// upper body at 8 192 envs with the episodes spread out, as in training: 14.1 us per step against 12.2 in lock-step)
__device__ __forceinline__ void stat_add(double *p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void stat_add(uint32_t *p, uint32_t v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A value the optimiser cannot look into (no instruction; removable when unused): the set-point offsets of the env-per-lane kernels
// pass through it, so that their last multiplication is ROUNDED in the plain step and in the fused env step alike - otherwise the
// compiler may contract it into the first use inside the acceleration in one kernel and not in the other (different code around it),
// and the two differ in the last bit (seen with hiprtc-built kernels of a random robot: 6e-7 after two steps)
__device__ __forceinline__ float rounded_here(float x) { asm("" : "+v"(x)); return x; }
struct GoalBox { float lo[32]; float hi[32]; };

struct EnvParams {
    int vel_penalty, bonus, max_len, auto_reset;
    float penalty, bonus_val;
    float a_lo, a_hi, v_lo, v_hi;    // joint angle / velocity boxes
    float act_hi, slope;             // set-point box upper bound, (hi-lo)/(1-(-1)) in fp32
    float tol_a2, tol_v2;            // squared goal tolerances
    float a_scale, v_scale;          // 2 / (hi - lo) of the angle / velocity box
};

// The env-step kernels of the joint-tree forms take ONE kernel argument, this struct: what the accounting behind the step needs -
// eleven pointers, the goal box (64 values), the env parameters, seed and env id - is then read from the kernel-argument segment
// THERE, through a pointer the compiler cannot trace back across the step (late_args).  As 21 separate arguments everything was
// loaded at the kernel's entry and kept alive across the step: 212 / 226 scalar registers of the split kernel (43 / 71 of the
// one-wave kernel) went into vector-register lanes and back, in every wave, ~430 vector-instruction slots of a wave that gets one
// per 5.5 cycles.
struct TreeEnvArgs {
    EnvParams ep;
    GoalBox box;
    float *q, *qd;
    uint32_t *feas;
    float *goal;
    uint32_t *step_num;
    float *ep_ret;
    uint32_t *goal_count;
    const float *act;
    float *obs, *reward;
    uint32_t *done;
    double *ep_sum;
    uint32_t *ep_cnt, *infeas_n;
    float h;
    int nsub;
    long n;                          // the envs of this launch
    uint64_t seed, env_id0;
    long stat_stride;                // the batch's env count: stride of the statistics planes ep_sum[2][.], ep_cnt[3][.]
};
typedef const __attribute__((address_space(4))) TreeEnvArgs *tree_env_kernarg_ptr;
__device__ __forceinline__ tree_env_kernarg_ptr late_args() {
    tree_env_kernarg_ptr p = (tree_env_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                      // (loads through p stay behind this point)
    return p;
}
__device__ __forceinline__ EnvParams late_env_params(tree_env_kernarg_ptr p) {      // (field by field: the source lives in the constant address space)
    EnvParams ep;
    ep.vel_penalty = p->ep.vel_penalty; ep.bonus = p->ep.bonus; ep.max_len = p->ep.max_len; ep.auto_reset = p->ep.auto_reset;
    ep.penalty = p->ep.penalty; ep.bonus_val = p->ep.bonus_val;
    ep.a_lo = p->ep.a_lo; ep.a_hi = p->ep.a_hi; ep.v_lo = p->ep.v_lo; ep.v_hi = p->ep.v_hi;
    ep.act_hi = p->ep.act_hi; ep.slope = p->ep.slope; ep.tol_a2 = p->ep.tol_a2; ep.tol_v2 = p->ep.tol_v2;
    ep.a_scale = p->ep.a_scale; ep.v_scale = p->ep.v_scale;
    return ep;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}


// reward / done of RoboyEnv.step (roboy_env.py:92-112,125-134) from the squared
// joint-space distances; fp32.  The normalisation (2v - hi - lo)/(hi - lo)
// (roboy_robot.py:93-95) is affine, so the distance of two normalised vectors is
// 2/(hi - lo) times the plain distance; compares use squared distances.
__device__ __forceinline__ float env_reward(const EnvParams &e, float dq2, float dv2, bool feasible, bool &reached) {
    const float LOG2E = 1.4426950408889634f;
    float r = -__builtin_amdgcn_exp2f(LOG2E * e.a_scale * __builtin_amdgcn_sqrtf(dq2));
    if (e.vel_penalty)
        r = (e.v_scale * __builtin_amdgcn_sqrtf(dv2) + 1.0f) * (r - __builtin_amdgcn_exp2f(LOG2E * r));
    if (!feasible) r -= fabsf(e.penalty);
    reached = (dq2 < e.tol_a2) && (dv2 < e.tol_v2);
    if (reached && e.bonus) r += e.bonus_val;
    return r;
}

}  // namespace rbe
