// msj_kernels.hpp - the env-per-lane kernels of the ball-joint class (MsjRobot): physics step, open-loop
// fused rollout, fused env layer.  In a header of their own because they are compiled twice: by hipcc into
// libroboy_sim.so (instances on kernarg constants and on MsjRobot's baked table, msj_baked.hpp) and, at run
// time, by hiprtc for any other 8-tendon ball-joint robot with THAT robot's constants as literals (msj_jit.hpp:
// RB_JIT_TABLE is then defined in front of this header).  DESIGN.md §5.
#pragma once
#include "rtc_compat.hpp"
#include "msj_math.hpp"
#include "philox.hpp"
#include "env_common.hpp"

#if defined(RB_JIT_TABLE)
namespace rbk { __device__ constexpr rb::MsjConst<float, 8> BAKED = RB_JIT_TABLE; }
#else
#include "msj_baked.hpp"
#endif

namespace rbk {

using rbe::EnvParams;
using rbe::GoalBox;
using rbe::goal_value;
using rbe::mul_then_add;

constexpr int NT8 = 8;
using Const8 = rb::MsjConst<float, NT8>;
// ball-joint robots with another tendon count: kernels instantiated for up to NTX
// tendons, the count itself (c.nt) read at run time (UNROLL = 0, rolled loop)
constexpr int NTX = 16;
using ConstX = rb::MsjConst<float, NTX>;


// robot constants of a kernel instance: the kernarg copy, or (BK) the compile-time table (MsjRobot's, or the
// handle's own when hiprtc compiles this header), which the compiler folds into the instruction stream
template <bool BK, typename CONST>
__device__ __forceinline__ const CONST &robot_consts(const CONST &kernarg) {
    if constexpr (BK) return rbk::BAKED; else return kernarg;
}

// Set-points of one env staged as an LDS column: the rolled tendon loop reads sp(k) with one ds_read_b32
// instead of selecting among 8 registers with a runtime index (7 v_cndmask + 14 SALU per trip before).
// [NT8][BLOCK] floats, lane-contiguous rows: conflict-free.
struct SpLds {
    const float *col;   // &lds[0][threadIdx.x]
    int stride;         // BLOCK
    __device__ __forceinline__ float operator()(int k) const { return col[k * stride]; }
};

// set-point -> activation offset, per tendon: act_scale * ksg_k, multiplied out on the host once per launch
// (on the device the product of two kernarg scalars costs a v_mov and a v_mul per lane and tendon)
struct Scale8 { float v[NT8]; };

// One env per lane.  Loads: q, qd planes (dword per lane, 256 B contiguous per wave and plane) and the env's
// 32-byte action record (two dwordx4).  Stores: q', qd' planes and the feasibility word.  84 algorithmic
// bytes per env step.
template <int INTEG, int BLOCK, int UNROLL, bool BK = false>
__global__ void __launch_bounds__(BLOCK)
msj_step_env_per_lane(const Const8 c_arg, float *__restrict__ q, float *__restrict__ qd,
                      uint32_t *__restrict__ feas, const float *__restrict__ act, const Scale8 us, long n, long cnt) {
    // n: the handle's envs = the stride of the state planes; cnt: the envs of THIS launch - all of them, or one of the two
    // halves that rb_rollout_dev steps as two independent chains of launches (roboy_sim.hip: chains; the pointers then
    // point at the half's first env)
    const Const8 &c = robot_consts<BK>(c_arg);
    const long i = long(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= cnt) return;
    float qq[3], vv[3], sp[NT8];
    const float4 a0 = reinterpret_cast<const float4 *>(act)[2 * i];
    const float4 a1 = reinterpret_cast<const float4 *>(act)[2 * i + 1];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + i]; vv[j] = qd[j * n + i]; }
    // activation offsets u_k = (act_scale * ksg_k) * action_k: what the model's tendon loop consumes
    const float a[NT8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int k = 0; k < NT8; ++k) sp[k] = a[k] * us.v[k];
    bool ok;
    if (UNROLL >= NT8) {
        ok = rb::MsjModel<float, NT8>::template step_sp<INTEG, UNROLL>(c, qq, vv, rb::SpArray<float, NT8>{sp});
    } else {
        __shared__ float lds_sp[NT8][BLOCK];
#pragma unroll
        for (int k = 0; k < NT8; ++k) lds_sp[k][threadIdx.x] = sp[k];
        // each lane reads back only what it wrote: no barrier needed
        ok = rb::MsjModel<float, NT8>::template step_sp<INTEG, UNROLL>(c, qq, vv, SpLds{&lds_sp[0][threadIdx.x], BLOCK});
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { q[j * n + i] = qq[j]; qd[j * n + i] = vv[j]; }
    feas[i] = ok ? 1u : 0u;
}




// Open-loop rollout fused into one launch (rb_rollout_fused_dev): the env-per-lane
// step applied n_steps times with the state held in registers; per step only the
// env's 32-byte action record is read.  Instantiated with the same BLOCK/UNROLL
// (and the same set-point source) as msj_step_env_per_lane uses for the batch
// size, so the arithmetic and hence the result is bit-identical to n_steps single steps.
template <int INTEG, int BLOCK, int UNROLL, bool BK = false>
__global__ void __launch_bounds__(BLOCK)
msj_rollout_fused(const Const8 c_arg, float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas,
                  const float *__restrict__ act_ring, int ring, int n_steps, const Scale8 us, long n) {
    const Const8 &c = robot_consts<BK>(c_arg);
    const long i = long(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    float qq[3], vv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + i]; vv[j] = qd[j * n + i]; }
    bool ok = true;
    const float4 *rec = reinterpret_cast<const float4 *>(act_ring) + 2 * i;
    float4 a0 = rec[0], a1 = rec[1];
    __shared__ float lds_sp[UNROLL >= NT8 ? 1 : NT8][BLOCK];
    int slab = 0;
    for (int t = 0; t < n_steps; ++t) {
        const float a[NT8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        float sp[NT8];
#pragma unroll
        for (int k = 0; k < NT8; ++k) sp[k] = a[k] * us.v[k];
        if (t + 1 < n_steps) {   // next step's action: in flight under this step's arithmetic
            slab = slab + 1 == ring ? 0 : slab + 1;
            const float4 *nx = rec + long(slab) * 2 * n;
            a0 = nx[0]; a1 = nx[1];
        }
        if (UNROLL >= NT8) {
            ok = rb::MsjModel<float, NT8>::template step_sp<INTEG, UNROLL>(c, qq, vv, rb::SpArray<float, NT8>{sp});
        } else {
#pragma unroll
            for (int k = 0; k < NT8; ++k) lds_sp[k][threadIdx.x] = sp[k];   // lane-private column: no barrier
            ok = rb::MsjModel<float, NT8>::template step_sp<INTEG, UNROLL>(c, qq, vv, SpLds{&lds_sp[0][threadIdx.x], BLOCK});
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { q[j * n + i] = qq[j]; qd[j * n + i] = vv[j]; }
    feas[i] = ok ? 1u : 0u;    // feasibility of the last step, as after n_steps single steps
}



// ------------------------------------------------------------ fused env layer
// RoboyEnv.step for a batch (reference gym_roboy/envs/roboy_env.py:51-70):
// rescale action -> physics step -> obs / reward / done -> goal resampling,
// plus the reset the reference's SubprocVecEnv workers apply on done
// (train_parallel.py:29) when auto_reset is set.  DESIGN.md §6.
// obs/goal helper: draw goal number `draw` of env `gid`
__device__ __forceinline__ void draw_goal3(const GoalBox &box, uint64_t seed, uint64_t gid, uint32_t draw, float g[3]) {
    const rb::Philox4 r = rb::philox_draw(seed, gid, draw, rb::STREAM_GOALS, 0u);
#pragma unroll
    for (int j = 0; j < 3; ++j) g[j] = goal_value(box.lo[j], box.hi[j], r.v[j]);
}

// UNROLL = 0: run-time tendon count (ConstX, c.nt tendons, action rows of c.nt floats)
template <int INTEG, int BLOCK, int UNROLL, typename CONST = Const8, bool BK = false>
__global__ void __launch_bounds__(BLOCK)
msj_env_step_kernel(const CONST c_arg, const EnvParams e, const GoalBox box,
                    float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas,
                    float *__restrict__ goal, uint32_t *__restrict__ step_num, float *__restrict__ ep_ret,
                    uint32_t *__restrict__ goal_count, const float *__restrict__ act,
                    float *__restrict__ obs, float *__restrict__ reward, uint32_t *__restrict__ done,
                    double *__restrict__ ep_sum, uint32_t *__restrict__ ep_cnt, uint32_t *__restrict__ infeas_n,
                    long n, uint64_t seed, uint64_t env0) {
    const CONST &c = robot_consts<BK>(c_arg);
    const long i = long(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    float qq[3], vv[3], gg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + i]; vv[j] = qd[j * n + i]; gg[j] = goal[j * n + i]; }
    // the reference asserts the action lies in [-1,1] (roboy_env.py:52); a batched kernel
    // cannot raise, so it clamps.  Then slope * (x - in_high) + out_high, each op rounded
    // (roboy_env.py:157-158)
    auto rescale = [&](float a) { return mul_then_add(e.slope, fminf(fmaxf(a, -1.0f), 1.0f) - 1.0f, e.act_hi); };
    bool ok;
    if constexpr (UNROLL == 0) {
        __shared__ float lds_sp[NTX][BLOCK];
        const int nt = c.nt;
        const float *row = act + i * nt;
        for (int k = 0; k < nt; ++k) lds_sp[k][threadIdx.x] = rescale(row[k]) * c.ten[k].ksg;
        ok = rb::MsjModel<float, NTX>::template step_sp<INTEG, 0>(c, qq, vv, SpLds{&lds_sp[0][threadIdx.x], BLOCK});
    } else {
    float sp[NT8];
    const float4 a0 = reinterpret_cast<const float4 *>(act)[2 * i];
    const float4 a1 = reinterpret_cast<const float4 *>(act)[2 * i + 1];
    const float a[NT8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int k = 0; k < NT8; ++k) sp[k] = rescale(a[k]) * c.ten[k].ksg;   // set-point -> activation offset
    if (UNROLL >= NT8) {
        ok = rb::MsjModel<float, NT8>::template step_sp<INTEG, UNROLL>(c, qq, vv, rb::SpArray<float, NT8>{sp});
    } else {
        // rolled tendon loop: the set-points are indexed at run time, keep them as an LDS column
        // (as msj_step_env_per_lane does); each lane reads back only what it wrote
        __shared__ float lds_sp[NT8][BLOCK];
#pragma unroll
        for (int k = 0; k < NT8; ++k) lds_sp[k][threadIdx.x] = sp[k];
        ok = rb::MsjModel<float, NT8>::template step_sp<INTEG, UNROLL>(c, qq, vv, SpLds{&lds_sp[0][threadIdx.x], BLOCK});
    }
    }
    uint32_t sn = step_num[i] + 1u;

    // reward (roboy_env.py:92-112), fp32.  The normalisation (2v - hi - lo)/(hi - lo)
    // (roboy_robot.py:93-95) is affine, so a difference of two normalised values is
    // 2 (v1 - v2)/(hi - lo): one multiply by a host-computed scale instead of two
    // divisions per joint; compares use squared distances (no sqrt); exp is exp2.
    float dq2 = 0.0f, dv2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float dq = qq[j] - gg[j];
        dq2 += dq * dq;
        dv2 += vv[j] * vv[j];
    }
    bool reached;
    float r = rbe::env_reward(e, dq2, dv2, ok, reached);
    const bool dn = reached || (sn > uint32_t(e.max_len));

    float o[9] = {qq[0], qq[1], qq[2], vv[0], vv[1], vv[2], gg[0], gg[1], gg[2]};
    float ret = ep_ret[i] + r;
    uint32_t fz = ok ? 1u : 0u;
    if (!ok) infeas_n[i] += 1u;
    if (dn) {
        // per-env episode accumulators, touched only when an episode ends;
        // rb_env_stats reduces them (no atomics in the step kernel).  Sums of returns
        // in fp64 (a return carries the +1000 bonus, its square overflows fp32's 24 bits
        // after a few episodes), counts as integers (fp32 counters stop at 2^24)
        ep_sum[i] += double(ret); ep_sum[n + i] += double(ret) * double(ret);
        ep_cnt[i] += 1u; ep_cnt[n + i] += sn - 1u; ep_cnt[2 * n + i] += reached ? 1u : 0u;
        const uint64_t gid = env0 + uint64_t(i);
        uint32_t draw = goal_count[i];
        draw_goal3(box, seed, gid, draw++, gg);          // RoboyEnv.step: _set_new_goal (:67-68)
        if (e.auto_reset) {                              // VecEnv worker: env.reset() (:82-87)
            draw_goal3(box, seed, gid, draw++, gg);
#pragma unroll
            for (int j = 0; j < 3; ++j) { qq[j] = 0.0f; vv[j] = 0.0f; o[j] = 0.0f; o[3 + j] = 0.0f; o[6 + j] = gg[j]; }
            sn = 1u; fz = 1u;
        }
        ret = 0.0f;
        goal_count[i] = draw;
#pragma unroll
        for (int j = 0; j < 3; ++j) goal[j * n + i] = gg[j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { q[j * n + i] = qq[j]; qd[j * n + i] = vv[j]; }
    feas[i] = fz; step_num[i] = sn; ep_ret[i] = ret;
    // 36-byte observation record: two 16-byte stores and one dword (dword-aligned
    // vector stores are legal for global memory) instead of nine strided dwords
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    float *orow = obs + i * 9;
    *reinterpret_cast<f4u *>(orow) = f4u{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f4u *>(orow + 4) = f4u{o[4], o[5], o[6], o[7]};
    orow[8] = o[8];
    reward[i] = r; done[i] = dn ? 1u : 0u;
}


}  // namespace rbk
