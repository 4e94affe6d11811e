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
constexpr int RS = rb::MsjModel<float, 8>::RS;      // UNROLL = RS: the "rolled stages" form (msj_math.hpp: step_rs)
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




// a workgroup's slice of an array as a buffer resource (base and size wave-uniform: scalar registers)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const void *base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

// One env per lane, second form (round 4): the integrator's four RK4 stages as a ROLLED loop (integrate_acc), all eight
// tendons written out inside it one after the other, loads and stores through workgroup buffer resources.  With the stage
// loop rolled the body is a quarter of the code and ~60 registers, so the tendon loop can be written out WITHOUT the register
// bill that made the fully unrolled RK4 body of the first form lose (roboy_sim.hip: RB_BAKED_UNROLL_RK4) - and written out on
// baked constants every tendon constant is a literal instead of the scalar-register operand a rolled loop's table read gives:
// an instruction with a scalar-register source issues at 4.4 cycles against 2.8 at four waves per SIMD
// (profiles/r3_a/issue_forms_probe.log), and they were 30 % of the first form's instructions.
template <int INTEG, int BLOCK, bool BK = false>
__global__ void __launch_bounds__(BLOCK)
msj_step_env_per_lane_rs(const Const8 c_arg, float *__restrict__ q, float *__restrict__ qd,
                         uint32_t *__restrict__ feas, const float *__restrict__ act, const Scale8 us, long n, long cnt) {
    const Const8 &c = robot_consts<BK>(c_arg);
    const long env0 = long(blockIdx.x) * BLOCK;
    const long left = cnt - env0;
    const int live = int(left < BLOCK ? left : BLOCK);
    const int le = int(threadIdx.x);
    if (le >= live) return;
    const int off = le * 4;
    float qq[3], vv[3], u[NT8];
    const __amdgpu_buffer_rsrc_t ra = wg_rsrc(act + env0 * NT8, live * NT8 * 4);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 a0 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ra, off * NT8, 0, 0));
    const f4 a1 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ra, off * NT8 + 16, 0, 0));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        qq[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wg_rsrc(q + j * n + env0, live * 4), off, 0, 0));
        vv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wg_rsrc(qd + j * n + env0, live * 4), off, 0, 0));
    }
    const float a[NT8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int k = 0; k < NT8; ++k) u[k] = a[k] * us.v[k];
    const bool ok = rb::MsjModel<float, NT8>::template step_rs<INTEG>(c, qq, vv, u);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(qq[j]), wg_rsrc(q + j * n + env0, live * 4), off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv[j]), wg_rsrc(qd + j * n + env0, live * 4), off, 0, 0);
    }
    __builtin_amdgcn_raw_buffer_store_b32(ok ? 1u : 0u, wg_rsrc(feas + env0, live * 4), off, 0, 0);
}


// ------------------------------------------------------------ two lanes per env: mirror pairs
// The form between "one env per lane" and "eight lanes per env" (round 4): a pair of adjacent lanes shares an env,
// each evaluates HALF of the tendons and both carry the rigid-body solve and the integrator.  Twice the waves for the
// same batch (262 144 envs: 8 per SIMD instead of 4 - what the issue rate of this instruction stream needs), a
// per-wave instruction chain 0.64x as long (mid-size batches, where every SIMD holds one or two waves).
//
// What keeps the two lanes on ONE instruction stream with the robot's constants as literals is a mirror symmetry of the
// robot: MsjRobot's tendon k' = 7 - k is tendon k reflected in the x-z plane (A' = S A, B' = S B, S = diag(1,-1,1), same
// muscle), the body's inertia is principal-axis with the centre of mass on z, gravity along z, and the joint limits of
// the two axes the reflection turns round are symmetric.  A reflection maps solutions to solutions: the env seen in the
// mirror is the same robot in the state q^ = (-q0, q1, -q2) (rotations about an axis in the mirror plane change sense),
// and its tendons 0..3 are the real env's tendons 7..4.  So the odd lane of a pair simply steps the MIRRORED env:
// same code, same constants, its own four tendons - and once per acceleration the lanes swap their torque sums (one
// DPP quad_perm each way; a torque is a pseudovector: (x, y, z) -> (-x, y, -z) through this mirror), after which both
// hold the full torque of their own world.  No per-lane tables, no selects, no LDS, no barrier.  The x-z mirror (MIRROR = 0)
// and the y-z mirror (MIRROR = 1: S = diag(-1,1,1), q^ = (q0, -q1, -q2), torque (x, -y, -z)) are instantiated; the host
// looks for either in the handle's constants (roboy_sim.hip: find_mirror_pairs) and robots without one keep the other forms.
struct PairMap { int a[4], d[4]; };      // byte offsets in an action row: a[k] = the even lane's tendon k, a[k] + d[k] = its mirror image (odd lane's tendon k)
struct Scale4 { float v[4]; };           // act_scale * ksg of the even lane's four tendons (their images' are equal)

template <int CTRL>
__device__ __forceinline__ float rbk_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int MIRROR>
struct AccelMirrorHalf {
    const Const8 &c;
    const float *u;      // activation offsets of this lane's four tendons
    __device__ __forceinline__ void operator()(const float q[3], const float qd[3], float qdd[3]) const {
        using M = rb::MsjModel<float, NT8>;
        const M::Frame f = M::frame(q, qd);
        float tx, ty, tz;
        M::half_torque(c, f, u, tx, ty, tz);
        // the partner lane's sum (quad_perm [1,0,3,2]), seen through the mirror
        M::template mirror_combine<MIRROR>(tx, ty, tz, rbk_dpp<0xB1>(tx), rbk_dpp<0xB1>(ty), rbk_dpp<0xB1>(tz));
        M::rigid_body(c, f, qd, tx, ty, tz, qdd);
    }
};

// c_arg.ten[0..3]: the even lane's tendons (the host orders the table; the baked table is used when its own first four
// tendons are a mirror half).  Loads: both lanes of a pair read the env's six state words (same addresses: one request)
// and their own four action words; stores: the even lane's.  84 algorithmic bytes per env step as before.
template <int INTEG, int BLOCK, int MIRROR, bool BK = false>
__global__ void __launch_bounds__(BLOCK)
msj_step_mirror_pairs(const Const8 c_arg, const PairMap pm, float *__restrict__ q, float *__restrict__ qd,
                      uint32_t *__restrict__ feas, const float *__restrict__ act, const Scale4 us, long n, long cnt) {
    const Const8 &c = robot_consts<BK>(c_arg);
    // Every access is (workgroup's base, in scalar registers) + (a lane offset of a few hundred bytes) through a buffer
    // resource whose record count is the workgroup's live bytes: one address register for everything, nothing that has to
    // stay alive until the stores (seven 64-bit address pairs otherwise: 14 of the 64 registers), and no guard - a load past
    // the end returns 0, a store past the end is dropped.
    const long env0 = long(blockIdx.x) * (BLOCK / 2);
    const long left = cnt - env0;
    const int live = int(left < BLOCK / 2 ? left : BLOCK / 2);
    const int le = int(threadIdx.x >> 1);      // the pair's env within the workgroup
    if (le >= live) return;                    // whole pairs leave together
    const bool odd = (threadIdx.x & 1) != 0;
    const int off = le * 4, oddi = int(threadIdx.x & 1);
    float qq[3], vv[3], u[4];
    const __amdgpu_buffer_rsrc_t ra = wg_rsrc(act + env0 * NT8, live * NT8 * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, off * NT8 + pm.a[k] + oddi * pm.d[k], 0, 0)) * us.v[k];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        qq[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wg_rsrc(q + j * n + env0, live * 4), off, 0, 0));
        vv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wg_rsrc(qd + j * n + env0, live * 4), off, 0, 0));
    }
    // the odd lane's env is the mirror image: the two joints whose axes lie in the mirror plane change sign
    constexpr int F0 = MIRROR == 0 ? 0 : 1, F1 = 2;
    const uint32_t flip = odd ? 0x80000000u : 0u;
    qq[F0] = __uint_as_float(__float_as_uint(qq[F0]) ^ flip); vv[F0] = __uint_as_float(__float_as_uint(vv[F0]) ^ flip);
    qq[F1] = __uint_as_float(__float_as_uint(qq[F1]) ^ flip); vv[F1] = __uint_as_float(__float_as_uint(vv[F1]) ^ flip);
    const bool ok = rb::MsjModel<float, NT8>::template integrate_acc<INTEG>(c, qq, vv, AccelMirrorHalf<MIRROR>{c, u});
    // the even lane holds the env itself: its stores (the odd lane's offset is pushed past the end: dropped)
    const int soff = odd ? 0x7fffff00 : le * 4;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(qq[j]), wg_rsrc(q + j * n + env0, live * 4), soff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv[j]), wg_rsrc(qd + j * n + env0, live * 4), soff, 0, 0);
    }
    __builtin_amdgcn_raw_buffer_store_b32(ok ? 1u : 0u, wg_rsrc(feas + env0, live * 4), soff, 0, 0);
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
        if (UNROLL == RS) {
            ok = rb::MsjModel<float, NT8>::template step_rs<INTEG>(c, qq, vv, sp);
        } else if (UNROLL >= NT8) {
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
__device__ __forceinline__ void draw_goal3(const float (&lo)[3], const float (&hi)[3], uint64_t seed, uint64_t gid, uint32_t draw, float g[3]) {
    const rb::Philox4 r = rb::philox_draw(seed, gid, draw, rb::STREAM_GOALS, 0u);
#pragma unroll
    for (int j = 0; j < 3; ++j) g[j] = goal_value(lo[j], hi[j], r.v[j]);
}
__device__ __forceinline__ void draw_goal3(const GoalBox &box, uint64_t seed, uint64_t gid, uint32_t draw, float g[3]) {
    const float lo[3] = {box.lo[0], box.lo[1], box.lo[2]}, hi[3] = {box.hi[0], box.hi[1], box.hi[2]};
    draw_goal3(lo, hi, seed, gid, draw, g);
}

// Everything the fused env kernels of the ball-joint class take besides the robot's constants, as ONE kernel argument (the LAST
// one).  The step in front needs a handful of it (state, goal and action pointers, the rescale's two numbers, the counts); the
// accounting behind the step needs the rest.  Kernels whose robot constants come through the kernarg (BK = false: 64 scalar registers
// held across the step) read that rest LATE, through a pointer into the kernel-argument segment that the compiler cannot trace back
// across the step (late_env_args): loaded up front like any other argument, the ~40 dwords of it were kept alive in scalar registers
// beside the constants - 38-86 scalar spills into vector-register lanes and a private segment of 36-52 bytes per lane in every
// kernarg instance (rounds 4-5; a dispatch with a private segment has the runtime set up scratch for it).  The baked / hiprtc
// instances (BK = true) have the registers and read the argument directly.
struct MsjEnvArgs {
    EnvParams e;
    float box_lo[3], box_hi[3];      // the goal box of the three joints (rbe::GoalBox holds 32: a kernarg of 256 bytes for six values)
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
    long n, cnt;                     // n: the handle's envs = the stride of the state / goal / statistics planes; cnt: the envs of THIS launch
    uint64_t seed, env0;             // env0: global id of the launch's first env (a sub-range arrives on shifted pointers)
};
// byte offset of the MsjEnvArgs argument in the kernel-argument segment: behind `lead` bytes of leading arguments, at its own alignment
// (checked against the code objects' metadata by tests/test_code_objects.py)
__host__ __device__ constexpr int msj_env_args_offset(int lead) { return (lead + int(alignof(MsjEnvArgs)) - 1) / int(alignof(MsjEnvArgs)) * int(alignof(MsjEnvArgs)); }
typedef const __attribute__((address_space(4))) MsjEnvArgs *msj_env_kernarg_ptr;
__device__ __forceinline__ msj_env_kernarg_ptr late_env_args(int offset) {
    const __attribute__((address_space(4))) char *p = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                      // (loads through p stay behind this point)
    return (msj_env_kernarg_ptr)(p + offset);
}

// What RoboyEnv.step does around the simulator's answer for env i of this launch (qq, vv: the new state; gg: the env's goal; ok:
// feasible): reward, done, episode accounting, goal redraw / reset on done, and every row back.  Shared by the env-per-lane
// kernel and the two-lanes-per-env kernel (whose even lanes call it).
template <typename ARGS>      // const MsjEnvArgs * (the argument itself) or msj_env_kernarg_ptr (the late view of it)
__device__ __forceinline__ void env_account(ARGS a, long i, float (&qq)[3], float (&vv)[3], float (&gg)[3], bool ok) {
    EnvParams e;                  // (field by field: the source may live in the constant address space)
    e.vel_penalty = a->e.vel_penalty; e.bonus = a->e.bonus; e.max_len = a->e.max_len; e.auto_reset = a->e.auto_reset;
    e.penalty = a->e.penalty; e.bonus_val = a->e.bonus_val;
    e.a_lo = a->e.a_lo; e.a_hi = a->e.a_hi; e.v_lo = a->e.v_lo; e.v_hi = a->e.v_hi;
    e.act_hi = a->e.act_hi; e.slope = a->e.slope; e.tol_a2 = a->e.tol_a2; e.tol_v2 = a->e.tol_v2;
    e.a_scale = a->e.a_scale; e.v_scale = a->e.v_scale;
    const long n = a->n;
    float *__restrict__ q = a->q, *__restrict__ qd = a->qd, *__restrict__ goal = a->goal, *__restrict__ ep_ret = a->ep_ret;
    float *__restrict__ obs = a->obs, *__restrict__ reward = a->reward;
    uint32_t *__restrict__ feas = a->feas, *__restrict__ step_num = a->step_num, *__restrict__ goal_count = a->goal_count;
    uint32_t *__restrict__ done = a->done, *__restrict__ ep_cnt = a->ep_cnt, *__restrict__ infeas_n = a->infeas_n;
    double *__restrict__ ep_sum = a->ep_sum;
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
        rbe::stat_add(&ep_sum[i], double(ret)); rbe::stat_add(&ep_sum[n + i], double(ret) * double(ret));
        rbe::stat_add(&ep_cnt[i], 1u); rbe::stat_add(&ep_cnt[n + i], sn - 1u); rbe::stat_add(&ep_cnt[2 * n + i], reached ? 1u : 0u);
        const uint64_t gid = a->env0 + uint64_t(i);
        const uint64_t seed = a->seed;
        const float lo[3] = {a->box_lo[0], a->box_lo[1], a->box_lo[2]}, hi[3] = {a->box_hi[0], a->box_hi[1], a->box_hi[2]};
        uint32_t draw = goal_count[i];
        // RoboyEnv.step draws a goal (_set_new_goal, :67-68); the VecEnv worker's env.reset() (:82-87) then draws another one, which
        // replaces it before anybody saw it: the counter advances by two, only the SECOND draw is evaluated
        draw_goal3(lo, hi, seed, gid, draw + (e.auto_reset ? 1u : 0u), gg);
        draw += e.auto_reset ? 2u : 1u;
        if (e.auto_reset) {
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

// UNROLL = 0: run-time tendon count (ConstX, c.nt tendons, action rows of c.nt floats)
template <int INTEG, int BLOCK, int UNROLL, typename CONST = Const8, bool BK = false>
__global__ void __launch_bounds__(BLOCK)
msj_env_step_kernel(const CONST c_arg, const MsjEnvArgs a) {
    // a.n: the handle's envs = the stride of the state / goal / statistics planes; a.cnt: the envs of THIS launch - all of them, or a
    // sub-range (rb_env_step_range_dev: every pointer then points at the range's first env, a.env0 is its global id)
    const float *__restrict__ q = a.q, *__restrict__ qd = a.qd, *__restrict__ goal = a.goal, *__restrict__ act = a.act;
    const long n = a.n, cnt = a.cnt;
    const float slope = a.e.slope, act_hi = a.e.act_hi;
    const CONST &c = robot_consts<BK>(c_arg);
    const long i = long(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= cnt) return;
    float qq[3], vv[3], gg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + i]; vv[j] = qd[j * n + i]; gg[j] = goal[j * n + i]; }
    // the reference asserts the action lies in [-1,1] (roboy_env.py:52); a batched kernel
    // cannot raise, so it clamps.  Then slope * (x - in_high) + out_high, each op rounded
    // (roboy_env.py:157-158)
    auto rescale = [&](float x) { return mul_then_add(slope, fminf(fmaxf(x, -1.0f), 1.0f) - 1.0f, act_hi); };
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
    const float av[NT8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int k = 0; k < NT8; ++k) sp[k] = rescale(av[k]) * c.ten[k].ksg;   // set-point -> activation offset
    if (UNROLL == RS) {
        ok = rb::MsjModel<float, NT8>::template step_rs<INTEG>(c, qq, vv, sp);
    } else if (UNROLL >= NT8) {
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
    if constexpr (BK) env_account(&a, i, qq, vv, gg, ok);
    else env_account(late_env_args(msj_env_args_offset(int(sizeof(CONST)))), i, qq, vv, gg, ok);
}


// The fused env layer in the two-lanes-per-env form (round 5; msj_step_mirror_pairs above): both lanes of a pair load the env's
// state, each rescales and evaluates ITS four tendons (the odd lane steps the mirrored env), and behind the integrator the even
// lane - which holds the env itself - does the accounting of env_account() while the odd lane is done.  What the mid-size
// batches of a PPO rollout (12 288 - 32 768 envs, RK4) step with: the plain step's mirror-pair form takes 4.5 us there against
// 5.2 for one env per lane (profiles/r4_a/mid_sweep.log) and the env layer used to be an env-per-lane kernel whatever the step was.
template <int INTEG, int BLOCK, int MIRROR, bool BK = false>
__global__ void __launch_bounds__(BLOCK)
msj_env_step_mirror_pairs(const Const8 c_arg, const PairMap pm, const MsjEnvArgs a) {
    const float *__restrict__ q = a.q, *__restrict__ qd = a.qd, *__restrict__ goal = a.goal, *__restrict__ act = a.act;
    const long n = a.n, cnt = a.cnt;
    const float slope = a.e.slope, act_hi = a.e.act_hi;
    const Const8 &c = robot_consts<BK>(c_arg);
    const long wg0 = long(blockIdx.x) * (BLOCK / 2);
    const long left = cnt - wg0;
    const int live = int(left < BLOCK / 2 ? left : BLOCK / 2);
    const int le = int(threadIdx.x >> 1);      // the pair's env within the workgroup
    if (le >= live) return;                    // whole pairs leave together
    const bool odd = (threadIdx.x & 1) != 0;
    const int off = le * 4, oddi = int(threadIdx.x & 1);
    const long i = wg0 + le;
    float qq[3], vv[3], gg[3], u[4];
    const __amdgpu_buffer_rsrc_t ra = wg_rsrc(act + wg0 * NT8, live * NT8 * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // clamp, slope * (x - in_high) + out_high with two roundings (roboy_env.py:157-158), then set-point -> activation offset
        const float x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, off * NT8 + pm.a[k] + oddi * pm.d[k], 0, 0));
        u[k] = mul_then_add(slope, fminf(fmaxf(x, -1.0f), 1.0f) - 1.0f, act_hi) * c.ten[k].ksg;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        qq[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wg_rsrc(q + j * n + wg0, live * 4), off, 0, 0));
        vv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wg_rsrc(qd + j * n + wg0, live * 4), off, 0, 0));
        gg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wg_rsrc(goal + j * n + wg0, live * 4), off, 0, 0));
    }
    constexpr int F0 = MIRROR == 0 ? 0 : 1, F1 = 2;
    const uint32_t flip = odd ? 0x80000000u : 0u;
    qq[F0] = __uint_as_float(__float_as_uint(qq[F0]) ^ flip); vv[F0] = __uint_as_float(__float_as_uint(vv[F0]) ^ flip);
    qq[F1] = __uint_as_float(__float_as_uint(qq[F1]) ^ flip); vv[F1] = __uint_as_float(__float_as_uint(vv[F1]) ^ flip);
    const bool ok = rb::MsjModel<float, NT8>::template integrate_acc<INTEG>(c, qq, vv, AccelMirrorHalf<MIRROR>{c, u});
    if (odd) return;                           // the even lane holds the env itself
    if constexpr (BK) env_account(&a, i, qq, vv, gg, ok);
    else env_account(late_env_args(msj_env_args_offset(int(sizeof(Const8) + sizeof(PairMap)))), i, qq, vv, gg, ok);
}


}  // namespace rbk
