// roboy_sim.hip - MI355X (gfx950) implementation of include/roboy_sim.h.
//
// Replaces the per-step ROS round-trip of the reference's RosSimulationClient
// (gym_roboy/envs/simulations/ros_simulation_client.py:32-81) by kernels that
// advance N independent environments in lock-step.  Ball-joint robots: state
// struct-of-arrays in HBM, one env per lane (msj_kernels.hpp; one tendon per lane
// for small batches, below), robot constants wave-uniform - through the kernarg
// (scalar loads -> SGPRs), or folded into the code: MsjRobot's table ahead of time
// (msj_baked.hpp), any other 8-tendon robot's by hiprtc at run time (msj_jit.hpp);
// LDS holds only what is indexed at run time (the set-points of the rolled tendon
// loop).  Generic joint trees: one env per lane running straight-line code generated
// for the robot (tree_lane_gen.hpp, tree_lane.hpp: the committed upper body ahead of
// time, other robots by hiprtc) or, without that specialisation, two envs per wave,
// eight lanes per link, working set in LDS (tree_aba.hpp).  DESIGN.md §4-§5.
// Which kernel instance a call launches is ONE table (roboy_dispatch.hpp, included below): 98 rows keyed by robot class / entry kind /
// kernel form / integrator / workgroup size / constants source / variant, RB_KERNEL_AUTO's thresholds as a list of rules; this file
// holds the kernels that are not in a header of their own, the handle, the availability predicates and the C ABI.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/roboy_sim.h"
#include "msj_build.hpp"
#include "msj_kernels.hpp"
#include "msj_jit.hpp"
#include "tree_aba.hpp"
// env-per-lane form of the joint-tree kernels: generated per robot (tree_lane_gen.hpp); the committed upper body's
// instances are compiled here ahead of time, any other robot's by hiprtc (tree_lane_jit.hpp)
#include "tree_lane_defs.hpp"
#define RBL_NS rbl_baked
#include "tree_lane_baked.hpp"
#include "tree_lane.hpp"
#undef RBL_NS
// ... and its split form (several waves per group of 64 envs, for small batches): tree_lane_split.hpp
#define RBL_NS rbl_split_baked
#include "tree_lane_split_baked.hpp"
#include "tree_lane_split.hpp"
#undef RBL_NS
#include "tree_lane_jit.hpp"
#include "tree_lane_split2.hpp"      // the lean two-part split instances: a translation unit of their own (roboy_sim_split2.hip)

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define RB_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(RB_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

#ifndef RB_SMALL_BATCH
#define RB_SMALL_BATCH 65536   // <= this many envs: latency-oriented launch configuration
#endif
// AUTO picks the tendon-per-lane form up to this many envs (measured crossover,
// profiles/r1_b/sweep.log: Euler 2.9 vs 3.5 us at 8 192 and a tie at 16 384; RK4
// 6.2 vs 7.0 us at 16 384 and 10.2 vs 7.1 us at 32 768)
// (round 4, robots with a mirror plane: the two-lanes-per-env form takes over from 4 096 / 12 288 envs - roboy_dispatch.hpp: AUTO_RULES)
#ifndef RB_TENDON_LANE_BATCH_EULER
#define RB_TENDON_LANE_BATCH_EULER 8192
#endif
#ifndef RB_TENDON_LANE_BATCH_RK4
#define RB_TENDON_LANE_BATCH_RK4 16384
#endif
#ifndef RB_TENDON_LANE_BATCH_PAIR_EULER
#define RB_TENDON_LANE_BATCH_PAIR_EULER 4096
#endif
#ifndef RB_TENDON_LANE_BATCH_PAIR_RK4
#define RB_TENDON_LANE_BATCH_PAIR_RK4 12288
#endif
// launch configuration of the env-per-lane form above RB_SMALL_BATCH envs: workgroup size and
// unroll factor of the tendon loop, per integrator (A/B: profiles/r2_a/kernel_ab_experiments.log)
#ifndef RB_BIG_BLOCK_EULER
#define RB_BIG_BLOCK_EULER 256
#endif
#ifndef RB_BIG_UNROLL_EULER
#define RB_BIG_UNROLL_EULER 2
#endif
#ifndef RB_BIG_BLOCK_RK4
#define RB_BIG_BLOCK_RK4 256
#endif
#ifndef RB_BIG_UNROLL_RK4
#define RB_BIG_UNROLL_RK4 2
#endif
// kernels instantiated on MsjRobot's baked constant table (msj_baked.hpp) unroll the tendon loop by this much, per integrator
// (round 2, U = 2 / 4 / 8: RK4 16.96 / 16.63 / 16.77 us, Euler 2M 36.9 / 35.2 / 35.3 us; round 3, U = 4 / 8: RK4 262 144 envs
// 16.58 / 16.92 us and 2 M envs 97.8 / 110.9 us, Euler 262 144 envs 7.01 / 6.88 us and 2 M envs 35.29 / 34.63 us - fully
// unrolled, the robot's constants are literals instead of scalar-register operands, which issue at half rate (profiles/r3_a:
// issue_forms_probe.log); RK4 pays more for the registers of the longer body than it gains)
#ifndef RB_BAKED_UNROLL_EULER
#define RB_BAKED_UNROLL_EULER 8
#endif
// (round 4: RK4 = RS, the "rolled stages" form - msj_math.hpp: step_rs, msj_kernels.hpp: msj_step_env_per_lane_rs - the four
// stages as a loop over running sums, the eight tendons written out inside it: a quarter of the code and 64 registers, so
// the constants are literals without the register bill that sank U = 8; 262 144 envs 16.6 -> 15.9 us per step with one launch,
// 12.9 -> 12.0 us in two chains, 131 072 envs 9.5 -> 8.6 / 7.4 us; profiles/r4_a/rs_sweep.log)
#ifndef RB_BAKED_UNROLL_RK4
#define RB_BAKED_UNROLL_RK4 RS
#endif
// (the fused open-loop rollout is instantiated with the SAME unroll factor as the step kernel - its results are bit-identical to
// single steps only then: the fully unrolled body contracts its products differently - although it loses to it: it keeps its
// state in registers across steps, 2 M envs Euler 8.6e10 env-steps/s at U = 4, 7.4e10 at U = 8)
// joint-tree robots without ahead-of-time instances: AUTO builds the env-per-lane kernels with hiprtc (~10 s each)
// from this many envs on (ROBOY_SIM_JIT=2: at any batch size, =0: never); below it the octet kernels run
#ifndef RB_TREE_JIT_BATCH
#define RB_TREE_JIT_BATCH 16384
#endif
// ... and only for robots whose generated code keeps at most this many values alive at once (rblg::Generated::max_live;
// the upper body: 332): beyond it the kernel spills to scratch - minutes to build, and every reload a full memory
// latency on a SIMD with one wave - and the octets are the better form.  An explicit rb_select_kernel(1) builds anyway.
#ifndef RB_TREE_LANE_MAX_LIVE
#define RB_TREE_LANE_MAX_LIVE 400
#endif
// joint-tree robots with split-form instances (several waves per 64 envs): AUTO uses them up to this many envs - while
// the waves of a launch still find a SIMD each (measured crossover against the one-wave form: profiles/r3_a)
#ifndef RB_TREE_SPLIT_BATCH
#define RB_TREE_SPLIT_BATCH 16384
#endif
// tendon-helper waves of the split form (tree_lane_gen.hpp: generate_split): at most this many, for the longest parts.  Must be the
// number the ahead-of-time text was generated with (tools/gen_tree_lane_baked.py: SPLIT_HELPERS).
#ifndef RB_SPLIT_HELPERS
#define RB_SPLIT_HELPERS 2
#endif
// ... and the share (percent, proximal first) of a helped part's tendons its helper takes (SPLIT_HELPER_SHARE there)
#ifndef RB_SPLIT_HELPER_SHARE
#define RB_SPLIT_HELPER_SHARE 80
#endif
// ... and whether the parts run their backward pass in two sweeps around barrier T (SPLIT_TWO_SWEEPS there): everything but the tendon
// wrenches' part before it, beside the helpers (upper body, 8 192 envs: 9.15 -> 8.84 us Euler, 26.5 -> 25.1 us RK4 at a share of 70 %)
#ifndef RB_SPLIT_TWO_SWEEPS
#define RB_SPLIT_TWO_SWEEPS 1
#endif
// ... and whether ONE part evaluates the trunk links' inertias / bias forces for all (SPLIT_SHARE_TRUNK there): upper body, 8 192 envs,
// share 55 ... 90 % with and without (profiles/r4_a/share_sweep_fine.log, two passes, +-0.05 us): 70 % 8.76 / 24.91 us Euler / RK4,
// 70 % + trunk 8.62 / 24.52, 75 % + trunk 8.62 / 24.50, 80 % 8.71 / 24.37, 80 % + trunk 8.74 / 24.25, 85 % and more 9.2 / 25.6
#ifndef RB_SPLIT_SHARE_TRUNK
#define RB_SPLIT_SHARE_TRUNK 1
#endif
// ... or the CUT form instead (generate_split_cut: the RB_SPLIT_HELPERS heaviest parts as a proximal and a distal wave each;
// RB_SPLIT_HELPER_SHARE is then the distal waves' share of the tendons).  SPLIT_CUT there.  Measured and NOT selected: 22 % fewer
// vector instructions on the longest path of the upper body, and 10.2 / 30.1 us against 8.84 / 25.1 - all five waves are busy at once
// there, and the two that share a SIMD run at 7 cycles per instruction instead of 5.5 (profiles/r4_a/cut_form.log).
// The lean two-part split form (roboy_sim_split2.hip): two part waves per 64 envs and two workgroups per CU - one generation up to
// 32 768 envs, where one wave per 64 envs leaves half of the SIMDs idle (16.1 us for the upper body's Euler step) and the five-wave
// form needs two generations (17.5 us).  Measured: profiles/r5_a/split2_sweep.log.
#ifndef RB_TREE_SPLIT2_BATCH
#define RB_TREE_SPLIT2_BATCH 32768
#endif
#ifndef RB_SPLIT2_PARTS
#define RB_SPLIT2_PARTS 2            // (tools/gen_tree_lane_baked.py: SPLIT2_PARTS, SPLIT2_SHARE_TRUNK)
#endif
#ifndef RB_SPLIT2_SHARE_TRUNK
#define RB_SPLIT2_SHARE_TRUNK 1
#endif
#ifndef RB_SPLIT_MAX_PARTS
#define RB_SPLIT_MAX_PARTS 4      // part waves per env group at most (the upper body has three branches: three parts)
#endif
#ifndef RB_SPLIT_CUT
#define RB_SPLIT_CUT 0
#endif
// the two-lanes-per-env form launches one-wave workgroups up to this many envs (spread over the CUs), 256-thread ones above
#ifndef RB_PAIR_SMALL_BATCH
#define RB_PAIR_SMALL_BATCH 65536
#endif
// AUTO picks the two-lanes-per-env form (robots with a mirror plane) above the tendon-per-lane range up to this many envs
// (0: never; set from the sweep in profiles/r4_a)
#ifndef RB_PAIR_BATCH_EULER
#define RB_PAIR_BATCH_EULER 16384
#endif
#ifndef RB_PAIR_BATCH_RK4
#define RB_PAIR_BATCH_RK4 32768
#endif
// The fused env layer of an 8-tendon ball-joint robot has its own AUTO thresholds (the accounting behind the step shifts the
// crossovers): two lanes per env up to here where the robot has a mirror plane (profiles/r5_a/env_pairs_sweep.log) ...
#ifndef RB_PAIR_ENV_BATCH_EULER
#define RB_PAIR_ENV_BATCH_EULER 24576
#endif
#ifndef RB_PAIR_ENV_BATCH_RK4
#define RB_PAIR_ENV_BATCH_RK4 32768
#endif
// ... and eight lanes per env below that (any 8-tendon ball-joint robot; profiles/r5_a/env_octets_sweep.log)
#ifndef RB_OCTET_ENV_BATCH_EULER
#define RB_OCTET_ENV_BATCH_EULER 8192
#endif
#ifndef RB_OCTET_ENV_BATCH_RK4
#define RB_OCTET_ENV_BATCH_RK4 8192
#endif
using namespace rbk;    // the env-per-lane kernels (msj_kernels.hpp), EnvParams, GoalBox, ...

// ------------------------------------------------------------------ kernels

// Env-per-lane step for ball-joint robots with 1..NTX tendons (count in c.nt): the rolled
// tendon loop of the large-batch form with a run-time trip count; actions are rows of
// c.nt floats, staged (scaled) as the lane's LDS column.
template <int INTEG, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
msj_step_env_per_lane_nt(const ConstX c, float *__restrict__ q, float *__restrict__ qd,
                         uint32_t *__restrict__ feas, const float *__restrict__ act, float act_scale, long n) {
    const long i = long(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    __shared__ float lds_sp[NTX][BLOCK];
    const int nt = c.nt;
    const float *row = act + i * nt;
    for (int k = 0; k < nt; ++k) lds_sp[k][threadIdx.x] = row[k] * (act_scale * c.ten[k].ksg);
    float qq[3], vv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + i]; vv[j] = qd[j * n + i]; }
    const bool ok = rb::MsjModel<float, NTX>::template step_sp<INTEG, 0>(c, qq, vv, SpLds{&lds_sp[0][threadIdx.x], BLOCK});
#pragma unroll
    for (int j = 0; j < 3; ++j) { q[j * n + i] = qq[j]; qd[j * n + i] = vv[j]; }
    feas[i] = ok ? 1u : 0u;
}

// ---------------------------------------------------------------------------
// Tendon-per-lane form for small batches.  8 consecutive lanes share one env:
// lane k evaluates tendon k (routing, Hill force, torque contribution), the 3
// torque components are summed over the 8 lanes with DPP adds (no LDS), and
// every lane then carries the (cheap, redundant) rigid-body solve.  A wave
// advances 8 envs with ~1/3 of the per-wave instruction chain of the
// env-per-lane form, and a 4 096-env batch becomes 512 waves instead of 64:
// what a latency-bound launch needs.  The per-tendon record is lane-dependent
// here, so it is read from a small device table (one 64-byte record per lane,
// L1/L2-resident) instead of SGPRs.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 8 lanes of an env group, result in every lane:
// quad_perm(1,0,3,2), quad_perm(2,3,0,1), then row_half_mirror (lane i <-> 7-i)
__device__ __forceinline__ float sum8(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v;
}

struct AccelOneTendon {
    const Const8 &c;
    const rb::MsjTendon<float> &t;
    float spk;
    __device__ __forceinline__ void operator()(const float q[3], const float qd[3], float qdd[3]) const {
        using M = rb::MsjModel<float, NT8>;
        const M::Frame f = M::frame(q, qd);
        float tx = 0.0f, ty = 0.0f, tz = 0.0f;
        M::tendon(c, f, t, spk, tx, ty, tz);
        tx = sum8(tx); ty = sum8(ty); tz = sum8(tz);
        M::rigid_body(c, f, qd, tx, ty, tz, qdd);
    }
};

template <int INTEG>
__global__ void __launch_bounds__(64)
msj_step_tendon_per_lane(const Const8 c, const rb::MsjTendon<float> *__restrict__ ten,
                         float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas,
                         const float *__restrict__ act, float act_scale, long n) {
    // XCD-aware block -> env-group map.  A wave touches only 32 bytes of each
    // state plane, so four consecutive waves share every 128-byte line; workgroups
    // are dealt round-robin over the 8 XCDs (private L2s), which made each line
    // travel to four L2s (PMC: 2.5x the algorithmic read bytes).  Blocks that share
    // an XCD (equal blockIdx % 8) now take one contiguous range of env groups
    // (bijective form of cdna_hip_programming.md T1; affects traffic only).
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7u, qn = nb >> 3, rn = nb & 7u;
    const unsigned blk = (xcd < rn ? xcd * (qn + 1u) : rn * (qn + 1u) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const long t = long(blk) * 64 + threadIdx.x;
    const int k = threadIdx.x & 7;
    long e = t >> 3;
    const bool live = e < n;          // whole 8-lane groups are live or not; dead groups
    if (!live) e = n - 1;             // shadow the last env so every DPP partner is active
    const rb::MsjTendon<float> rec = ten[k];
    const float spk = act[e * NT8 + k] * (act_scale * rec.ksg);   // activation offset of the own tendon
    float qq[3], vv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + e]; vv[j] = qd[j * n + e]; }
    const bool ok = rb::MsjModel<float, NT8>::template integrate<INTEG>(c, qq, vv, AccelOneTendon{c, rec, spk});
    // lanes 0-2 store q, 3-5 store qd, 6 stores the feasibility word: one store each
    if (live && k < 6) {
        const float val = k == 0 ? qq[0] : k == 1 ? qq[1] : k == 2 ? qq[2] : k == 3 ? vv[0] : k == 4 ? vv[1] : vv[2];
        float *plane = (k < 3 ? q : qd) + long(k < 3 ? k : k - 3) * n;
        plane[e] = val;
    }
    if (live && k == 6) feas[e] = ok ? 1u : 0u;
}


// The fused env layer in the eight-lanes-per-env form (round 5): the step as above - lane k rescales and evaluates tendon k - and
// behind the integrator lane 0 of each group does RoboyEnv.step's accounting (rbk::env_account, shared with the other two forms).
// For small batches, where a launch is one wave's dependent chain: the chain of this form is a third of the env-per-lane kernel's.
// n: the handle's envs (plane stride); cnt: the envs of this launch (a sub-range arrives on shifted pointers, env0 = its global id).
template <int INTEG>
__global__ void __launch_bounds__(64)
msj_env_step_tendon_per_lane(const Const8 c, const rb::MsjTendon<float> *__restrict__ ten, const MsjEnvArgs a) {
    const float *__restrict__ q = a.q, *__restrict__ qd = a.qd, *__restrict__ goal = a.goal, *__restrict__ act = a.act;
    const long n = a.n, cnt = a.cnt;
    const float slope = a.e.slope, act_hi = a.e.act_hi;
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7u, qn = nb >> 3, rn = nb & 7u;      // XCD-aware block -> env-group map (above)
    const unsigned blk = (xcd < rn ? xcd * (qn + 1u) : rn * (qn + 1u) + (xcd - rn) * qn) + (blockIdx.x >> 3);
    const long t = long(blk) * 64 + threadIdx.x;
    const int k = threadIdx.x & 7;
    long e = t >> 3;
    const bool live = e < cnt;
    if (!live) e = cnt - 1;           // dead groups shadow the last env so every DPP partner is active
    const rb::MsjTendon<float> rec = ten[k];
    // clamp, slope * (x - in_high) + out_high with two roundings (roboy_env.py:157-158), then set-point -> activation offset
    const float x = act[e * NT8 + k];
    const float spk = rbk::mul_then_add(slope, fminf(fmaxf(x, -1.0f), 1.0f) - 1.0f, act_hi) * rec.ksg;
    float qq[3], vv[3], gg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + e]; vv[j] = qd[j * n + e]; gg[j] = goal[j * n + e]; }
    const bool ok = rb::MsjModel<float, NT8>::template integrate<INTEG>(c, qq, vv, AccelOneTendon{c, rec, spk});
    // (kernarg constants: the accounting reads its arguments late - msj_kernels.hpp, MsjEnvArgs)
    if (live && k == 0) rbk::env_account(rbk::late_env_args(rbk::msj_env_args_offset(int(sizeof(Const8) + sizeof(void *)))), e, qq, vv, gg, ok);
}


// State layout: SoA planes [n_q][n] for the env-per-lane / tendon-per-lane kernels, env-major rows
// [n][n_q] for the joint-tree kernels (rows != 0), tree_aba.hpp.
__device__ __forceinline__ long state_index(long i, int j, int n_q, long n, int rows) { return rows ? i * n_q + j : long(j) * n + i; }

__global__ void reset_kernel(float *q, float *qd, uint32_t *feas, const uint8_t *mask, int n_q, long n, int rows) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    for (int j = 0; j < n_q; ++j) { q[state_index(i, j, n_q, n, rows)] = 0.0f; qd[state_index(i, j, n_q, n, rows)] = 0.0f; }
    feas[i] = 1u;
}

// read-back staging: [n][n_q] q rows | [n][n_q] qd rows | [n] feasibility bytes in
// one buffer: one kernel, one device-to-host copy, one synchronisation
__global__ void pack_state_kernel(const float *q, const float *qd, const uint32_t *feas, float *rows, int n_q, long n, int src_rows) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float *rq = rows + i * n_q, *rv = rows + n * n_q + i * n_q;
    for (int j = 0; j < n_q; ++j) { rq[j] = q[state_index(i, j, n_q, n, src_rows)]; rv[j] = qd[state_index(i, j, n_q, n, src_rows)]; }
    reinterpret_cast<uint8_t *>(rows + 2 * n * n_q)[i] = feas[i] ? 1 : 0;
}
__global__ void unpack_rows_kernel(const float *rows, float *planes, int n_q, long n, int dst_rows) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < n_q; ++j) planes[state_index(i, j, n_q, n, dst_rows)] = rows[i * n_q + j];
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


// rb_env_stats: reduce the per-env accumulators in fp64, ONE launch and nothing else (no memset, no copy:
// the statistics run at the end of every reporting interval of a rollout).  Every block leaves its 8 partial
// sums in `partials`, takes a ticket, and the block that arrives last adds the partials up - thread b owns
// block b's row, then a fixed shuffle / LDS tree - and writes the result to `out` and, if given, `out2`
// (the caller's device buffer).  The sum has a fixed order: bit-reproducible, unlike float atomics.
// Hand-off as MI355X_MICROARCH.md prescribes: stores -> agent-scope release -> vmcnt(0) -> ticket; the last
// block: ticket value -> agent-scope acquire -> barrier -> plain loads.
constexpr int STATS_BLOCKS = 256;
// ENV: the handle runs the fused env layer (all eight slots live); plain rollouts only count infeasible envs, so
// their reduction moves one value instead of eight through the shuffles (8 -> 4 us at 262 144 envs)
template <bool ENV>
__global__ void __launch_bounds__(256)
stats_reduce_kernel(const double *ep_sum, const uint32_t *ep_cnt, const float *ep_ret, const uint32_t *infeas_n,
                    const uint32_t *feas, double *partials, unsigned int *ticket, double *out, double *out2,
                    double env_steps, long n) {
    __shared__ double sh[4][8];
    __shared__ bool last;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // [sum return, sum return^2, n_episodes, sum length, n_goal, n_infeasible, -, running returns]
    for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n; i += long(gridDim.x) * 256) {
        if (ENV) {
            v[0] += ep_sum[i]; v[1] += ep_sum[n + i];
#pragma unroll
            for (int k = 0; k < 3; ++k) v[2 + k] += double(ep_cnt[k * n + i]);
            v[7] += double(ep_ret[i]);
            v[5] += double(infeas_n[i]);     // env layer: infeasible env-steps since the reset
        } else {
            v[5] += feas[i] ? 0.0 : 1.0;     // plain rollouts: envs flagged infeasible right now
        }
    }
    auto block_sum = [&]() {                 // -> sh[0][k] in threads 0..7 after the barrier
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (ENV || k == 5)
                for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 8; ++k) sh[w][k] = v[k];
        __syncthreads();
    };
    block_sum();
    if (threadIdx.x < 8)
        partials[blockIdx.x * 8 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    if (threadIdx.x == 0) last = false;
    __syncthreads();
    if (threadIdx.x == 0) {
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
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x < gridDim.x ? __hip_atomic_load(partials + threadIdx.x * 8 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
    block_sum();
    if (threadIdx.x < 8) {
        double t = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        // out: [sum return, sum return^2, n_episodes, sum length, n_goal, n_infeasible, n_env_steps, sum reward]
        if (threadIdx.x == 6) t = env_steps;
        if (threadIdx.x == 7) t += sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];      // sum reward = finished + running returns
        out[threadIdx.x] = t;
        if (out2) out2[threadIdx.x] = t;
    }
    if (threadIdx.x == 0) *ticket = 0u;       // ready for the next launch (same stream: ordered behind this kernel)
}

__global__ void env_reset_kernel(const GoalBox box, float *q, float *qd, uint32_t *feas, float *goal,
                                 uint32_t *step_num, float *ep_ret, uint32_t *goal_count, float *obs,
                                 long n, uint64_t seed, uint64_t env0) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g[3];
    const uint32_t draw = goal_count[i];
    draw_goal3(box, seed, env0 + uint64_t(i), draw, g);
    goal_count[i] = draw + 1u;
    for (int j = 0; j < 3; ++j) {
        q[j * n + i] = 0.0f; qd[j * n + i] = 0.0f; goal[j * n + i] = g[j];
        if (obs) { obs[i * 9 + j] = 0.0f; obs[i * 9 + 3 + j] = 0.0f; obs[i * 9 + 6 + j] = g[j]; }
    }
    feas[i] = 1u; step_num[i] = 1u; ep_ret[i] = 0.0f;
}

inline unsigned blocks_for(long n, int block) { return unsigned((n + block - 1) / block); }

}  // namespace

// --------------------------------------------------------------------- handle
struct rb_sim {
    int device = 0, n_cu = 256;
    long n = 0;
    int n_q = 0, n_t = 0;
    int integrator = 0, nsub = 1, kernel = RB_KERNEL_ENV_PER_LANE;
    double step_size = 0.1;
    uint64_t seed = 0;
    int64_t env0 = 0;
    Const8 c8;
    ConstX cx;               // ball-joint robots with n_t != 8 (ntx = true)
    bool ntx = false;
    bool baked = false;      // c8 equals MsjRobot's compile-time table (msj_baked.hpp): the BK kernel instances apply
    // any other 8-tendon ball-joint robot: the same instances compiled at run time on ITS constants (msj_jit.hpp),
    // built by the first large-batch call; 0 = not tried yet, 1 = in use, -1 = not available (kernarg instances run)
    int jit_state = 0;
    rbj::Module jit;
    std::string jit_why;
    rb::MsjTendon<float> *d_ten = nullptr;   // device copy of c8.ten for the tendon-per-lane form
    // two lanes per env (msj_kernels.hpp: mirror pairs): available when the robot has a mirror plane (find_mirror_pairs)
    bool pair_ok = false, pair_baked = false;   // pair_baked: the baked table's own tendons 0..3 are a mirror half (BK instances apply)
    int pair_mirror = 0;                        // 0: x-z plane, 1: y-z plane
    int pair_half[4] = {0, 1, 2, 3}, pair_image[4] = {7, 6, 5, 4};   // the even lane's tendons and their mirror images
    Const8 c8p;                                 // c8 with the even lane's tendons first (kernarg instances)
    int kernel_choice = RB_KERNEL_AUTO;
    // generic joint-tree robots (tree_aba.hpp): a few envs per wave, articulated-body algorithm
    bool tree = false;
    rbt::TreeHost tree_host;
    uint32_t *d_tree_words = nullptr;   // the robot tables, staged into LDS by every workgroup
    int tree_waves = 1;                 // waves per workgroup
    // env-per-lane form of the joint-tree kernels (tree_lane.hpp): the text generated for this robot, whether it is
    // the text the library's ahead-of-time instances were compiled from, and the hiprtc-built kernels otherwise
    rblg::Generated lane_gen;
    bool lane_ok = false;               // lane_gen is valid (the generator supports the robot)
    bool lane_baked = false;
    rblj::Kernel lane_step_k, lane_env_k;
    // the split form of the same (tree_lane_split.hpp): several waves per group of 64 envs, for small batches
    rblg::SplitGenerated split_gen;
    bool split_ok = false, split_baked = false;
    rblj::Kernel split_step_k, split_env_k;
    // ... and its lean two-part form (two workgroups per CU): the ahead-of-time instances of roboy_sim_split2.hip for the committed
    // upper body, hiprtc-built ones for any other robot that has a split plan (on request: rb_select_kernel(6))
    rblg::SplitGenerated split2_gen;
    bool split2_ok = false, split2_baked = false;
    rblj::Kernel split2_step_k, split2_env_k;
    GoalBox box;
    hipStream_t own_stream = nullptr, stream = nullptr;
    static constexpr int MAX_CHAINS = 4;
    int chains_choice = 0;                                 // rb_set_rollout_chains: 0 = the library's choice, 1..MAX_CHAINS = that many
    hipStream_t chain_stream[MAX_CHAINS] = {};             // the further chains of rb_rollout_dev ([0] unused: the handle's stream; created on first use)
    hipEvent_t chain_fork = nullptr, chain_join[MAX_CHAINS] = {};
    // caller streams handed to rb_step_range_dev / rb_env_step_range_dev: the handle remembers each distinct one with an event
    // recorded behind its last range launch, so that drain() (rb_destroy, rb_select_kernel, rb_set_stream, graph eviction) waits
    // for that work too.  A small set (a closed-loop caller uses one stream per chain); beyond it the oldest entry is waited
    // for and reused.
    static constexpr int MAX_CALLER_STREAMS = 8;
    struct CallerStream { hipStream_t stream = nullptr; hipEvent_t done = nullptr; bool pending = false; uint64_t tick = 0; };
    CallerStream caller[MAX_CALLER_STREAMS];
    uint64_t caller_tick = 0;
    hipStream_t call_stream = nullptr;                     // the stream of the range call in progress (capturing() looks at it too)
    float *d_q = nullptr, *d_qd = nullptr;
    uint32_t *d_feas = nullptr, *d_goal_count = nullptr;
    // fused env layer (rb_env_*)
    bool env_ready = false;
    EnvParams env;
    float *d_goal = nullptr, *d_ep_ret = nullptr;
    double *d_ep_sum = nullptr;      // [2][n]: sum of episode returns, sum of squared returns
    uint32_t *d_ep_cnt = nullptr;    // [3][n]: episodes, summed episode length, goals reached
    uint32_t *d_step_num = nullptr, *d_infeas_n = nullptr;
    double *d_stats = nullptr;   // [8] result of the reduction, then [STATS_BLOCKS][8] block partials, then the ticket word
    double env_steps = 0.0;      // env steps issued since the statistics were reset
    // host I/O staging
    float *d_rows = nullptr;   // [n][max(n_q, n_t)]
    float *d_state_rows = nullptr, *h_state_rows = nullptr;   // read-back staging, device / pinned host
    uint8_t *d_u8 = nullptr;   // [n]
    // rollout graph cache
    struct GraphKey {
        const float *ring_ptr; int ring; int chunk; float scale; int kernel;
        bool operator<(const GraphKey &o) const {
            return std::memcmp(this, &o, sizeof(GraphKey)) < 0;
        }
    };
    struct ChainGraphs {                                   // one LINEAR graph per chain (a chain = a stream stepping its share of the envs)
        hipGraphExec_t exec[MAX_CHAINS] = {};
        int chains = 1;
        void destroy() { for (hipGraphExec_t &e : exec) { if (e) (void)hipGraphExecDestroy(e); e = nullptr; } }
    };
    std::map<GraphKey, ChainGraphs> graphs;
};

namespace {

// Build the run-time specialised kernels of this robot once, outside any stream capture (hipModuleLoadData
// is not capturable): called by the entry points before they launch or capture.
// A stream that is being captured must not see the build: the attempt is left for a later call (the state stays
// "not tried") and this call launches the instances it already has.
bool stream_capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return stream && hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
// the handle's stream, or the caller's stream of the range call in progress (rb_*_range_dev set call_stream around their work)
bool capturing(rb_sim *s) { return stream_capturing(s->stream) || (s->call_stream && stream_capturing(s->call_stream)); }
struct CallStreamScope {          // the stream a range entry point launches on, visible to the build guards for the call's duration
    rb_sim *s;
    CallStreamScope(rb_sim *sim, hipStream_t st) : s(sim) { s->call_stream = st; }
    ~CallStreamScope() { s->call_stream = nullptr; }
};

int jit_level() {      // ROBOY_SIM_JIT: 0 = never, 1 (default) = from the batch thresholds on, 2 = at any batch size
    const char *e = std::getenv("ROBOY_SIM_JIT");
    return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1;
}

// AUTO's own condition for the one-wave-per-64-envs form of the joint-tree kernels: instances at hand, or worth building
bool tree_wants_lane_auto(const rb_sim *s) {
    if (!s->tree || !s->lane_ok) return false;
    if (s->lane_baked) return true;
    const int lvl = jit_level();
    return s->lane_gen.max_live <= RB_TREE_LANE_MAX_LIVE && (lvl == 2 || (lvl == 1 && s->n >= RB_TREE_JIT_BATCH));
}
// may this handle run that form?  (an explicit choice, AUTO's condition, or as what the split forms degrade to)
bool tree_wants_lane(const rb_sim *s) {
    if (!s->tree || !s->lane_ok || s->kernel_choice == RB_KERNEL_ENV_PER_WAVE) return false;
    return s->kernel_choice == RB_KERNEL_ENV_PER_LANE || tree_wants_lane_auto(s);
}

bool tree_wants_split(const rb_sim *s);      // (defined behind the dispatch table: AUTO's rules decide)
bool tree_wants_split2(const rb_sim *s);
// the lean layout's formula (tree_lane_split.hpp, RBL_LEAN): q | qd | goal, the exchange area over the action / observation image, flags
size_t split_lean_lds_bytes(const rblg::SplitGenerated &g) {
    const int img = 3 * g.n_q + (3 * g.n_q > g.n_t ? 3 * g.n_q : g.n_t);
    int shared = g.x_buffers * g.x_slots > img - 3 * g.n_q ? g.x_buffers * g.x_slots : img - 3 * g.n_q;
    if (shared < 4 * g.n_q) shared = 4 * g.n_q;
    return size_t(3 * g.n_q + shared + 3 * g.n_parts + 1) * 64 * 4;
}
size_t split_lds_bytes(const rblg::SplitGenerated &g) {      // the formula of tree_lane_split.hpp: SP_LDS_BYTES
    const int img = 3 * g.n_q + (3 * g.n_q > g.n_t ? 3 * g.n_q : g.n_t);
    return size_t(img + g.x_buffers * g.x_slots + g.n_parts * (g.part_lds + (g.acc_slots ? g.acc_slots : 2 * g.n_q)) + 3 * g.n_parts + 1 +
                  (g.n_helpers > 0 ? g.n_q : 0)) * 64 * 4;
}
// the hiprtc-built split kernels of a robot without ahead-of-time instances (explicit choice only); kind: 0 = step, 1 = env step
bool build_split_kernel(rb_sim *s, int kind = 0) {
    rblj::Kernel &k = kind == 0 ? s->split_step_k : s->split_env_k;
    if (k.state != 0) return k.state == 1;
    if (capturing(s)) return false;                      // try again outside the capture
    if (hipSetDevice(s->device) != hipSuccess) { k.state = -1; k.why = "hipSetDevice failed"; return false; }
    // (the static_assert: the host's LDS formula - split_lds_bytes, what the launch asks for - must be the kernels' layout)
    const std::string src = "#include \"tree_lane_defs.hpp\"\n#define RBL_NS rbl_jit_split\n" + s->split_gen.text + "#include \"tree_lane_split.hpp\"\n" +
                            "static_assert(rbl_jit_split::SP_LDS_BYTES == " + std::to_string(split_lds_bytes(s->split_gen)) + ", \"host and kernel LDS layouts differ\");\n";
    const std::string name = std::string(kind == 0 ? "rbl_jit_split::tree_split_step<" : "rbl_jit_split::tree_split_env_step<") + (s->integrator == RB_EULER ? "0>" : "1>");
    const char *names[1] = {name.c_str()};
    hipFunction_t *slots[1] = {&k.fn};
    k.state = rbj::compile_and_load(src, "roboy_tree_split_jit.hip", names, 1, k.mod, slots, k.why) ? 1 : -1;
    if (k.state == 1 && split_lds_bytes(s->split_gen) > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k.fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(split_lds_bytes(s->split_gen))) != hipSuccess) {
        k.state = -1; k.why = "LDS of the split kernel not granted";
    }
    return k.state == 1;
}

// the hiprtc-built lean two-part kernels of a robot without ahead-of-time instances (explicit choice only); kind: 0 = step, 1 = env step
bool build_split2_kernel(rb_sim *s, int kind = 0) {
    rblj::Kernel &k = kind == 0 ? s->split2_step_k : s->split2_env_k;
    if (k.state != 0) return k.state == 1;
    if (capturing(s)) return false;                      // try again outside the capture
    if (hipSetDevice(s->device) != hipSuccess) { k.state = -1; k.why = "hipSetDevice failed"; return false; }
    const size_t lds = split_lean_lds_bytes(s->split2_gen);
    const std::string src = "#include \"tree_lane_defs.hpp\"\n#define RBL_NS rbl_jit_split2\n#define RBL_LEAN 1\n" + s->split2_gen.text + "#include \"tree_lane_split.hpp\"\n" +
                            "static_assert(rbl_jit_split2::SP_LDS_BYTES == " + std::to_string(lds) + ", \"host and kernel LDS layouts differ\");\n";
    const std::string name = std::string(kind == 0 ? "rbl_jit_split2::tree_split_step<" : "rbl_jit_split2::tree_split_env_step<") + (s->integrator == RB_EULER ? "0>" : "1>");
    const char *names[1] = {name.c_str()};
    hipFunction_t *slots[1] = {&k.fn};
    k.state = rbj::compile_and_load(src, "roboy_tree_split2_jit.hip", names, 1, k.mod, slots, k.why) ? 1 : -1;
    if (k.state == 1 && lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k.fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) {
        k.state = -1; k.why = "LDS of the lean split kernel not granted";
    }
    return k.state == 1;
}

// kind: 0 = step, 1 = env step.  The kernel to launch, or nullptr for the ahead-of-time instances / the octet kernels.
rblj::Kernel *lane_kernel(rb_sim *s, int kind) {
    rblj::Kernel &k = kind == 0 ? s->lane_step_k : s->lane_env_k;
    if (k.state == 0 && !s->lane_baked && tree_wants_lane(s) && !capturing(s) && hipSetDevice(s->device) == hipSuccess) {
        if (!rblj::build(s->lane_gen, kind, s->integrator == RB_EULER ? 0 : 1, k)) s->jit_why = k.why;
        else {
            // a workgroup is one wave; more than 64 KiB of dynamic LDS (robots with many joints) has to be granted
            const size_t lds = rblg::lane_lds_bytes_per_wave(s->lane_gen);
            if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k.fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) {
                k.state = -1; s->jit_why = k.why = "LDS region of the lane kernel not granted";
            }
        }
    }
    return &k;
}
void maybe_jit(rb_sim *s) {
    if (s->tree) { if (tree_wants_lane(s) && !s->lane_baked) (void)lane_kernel(s, 0); return; }
    if (s->jit_state != 0) return;
    if (s->ntx || s->baked || s->n <= RB_SMALL_BATCH || !rbj::enabled() || jit_level() == 0) { s->jit_state = -1; return; }
    if (capturing(s)) return;                     // try again outside the capture
    s->jit_state = -1;
    if (hipSetDevice(s->device) != hipSuccess) return;
    if (rbj::build(s->c8, s->jit, s->jit_why)) s->jit_state = 1;
}

// Chains: rb_rollout_dev steps large batches (ball joints; joint trees in the one-wave form) as TWO independent chains of
// half-batch launches on two streams (one linear graph per chain).  One launch per step leaves 1.8 us between launches and
// ~1.1 us of load / store phases that nothing overlaps (profiles/r3_a/headline_stamps.log: one generation of waves); with two
// chains one half's gaps lie under the other half's arithmetic, and two concurrent launches interleave their generations of
// waves.  Measured with bench.py's regions (profiles/r3_a/chain_thresholds.log, us per step, one launch -> two chains):
// MsjRobot RK4 98 304 envs 9.4 -> 9.0, 196 608 envs 12.8 -> 11.6, 262 144 envs 16.6 -> 13.0, 2 M envs 97 -> 92; Euler 131 072 envs
// 3.9 -> 4.1 (slower: a chain cannot step faster than ~3.6 us per launch), 262 144 envs 6.9 -> 5.6, 524 288 envs 11.0 -> 7.9,
// 2 M envs 33.7 -> 30.7; upper body (one wave per 64 envs) Euler 32 768 envs 15.8 -> 15.3, 65 536 envs 18.7 -> 16.8, 262 144 envs
// 70.3 -> 56.9; RK4 65 536 envs 55.6 -> 53.7, 131 072 envs 109.3 -> 102.1.  Three and four chains are no better anywhere
// (chain_count.log).  Envs are independent, so the results are those of one launch per step, bit for bit.
// ROBOY_SIM_CHAINS = 1 switches it off (2-4 force a count).
// Flags of the events that fork and join the chains.  Producer and consumer of such an event are kernels on the SAME device; the
// event's own system-scope release + acquire (for the host and other devices) is a fixed cost of ~3 us per event: without it a 20-step
// rollout of the headline batch takes 13.3-13.4 instead of 13.5-13.7 us per step (profiles/r5_a/event_flags_ab.log).  What the
// fence-free form rests on is read off the packets (profiles/r6_a/chain_fence_scopes.log: ROCclr's packet log of a two-chain rollout,
// HIP 7.2.26015 / ROCm 7.2.0): EVERY kernel dispatch packet of a chain - eager or replayed from a graph, 75 of 75 - carries header
// 0xb02 = barrier, acquire scope AGENT, release scope AGENT, with or without the flag; the event's marker is a barrier packet whose
// header goes from 0x1500 (acquire / release scope system) to 0x100 (completion only) with the flag; the waiting stream's barrier
// packet carries no fence in either mode.  So the join is: producer kernel (agent-scope release at its end) -> marker (completion
// signal) -> wait -> consumer kernel (agent-scope acquire at its start) - the XCD-private L2s are written back / invalidated at agent
// scope by the kernels' own packets, which is all a same-device consumer needs.  Host visibility is not these events' job: every
// host-facing entry point ends in hipStreamSynchronize (its barrier packet keeps system scope).  Guard tests with readers on other XCDs
// than the writers: tests/test_full_size_gpu.py::test_consumers_behind_the_join_see_the_other_chains_writes (graph chains; eager head +
// graphs + trailing whole-batch launches; the upper body's chains).  ROBOY_SIM_EVENT_SYSTEM_FENCE=1 restores the system-scope events at
// run time (another ROCm, a doubt, an A/B); a HIP without the flag builds the fenced form.
#ifndef RB_CHAIN_EVENT_FLAGS
#ifdef hipEventDisableSystemFence
#define RB_CHAIN_EVENT_FLAGS (hipEventDisableTiming | hipEventDisableSystemFence)
#else
#define RB_CHAIN_EVENT_FLAGS hipEventDisableTiming
#endif
#endif
inline unsigned chain_event_flags() {
    static const bool fenced = [] { const char *e = std::getenv("ROBOY_SIM_EVENT_SYSTEM_FENCE"); return e && e[0] == '1'; }();
    return fenced ? unsigned(hipEventDisableTiming) : unsigned(RB_CHAIN_EVENT_FLAGS);
}
#ifndef RB_CHAIN_BATCH_RK4
#define RB_CHAIN_BATCH_RK4 98304
#endif
#ifndef RB_CHAIN_BATCH_EULER
#define RB_CHAIN_BATCH_EULER 262144
#endif
#ifndef RB_CHAIN_BATCH_TREE_EULER
#define RB_CHAIN_BATCH_TREE_EULER 32768
#endif
#ifndef RB_CHAIN_BATCH_TREE_RK4
#define RB_CHAIN_BATCH_TREE_RK4 65536
#endif
#include "roboy_dispatch.hpp"

// does this handle run the split form (several waves per 64 envs)?  An explicit choice, or AUTO's rule (ahead-of-time instances
// and a batch small enough that its waves still find a SIMD each) ...
bool tree_wants_split(const rb_sim *s) {
    if (!s->tree || !s->split_ok) return false;
    if (s->kernel_choice == RB_KERNEL_ENV_PER_LANE_SPLIT) return true;
    return s->kernel_choice == RB_KERNEL_AUTO && auto_form(s, ENTRY_STEP) == F_SPLIT;
}
// ... or the lean two-part split form?  An explicit choice (any robot with a split plan: hiprtc), or AUTO between the five-wave
// form's batch and a wave on every SIMD where ahead-of-time instances exist
bool tree_wants_split2(const rb_sim *s) {
    if (!s->tree || !s->split2_ok) return false;
    if (s->kernel_choice == RB_KERNEL_ENV_PER_LANE_SPLIT2) return true;      // (rb_select_kernel has built the kernels of a robot without instances)
    return s->kernel_choice == RB_KERNEL_AUTO && auto_form(s, ENTRY_STEP) == F_SPLIT2;
}

// kernel forms that step a sub-range of the batch (shifted pointers, own env count): what chains and the rb_*_range_dev entry points
// need.  A pure query: the row of what is at hand, no build (rb_rollout_dev / rb_step_range_dev / rb_range_capable build first).
bool range_capable(const rb_sim *s) {
    const Row *r = row_for(const_cast<rb_sim *>(s), ENTRY_STEP, /*build=*/false);
    return r && r->ranges;
}
bool chainable(const rb_sim *s) { return range_capable(s) && (s->tree || s->n > RB_SMALL_BATCH); }
int rollout_chains(const rb_sim *s) {
    static const int forced = [] { const char *e = getenv("ROBOY_SIM_CHAINS"); return e ? atoi(e) : 0; }();
    if (!chainable(s)) return 1;
    if (s->chains_choice >= 1) return s->chains_choice;
    if (forced >= 1 && forced <= rb_sim::MAX_CHAINS) return forced;
    if (s->tree) return s->n >= (s->integrator == RB_EULER ? RB_CHAIN_BATCH_TREE_EULER : RB_CHAIN_BATCH_TREE_RK4) ? 2 : 1;
    return s->n >= (s->integrator == RB_EULER ? RB_CHAIN_BATCH_EULER : RB_CHAIN_BATCH_RK4) ? 2 : 1;
}

// does rb_rollout_dev put one ring turn of plain launches in front of its graphs?  Where a step kernel outlasts the host's
// launches of a step (~3.5 us each, one per chain) with room to spare: ball joints, RK4, from 196 608 envs on (20-step rollouts
// of the headline batch 13.7 -> 13.4 us per step; at 131 072 envs and for the joint trees it is a wash: head_sweep.log);
// ROBOY_SIM_EAGER_HEAD=0 / 1 forces it off / on
#ifndef RB_EAGER_HEAD_BATCH_RK4
#define RB_EAGER_HEAD_BATCH_RK4 196608
#endif
bool rollout_eager_head(const rb_sim *s, int chains) {
    static const int forced = [] { const char *e = getenv("ROBOY_SIM_EAGER_HEAD"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return forced != 0;
    if (s->tree) return false;
    return !s->ntx && s->integrator == RB_RK4 && s->n >= RB_EAGER_HEAD_BATCH_RK4;
}

// the physics step over envs [i0, i1) on `stream` (i1 < 0: the whole batch on the handle's stream): one row of the dispatch table
int launch_step(rb_sim *s, const float *d_act, float act_scale, long i0 = 0, long i1 = -1, hipStream_t stream = nullptr) {
    Launch L;
    if (i1 < 0) { L.i0 = 0; L.cnt = s->n; L.stream = s->stream; }
    else        { L.i0 = i0; L.cnt = i1 - i0; L.stream = stream; }
    L.act = d_act; L.act_scale = act_scale;
    return dispatch(s, ENTRY_STEP, L);
}

int check(const rb_sim *s) {
    if (!s) return fail(RB_EINVAL, "null simulation handle");
    return RB_OK;
}

int read_state_host(rb_sim *s, float *q, float *qd, uint8_t *feasible) {
    const long n = s->n;
    const size_t plane = sizeof(float) * size_t(n) * s->n_q;
    hipLaunchKernelGGL(pack_state_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s->stream,
                       s->d_q, s->d_qd, s->d_feas, s->d_state_rows, s->n_q, n, s->tree ? 1 : 0);
    RB_HIP(hipGetLastError());
    RB_HIP(hipMemcpyAsync(s->h_state_rows, s->d_state_rows, 2 * plane + size_t(n), hipMemcpyDeviceToHost, s->stream));
    RB_HIP(hipStreamSynchronize(s->stream));
    const char *h = reinterpret_cast<const char *>(s->h_state_rows);
    if (q) std::memcpy(q, h, plane);
    if (qd) std::memcpy(qd, h + plane, plane);
    if (feasible) std::memcpy(feasible, h + 2 * plane, size_t(n));
    return RB_OK;
}

// Every stream this handle may still have work on - the stream in use (the caller's after rb_set_stream), its own, and the
// chain streams of rb_rollout_dev / rb_env_step_dev - drained; then the cached graph executables may go.
int drain(rb_sim *s) {
    RB_HIP(hipSetDevice(s->device));
    RB_HIP(hipStreamSynchronize(s->stream));
    if (s->own_stream && s->own_stream != s->stream) RB_HIP(hipStreamSynchronize(s->own_stream));
    for (int c = 1; c < rb_sim::MAX_CHAINS; ++c)
        if (s->chain_stream[c]) RB_HIP(hipStreamSynchronize(s->chain_stream[c]));
    // range launches on caller streams: the event behind the last one on each (the stream itself may be gone by now - an
    // event outlives its stream - so the EVENT is waited for, never the stream)
    for (rb_sim::CallerStream &c : s->caller)
        if (c.pending) { RB_HIP(hipEventSynchronize(c.done)); c.pending = false; }
    return RB_OK;
}
// Remember that a range launch has just been enqueued on `st`, a stream that is not one of the handle's own: record the
// entry's event behind it.  Not while `st` is being captured (the launch is a graph node then, not work in flight: whoever
// replays that graph keeps the handle alive meanwhile - roboy_sim.h).
int note_caller_stream(rb_sim *s, hipStream_t st) {
    if (st == s->stream || st == s->own_stream) return RB_OK;
    for (int c = 1; c < rb_sim::MAX_CHAINS; ++c) if (st == s->chain_stream[c] && st) return RB_OK;
    if (stream_capturing(st)) return RB_OK;
    rb_sim::CallerStream *slot = nullptr;
    for (rb_sim::CallerStream &c : s->caller) if (c.done && c.stream == st) { slot = &c; break; }
    if (!slot) {
        for (rb_sim::CallerStream &c : s->caller) if (!c.done) { slot = &c; break; }
        if (!slot) {                                   // every entry taken: the least recently used one is waited for and reused
            slot = &s->caller[0];
            for (rb_sim::CallerStream &c : s->caller) if (c.tick < slot->tick) slot = &c;
            if (slot->pending) { RB_HIP(hipEventSynchronize(slot->done)); slot->pending = false; }
        } else {
            // (completion is all drain() needs from it - not host visibility of the launch's writes: no system-scope fence on the
            // caller's stream behind every range launch, ~3 us each)
            RB_HIP(hipEventCreateWithFlags(&slot->done, chain_event_flags()));
        }
        slot->stream = st;
    }
    RB_HIP(hipEventRecord(slot->done, st));
    slot->pending = true;
    slot->tick = ++s->caller_tick;
    return RB_OK;
}
int drop_graphs(rb_sim *s) {
    if (s->graphs.empty()) return RB_OK;
    int rc = drain(s);                 // none may be in flight when destroyed
    if (rc) return rc;
    for (auto &kv : s->graphs) kv.second.destroy();
    s->graphs.clear();
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
    if (n_envs > (int64_t(1) << 30)) return fail(RB_EINVAL, "n_envs too large");
    if (env_id_offset < 0) return fail(RB_EINVAL, "env_id_offset must be >= 0");

    rb_sim *s = new (std::nothrow) rb_sim();
    if (!s) return fail(RB_ENOMEM, "out of host memory");
    std::string why;
    int rc = rb::msj_build<float, NT8>(robot, step_size, n_substeps, &s->c8, why);
    if (rc == RB_OK) {
        // bit-for-bit equal to the baked table?  (field by field: the struct has alignment padding)
        const Const8 &a = s->c8, &b = rbk::BAKED_HOST;
        auto same = [](const float *x, const float *y, int n) { return std::memcmp(x, y, sizeof(float) * size_t(n)) == 0; };
        bool eq = same(a.IO, b.IO, 6) && same(a.mc, b.mc, 3) && same(a.g, b.g, 3) && same(a.arm, b.arm, 3) && same(a.damp, b.damp, 3) &&
                  same(a.qlo, b.qlo, 3) && same(a.qhi, b.qhi, 3) && same(a.qdmax, b.qdmax, 3) && same(&a.kps, &b.kps, 8) &&
                  a.nsub == b.nsub && a.simple == b.simple && a.nt == b.nt;
        for (int k = 0; k < NT8 && eq; ++k) eq = same(a.ten[k].A, b.ten[k].A, 16);
        s->baked = eq;
        s->pair_ok = rb::find_mirror_pairs(s->c8, s->pair_mirror, s->pair_half, s->pair_image);
        if (s->pair_ok) {
            s->c8p = s->c8;
            bool first_four = true;
            for (int k = 0; k < 4; ++k) { s->c8p.ten[k] = s->c8.ten[s->pair_half[k]]; first_four = first_four && s->pair_half[k] == k; }
            s->pair_baked = s->baked && first_four;      // the BK instances read BAKED.ten[0..3]
        }
    }
    if (rc == RB_EUNSUPPORTED && robot->n_t != NT8) {
        // a ball-joint robot with another tendon count: same closed form, run-time count
        std::string why_x;
        if (rb::msj_build<float, NTX>(robot, step_size, n_substeps, &s->cx, why_x, /*exact=*/false) == RB_OK) {
            s->ntx = true;
            rc = RB_OK;
        }
    }
    if (rc == RB_EUNSUPPORTED) {
        // not a ball-joint robot: the generic joint-tree kernel (one env per wave)
        std::string why_tree;
        rc = rbt::tree_build(robot, step_size, n_substeps, s->tree_host, why_tree);
        if (rc == RB_OK) {
            s->tree = true;
            std::string why_gen;
            s->lane_ok = rblg::generate(robot, true, s->lane_gen, why_gen) == RB_OK;
            s->lane_baked = s->lane_ok && s->lane_gen.hash == RBL_TEXT_HASH && rblg::lane_lds_slots(s->lane_gen) == rbl_baked::LDS_SLOTS;
            s->split_ok = (RB_SPLIT_CUT ? rblg::generate_split_cut(robot, RB_SPLIT_MAX_PARTS, s->split_gen, why_gen, RB_SPLIT_HELPERS, RB_SPLIT_HELPER_SHARE)
                                        : rblg::generate_split(robot, RB_SPLIT_MAX_PARTS, s->split_gen, why_gen, RB_SPLIT_HELPERS, RB_SPLIT_HELPER_SHARE, RB_SPLIT_TWO_SWEEPS != 0, RB_SPLIT_SHARE_TRUNK != 0)) == RB_OK &&
                          split_lds_bytes(s->split_gen) <= 160 * 1024;
            if (!s->split_ok && RB_SPLIT_HELPERS > 0)        // (the exchange area of the helper form does not fit: the three-barrier-less form)
                s->split_ok = rblg::generate_split(robot, RB_SPLIT_MAX_PARTS, s->split_gen, why_gen, 0) == RB_OK;
            // (... and the host's LDS formula is the kernels': a launch with less LDS than tree_lane_split.hpp lays out would write past it)
            s->split_baked = s->split_ok && s->split_gen.hash == RBL_SPLIT_TEXT_HASH && s->split_gen.n_parts == RBL_NPARTS &&
                             s->split_gen.n_helpers == RBL_NHELPERS && split_lds_bytes(s->split_gen) == size_t(rbl_split_baked::SP_LDS_BYTES);
            // the lean two-part form: only if the text generated for THIS robot is the text the second translation unit was compiled from
            if (s->split_ok) {
                std::string why2;
                s->split2_ok = rblg::generate_split(robot, RB_SPLIT2_PARTS, s->split2_gen, why2, 0, 45, false, RB_SPLIT2_SHARE_TRUNK != 0) == RB_OK &&
                               split_lean_lds_bytes(s->split2_gen) <= 160 * 1024;
                s->split2_baked = s->split2_ok && robot->n_q == rbs2::n_q() && robot->n_t == rbs2::n_t() && s->split2_gen.hash == rbs2::text_hash() &&
                                  s->split2_gen.n_parts == rbs2::n_parts() && split_lean_lds_bytes(s->split2_gen) == rbs2::lds_bytes();
                if (!s->split2_ok) s->split2_gen = rblg::SplitGenerated();
            }
        }
        else why = "not a ball-joint robot (" + why + ") and not a supported joint tree (" + why_tree + ")";
    }
    if (rc != RB_OK) {
        delete s;
        return fail(rc, "no HIP kernel for this robot structure: " + why);
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
    {
        hipDeviceProp_t prop;
        RB_TRY(hipGetDeviceProperties(&prop, device));
        s->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
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
    RB_TRY(hipMalloc(&s->d_state_rows, plane * (2 * s->n_q + 1)));   // q rows | qd rows | n bytes
    RB_TRY(hipHostMalloc(reinterpret_cast<void **>(&s->h_state_rows), plane * (2 * s->n_q + 1), hipHostMallocDefault));
    if (!s->tree) {
        RB_TRY(hipMalloc(&s->d_ten, sizeof(s->c8.ten)));
        RB_TRY(hipMemcpyAsync(s->d_ten, s->c8.ten, sizeof(s->c8.ten), hipMemcpyHostToDevice, s->stream));
    } else {
        rbt::TreeHost &th = s->tree_host;
        RB_TRY(hipMalloc(&s->d_tree_words, sizeof(uint32_t) * th.words.size()));
        RB_TRY(hipMemcpy(s->d_tree_words, th.words.data(), sizeof(uint32_t) * th.words.size(), hipMemcpyHostToDevice));
        th.dev.g_words = reinterpret_cast<const float4 *>(s->d_tree_words);
        if (rbt::tree_lds_bytes(th, 1) > 160 * 1024) {
            rb_destroy(s);
            return fail(RB_EUNSUPPORTED, "robot working set exceeds the 160 KiB LDS of a CU");
        }
        s->tree_waves = rbt::tree_pick_waves(th);
        // more than 64 KiB of dynamic LDS per workgroup has to be granted per kernel
        const int lds_max = 160 * 1024;
#define RB_TREE_ATTR(K) RB_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max))
        RB_TREE_ATTR((rbt::tree_step_aba<0, rbt::TREE_E, true>)); RB_TREE_ATTR((rbt::tree_step_aba<1, rbt::TREE_E, true>));
        RB_TREE_ATTR((rbt::tree_step_aba<0, rbt::TREE_E, false>)); RB_TREE_ATTR((rbt::tree_step_aba<1, rbt::TREE_E, false>));
        RB_TREE_ATTR((rbt::tree_env_step_aba<0, rbt::TREE_E, true>)); RB_TREE_ATTR((rbt::tree_env_step_aba<1, rbt::TREE_E, true>));
        RB_TREE_ATTR((rbt::tree_env_step_aba<0, rbt::TREE_E, false>)); RB_TREE_ATTR((rbt::tree_env_step_aba<1, rbt::TREE_E, false>));
        if (s->split_baked) {
            RB_TREE_ATTR(rbl_split_baked::tree_split_step<0>); RB_TREE_ATTR(rbl_split_baked::tree_split_step<1>);
            RB_TREE_ATTR(rbl_split_baked::tree_split_env_step<0>); RB_TREE_ATTR(rbl_split_baked::tree_split_env_step<1>);
        }
#undef RB_TREE_ATTR
    }
    RB_TRY(hipMemsetAsync(s->d_goal_count, 0, sizeof(uint32_t) * size_t(n_envs), s->stream));
    RB_TRY(hipMalloc(&s->d_infeas_n, sizeof(uint32_t) * size_t(n_envs)));
    RB_TRY(hipMemsetAsync(s->d_infeas_n, 0, sizeof(uint32_t) * size_t(n_envs), s->stream));
    RB_TRY(hipMalloc(&s->d_stats, sizeof(double) * 8 * (STATS_BLOCKS + 2)));
    RB_TRY(hipMemsetAsync(s->d_stats, 0, sizeof(double) * 8 * (STATS_BLOCKS + 2), s->stream));
#undef RB_TRY
    *out = s;
    (void)rb_select_kernel(s, RB_KERNEL_AUTO);
    rc = rb_reset(s, nullptr);
    if (rc != RB_OK) { std::string m = g_err; rb_destroy(s); *out = nullptr; return fail(rc, m); }
    return RB_OK;
}

void rb_destroy(rb_sim *s) {
    if (!s) return;
    // whichever stream is in use (a caller's after rb_set_stream - it must still exist), the handle's own and the chain
    // streams: nothing of this handle may be in flight when its graph executables, streams and buffers go
    (void)drain(s);
    for (auto &kv : s->graphs) kv.second.destroy();
    rbj::unload(s->jit);
    rblj::unload(s->lane_step_k); rblj::unload(s->lane_env_k); rblj::unload(s->split_step_k); rblj::unload(s->split_env_k);
    rblj::unload(s->split2_step_k); rblj::unload(s->split2_env_k);
    (void)hipFree(s->d_q); (void)hipFree(s->d_qd); (void)hipFree(s->d_feas);
    (void)hipFree(s->d_goal_count); (void)hipFree(s->d_rows); (void)hipFree(s->d_u8); (void)hipFree(s->d_ten);
    (void)hipFree(s->d_tree_words);
    (void)hipFree(s->d_state_rows); (void)hipHostFree(s->h_state_rows);
    (void)hipFree(s->d_goal); (void)hipFree(s->d_ep_ret); (void)hipFree(s->d_ep_sum); (void)hipFree(s->d_ep_cnt);
    (void)hipFree(s->d_step_num); (void)hipFree(s->d_infeas_n); (void)hipFree(s->d_stats);
    for (rb_sim::CallerStream &c : s->caller) if (c.done) (void)hipEventDestroy(c.done);
    if (s->chain_fork) (void)hipEventDestroy(s->chain_fork);
    for (int c = 1; c < rb_sim::MAX_CHAINS; ++c) {
        if (s->chain_join[c]) (void)hipEventDestroy(s->chain_join[c]);
        if (s->chain_stream[c]) (void)hipStreamDestroy(s->chain_stream[c]);
    }
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

#ifdef RB_TREE_DEBUG
// debug builds only: arm a dump of the LDS working set of the wave that owns `env` (joint-tree robots; filled by
// the next rb_step*), then fetch it.  Layout: TREE_E blocks of dev.ES floats (tree_build.hpp); *dev_out gets the
// table offsets.
int rb_debug_tree_arm(rb_sim *s, long env) {
    if (!s || !s->tree) return fail(RB_EINVAL, "not a joint-tree handle");
    RB_HIP(hipSetDevice(s->device));
    float *buf = nullptr;
    const size_t nf = size_t(rbt::TREE_E) * s->tree_host.dev.ES;
    RB_HIP(hipMalloc(&buf, nf * sizeof(float)));
    RB_HIP(hipMemset(buf, 0, nf * sizeof(float)));
    RB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rbt::rb_tree_dbg_out), &buf, sizeof(buf)));
    RB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rbt::rb_tree_dbg_env), &env, sizeof(env)));
    return RB_OK;
}
int rb_debug_tree_fetch(rb_sim *s, float *out, int n_floats, rbt::TreeDev *dev_out) {
    if (!s || !s->tree) return fail(RB_EINVAL, "not a joint-tree handle");
    RB_HIP(hipSetDevice(s->device));
    RB_HIP(hipStreamSynchronize(s->stream));
    float *buf = nullptr;
    RB_HIP(hipMemcpyFromSymbol(&buf, HIP_SYMBOL(rbt::rb_tree_dbg_out), sizeof(buf)));
    if (!buf) return fail(RB_EINVAL, "not armed");
    const int nf = rbt::TREE_E * s->tree_host.dev.ES;
    RB_HIP(hipMemcpy(out, buf, sizeof(float) * size_t(n_floats < nf ? n_floats : nf), hipMemcpyDeviceToHost));
    if (dev_out) *dev_out = s->tree_host.dev;
    float *none = nullptr;
    RB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rbt::rb_tree_dbg_out), &none, sizeof(none)));
    (void)hipFree(buf);
    return RB_OK;
}
#endif

#ifdef RB_SPLIT_STAMPS
// diagnostic builds only: the barrier stamps of the split kernels' workgroup 0 (tree_lane_defs.hpp), 8 waves x 128 entries
extern "C" int rb_debug_stamps(unsigned long long *out) {
    RB_HIP(hipDeviceSynchronize());
    RB_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rbl_stamp_buf), sizeof(unsigned long long) * 8 * 128));
    return RB_OK;
}
#endif

#ifdef RB_LANE_STAMPS
// diagnostic builds only: the phase stamps of the one-wave joint-tree kernels (tree_lane.hpp), RBL_STAMP_WAVES waves x 4 x (100 MHz clock, shader clock)
extern "C" int rb_debug_lane_stamps(unsigned long long *out, int n_words) {
    RB_HIP(hipDeviceSynchronize());
    const size_t all = sizeof(unsigned long long) * RBL_STAMP_WAVES * 8, want = sizeof(unsigned long long) * size_t(n_words);
    RB_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rbl_baked::rbl_lane_stamp_buf), want < all ? want : all));
    return RB_OK;
}
#endif

void rb_jit_cache_stats(int64_t *hits, int64_t *compiles, int64_t *stores) {
    rbj::CacheStats &c = rbj::cache_stats();
    if (hits) *hits = c.hits.load();
    if (compiles) *compiles = c.compiles.load();
    if (stores) *stores = c.stores.load();
}

int rb_specialization(rb_sim *s) {
    if (check(s)) return -1;
    if (s->tree) {
        if (tree_wants_split2(s)) return s->split2_baked ? RB_SPEC_TABLE : (s->split2_step_k.state == 1 ? RB_SPEC_JIT : RB_SPEC_NONE);
        if (tree_wants_split(s)) return s->split_baked ? RB_SPEC_TABLE : (s->split_step_k.state == 1 ? RB_SPEC_JIT : RB_SPEC_NONE);
        if (!tree_wants_lane(s)) { g_err = s->lane_ok ? "the octet kernels are selected (batch below the build threshold, or by choice)" : "no generator for this robot"; return RB_SPEC_NONE; }
        if (s->lane_baked) return RB_SPEC_TABLE;
        if (lane_kernel(s, 0)->state == 1) return RB_SPEC_JIT;
        g_err = s->lane_step_k.why;
        return RB_SPEC_NONE;
    }
    if (s->baked) return RB_SPEC_TABLE;
    if (s->jit_state == 0) maybe_jit(s);
    if (s->jit_state == 1) return RB_SPEC_JIT;
    if (!s->jit_why.empty()) g_err = s->jit_why;
    return RB_SPEC_NONE;
}

int rb_select_kernel(rb_sim *s, int kernel) {
    if (check(s)) return RB_EINVAL;
    // validate first: a refused request leaves the handle as it was (kernel_choice, rb_info, the graph cache)
    if (kernel < RB_KERNEL_AUTO || kernel > RB_KERNEL_ENV_PER_LANE_SPLIT2) return fail(RB_EINVAL, "unknown kernel variant");
    if (kernel == RB_KERNEL_ENV_PER_LANE_SPLIT2 && !(s->tree && s->split2_ok))
        return fail(RB_EUNSUPPORTED, "no split form for this robot (a ball-joint robot, a serial chain, or branches tied together by tendons)");
    if (kernel == RB_KERNEL_LANE_PAIR && !(s->pair_ok && !s->tree && !s->ntx))
        return fail(RB_EUNSUPPORTED, "the two-lanes-per-env form needs an 8-tendon ball-joint robot with a mirror plane (tendons in mirror-image pairs, "
                                     "principal-axis inertia, symmetric joint limits)");
    if (kernel == RB_KERNEL_ENV_PER_LANE_SPLIT && !(s->tree && s->split_ok))
        return fail(RB_EUNSUPPORTED, "no split form for this robot (a ball-joint robot, a serial chain, or branches tied together by tendons)");
    if (s->tree) {
        if (kernel == RB_KERNEL_TENDON_PER_LANE) return fail(RB_EUNSUPPORTED, "joint-tree robots have no tendon-per-lane kernel");
        if (kernel == RB_KERNEL_ENV_PER_LANE && !s->lane_ok) return fail(RB_EUNSUPPORTED, "no env-per-lane kernel can be generated for this robot");
        const int before = s->kernel_choice;
        s->kernel_choice = kernel;           // lane_kernel() / tree_wants_*() read it; restored on every failure below
        if (kernel == RB_KERNEL_ENV_PER_LANE && !s->lane_baked) {
            // an explicit choice builds the step kernel now (and fails loudly if that is not possible)
            if (capturing(s)) { s->kernel_choice = before; return fail(RB_EINVAL, "the env-per-lane kernels cannot be built during a stream capture"); }
            if (lane_kernel(s, 0)->state != 1) {
                s->kernel_choice = before;
                return fail(RB_EUNSUPPORTED, "env-per-lane kernel not available: " + s->lane_step_k.why);
            }
        }
        if (kernel == RB_KERNEL_ENV_PER_LANE_SPLIT2 && !s->split2_baked) {
            if (capturing(s)) { s->kernel_choice = before; return fail(RB_EINVAL, "the lean split-form kernel cannot be built during a stream capture"); }
            if (!build_split2_kernel(s)) {
                s->kernel_choice = before;
                return fail(RB_EUNSUPPORTED, "lean split-form kernel not available: " + s->split2_step_k.why);
            }
        }
        if (kernel == RB_KERNEL_ENV_PER_LANE_SPLIT && !s->split_baked) {
            if (capturing(s)) { s->kernel_choice = before; return fail(RB_EINVAL, "the split-form kernel cannot be built during a stream capture"); }
            if (!build_split_kernel(s)) {
                s->kernel_choice = before;
                return fail(RB_EUNSUPPORTED, "split-form kernel not available: " + s->split_step_k.why);
            }
        }
        int rc = drop_graphs(s);             // graphs captured with another variant must not be replayed
        if (rc) { s->kernel_choice = before; return rc; }
        { const int form = resolve_form(s, ENTRY_STEP, /*build=*/false, nullptr); s->kernel = form > 0 ? form : RB_KERNEL_ENV_PER_WAVE; }
        return RB_OK;
    }
    if (kernel == RB_KERNEL_ENV_PER_WAVE) return fail(RB_EUNSUPPORTED, "ball-joint robots have no env-per-wave kernel");
    if (s->ntx && kernel == RB_KERNEL_TENDON_PER_LANE)    // 8 lanes per env is the 8-tendon form
        return fail(RB_EUNSUPPORTED, "the tendon-per-lane kernel is built for 8 tendons");
    int rc = drop_graphs(s);
    if (rc) return rc;
    s->kernel_choice = kernel;
    if (s->ntx) { s->kernel = RB_KERNEL_ENV_PER_LANE; return RB_OK; }
    s->kernel = kernel != RB_KERNEL_AUTO ? kernel : auto_form(s, ENTRY_STEP);
    return RB_OK;
}

int rb_set_stream(rb_sim *s, void *hip_stream) {
    if (check(s)) return RB_EINVAL;
    { int rc = drain(s); if (rc) return rc; }                            // the previous stream and the chain streams
    if (hip_stream == RB_STREAM_DEVICE_DEFAULT) s->stream = nullptr;     // the null stream
    else s->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->own_stream;
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
                       s->d_q, s->d_qd, s->d_feas, d_mask, s->n_q, s->n, s->tree ? 1 : 0);
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
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(g), dim3(256), 0, s->stream, s->d_rows, s->d_q, s->n_q, n, s->tree ? 1 : 0);
    RB_HIP(hipStreamSynchronize(s->stream));
    RB_HIP(hipMemcpyAsync(s->d_rows, qd, sizeof(float) * n * s->n_q, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(g), dim3(256), 0, s->stream, s->d_rows, s->d_qd, s->n_q, n, s->tree ? 1 : 0);
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
    maybe_jit(s);
    RB_HIP(hipMemcpyAsync(s->d_rows, act, sizeof(float) * s->n * s->n_t, hipMemcpyHostToDevice, s->stream));
    int rc = launch_step(s, s->d_rows, act_scale);
    if (rc) return rc;
    s->env_steps += double(s->n);
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
    RB_HIP(hipSetDevice(s->device));
    if (reinterpret_cast<uintptr_t>(d_act) % 16) return fail(RB_EINVAL, "action slab must be 16-byte aligned");
    maybe_jit(s);
    s->env_steps += double(s->n);
    return launch_step(s, d_act, act_scale);
}

int rb_rollout_dev(rb_sim *s, const float *d_ring, int ring, int n_steps, float act_scale, int use_graph) {
    if (check(s) || !d_ring) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    if (ring < 1 || n_steps < 0) return fail(RB_EINVAL, "ring must be >= 1 and n_steps >= 0");
    if (reinterpret_cast<uintptr_t>(d_ring) % 16) return fail(RB_EINVAL, "action ring must be 16-byte aligned");
    maybe_jit(s);                    // before any capture below
    const size_t slab = size_t(s->n) * s->n_t;
    const int chains = use_graph ? rollout_chains(s) : 1;
    if (use_graph && n_steps >= 8 && !s->stream) return fail(RB_EINVAL, "hipGraph capture needs a non-default stream (rb_set_stream)");
    // chain c steps envs [lo_c, lo_{c+1}) (cuts at multiples of 256) on its own stream (chain 0: the handle's)
    long lo[rb_sim::MAX_CHAINS + 1];
    for (int c = 0; c <= chains; ++c) lo[c] = c == chains ? s->n : (chains == 1 ? 0 : ((s->n * c / chains + 255) / 256) * 256);
    auto chain_stream = [&](int c) { return c ? s->chain_stream[c] : s->stream; };
    bool forked = false;
    auto fork = [&]() -> int {       // the further chains start behind everything the handle's stream holds so far
        if (chains == 1 || forked) return RB_OK;
        if (!s->chain_fork) RB_HIP(hipEventCreateWithFlags(&s->chain_fork, chain_event_flags()));
        for (int c = 1; c < chains; ++c)
            if (!s->chain_stream[c]) {
                RB_HIP(hipStreamCreateWithFlags(&s->chain_stream[c], hipStreamNonBlocking));
                RB_HIP(hipEventCreateWithFlags(&s->chain_join[c], chain_event_flags()));
            }
        // (a stream that reports everything done has nothing for the chains to wait for: saves the two calls' ~5 us of host time in
        // front of the first launch of a short rollout)
        if (capturing(s) || hipStreamQuery(s->stream) != hipSuccess) {
            RB_HIP(hipEventRecord(s->chain_fork, s->stream));
            for (int c = 1; c < chains; ++c) if (lo[c + 1] > lo[c]) RB_HIP(hipStreamWaitEvent(s->chain_stream[c], s->chain_fork, 0));
        }
        (void)hipGetLastError();         // hipErrorNotReady from the query is not an error
        forked = true;
        return RB_OK;
    };
    auto join = [&]() -> int {       // ... and the handle's stream continues behind all of them
        if (!forked) return RB_OK;
        for (int c = 1; c < chains; ++c)
            if (lo[c + 1] > lo[c]) {
                RB_HIP(hipEventRecord(s->chain_join[c], s->chain_stream[c]));
                RB_HIP(hipStreamWaitEvent(s->stream, s->chain_join[c], 0));
            }
        forked = false;
        return RB_OK;
    };
    int t = 0;
    // An eager head in front of the graphs (large batches whose step kernel outlasts a host-side launch): replaying a graph
    // costs ~10 us between the call and its first kernel, and the second chain's graph is launched after the first one's - a
    // 20-step rollout of the headline batch lost 38 us to that (13.9 instead of 12.0 us per step, profiles/r4_a).  One ring
    // turn of plain launches, the chains taking turns, has the device busy after ~4 us; the graphs are enqueued behind them
    // while they run.  The same kernels on the same streams: nothing changes in the results.
    if (use_graph && rollout_eager_head(s, chains) && n_steps >= ring + 8 && ring <= 8) {
        int rc = fork();
        for (int k = 0; k < ring && rc == RB_OK; ++k)
            for (int c = chains - 1; c >= 0 && rc == RB_OK; --c)      // the further chains first (below)
                if (lo[c + 1] > lo[c])
                    rc = chains == 1 ? launch_step(s, d_ring + size_t(k) * slab, act_scale)
                                     : launch_step(s, d_ring + size_t(k) * slab, act_scale, lo[c], lo[c + 1], chain_stream(c));
        if (rc) return rc;
        t = ring;
    }
    // graphs of up to 128 per-step kernel nodes, each a whole number of ring turns (so every
    // graph starts at ring slot 0); what is left over (< 8 steps or a partial turn) is launched eagerly
    while (use_graph) {
        const int left = n_steps - t;
        const int chunk = ((left < 128 ? left : 128) / ring) * ring;
        if (chunk < 8) break;
        rb_sim::GraphKey key;
        std::memset(&key, 0, sizeof(key));
        key.ring_ptr = d_ring; key.ring = ring; key.chunk = chunk; key.scale = act_scale; { const Row *row = row_for(s, ENTRY_STEP, true); key.kernel = (row ? int(row - TABLE) : 0xffff) | (chains << 16); }
        auto it = s->graphs.find(key);
        if (it == s->graphs.end()) {
            { int rc = fork(); if (rc) return rc; }     // creates the chain streams
            // one LINEAR graph of `chunk` launches per chain, captured on the chain's own stream.  (One graph with parallel
            // branches is replayed badly - its second branch starts late: 16.2 us per step in a 20-step region against 13.9 for two
            // linear graphs on two streams.)
            rb_sim::ChainGraphs cg;
            cg.chains = chains;
            for (int c = 0; c < chains; ++c) {
                if (lo[c + 1] <= lo[c]) continue;
                hipStream_t st = chain_stream(c);
                hipGraph_t graph = nullptr;
                hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
                if (e != hipSuccess) { cg.destroy(); return fail(RB_EHIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e)); }
                int rc = RB_OK;
                for (int k = 0; k < chunk && rc == RB_OK; ++k) {
                    const float *slab_k = d_ring + size_t(k % ring) * slab;
                    rc = chains == 1 ? launch_step(s, slab_k, act_scale) : launch_step(s, slab_k, act_scale, lo[c], lo[c + 1], st);
                }
                e = hipStreamEndCapture(st, &graph);
                if (rc != RB_OK) { if (graph) (void)hipGraphDestroy(graph); cg.destroy(); return rc; }
                if (e != hipSuccess) { cg.destroy(); return fail(RB_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); }
                e = hipGraphInstantiate(&cg.exec[c], graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (e != hipSuccess) { cg.destroy(); return fail(RB_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
            }
            if (s->graphs.size() >= 16) {   // bound the cache: callers that keep changing slabs get re-captures, not a leak
                int rc = drop_graphs(s);                   // a cached exec may still be in flight (launched by an earlier chunk, on any chain)
                if (rc) { cg.destroy(); return rc; }
                forked = false;                            // (drained: fork again below)
            }
            it = s->graphs.emplace(key, cg).first;
        }
        // all the chunks of this size: the chains replay their graphs back to back; they are joined once, at the end
        const rb_sim::ChainGraphs &cg = it->second;
        { int rc = fork(); if (rc) return rc; }
        // The further chains' graphs are launched FIRST: with the handle's stream first its chain ran ahead and the other one
        // finished 13 us behind it - a 20-step rollout of the headline batch took 308 us against 272 us in this order, where the
        // two chains end within 6 us of each other (tools/proto/region_timeline_probe.hip, profiles/r4_a/region_timeline.log)
        for (; t + chunk <= n_steps; t += chunk)
            for (int c = cg.chains - 1; c >= 0; --c)
                if (cg.exec[c]) RB_HIP(hipGraphLaunch(cg.exec[c], chain_stream(c)));
    }
    { int rc = join(); if (rc) return rc; }
    for (; t < n_steps; ++t) {
        int rc = launch_step(s, d_ring + size_t(t % ring) * slab, act_scale);
        if (rc) return rc;
    }
    s->env_steps += double(s->n) * n_steps;
    return RB_OK;
}

int rb_rollout_chains(rb_sim *s) {
    if (check(s)) return -1;
    maybe_jit(s);                    // (what rb_rollout_dev would build first)
    return rollout_chains(s);
}

int rb_set_rollout_chains(rb_sim *s, int chains) {
    if (check(s)) return RB_EINVAL;
    if (chains < 0 || chains > rb_sim::MAX_CHAINS) return fail(RB_EINVAL, "chains must be 0 (the library's choice) or 1..4");
    s->chains_choice = chains;      // (graphs are cached per chain count: nothing to drop)
    return RB_OK;
}

int rb_rollout_fused_dev(rb_sim *s, const float *d_ring, int ring, int n_steps, float act_scale) {
    if (check(s) || !d_ring) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    if (ring < 1 || n_steps < 0) return fail(RB_EINVAL, "ring must be >= 1 and n_steps >= 0");
    if (reinterpret_cast<uintptr_t>(d_ring) % 16) return fail(RB_EINVAL, "action ring must be 16-byte aligned");
    if (s->tree || s->ntx) return fail(RB_EUNSUPPORTED, "fused rollout is built for 8-tendon ball-joint robots");
    if (n_steps == 0) return RB_OK;
    const long n = s->n;
    maybe_jit(s);
    Launch L;
    L.i0 = 0; L.cnt = n; L.stream = s->stream; L.act = d_ring; L.act_scale = act_scale; L.ring = ring; L.n_steps = n_steps;
    { int rc = dispatch(s, ENTRY_FUSED, L); if (rc) return rc; }
    s->env_steps += double(n) * n_steps;
    return RB_OK;
}

int rb_fill_actions_dev(rb_sim *s, float *d_act, uint32_t step) {
    if (check(s) || !d_act) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    hipLaunchKernelGGL(fill_actions_kernel, dim3(blocks_for(s->n, 256)), dim3(256), 0, s->stream,
                       d_act, s->n_t, s->n, s->seed, uint64_t(s->env0), step);
    RB_HIP(hipGetLastError());
    return RB_OK;
}

int rb_sample_goals_dev(rb_sim *s, const uint8_t *d_mask, float *d_goal_q) {
    if (check(s) || !d_goal_q) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    hipLaunchKernelGGL(sample_goals_kernel, dim3(blocks_for(s->n, 256)), dim3(256), 0, s->stream,
                       d_goal_q, s->d_goal_count, d_mask, s->box, s->n_q, s->n, s->seed, uint64_t(s->env0), 0);
    RB_HIP(hipGetLastError());
    return RB_OK;
}

int rb_env_configure(rb_sim *s, const rb_env_config *cfg) {
    if (check(s) || !cfg) return fail(RB_EINVAL, "null argument");
    if (!s->tree && s->n_q != 3) return fail(RB_EUNSUPPORTED, "fused env layer: unsupported robot");
    if (cfg->max_episode_length < 1) return fail(RB_EINVAL, "max_episode_length must be >= 1");
    if (!(cfg->angle_hi > cfg->angle_lo) || !(cfg->vel_hi > cfg->vel_lo) || !(cfg->action_hi > cfg->action_lo))
        return fail(RB_EINVAL, "empty box in env config");
    RB_HIP(hipSetDevice(s->device));
    EnvParams &e = s->env;
    e.vel_penalty = cfg->joint_vel_penalty != 0; e.bonus = cfg->goal_bonus != 0;
    e.max_len = cfg->max_episode_length; e.auto_reset = cfg->auto_reset != 0;
    e.penalty = cfg->penalty_boundary; e.bonus_val = cfg->bonus_goal;
    e.a_lo = cfg->angle_lo; e.a_hi = cfg->angle_hi; e.v_lo = cfg->vel_lo; e.v_hi = cfg->vel_hi;
    e.act_hi = cfg->action_hi;
    // (out.high - out.low) / (in.high - in.low) evaluated in fp32 like the reference's float32 Boxes
    e.slope = (cfg->action_hi - cfg->action_lo) / (1.0f - (-1.0f));
    e.tol_a2 = cfg->goal_angle_tol * cfg->goal_angle_tol; e.tol_v2 = cfg->goal_vel_tol * cfg->goal_vel_tol;
    e.a_scale = 2.0f / (cfg->angle_hi - cfg->angle_lo); e.v_scale = 2.0f / (cfg->vel_hi - cfg->vel_lo);
    if (!s->d_goal) {
        const size_t plane = sizeof(float) * size_t(s->n);
        RB_HIP(hipMalloc(&s->d_goal, plane * s->n_q));
        RB_HIP(hipMalloc(&s->d_ep_ret, plane));
        RB_HIP(hipMalloc(&s->d_step_num, sizeof(uint32_t) * size_t(s->n)));
        RB_HIP(hipMalloc(&s->d_ep_sum, sizeof(double) * size_t(s->n) * 2));
        RB_HIP(hipMalloc(&s->d_ep_cnt, sizeof(uint32_t) * size_t(s->n) * 3));
    }
    RB_HIP(hipMemsetAsync(s->d_ep_sum, 0, sizeof(double) * size_t(s->n) * 2, s->stream));
    RB_HIP(hipMemsetAsync(s->d_ep_cnt, 0, sizeof(uint32_t) * size_t(s->n) * 3, s->stream));
    RB_HIP(hipMemsetAsync(s->d_infeas_n, 0, sizeof(uint32_t) * size_t(s->n), s->stream));
    s->env_steps = 0.0;
    s->env_ready = true;
    // the run-time specialised kernels are built here, outside any capture a caller may wrap around its first step
    maybe_jit(s);
    if (s->tree && tree_wants_lane(s) && !s->lane_baked) (void)lane_kernel(s, 1);
    if (s->tree && tree_wants_split(s) && !s->split_baked) (void)build_split_kernel(s, 1);
    return rb_env_reset_dev(s, nullptr);
}

int rb_env_reset_dev(rb_sim *s, float *d_obs) {
    if (check(s)) return RB_EINVAL;
    RB_HIP(hipSetDevice(s->device));
    if (!s->env_ready) return fail(RB_EINVAL, "rb_env_configure has not been called");
    if (s->tree)
        hipLaunchKernelGGL(rbt::tree_env_reset_kernel, dim3(blocks_for(s->n, 256)), dim3(256), 0, s->stream,
                           s->box, s->d_q, s->d_qd, s->d_feas, s->d_goal, s->d_step_num, s->d_ep_ret,
                           s->d_goal_count, d_obs, s->n_q, s->n, s->seed, uint64_t(s->env0));
    else
        hipLaunchKernelGGL(env_reset_kernel, dim3(blocks_for(s->n, 256)), dim3(256), 0, s->stream,
                           s->box, s->d_q, s->d_qd, s->d_feas, s->d_goal, s->d_step_num, s->d_ep_ret,
                           s->d_goal_count, d_obs, s->n, s->seed, uint64_t(s->env0));
    RB_HIP(hipGetLastError());
    return RB_OK;
}

int rb_env_set_goal(rb_sim *s, const float *goal_q, const uint32_t *step_num) {
    if (check(s) || !goal_q) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    if (!s->env_ready) return fail(RB_EINVAL, "rb_env_configure has not been called");
    const long n = s->n;
    RB_HIP(hipMemcpyAsync(s->d_rows, goal_q, sizeof(float) * n * s->n_q, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s->stream, s->d_rows, s->d_goal, s->n_q, n, s->tree ? 1 : 0);
    RB_HIP(hipGetLastError());
    if (step_num)
        RB_HIP(hipMemcpyAsync(s->d_step_num, step_num, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s->stream));
    RB_HIP(hipStreamSynchronize(s->stream));
    return RB_OK;
}

// RoboyEnv.step for envs [i0, i0 + cnt) on `stream`; the array arguments are those of the whole batch (the launchers apply the
// offsets).  Sub-ranges: every ball-joint form and the joint trees' one-wave-per-64-envs form (Row::ranges).
static int env_step_launch(rb_sim *s, long i0, long cnt, hipStream_t stream, const float *d_act, float *d_obs, float *d_reward, uint32_t *d_done) {
    Launch L;
    L.i0 = i0; L.cnt = cnt; L.stream = stream; L.act = d_act; L.obs = d_obs; L.reward = d_reward; L.done = d_done;
    return dispatch(s, ENTRY_ENV, L);
}

int rb_env_step_dev(rb_sim *s, const float *d_act, float *d_obs, float *d_reward, uint32_t *d_done) {
    if (check(s) || !d_act || !d_obs || !d_reward || !d_done) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    if (!s->env_ready) return fail(RB_EINVAL, "rb_env_configure has not been called");
    if (!s->tree && reinterpret_cast<uintptr_t>(d_act) % 16) return fail(RB_EINVAL, "action slab must be 16-byte aligned");
    int rc = env_step_launch(s, 0, s->n, s->stream, d_act, d_obs, d_reward, d_done);
    if (rc == RB_OK) s->env_steps += double(s->n);
    return rc;
}

// a sub-range on a stream of the caller's choice: envs are independent, so disjoint ranges may be stepped concurrently
static int range_ok(const rb_sim *s, int64_t first, int64_t count) {
    if (first < 0 || count < 1 || first + count > s->n) return fail(RB_EINVAL, "range outside the batch");
    if (first % 256) return fail(RB_EINVAL, "a range starts at a multiple of 256 envs");
    return RB_OK;
}
int rb_env_step_range_dev(rb_sim *s, int64_t first_env, int64_t n_envs, void *hip_stream, const float *d_act, float *d_obs, float *d_reward, uint32_t *d_done) {
    if (check(s) || !d_act || !d_obs || !d_reward || !d_done) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    if (!s->env_ready) return fail(RB_EINVAL, "rb_env_configure has not been called");
    if (!s->tree && reinterpret_cast<uintptr_t>(d_act) % 16) return fail(RB_EINVAL, "action slab must be 16-byte aligned");
    if (range_ok(s, first_env, n_envs)) return RB_EINVAL;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->stream;
    CallStreamScope scope(s, st);          // a first call captured on the caller's stream defers the run-time builds (capturing())
    int rc = env_step_launch(s, long(first_env), long(n_envs), st, d_act, d_obs, d_reward, d_done);
    if (rc == RB_OK) { s->env_steps += double(n_envs); rc = note_caller_stream(s, st); }
    return rc;
}
int rb_step_range_dev(rb_sim *s, int64_t first_env, int64_t n_envs, void *hip_stream, const float *d_act, float act_scale) {
    if (check(s) || !d_act) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    if (reinterpret_cast<uintptr_t>(d_act) % 16) return fail(RB_EINVAL, "action slab must be 16-byte aligned");
    if (range_ok(s, first_env, n_envs)) return RB_EINVAL;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->stream;
    CallStreamScope scope(s, st);          // a first call captured on the caller's stream defers the run-time builds (capturing())
    maybe_jit(s);
    // (a whole batch is taken by every form, on the caller's stream too; sub-ranges by the forms of rb_range_capable - dispatch() refuses others)
    int rc = launch_step(s, d_act, act_scale, long(first_env), long(first_env + n_envs), st);
    if (rc == RB_OK) { s->env_steps += double(n_envs); rc = note_caller_stream(s, st); }
    return rc;
}
int rb_range_capable(rb_sim *s) {
    if (check(s)) return -1;
    // builds what a launch would build (outside captures), then asks the table: bit 0 = the step's row takes sub-ranges, bit 1 = the env step's
    maybe_jit(s);
    const Row *step = row_for(s, ENTRY_STEP, true), *env = row_for(s, ENTRY_ENV, s->env_ready);
    return (step && step->ranges ? 1 : 0) | (env && env->ranges ? 2 : 0);
}

static int stats_launch(rb_sim *s, double *d_out2) {
    unsigned g = blocks_for(s->n, 256);
    if (g > unsigned(STATS_BLOCKS)) g = STATS_BLOCKS;
    double *partials = s->d_stats + 8;
    unsigned int *ticket = reinterpret_cast<unsigned int *>(s->d_stats + 8 * (STATS_BLOCKS + 1));
    if (s->env_ready)
        hipLaunchKernelGGL(stats_reduce_kernel<true>, dim3(g), dim3(256), 0, s->stream, s->d_ep_sum, s->d_ep_cnt, s->d_ep_ret,
                           s->d_infeas_n, s->d_feas, partials, ticket, s->d_stats, d_out2, s->env_steps, s->n);
    else
        hipLaunchKernelGGL(stats_reduce_kernel<false>, dim3(g), dim3(256), 0, s->stream, nullptr, s->d_ep_cnt, s->d_ep_ret,
                           s->d_infeas_n, s->d_feas, partials, ticket, s->d_stats, d_out2, s->env_steps, s->n);
    RB_HIP(hipGetLastError());
    return RB_OK;
}
static int stats_reset(rb_sim *s) {
    if (s->env_ready) {
        RB_HIP(hipMemsetAsync(s->d_ep_sum, 0, sizeof(double) * size_t(s->n) * 2, s->stream));
        RB_HIP(hipMemsetAsync(s->d_ep_cnt, 0, sizeof(uint32_t) * size_t(s->n) * 3, s->stream));
    }
    RB_HIP(hipMemsetAsync(s->d_infeas_n, 0, sizeof(uint32_t) * size_t(s->n), s->stream));
    s->env_steps = 0.0;
    return RB_OK;
}

int rb_env_stats_dev(rb_sim *s, double *d_stats8, int reset) {
    if (check(s) || !d_stats8) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    int rc = stats_launch(s, d_stats8);
    if (rc) return rc;
    return reset ? stats_reset(s) : RB_OK;
}

int rb_env_stats(rb_sim *s, double *stats8, int reset) {
    if (check(s) || !stats8) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    int rc = stats_launch(s, nullptr);
    if (rc) return rc;
    RB_HIP(hipMemcpyAsync(stats8, s->d_stats, sizeof(double) * 8, hipMemcpyDeviceToHost, s->stream));
    RB_HIP(hipStreamSynchronize(s->stream));
    return reset ? stats_reset(s) : RB_OK;
}

static rb_dispatch_row g_rows[N_ROWS];
static rb_dispatch_row public_row(const Row &r) {
    return rb_dispatch_row{r.key.cls, r.key.entry, r.key.form, r.key.integ, r.key.block, r.key.src, r.key.variant, r.ranges ? 1 : 0};
}
int rb_dispatch_rows(const rb_dispatch_row **rows) {
    static const bool filled = [] { for (int i = 0; i < N_ROWS; ++i) g_rows[i] = public_row(TABLE[i]); return true; }();
    (void)filled;
    if (rows) *rows = g_rows;
    return N_ROWS;
}
int rb_auto_rules(const rb_auto_rule **rules) {
    if (rules) *rules = AUTO_RULES;
    return N_AUTO_RULES;
}
int rb_get_launch_thresholds(rb_launch_thresholds *out) {
    if (!out) return fail(RB_EINVAL, "null argument");
    *out = rb_launch_thresholds{RB_SMALL_BATCH, RB_PAIR_SMALL_BATCH, RB_CHAIN_BATCH_RK4, RB_CHAIN_BATCH_EULER, RB_CHAIN_BATCH_TREE_RK4,
                                RB_CHAIN_BATCH_TREE_EULER, RB_EAGER_HEAD_BATCH_RK4, RB_TREE_JIT_BATCH};
    return RB_OK;
}
int rb_dispatch_current(rb_sim *s, int entry, rb_dispatch_row *out) {
    if (check(s) || !out) return fail(RB_EINVAL, "null argument");
    if (entry < ENTRY_STEP || entry > ENTRY_FUSED) return fail(RB_EINVAL, "unknown entry kind");
    if (entry == ENTRY_FUSED && (s->tree || s->ntx)) return fail(RB_EUNSUPPORTED, "fused rollout is built for 8-tendon ball-joint robots");
    RB_HIP(hipSetDevice(s->device));
    if (entry != ENTRY_ENV || s->env_ready) maybe_jit(s);
    std::string why;
    const Row *r = row_for(s, entry, /*build=*/entry != ENTRY_ENV || s->env_ready, &why);
    if (!r) return fail(RB_EUNSUPPORTED, why);
    *out = public_row(*r);
    return RB_OK;
}

int rb_malloc(rb_sim *s, int64_t bytes, void **d_ptr) {
    if (check(s) || !d_ptr || bytes < 0) return fail(RB_EINVAL, "bad argument");
    RB_HIP(hipSetDevice(s->device));
    RB_HIP(hipMalloc(d_ptr, size_t(bytes)));
    return RB_OK;
}
int rb_free(rb_sim *s, void *d_ptr) {
    if (check(s)) return RB_EINVAL;
    RB_HIP(hipSetDevice(s->device));
    RB_HIP(hipFree(d_ptr));
    return RB_OK;
}
int rb_memcpy_h2d(rb_sim *s, void *d_dst, const void *h_src, int64_t bytes) {
    if (check(s) || !d_dst || !h_src) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    RB_HIP(hipMemcpyAsync(d_dst, h_src, size_t(bytes), hipMemcpyHostToDevice, s->stream));
    RB_HIP(hipStreamSynchronize(s->stream));
    return RB_OK;
}
int rb_memcpy_d2h(rb_sim *s, void *h_dst, const void *d_src, int64_t bytes) {
    if (check(s) || !h_dst || !d_src) return fail(RB_EINVAL, "null argument");
    RB_HIP(hipSetDevice(s->device));
    RB_HIP(hipMemcpyAsync(h_dst, d_src, size_t(bytes), hipMemcpyDeviceToHost, s->stream));
    RB_HIP(hipStreamSynchronize(s->stream));
    return RB_OK;
}

}  // extern "C"
