// tree_lane.hpp - env-per-lane kernels for joint-tree robots around a GENERATED acceleration function.
//
// Included after the text tree_lane_gen.hpp writes for one robot (RBL_NS, RBL_NQ, RBL_NT, RBL_ACCEL_LDS, the
// tables KSG / QLO / QHI / VMAX and RBL_NS::rbl_accel): by roboy_sim.hip for the committed upper body
// (tree_lane_baked.hpp) and by hiprtc for any other robot (tree_lane_jit.hpp).  One lane owns one env for the
// whole step; a wave owns 64 consecutive envs.
//
//   * Registers: a workgroup is one wave (64 threads) and may use the whole 512-entry VGPR + AGPR file of a SIMD
//     (one wave per SIMD, four workgroups per CU by their LDS regions); the acceleration's live set (~450 values
//     on the upper body) stays on chip.
//   * LDS: RBL_ACCEL_LDS lane-private slots for values the acceleration parks between its sweeps, plus
//     2 RBL_NQ slots for the RK4 accumulators.  Slot s of lane l is word s * 64 + l of the wave's region:
//     every access is conflict-free and needs no address arithmetic beyond the immediate offset.
//   * HBM: the state rows q[n][n_q], qd[n][n_q], the action rows [n][n_t] (and goal / observation rows in the
//     env-layer kernel) are env-major, as the octet kernels of tree_aba.hpp keep them.  A wave moves its 64
//     rows as one contiguous run with coalesced dword accesses and transposes it through the (then idle) LDS
//     region, all input rows of a step in one batch of loads (one memory latency).
// Algorithmic HBM bytes per env step: 4 (4 n_q + n_t + 1), as for every kernel of the library.
#pragma once
#include "env_common.hpp"
#include "philox.hpp"

#ifndef RBL_NS
#error "include the generated robot header (tree_lane_gen.hpp) first"
#endif

namespace RBL_NS {

struct LaneLds {
    float *p;   // the wave's region + lane
    __device__ __forceinline__ float &operator()(int slot) const { return p[slot * 64]; }
};

constexpr int STAGE_SLOTS = 5 * RBL_NQ > 3 * RBL_NQ + RBL_NT ? 5 * RBL_NQ : 3 * RBL_NQ + RBL_NT;   // widest row set a wave transposes (q, qd, obs out | q, qd, goal, act in)
constexpr int REGION_SLOTS = RBL_ACCEL_LDS > STAGE_SLOTS ? RBL_ACCEL_LDS : STAGE_SLOTS;  // acceleration slots, aliased by the transposes
constexpr int ACC_SLOT = REGION_SLOTS;                                                   // RK4 accumulators: qa at ACC_SLOT + j, va at ACC_SLOT + NQ + j
constexpr int LDS_SLOTS = REGION_SLOTS + 2 * RBL_NQ;
constexpr int LDS_BYTES_PER_WAVE = LDS_SLOTS * 64 * 4;

__device__ __forceinline__ void lane_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The rows of a wave as a buffer resource: base = the wave's first row, records = the bytes of its live rows.  A load
// past the end returns 0 and a store past the end is dropped by the range check, so neither needs a guard: a guarded
// load is a branch with its own s_waitcnt vmcnt(0), and W of those in a row cost W memory latencies with nothing else
// on the SIMD to cover them (measured: 63 % of a wave's cycles), and one 64-bit address pair per load.  The
// descriptor is built from readfirstlane'd values (the wave index comes from threadIdx, which the compiler cannot
// prove uniform: cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const float *g, long env0, int width, int live) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(g + env0 * width);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(a)), hi = __builtin_amdgcn_readfirstlane(uint32_t(a >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane(live * width * 4);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>((uintptr_t(hi) << 32) | lo), 0, bytes, 0x00020000);
}

// The lane index as a value the optimiser cannot trace: the LDS addresses derived from it are then recomputed where
// they are used (a few integer adds) instead of being hoisted out of the stage / substep loops as loop invariants -
// two dozen registers held across the acceleration, which is where the kernel has none to spare (they were spilled
// to scratch and reloaded inside the loop, each reload a full memory latency with nothing to cover it).
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// Nothing moves across this point, neither in the optimiser (memory accesses) nor in the instruction scheduler: what
// follows the acceleration (accumulator slots, row loads) must not be started before it, into its register budget.
__device__ __forceinline__ void fence_code() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// out[W] of each lane -> rows [64][W] of a wave, env-major in HBM (live = envs of this wave inside the batch)
template <int W>
__device__ __forceinline__ void store_rows(float *__restrict__ g, long env0, int live, float *region, int lane_, const float (&in)[W]) {
    const int lane = opaque(lane_);
#pragma unroll
    for (int j = 0; j < W; ++j) region[lane * W + j] = in[j];
    lane_wave_sync();
    const __amdgpu_buffer_rsrc_t r = rows_rsrc(g, env0, W, live);
#pragma unroll
    for (int k = 0; k < W; ++k) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(region[k * 64 + lane]), r, lane * 4, k * 256, 0);
    lane_wave_sync();
}

// rows [64][W] of a wave, env-major in HBM -> out[W] of each lane (the inverse of store_rows; lanes past the batch read the last live row)
template <int W>
__device__ __forceinline__ void load_rows(const float *__restrict__ g, long env0, int live, float *region, int lane_, float (&out)[W]) {
    const int lane = opaque(lane_);
    const __amdgpu_buffer_rsrc_t r = rows_rsrc(g, env0, W, live);
    float t[W];
#pragma unroll
    for (int k = 0; k < W; ++k) t[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, k * 256, 0));
#pragma unroll
    for (int k = 0; k < W; ++k) region[k * 64 + lane] = t[k];
    lane_wave_sync();
    const int row = lane < live ? lane : live - 1;
#pragma unroll
    for (int j = 0; j < W; ++j) out[j] = region[row * W + j];
    lane_wave_sync();
}

__device__ __forceinline__ float sat(float v, int j) { return __builtin_amdgcn_fmed3f(v, -VMAX[j], VMAX[j]); }

// One env step of this lane's env: n_sub integrator substeps with the activation offsets held; velocity saturation
// and joint limits as in tree_aba.hpp tree_integrate.  q / v hold the state on entry and the new state on return;
// returns false if a joint hit a limit.  What is live across the acceleration besides its own ~370 registers: the
// activation offsets (its input anyway), q (Euler) or q0, v0 (RK4: 2 n_q) - inside the 512-register file of a
// SIMD's only wave on the upper body; the RK4 sums go to LDS.
template <int INTEG>
__device__ __forceinline__ bool lane_step(const LaneLds &L, const float (&spu)[RBL_NT], float h, int nsub, float (&q)[RBL_NQ], float (&v)[RBL_NQ]) {
    bool ok = true;
    for (int sub = 0; sub < nsub; ++sub) {
        if (INTEG == 0) {
            float a[RBL_NQ];
            fence_code();
            rbl_accel(q, v, spu, a, L);
            fence_code();
#pragma unroll
            for (int j = 0; j < RBL_NQ; ++j) { v[j] = sat(v[j] + h * a[j], j); q[j] = q[j] + h * v[j]; }
        } else {
            // RK4 with every stage velocity saturated, as a loop over the four stages (one copy of the acceleration
            // code); the weighted sums live in LDS and accumulate in the order k1 + 2 k2 + 2 k3 + k4
            const float h6 = h * (1.0f / 6.0f);
            float kq[RBL_NQ], kv[RBL_NQ];
#pragma unroll
            for (int j = 0; j < RBL_NQ; ++j) { kq[j] = 0.0f; kv[j] = 0.0f; L(ACC_SLOT + j) = 0.0f; L(ACC_SLOT + RBL_NQ + j) = 0.0f; }
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                const float wgt = (st == 0 || st == 3) ? 1.0f : 2.0f, cst = st == 0 ? 0.0f : (st == 3 ? h : 0.5f * h);
                float qs[RBL_NQ];
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j) {
                    qs[j] = q[j] + cst * kq[j];                   // state of this stage ...
                    kq[j] = sat(v[j] + cst * kv[j], j);            // ... and its (saturated) velocity
                }
                // the velocity's share of the sums before the acceleration
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j) L(ACC_SLOT + j) += wgt * kq[j];
                fence_code();
                rbl_accel(qs, kq, spu, kv, L);
                fence_code();
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j) L(ACC_SLOT + RBL_NQ + j) += wgt * kv[j];
            }
#pragma unroll
            for (int j = 0; j < RBL_NQ; ++j) { q[j] = q[j] + h6 * L(ACC_SLOT + j); v[j] = v[j] + h6 * L(ACC_SLOT + RBL_NQ + j); }
        }
#pragma unroll
        for (int j = 0; j < RBL_NQ; ++j) {
            float vv = sat(v[j], j);
            const bool over = q[j] > QHI[j], under = q[j] < QLO[j];
            if (over) { q[j] = QHI[j]; vv = fminf(vv, 0.0f); }
            if (under) { q[j] = QLO[j]; vv = fmaxf(vv, 0.0f); }
            v[j] = vv;
            ok = ok && !(over || under);
        }
    }
    return ok;
}

// the input row sets of a step at once (GOAL: the env layer's goal rows too): every load is in flight before the
// first is waited for - one memory latency per step, not one per array
template <bool GOAL>
__device__ __forceinline__ void load_inputs(const float *__restrict__ gq, const float *__restrict__ gqd, const float *__restrict__ act,
                                            const float *__restrict__ ggoal, long env0, int live, float *region, int lane_,
                                            float (&q)[RBL_NQ], float (&v)[RBL_NQ], float (&a)[RBL_NT], float (&gl)[RBL_NQ]) {
    const int lane = opaque(lane_);
    const __amdgpu_buffer_rsrc_t rq = rows_rsrc(gq, env0, RBL_NQ, live), rv = rows_rsrc(gqd, env0, RBL_NQ, live), ra = rows_rsrc(act, env0, RBL_NT, live);
    float tq[RBL_NQ], tv[RBL_NQ], ta[RBL_NT], tg[RBL_NQ];
#pragma unroll
    for (int k = 0; k < RBL_NQ; ++k) {
        tq[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rq, lane * 4, k * 256, 0));
        tv[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, lane * 4, k * 256, 0));
    }
#pragma unroll
    for (int k = 0; k < RBL_NT; ++k) ta[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, lane * 4, k * 256, 0));
    if (GOAL) {
        const __amdgpu_buffer_rsrc_t rg = rows_rsrc(ggoal, env0, RBL_NQ, live);
#pragma unroll
        for (int k = 0; k < RBL_NQ; ++k) tg[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, lane * 4, k * 256, 0));
    }
    constexpr int OV = RBL_NQ * 64, OA = 2 * RBL_NQ * 64, OG = (2 * RBL_NQ + RBL_NT) * 64;
#pragma unroll
    for (int k = 0; k < RBL_NQ; ++k) { region[k * 64 + lane] = tq[k]; region[OV + k * 64 + lane] = tv[k]; }
#pragma unroll
    for (int k = 0; k < RBL_NT; ++k) region[OA + k * 64 + lane] = ta[k];
    if (GOAL) {
#pragma unroll
        for (int k = 0; k < RBL_NQ; ++k) region[OG + k * 64 + lane] = tg[k];
    }
    lane_wave_sync();
    const int row = lane < live ? lane : live - 1;
#pragma unroll
    for (int j = 0; j < RBL_NQ; ++j) { q[j] = region[row * RBL_NQ + j]; v[j] = region[OV + row * RBL_NQ + j]; }
#pragma unroll
    for (int k = 0; k < RBL_NT; ++k) a[k] = region[OA + row * RBL_NT + k];
    if (GOAL) {
#pragma unroll
        for (int j = 0; j < RBL_NQ; ++j) gl[j] = region[OG + row * RBL_NQ + j];
    }
    lane_wave_sync();
}

// the output row sets of a step at once: one transpose, then every store back to back (q | qd, then - env layer -
// the observation rows, which repeat them in front of the goal)
template <bool OBS>
__device__ __forceinline__ void store_outputs(float *__restrict__ gq, float *__restrict__ gqd, float *__restrict__ gobs, long env0, int live,
                                              float *region, int lane_, const float (&q)[RBL_NQ], const float (&v)[RBL_NQ],
                                              const float (&o)[3 * RBL_NQ]) {
    const int lane = opaque(lane_);
    constexpr int OV = RBL_NQ * 64, OO = 2 * RBL_NQ * 64;
#pragma unroll
    for (int j = 0; j < RBL_NQ; ++j) { region[lane * RBL_NQ + j] = q[j]; region[OV + lane * RBL_NQ + j] = v[j]; }
    if (OBS) {
#pragma unroll
        for (int j = 0; j < 3 * RBL_NQ; ++j) region[OO + lane * (3 * RBL_NQ) + j] = o[j];
    }
    lane_wave_sync();
    const __amdgpu_buffer_rsrc_t rq = rows_rsrc(gq, env0, RBL_NQ, live), rv = rows_rsrc(gqd, env0, RBL_NQ, live);
#pragma unroll
    for (int k = 0; k < RBL_NQ; ++k) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(region[k * 64 + lane]), rq, lane * 4, k * 256, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(region[OV + k * 64 + lane]), rv, lane * 4, k * 256, 0);
    }
    if (OBS) {
        const __amdgpu_buffer_rsrc_t ro = rows_rsrc(gobs, env0, 3 * RBL_NQ, live);
#pragma unroll
        for (int k = 0; k < 3 * RBL_NQ; ++k) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(region[OO + k * 64 + lane]), ro, lane * 4, k * 256, 0);
    }
    lane_wave_sync();
}

#if defined(RB_LANE_STAMPS)
// diagnostic builds only (tools/lane_stamps.py): per wave (the first RBL_STAMP_WAVES workgroups) the constant 100 MHz clock and the
// shader clock at the kernel's entry, behind the input rows, behind the step and behind the output rows; rb_debug_lane_stamps
// fetches the buffer.  No product build defines RB_LANE_STAMPS.
#define RBL_STAMP_WAVES 4096
__device__ unsigned long long rbl_lane_stamp_buf[RBL_STAMP_WAVES * 8];
__device__ __forceinline__ void rbl_lane_stamp(int k) {
    if (blockIdx.x < RBL_STAMP_WAVES && threadIdx.x == 0) {
        rbl_lane_stamp_buf[blockIdx.x * 8 + 2 * k] = __builtin_amdgcn_s_memrealtime();
        rbl_lane_stamp_buf[blockIdx.x * 8 + 2 * k + 1] = __builtin_amdgcn_s_memtime();
    }
}
#define RBL_LANE_STAMP(k) rbl_lane_stamp(k)
#else
#define RBL_LANE_STAMP(k)
#endif

// forward_step_command for a batch: act rows are set-points scaled by act_scale (tree_step_aba's contract)
template <int INTEG>
__global__ void __launch_bounds__(64)
tree_lane_step(float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas, const float *__restrict__ act,
               float act_scale, float h, int nsub, long n) {
    extern __shared__ float lds_lane[];
    const int lane = threadIdx.x;
    const long env0 = long(blockIdx.x) * 64;      // a workgroup is ONE wave: everything derived from the block index is scalar
    if (env0 >= n) return;
    const int live = n - env0 < 64 ? int(n - env0) : 64;
    float *region = lds_lane;
    const LaneLds L{region + lane};
    float qq[RBL_NQ], vv[RBL_NQ], spu[RBL_NT], none[RBL_NQ];
    RBL_LANE_STAMP(0);
    load_inputs<false>(q, qd, act, nullptr, env0, live, region, lane, qq, vv, spu, none);
    RBL_LANE_STAMP(1);
#pragma unroll
    for (int k = 0; k < RBL_NT; ++k) spu[k] = rbe::rounded_here((spu[k] * act_scale) * KSG[k]);
    const bool ok = lane_step<INTEG>(L, spu, h, nsub, qq, vv);
    RBL_LANE_STAMP(2);
    {
        float no_obs[3 * RBL_NQ];
        store_outputs<false>(q, qd, nullptr, env0, live, region, lane, qq, vv, no_obs);
    }
    if (lane < live) feas[env0 + lane] = ok ? 1u : 0u;
    RBL_LANE_STAMP(3);
}

// RoboyEnv.step fused around the step (semantics of tree_env_step_aba / msj_env_step_kernel)
template <int INTEG>
__global__ void __launch_bounds__(64)
tree_lane_env_step(const rbe::TreeEnvArgs a) {
    // a.n: the envs of this launch (the whole batch, or a sub-range on shifted pointers: rb_env_step_range_dev); a.stat_stride: the
    // batch's env count = the stride of the per-env statistics planes ep_sum[2][.], ep_cnt[3][.].  The front of the kernel takes what
    // it needs by name; the accounting behind the step reads its arguments through `late` (env_common.hpp: TreeEnvArgs)
    const float *__restrict__ q = a.q, *__restrict__ qd = a.qd, *__restrict__ act = a.act, *__restrict__ goal = a.goal;
    const uint32_t *__restrict__ step_num = a.step_num, *__restrict__ goal_count = a.goal_count;
    const float *__restrict__ ep_ret = a.ep_ret;
    const float h = a.h, slope = a.ep.slope, act_hi = a.ep.act_hi;
    const int nsub = a.nsub;
    const long n = a.n;
    extern __shared__ float lds_lane[];
    const int lane = threadIdx.x;
    const long env0 = long(blockIdx.x) * 64;      // a workgroup is ONE wave: everything derived from the block index is scalar
    if (env0 >= n) return;
    const int live = n - env0 < 64 ? int(n - env0) : 64;
    float *region = lds_lane;
    const LaneLds L{region + lane};
    float qq[RBL_NQ], vv[RBL_NQ], spu[RBL_NT], gg[RBL_NQ];
    // What the accounting needs of the env's OLD state - its goal row and three counters - is fetched in front of the step (one batch of
    // loads with the state rows: their latency passes behind the acceleration) where the register file can carry n_q + 3 more values
    // across the step: Euler.  RK4 holds q0, v0 and the stage state beside the acceleration's live set - with the goal on top the
    // compiler put 13 values into scratch (56 bytes per lane, stored in front of the step and reloaded behind it: a private segment,
    // and a memory round trip anyway): there the goal rows and the counters are fetched BEHIND the step (RBL_LATE_GOAL bit 1; bit 0 =
    // Euler too).  Same values either way - nobody writes them during the step.
#ifndef RBL_LATE_GOAL
#define RBL_LATE_GOAL 2
#endif
    constexpr bool LATE_GOAL = ((RBL_LATE_GOAL >> INTEG) & 1) != 0;
    RBL_LANE_STAMP(0);
    load_inputs<!LATE_GOAL>(q, qd, act, goal, env0, live, region, lane, qq, vv, spu, gg);
    RBL_LANE_STAMP(1);
    const bool mine = lane < live;
    const long me = env0 + (mine ? lane : live - 1);
    uint32_t sn_old = 0u, draw_old = 0u;
    float ret_old = 0.0f;
    if constexpr (!LATE_GOAL) {
        // the env's counters, requested now as well (their latency passes behind the acceleration)
        sn_old = step_num[me];
        ret_old = ep_ret[me];
        draw_old = goal_count[me];       // (needed when the episode ends only - but then it would be a memory latency of its own)
    }
#pragma unroll
    for (int k = 0; k < RBL_NT; ++k) {
        // clamp to the action box, then slope * (x - in_high) + out_high with two roundings (roboy_env.py:157-158)
        const float x = fminf(fmaxf(spu[k], -1.0f), 1.0f);
        spu[k] = rbe::rounded_here(rbe::mul_then_add(slope, x - 1.0f, act_hi) * KSG[k]);
    }
    const bool ok = lane_step<INTEG>(L, spu, h, nsub, qq, vv);
    RBL_LANE_STAMP(2);
    const rbe::tree_env_kernarg_ptr late = rbe::late_args();
    const rbe::EnvParams ep = rbe::late_env_params(late);
    const uint64_t seed = late->seed, env_id0 = late->env_id0;
    const long stat_stride = late->stat_stride;
    if constexpr (LATE_GOAL) {
        sn_old = late->step_num[me];
        ret_old = late->ep_ret[me];
        draw_old = late->goal_count[me];
        load_rows<RBL_NQ>(late->goal, env0, live, region, lane, gg);      // (the region is idle again)
    }
    // observation [q | qd | goal], reward, done; goal redraw (and reset) on done
    float o[3 * RBL_NQ];
#pragma unroll
    for (int j = 0; j < RBL_NQ; ++j) { o[j] = qq[j]; o[RBL_NQ + j] = vv[j]; o[2 * RBL_NQ + j] = gg[j]; }
    float dq2 = 0.0f, dv2 = 0.0f;
#pragma unroll
    for (int j = 0; j < RBL_NQ; ++j) {
        const float dq = o[j] - o[2 * RBL_NQ + j];
        dq2 += dq * dq; dv2 += o[RBL_NQ + j] * o[RBL_NQ + j];
    }
    uint32_t sn = sn_old + 1u;
    bool reached;
    const float r = rbe::env_reward(ep, dq2, dv2, ok, reached);
    const bool dn = reached || (sn > uint32_t(ep.max_len));
    float ret = ret_old + r;
    uint32_t fz = ok ? 1u : 0u;
    float gn[RBL_NQ];                                     // the goal after this step (redrawn on done)
#pragma unroll
    for (int j = 0; j < RBL_NQ; ++j) gn[j] = o[2 * RBL_NQ + j];
    if (dn) {
        const uint64_t gid = env_id0 + uint64_t(me);
        uint32_t draw = draw_old;
        auto draw_goals = [&](uint32_t dnum) {
#pragma unroll
            for (int b = 0; 4 * b < RBL_NQ; ++b) {
                const rb::Philox4 rnd = rb::philox_draw(seed, gid, dnum, rb::STREAM_GOALS, uint32_t(b));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * b + k < RBL_NQ) gn[4 * b + k] = rbe::goal_value(late->box.lo[4 * b + k], late->box.hi[4 * b + k], rnd.v[k]);
            }
        };
        // RoboyEnv.step: _set_new_goal (:67-68), AFTER the observation was made (:60); the VecEnv worker's env.reset() (:82-87) draws
        // again and the reset observation replaces the step's: the first draw is never seen - the counter advances by two, only the
        // second draw is evaluated (380 instead of 751 integer instructions in every wave that has a finished episode - and a launch
        // waits for its slowest wave)
        draw_goals(draw + (ep.auto_reset ? 1u : 0u));
        draw += ep.auto_reset ? 2u : 1u;
        if (ep.auto_reset) {
#pragma unroll
            for (int j = 0; j < RBL_NQ; ++j) { qq[j] = 0.0f; vv[j] = 0.0f; o[j] = 0.0f; o[RBL_NQ + j] = 0.0f; o[2 * RBL_NQ + j] = gn[j]; }
        }
        if (mine) {
            double *ep_sum = late->ep_sum;
            uint32_t *ep_cnt = late->ep_cnt;
            rbe::stat_add(&ep_sum[me], double(ret)); rbe::stat_add(&ep_sum[stat_stride + me], double(ret) * double(ret));
            rbe::stat_add(&ep_cnt[me], 1u); rbe::stat_add(&ep_cnt[stat_stride + me], sn - 1u); rbe::stat_add(&ep_cnt[2 * stat_stride + me], reached ? 1u : 0u);
            late->goal_count[me] = draw;
        }
        if (ep.auto_reset) { sn = 1u; fz = 1u; }
        ret = 0.0f;
    }
    // rows back: state, goal (only the lanes that redrew change it), observation
    store_outputs<true>(late->q, late->qd, late->obs, env0, live, region, lane, qq, vv, o);
    if (__builtin_amdgcn_ballot_w64(dn && mine) != 0ull) store_rows<RBL_NQ>(late->goal, env0, live, region, lane, gn);
    if (mine) {
        late->feas[me] = fz; late->step_num[me] = sn; late->ep_ret[me] = ret; late->reward[me] = r; late->done[me] = dn ? 1u : 0u;
        // (an atomic that returns nothing: a load-add-store here is a memory latency the wave - alone on its SIMD - sits out)
        if (!ok) __hip_atomic_fetch_add(&late->infeas_n[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RBL_LANE_STAMP(3);
}

}  // namespace RBL_NS
