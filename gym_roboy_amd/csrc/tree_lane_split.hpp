// tree_lane_split.hpp - joint-tree kernels with SEVERAL WAVES per group of 64 envs, around the split-form text of
// tree_lane_gen.hpp (generate_split): for batches too small to give every SIMD a wave (configs[3]: 8 192 upper-body envs are
// 128 waves for 1 024 SIMDs, and a step is then one wave's whole instruction stream, 18 us).  A workgroup is
// RBL_NPARTS waves, each on a SIMD of its own with the whole register file; wave p runs part p of the generated code:
// the trunk's forward sweep (every wave for itself), its own branches, one exchange of the branches' contributions to
// the trunk through LDS behind ONE workgroup barrier per acceleration, the trunk's backward / forward passes (every
// wave for itself, bit-identical) and its own branches' accelerations.  Every wave integrates the trunk's joints and
// its own; nothing else of the state crosses between waves until the rows are written back.
//
// Included after the generated split header (RBL_NS, RBL_NQ, RBL_NT, RBL_NPARTS, RBL_PART_LDS, RBL_X_SLOTS, the tables
// KSG / QLO / QHI / VMAX / PART_OF_JOINT and RBL_NS::rbl_part).  LDS of a workgroup, in 64-float slots:
//   image   3 n_q + max(3 n_q, n_t)   the rows of the 64 envs (q | qd | goal | action, the observation rows over the action's),
//                                     transposed in / out cooperatively by all waves
//   X       2 x RBL_X_SLOTS           exchange area, double-buffered: acceleration n uses buffer n & 1, so a wave that is
//                                     already writing for n + 1 cannot disturb one still reading n (it cannot reach n + 2
//                                     before the other has passed the barrier of n + 1)
//   per wave: RBL_PART_LDS parking slots + 2 n_q RK4 accumulators; then one flag slot per wave
//
// Helper waves (round 4, RBL_NHELPERS > 0): the workgroup has RBL_NPARTS + RBL_NHELPERS waves.  A helper (wave RBL_NPARTS + h) runs
// the generated function of "part" RBL_NPARTS + h: the kinematics of the trunk and of its part's links from the stage state the
// parts publish in the exchange area, its part's tendons, the wrench sums per link back into the exchange area - while the part's own
// wave runs its forward sweep - with three barriers per acceleration (S, T, X: all in the generated text, the same count in every
// wave).  It owns no joints, keeps no state between accelerations (no parking slots, no accumulators) and shares a SIMD with one of
// the parts - which costs that part nothing: a wave issues one vector instruction per 5.5 cycles at best, a SIMD takes one per 2.8
// from two.  With helpers the exchange area is SINGLE-buffered (between a wave's reads of one acceleration and anybody's writes
// of the next lies at least one barrier).
//
// The cut form (round 4, generate_split_cut; RBL_X_SINGLE, RBL_ACC_JOINTS): the heaviest parts are two waves each - the part's own
// (proximal links, tendons) and a distal one, which is a part like any other here (it owns and integrates its joints, and the trunk's
// like everybody); five barriers per acceleration, all in the generated text; single-buffered exchange area; the RK4 accumulators
// take a slot per joint a part integrates instead of one per joint of the robot.
//
// The LEAN layout (round 5, RBL_LEAN = 1; no helpers): for batches between "one workgroup per CU" and "a wave on every SIMD" - the upper
// body at 16 384 < n <= 32 768 envs, where the one-wave form leaves half of the SIMDs idle and the five-wave form needs two
// generations.  Two part waves per env group and TWO workgroups per CU: the workgroup must fit 80 KB of LDS.  A part wave is alone on
// its SIMD with the whole 512-register file and uses ~300 of it, so what the other layouts keep in LDS moves to registers - the
// parking slots of the generated code (a local array behind the same accessor: every slot index is a literal, the array becomes
// registers) and the RK4 accumulators - and the exchange area lies OVER the action / observation image (dead while the step runs;
// a barrier behind the waves' input reads and one in front of their output writes keep the two uses apart):
//   image   q | qd | goal          3 n_q slots, readable to the end of the step
//   shared  max(action / observation image, 2 x RBL_X_SLOTS exchange buffers, 4 n_q)   the goal rows a finished episode writes go behind the observation image
//   flags   3 per part + 1
// Upper body: 222 slots = 55.5 KB against 481 = 120 KB in the plain layout of the same two parts.
#pragma once
#include "env_common.hpp"
#include "philox.hpp"

#ifndef RBL_NPARTS
#error "include the generated split-form header (tree_lane_gen.hpp: generate_split) first"
#endif
#ifndef RBL_NHELPERS
#define RBL_NHELPERS 0
#endif
#ifndef RBL_LEAN
#define RBL_LEAN 0
#endif
#if RBL_LEAN && RBL_NHELPERS > 0
#error "the lean layout has no helper waves"
#endif

namespace RBL_NS {

// K2S: how the generated part functions instantiated with this accessor write their pair constants (tree_lane_defs.hpp: RBL_K2) -
// per half (true: the straight-line Euler step gains 1-2 %) or packed (false: RK4's stage loop keeps the pairs resident)
#ifndef RBL_K2_SPLIT_EULER
#define RBL_K2_SPLIT_EULER 1
#endif
// (RK4: per half in the part waves / the helper waves only - RBL_K2_SPLIT_RK4 bit 0 / bit 1 - was measured too: k2split_rk4_ab.log)
#ifndef RBL_K2_SPLIT_RK4
#define RBL_K2_SPLIT_RK4 0
#endif
constexpr bool sp_k2_split(int integ, bool helper = false) {
    return integ == 0 ? RBL_K2_SPLIT_EULER != 0 : ((RBL_K2_SPLIT_RK4 >> (helper ? 1 : 0)) & 1) != 0;
}
template <bool K2S>
struct SplitLdsT {
    float *p;   // the wave's private region + lane
    static constexpr bool k2_split = K2S;
    __device__ __forceinline__ float &operator()(int slot) const { return p[slot * 64]; }
};
using SplitLds = SplitLdsT<false>;      // (the exchange area's accessor: RBL_XA - the mode rides on RBL_L)
// the lean layout's parking "slots": a local array of the wave (slot indices are literals in the generated text: registers)
template <bool K2S>
struct SplitRegsT {
    float *p;
    static constexpr bool k2_split = K2S;
    __device__ __forceinline__ float &operator()(int slot) const { return p[slot]; }
};

// row images: q | qd | goal | action, the observation image [q | qd | goal as observed] over the action image (dead after a wave's
// first lines); the goal image stays readable to the end of the step (the waves fetch their joints' goals AFTER the step: ten values
// fewer alive across it in kernels that sit at their 256 registers)
constexpr int SP_IMG_SLOTS = 3 * RBL_NQ + (3 * RBL_NQ > RBL_NT ? 3 * RBL_NQ : RBL_NT);
constexpr int SP_OV = RBL_NQ * 64, SP_OG = 2 * RBL_NQ * 64, SP_OA = 3 * RBL_NQ * 64, SP_OO = SP_OA;
#ifndef RBL_ACC_JOINTS
#define RBL_ACC_JOINTS RBL_NQ        // joints a part integrates at most (the cut form says; else a slot per joint of the robot)
#endif
#ifndef RBL_X_SINGLE
#define RBL_X_SINGLE (RBL_NHELPERS > 0)
#endif
constexpr int SP_ACC_JOINTS = RBL_ACC_JOINTS;
constexpr bool SP_LEAN = RBL_LEAN != 0;
constexpr int SP_WAVE_SLOTS = SP_LEAN ? 0 : RBL_PART_LDS + 2 * SP_ACC_JOINTS;
constexpr int SP_ACC_SLOT = RBL_PART_LDS;
constexpr int SP_NWAVES = RBL_NPARTS + RBL_NHELPERS;
constexpr int SP_X_BUFFERS = RBL_X_SINGLE ? 1 : 2;
// lean: the exchange area over the action / observation image, behind q | qd | goal; the goal rows of finished episodes behind the observation image
constexpr int SP_X_OFF = SP_LEAN ? 3 * RBL_NQ : SP_IMG_SLOTS;
constexpr int SP_SHARED_SLOTS = (SP_X_BUFFERS * RBL_X_SLOTS > SP_IMG_SLOTS - 3 * RBL_NQ ? SP_X_BUFFERS * RBL_X_SLOTS : SP_IMG_SLOTS - 3 * RBL_NQ) > 4 * RBL_NQ
                                    ? (SP_X_BUFFERS * RBL_X_SLOTS > SP_IMG_SLOTS - 3 * RBL_NQ ? SP_X_BUFFERS * RBL_X_SLOTS : SP_IMG_SLOTS - 3 * RBL_NQ) : 4 * RBL_NQ;
constexpr int SP_WAVE_OFF = SP_LEAN ? 3 * RBL_NQ + SP_SHARED_SLOTS : SP_X_OFF + SP_X_BUFFERS * RBL_X_SLOTS;
constexpr int SP_GOAL_OUT_OFF = SP_LEAN ? 6 * RBL_NQ : SP_WAVE_OFF;     // where the accountant leaves the new goal rows (plain layout: its own parking region, free by then)
constexpr int SP_FLAG_OFF = SP_WAVE_OFF + RBL_NPARTS * SP_WAVE_SLOTS;
constexpr int SP_DRAW_OFF = SP_FLAG_OFF + 3 * RBL_NPARTS + 1;     // per part: limit flag, and (env layer) its joints' shares of |dq|^2, |qd|^2; then the "a goal changed" word
// ... then (helper form, env layer) the goals the helpers draw ahead for every env of the group, one slot per joint: a helper is idle
// from its last hand-over to the end of the step, the accountant's Philox draw - 450 instructions whenever ONE of its 64 envs ends an
// episode, which with the episodes spread out as in training happens in some workgroup of every launch - is not
constexpr int SP_DRAW_SLOTS = RBL_NHELPERS > 0 ? RBL_NQ : 0;
constexpr int SP_LDS_SLOTS = SP_DRAW_OFF + SP_DRAW_SLOTS;
constexpr int SP_LDS_BYTES = SP_LDS_SLOTS * 64 * 4;

__device__ __forceinline__ int sp_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ void sp_fence_code() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sp_rows_rsrc(const float *g, long env0, int width, int live) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(g + env0 * width);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(a)), hi = __builtin_amdgcn_readfirstlane(uint32_t(a >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane(live * width * 4);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>((uintptr_t(hi) << 32) | lo), 0, bytes, 0x00020000);
}
// `width` slots of a row image, moved by the waves of the workgroup together: wave w takes the slots k = w, w + K, ...
// (HBM -> image: 64 consecutive floats of the rows' run per slot; the range check of the resource guards the tail)
template <int W>
__device__ __forceinline__ void sp_load_image(const float *__restrict__ g, long env0, int live, float *img, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t r = sp_rows_rsrc(g, env0, W, live);
    constexpr int PER = (W + SP_NWAVES - 1) / SP_NWAVES;
    float t[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int k = wave + u * SP_NWAVES;
        t[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, (k < W ? k : W - 1) * 256, 0));
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int k = wave + u * SP_NWAVES;
        if (k < W) img[k * 64 + lane] = t[u];
    }
}
// The input images of a step at once: EVERY load of every image is in flight before the first is waited for - one memory round
// trip per step instead of one per image (round 4, barrier stamps of workgroup 0: 4 200 cycles from the wave's start to "rows in
// LDS" with the images loaded one after the other - three round trips - of a 24 000-cycle Euler step).
template <int W>
struct SpImageLoad {
    static constexpr int PER = (W + SP_NWAVES - 1) / SP_NWAVES;
    float t[PER];
    __device__ __forceinline__ void issue(const float *__restrict__ g, long env0, int live, int wave, int lane) {
        const __amdgpu_buffer_rsrc_t r = sp_rows_rsrc(g, env0, W, live);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = wave + u * SP_NWAVES;
            t[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, (k < W ? k : W - 1) * 256, 0));
        }
    }
    __device__ __forceinline__ void land(float *img, int wave, int lane) const {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = wave + u * SP_NWAVES;
            if (k < W) img[k * 64 + lane] = t[u];
        }
    }
};

template <int W>
__device__ __forceinline__ void sp_store_image(float *__restrict__ g, long env0, int live, const float *img, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t r = sp_rows_rsrc(g, env0, W, live);
    constexpr int PER = (W + SP_NWAVES - 1) / SP_NWAVES;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int k = wave + u * SP_NWAVES;
        if (k < W) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(img[k * 64 + lane]), r, lane * 4, k * 256, 0);
    }
}

// The output images of a step at once: every LDS read of every image first, then the stores back to back (one after the other the
// images cost an LDS latency per store: barrier stamps, 150-200 cycles per store of a wave that is alone on its SIMD)
template <int W>
struct SpImageStore {
    static constexpr int PER = (W + SP_NWAVES - 1) / SP_NWAVES;
    float t[PER];
    __device__ __forceinline__ void read(const float *img, int wave, int lane) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = wave + u * SP_NWAVES;
            t[u] = img[(k < W ? k : W - 1) * 64 + lane];
        }
    }
    __device__ __forceinline__ void store(float *__restrict__ g, long env0, int live, int wave, int lane) const {
        const __amdgpu_buffer_rsrc_t r = sp_rows_rsrc(g, env0, W, live);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = wave + u * SP_NWAVES;
            if (k < W) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t[u]), r, lane * 4, k * 256, 0);
        }
    }
};

__device__ __forceinline__ float sp_sat(float v, int j) { return __builtin_amdgcn_fmed3f(v, -VMAX[j], VMAX[j]); }
template <int PART>
__device__ __forceinline__ constexpr bool sp_mine(int j) { return PART_OF_JOINT[j] < 0 || PART_OF_JOINT[j] == PART; }
// the RK4 accumulator slot of joint j in part PART's region: one per joint the part integrates (the cut form: five waves' regions
// of 2 n_q slots each would not fit the LDS), or simply j
template <int PART>
__device__ __forceinline__ constexpr int sp_acc(int j) {
    if (SP_ACC_JOINTS == RBL_NQ) return j;
    int n = 0;
    for (int i = 0; i < j; ++i) n += sp_mine<PART>(i) ? 1 : 0;
    return n;
}

// One env step of the joints part PART integrates (the trunk's and its own); the other entries of q / v are not touched.
template <int INTEG, int PART, class PARK>
__device__ __forceinline__ bool split_step(const PARK &L, float *xbase, int lane, const float (&spu)[RBL_NT], float h, int nsub,
                                           float (&q)[RBL_NQ], float (&v)[RBL_NQ]) {
    bool ok = true;
    int n_acc = 0;                                   // accelerations so far: selects the exchange buffer
    auto accel = [&](const float (&qq)[RBL_NQ], const float (&vv)[RBL_NQ], float (&out)[RBL_NQ]) {
        const SplitLds X{xbase + (SP_X_BUFFERS == 2 ? (n_acc & 1) * (RBL_X_SLOTS * 64) : 0) + lane};
        ++n_acc;
        sp_fence_code();
        rbl_part(PART, qq, vv, spu, out, L, X);
        sp_fence_code();
    };
    for (int sub = 0; sub < nsub; ++sub) {
        if (INTEG == 0) {
            float a[RBL_NQ];
            accel(q, v, a);
#pragma unroll
            for (int j = 0; j < RBL_NQ; ++j)
                if (sp_mine<PART>(j)) { v[j] = sp_sat(v[j] + h * a[j], j); q[j] = q[j] + h * v[j]; }
        } else {
            const float h6 = h * (1.0f / 6.0f);
            float kq[RBL_NQ], kv[RBL_NQ];
            if constexpr (SP_LEAN) {
                // the weighted sums in registers (the wave has ~200 to spare), same order of accumulation as the LDS form: k1 + 2 k2 + 2 k3 + k4
                float qa[RBL_NQ], va[RBL_NQ];
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j) { kq[j] = 0.0f; kv[j] = 0.0f; qa[j] = 0.0f; va[j] = 0.0f; }
#pragma unroll 1
                for (int st = 0; st < 4; ++st) {
                    const float wgt = (st == 0 || st == 3) ? 1.0f : 2.0f, cst = st == 0 ? 0.0f : (st == 3 ? h : 0.5f * h);
                    float qs[RBL_NQ];
#pragma unroll
                    for (int j = 0; j < RBL_NQ; ++j) {
                        qs[j] = 0.0f;
                        if (sp_mine<PART>(j)) { qs[j] = q[j] + cst * kq[j]; kq[j] = sp_sat(v[j] + cst * kv[j], j); }
                    }
#pragma unroll
                    for (int j = 0; j < RBL_NQ; ++j) if (sp_mine<PART>(j)) qa[j] += wgt * kq[j];
                    accel(qs, kq, kv);
#pragma unroll
                    for (int j = 0; j < RBL_NQ; ++j) if (sp_mine<PART>(j)) va[j] += wgt * kv[j];
                }
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j)
                    if (sp_mine<PART>(j)) { q[j] = q[j] + h6 * qa[j]; v[j] = v[j] + h6 * va[j]; }
            } else {
#pragma unroll
            for (int j = 0; j < RBL_NQ; ++j) {
                kq[j] = 0.0f; kv[j] = 0.0f;
                if (sp_mine<PART>(j)) { L(SP_ACC_SLOT + sp_acc<PART>(j)) = 0.0f; L(SP_ACC_SLOT + SP_ACC_JOINTS + sp_acc<PART>(j)) = 0.0f; }
            }
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                const float wgt = (st == 0 || st == 3) ? 1.0f : 2.0f, cst = st == 0 ? 0.0f : (st == 3 ? h : 0.5f * h);
                float qs[RBL_NQ];
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j) {
                    qs[j] = 0.0f;
                    if (sp_mine<PART>(j)) { qs[j] = q[j] + cst * kq[j]; kq[j] = sp_sat(v[j] + cst * kv[j], j); }
                }
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j) if (sp_mine<PART>(j)) L(SP_ACC_SLOT + sp_acc<PART>(j)) += wgt * kq[j];
                accel(qs, kq, kv);
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j) if (sp_mine<PART>(j)) L(SP_ACC_SLOT + SP_ACC_JOINTS + sp_acc<PART>(j)) += wgt * kv[j];
            }
#pragma unroll
            for (int j = 0; j < RBL_NQ; ++j)
                if (sp_mine<PART>(j)) { q[j] = q[j] + h6 * L(SP_ACC_SLOT + sp_acc<PART>(j)); v[j] = v[j] + h6 * L(SP_ACC_SLOT + SP_ACC_JOINTS + sp_acc<PART>(j)); }
            }
        }
#pragma unroll
        for (int j = 0; j < RBL_NQ; ++j) {
            if (!sp_mine<PART>(j)) continue;
            float vv = sp_sat(v[j], j);
            const bool over = q[j] > QHI[j], under = q[j] < QLO[j];
            if (over) { q[j] = QHI[j]; vv = fminf(vv, 0.0f); }
            if (under) { q[j] = QLO[j]; vv = fmaxf(vv, 0.0f); }
            v[j] = vv;
            ok = ok && !(over || under);
        }
    }
    return ok;
}

// everything between the loaded rows and the row images of the results, for the wave that runs part PART
template <int INTEG, int PART>
__device__ __forceinline__ void split_wave(float *lds, int lane, int live, float act_scale, float h, int nsub, bool env_layer,
                                           const rbe::EnvParams *ep) {
    float *img = lds;
    const int row = sp_opaque(lane < live ? lane : live - 1);
    float q[RBL_NQ], v[RBL_NQ], spu[RBL_NT];
    constexpr int OV = SP_OV, OA = SP_OA, OG = SP_OG;
#pragma unroll
    for (int j = 0; j < RBL_NQ; ++j) { q[j] = img[row * RBL_NQ + j]; v[j] = img[OV + row * RBL_NQ + j]; }
#pragma unroll
    for (int k = 0; k < RBL_NT; ++k) {
        const float a = img[OA + row * RBL_NT + k];
        if (env_layer) {
            // clamp to the action box, then slope * (x - in_high) + out_high with two roundings (roboy_env.py:157-158)
            const float x = fminf(fmaxf(a, -1.0f), 1.0f);
            spu[k] = rbe::rounded_here(rbe::mul_then_add(ep->slope, x - 1.0f, ep->act_hi) * KSG[k]);
        } else {
            spu[k] = rbe::rounded_here((a * act_scale) * KSG[k]);
        }
    }
    bool ok;
    if constexpr (SP_LEAN) {
        // the exchange area lies over the action image: nobody writes it before everybody has read its actions ...
        RBL_PART_BARRIER;
        float park[RBL_PART_LDS > 0 ? RBL_PART_LDS : 1];
        ok = split_step<INTEG, PART>(SplitRegsT<sp_k2_split(INTEG)>{park}, lds + SP_X_OFF * 64, lane, spu, h, nsub, q, v);
        // ... and over the observation image: nobody writes that before everybody has read the last acceleration's exchange
        RBL_PART_BARRIER;
    } else {
        const SplitLdsT<sp_k2_split(INTEG)> L{lds + (SP_WAVE_OFF + PART * SP_WAVE_SLOTS) * 64 + lane};
        ok = split_step<INTEG, PART>(L, lds + SP_X_OFF * 64, lane, spu, h, nsub, q, v);
    }
    // (the accelerations' barriers lie between every wave's reads of the input image above and these writes)
    const int wl = sp_opaque(lane);
#pragma unroll
    for (int j = 0; j < RBL_NQ; ++j)
        if (PART_OF_JOINT[j] == PART || (PART_OF_JOINT[j] < 0 && PART == 0)) {
            img[wl * RBL_NQ + j] = q[j]; img[OV + wl * RBL_NQ + j] = v[j];
            // env layer: and into the observation rows [q | qd | goal] (the image over the action image, which every wave has read
            // before its first acceleration barrier); the accountant adds the goals
            if (env_layer) { img[OA + wl * (3 * RBL_NQ) + j] = q[j]; img[OA + wl * (3 * RBL_NQ) + RBL_NQ + j] = v[j]; }
        }
    lds[(SP_FLAG_OFF + 3 * PART) * 64 + lane] = ok ? 1.0f : 0.0f;
    if (env_layer) {
        // the goals of the joints this wave accounts for (it sums their squared distances)
        float dq2 = 0.0f, dv2 = 0.0f;
#pragma unroll
        for (int j = 0; j < RBL_NQ; ++j)
            if (PART_OF_JOINT[j] == PART || (PART_OF_JOINT[j] < 0 && PART == 0)) { const float dq = q[j] - img[OG + row * RBL_NQ + j]; dq2 += dq * dq; dv2 += v[j] * v[j]; }
        lds[(SP_FLAG_OFF + 3 * PART + 1) * 64 + lane] = dq2;
        lds[(SP_FLAG_OFF + 3 * PART + 2) * 64 + lane] = dv2;
    }
}

// a helper wave: one call of its generated function per acceleration of the step (the parts' count: n_sub x 1 or 4); it needs the
// activation offsets of its part's tendons and nothing else of the inputs - state comes from the exchange area, nothing goes out
// but the wrench sums it leaves there
template <int INTEG, int HELPER>
__device__ __forceinline__ void split_helper(float *lds, int lane, int live, float act_scale, int nsub, bool env_layer, const rbe::EnvParams *ep) {
    float *img = lds;
    const int row = sp_opaque(lane < live ? lane : live - 1);
    constexpr int OA = SP_OA;
    float spu[RBL_NT], none_q[RBL_NQ], none_a[RBL_NQ];
#pragma unroll
    for (int j = 0; j < RBL_NQ; ++j) { none_q[j] = 0.0f; none_a[j] = 0.0f; }
#pragma unroll
    for (int k = 0; k < RBL_NT; ++k) {
        const float a = img[OA + row * RBL_NT + k];
        if (env_layer) {
            const float x = fminf(fmaxf(a, -1.0f), 1.0f);
            spu[k] = rbe::rounded_here(rbe::mul_then_add(ep->slope, x - 1.0f, ep->act_hi) * KSG[k]);
        } else {
            spu[k] = rbe::rounded_here((a * act_scale) * KSG[k]);
        }
    }
#if !defined(RB_SPLIT_NO_HELPER_PRIO)
    __builtin_amdgcn_s_setprio(2);                      // a helper that shares a SIMD goes first: the wave beside it is the one with slack
#endif
    const SplitLdsT<sp_k2_split(INTEG, true)> L{lds + lane};  // (unused: a helper parks nothing; its type carries the pair-constant mode)
    const SplitLds X{lds + SP_X_OFF * 64 + lane};
    const int n_acc = nsub * (INTEG == 0 ? 1 : 4);
    // env layer: the goal counters of the group's envs, requested now (used behind the step)
    const long env0 = long(blockIdx.x) * 64;
    uint32_t draw_old = 0u;
    if (env_layer) draw_old = rbe::late_args()->goal_count[env0 + row];
#pragma unroll 1
    for (int a = 0; a < n_acc; ++a) {
        sp_fence_code();
        rbl_part(RBL_NPARTS + HELPER, none_q, none_q, spu, none_a, L, X);
        sp_fence_code();
    }
    if (env_layer) {
        // the goals every env of the group WOULD draw if its episode ended with this step (the observable draw: the second one under
        // auto_reset - see the accountant), left in LDS for the accountant; same Philox counters, same arithmetic: same values.  The
        // helpers share the Philox blocks (four goals each) between them: each is done before the arms are
        const rbe::tree_env_kernarg_ptr late = rbe::late_args();
        const uint64_t gid = late->env_id0 + uint64_t(env0 + row), seed = late->seed;
        const uint32_t dnum = draw_old + (late->ep.auto_reset ? 1u : 0u);
#pragma unroll
        for (int b = HELPER; 4 * b < RBL_NQ; b += RBL_NHELPERS) {
            const rb::Philox4 rnd = rb::philox_draw(seed, gid, dnum, rb::STREAM_GOALS, uint32_t(b));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * b + k < RBL_NQ) lds[(SP_DRAW_OFF + 4 * b + k) * 64 + lane] = rbe::goal_value(late->box.lo[4 * b + k], late->box.hi[4 * b + k], rnd.v[k]);
        }
    }
}

template <int INTEG, int PART>
__device__ __forceinline__ void split_dispatch(int wave, float *lds, int lane, int live, float act_scale, float h, int nsub, bool env_layer,
                                               const rbe::EnvParams *ep) {
    if constexpr (PART < RBL_NPARTS) {
        if (wave == WAVE_OF_PART[PART]) split_wave<INTEG, PART>(lds, lane, live, act_scale, h, nsub, env_layer, ep);
        else split_dispatch<INTEG, PART + 1>(wave, lds, lane, live, act_scale, h, nsub, env_layer, ep);
    } else if constexpr (PART < SP_NWAVES) {
        if (wave == WAVE_OF_PART[PART]) split_helper<INTEG, PART - RBL_NPARTS>(lds, lane, live, act_scale, nsub, env_layer, ep);
        else split_dispatch<INTEG, PART + 1>(wave, lds, lane, live, act_scale, h, nsub, env_layer, ep);
    }
}

// forward_step_command for a batch (the contract of tree_lane_step / tree_step_aba)
template <int INTEG>
__global__ void __launch_bounds__(64 * SP_NWAVES)
tree_split_step(float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas, const float *__restrict__ act,
                float act_scale, float h, int nsub, long n) {
    extern __shared__ float lds_split[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long env0 = long(blockIdx.x) * 64;
    if (env0 >= n) return;                          // (the whole workgroup: no barrier is left behind)
    const int live = n - env0 < 64 ? int(n - env0) : 64;
    float *lds = lds_split;
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark(true);                              // entry 0: the wave exists
#endif
    {
        SpImageLoad<RBL_NQ> lq, lv;
        SpImageLoad<RBL_NT> la;
        lq.issue(q, env0, live, wave, lane); lv.issue(qd, env0, live, wave, lane); la.issue(act, env0, live, wave, lane);
        lq.land(lds, wave, lane); lv.land(lds + SP_OV, wave, lane); la.land(lds + SP_OA, wave, lane);
    }
    __syncthreads();
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark();                                  // entry 1: the rows are in LDS
#endif
    split_dispatch<INTEG, 0>(wave, lds, lane, live, act_scale, h, nsub, false, nullptr);
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark();                                  // the wave's share of the step is done
#endif
    __syncthreads();
    {
        SpImageStore<RBL_NQ> sq, sv;
        sq.read(lds, wave, lane); sv.read(lds + SP_OV, wave, lane);
        sq.store(q, env0, live, wave, lane); sv.store(qd, env0, live, wave, lane);
    }
    if (wave == 0 && lane < live) {
        bool ok = true;
#pragma unroll
        for (int p = 0; p < RBL_NPARTS; ++p) ok = ok && lds[(SP_FLAG_OFF + 3 * p) * 64 + lane] != 0.0f;
        feas[env0 + lane] = ok ? 1u : 0u;
    }
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark();                                  // last entry: stores issued
#endif
}

// RoboyEnv.step fused around the split step (semantics of tree_lane_env_step / msj_env_step_kernel).  The waves
// step their joints; wave 0 then is the envs' accountant (one env per lane): reward and done from the waves' partial sums,
// goal redraw (and reset) on done, the observation rows; all waves write the row images back together.
// (one kernel argument, rbe::TreeEnvArgs: what the accountant needs is read from the kernel-argument segment behind the step)
template <int INTEG>
__global__ void __launch_bounds__(64 * SP_NWAVES)
tree_split_env_step(const rbe::TreeEnvArgs a) {
    // what the front of the kernel needs, by name (the rest: `late`, below)
    const float *__restrict__ q = a.q, *__restrict__ qd = a.qd, *__restrict__ act = a.act, *__restrict__ goal = a.goal;
    const uint32_t *__restrict__ step_num = a.step_num, *__restrict__ goal_count = a.goal_count;
    const float *__restrict__ ep_ret = a.ep_ret;
    const float h = a.h;
    const int nsub = a.nsub;
    const long n = a.n;
    rbe::EnvParams ep_front;                         // the action transform's two constants
    ep_front.slope = a.ep.slope; ep_front.act_hi = a.ep.act_hi;
    extern __shared__ float lds_split[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long env0 = long(blockIdx.x) * 64;
    if (env0 >= n) return;
    const int live = n - env0 < 64 ? int(n - env0) : 64;
    float *lds = lds_split;
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark(true);
#endif
    constexpr int OV = SP_OV, OA = SP_OA, OG = SP_OG, OO = SP_OO;     // (the observation image takes the place of the action image)
    const bool mine = wave == 0 && lane < live;
    const long me = env0 + (lane < live ? lane : live - 1);
    uint32_t sn_old = 0u, draw_old = 0u;
    float ret_old = 0.0f;
    {
        SpImageLoad<RBL_NQ> lq, lv, lg;
        SpImageLoad<RBL_NT> la;
        lq.issue(q, env0, live, wave, lane); lv.issue(qd, env0, live, wave, lane); la.issue(act, env0, live, wave, lane);
        lg.issue(goal, env0, live, wave, lane);
        // the accountant's counters ride on the same round trip: requested BEFORE the images are waited for (behind them they cost a
        // second memory latency in front of the barrier below, whose fence waits for every load: 2 000 cycles by the barrier stamps)
        if (wave == 0) { sn_old = step_num[me]; ret_old = ep_ret[me]; draw_old = goal_count[me]; }
        lq.land(lds, wave, lane); lv.land(lds + OV, wave, lane); la.land(lds + OA, wave, lane); lg.land(lds + OG, wave, lane);
    }
    __syncthreads();
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark();                                  // rows in LDS
#endif
    split_dispatch<INTEG, 0>(wave, lds, lane, live, 1.0f, h, nsub, true, &ep_front);
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark();                                  // the wave's share of the step is done
#endif
    // The accountant (wave 0 - the shortest part: it is here first) uses its wait for the others: its arguments from the kernel-
    // argument segment, the old goal row of its env (read HERE, not in front of the step - a value defined under `wave == 0` before
    // the dispatch is alive, as far as the register allocator can tell, through every other wave's code: 20 registers in kernels that
    // sit at their 256), and the goal part of the observation rows (the waves write q and qd themselves)
    const rbe::tree_env_kernarg_ptr late = rbe::late_args();
    rbe::EnvParams ep;
    float gg[RBL_NQ];
    if (wave == 0) {
        ep = rbe::late_env_params(late);
        const int row = sp_opaque(lane < live ? lane : live - 1), wl0 = sp_opaque(lane);
#pragma unroll
        for (int j = 0; j < RBL_NQ; ++j) gg[j] = lds[OG + row * RBL_NQ + j];
#pragma unroll
        for (int j = 0; j < RBL_NQ; ++j) lds[OO + wl0 * (3 * RBL_NQ) + 2 * RBL_NQ + j] = gg[j];
    }
    __syncthreads();
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark();                                  // every wave's is
#endif
    if (wave == 0) {
        const int wl = sp_opaque(lane);
        const uint64_t seed = late->seed, env_id0 = late->env_id0;
        double *__restrict__ ep_sum = late->ep_sum;
        uint32_t *__restrict__ ep_cnt = late->ep_cnt, *__restrict__ infeas_n = late->infeas_n, *__restrict__ feas = late->feas;
        uint32_t *__restrict__ step_num = late->step_num, *__restrict__ goal_count = late->goal_count, *__restrict__ done = late->done;
        float *__restrict__ ep_ret = late->ep_ret, *__restrict__ reward = late->reward;
        const long n = late->n;
        bool ok = true;
        float dq2 = 0.0f, dv2 = 0.0f;
#pragma unroll
        for (int p = 0; p < RBL_NPARTS; ++p) {
            ok = ok && lds[(SP_FLAG_OFF + 3 * p) * 64 + lane] != 0.0f;
            dq2 += lds[(SP_FLAG_OFF + 3 * p + 1) * 64 + lane];
            dv2 += lds[(SP_FLAG_OFF + 3 * p + 2) * 64 + lane];
        }
        uint32_t sn = sn_old + 1u;
        bool reached;
        const float r = rbe::env_reward(ep, dq2, dv2, ok, reached);
        const bool dn = reached || (sn > uint32_t(ep.max_len));
        float ret = ret_old + r;
        uint32_t fz = ok ? 1u : 0u;
        float gn[RBL_NQ];
#pragma unroll
        for (int j = 0; j < RBL_NQ; ++j) gn[j] = gg[j];
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
            // RoboyEnv.step: _set_new_goal (:67-68), AFTER the observation was made (:60); with the VecEnv worker's env.reset() (:82-87)
            // a second draw replaces the first unseen: only that one is evaluated (the counter still advances by two)
            if (SP_DRAW_SLOTS > 0) {                       // the helpers drew them while the arms finished the step
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j) gn[j] = lds[(SP_DRAW_OFF + j) * 64 + lane];
            } else {
                draw_goals(draw + (ep.auto_reset ? 1u : 0u));
            }
            draw += ep.auto_reset ? 2u : 1u;
            if (ep.auto_reset) {
#pragma unroll
                for (int j = 0; j < RBL_NQ; ++j) {
                    lds[wl * RBL_NQ + j] = 0.0f; lds[OV + wl * RBL_NQ + j] = 0.0f; gg[j] = gn[j];
                    lds[OO + wl * (3 * RBL_NQ) + j] = 0.0f; lds[OO + wl * (3 * RBL_NQ) + RBL_NQ + j] = 0.0f;
                    lds[OO + wl * (3 * RBL_NQ) + 2 * RBL_NQ + j] = gn[j];
                }
            }
            if (mine) {
                rbe::stat_add(&ep_sum[me], double(ret)); rbe::stat_add(&ep_sum[n + me], double(ret) * double(ret));
                rbe::stat_add(&ep_cnt[me], 1u); rbe::stat_add(&ep_cnt[n + me], sn - 1u); rbe::stat_add(&ep_cnt[2 * n + me], reached ? 1u : 0u);
                goal_count[me] = draw;
            }
            if (ep.auto_reset) { sn = 1u; fz = 1u; }
            ret = 0.0f;
        }
        // (observation rows [q | qd | goal as observed]: q and qd are the waves' own writes, the goals went in above)
        const bool any = __builtin_amdgcn_ballot_w64(dn && mine) != 0ull;
        if (lane == 0) lds[(SP_FLAG_OFF + 3 * RBL_NPARTS) * 64] = any ? 1.0f : 0.0f;
        if (any) {
            // the new goal rows go out from the accountant's registers through its own parking region (free now; lean layout: behind the
            // observation image, inside the exchange area nobody uses any more)
            float *gimg = lds + (SP_GOAL_OUT_OFF) * 64;
#pragma unroll
            for (int j = 0; j < RBL_NQ; ++j) gimg[wl * RBL_NQ + j] = gn[j];
        }
        if (mine) {
            feas[me] = fz; step_num[me] = sn; ep_ret[me] = ret; reward[me] = r; done[me] = dn ? 1u : 0u;
            // (an atomic that returns nothing: a load-add-store here is a memory latency on the accountant's serial stretch - 1 500 cycles
            // by the barrier stamps)
            if (!ok) __hip_atomic_fetch_add(&infeas_n[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark();                                  // (wave 0: the accountant is done)
#endif
    __syncthreads();
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark();
#endif
    {
        SpImageStore<RBL_NQ> sq, sv;
        SpImageStore<3 * RBL_NQ> so;
        sq.read(lds, wave, lane); sv.read(lds + OV, wave, lane); so.read(lds + OO, wave, lane);
        const bool new_goals = lds[(SP_FLAG_OFF + 3 * RBL_NPARTS) * 64] != 0.0f;
        sq.store(late->q, env0, live, wave, lane); sv.store(late->qd, env0, live, wave, lane); so.store(late->obs, env0, live, wave, lane);
        if (new_goals) sp_store_image<RBL_NQ>(late->goal, env0, live, lds + SP_GOAL_OUT_OFF * 64, wave, lane);
    }
#if defined(RB_SPLIT_STAMPS)
    rbl_stamp_mark();                                  // stores issued
#endif
}

}  // namespace RBL_NS
