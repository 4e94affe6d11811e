// tree_lane_defs.hpp - what the text written by tree_lane_gen.hpp expects to find: function qualifier, table
// declaration, the lane-private LDS slot accessor and the few math primitives, for the device build (hipcc and
// hiprtc).  tests/hostmath/tree_lane_host.cpp defines the same names for g++ (the CPU test of the generator).
#pragma once
#include "rtc_compat.hpp"

#define RBL_FN __device__ __forceinline__
#define RBL_TABLE(name, n) __device__ constexpr float name[n]
#define RBL_SCHED_BARRIER __builtin_amdgcn_sched_barrier(0)
#define RBL_LDS(slot) rbl_lds(slot)
// split form (several waves per env group, tree_lane_split.hpp): integer tables, the exchange area, the workgroup barrier
#define RBL_ITABLE(name, n) __device__ constexpr int name[n]
#define RBL_X(slot) rbl_x(slot)
#if defined(RB_SPLIT_STAMPS)
// diagnostic builds only (tools/gpu_r4_helpers_stamps.sh): s_memtime before and after every workgroup barrier of workgroup 0,
// per wave, into a global buffer (rb_debug_stamps fetches it); no product build defines RB_SPLIT_STAMPS
__device__ unsigned long long rbl_stamp_buf[8 * 128];
__device__ int rbl_stamp_cnt[8];
__device__ __forceinline__ void rbl_stamp_mark(bool reset = false) {        // one entry: now
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        const int k = reset ? 0 : rbl_stamp_cnt[wave];
        if (k < 127) { rbl_stamp_buf[wave * 128 + k] = __builtin_amdgcn_s_memtime(); rbl_stamp_cnt[wave] = k + 1; }
    }
}
__device__ __forceinline__ void rbl_stamped_barrier() {
    rbl_stamp_mark();
    __syncthreads();
    rbl_stamp_mark();
}
#define RBL_PART_BARRIER rbl_stamped_barrier()
#else
#define RBL_PART_BARRIER __syncthreads()
#endif

RBL_FN float rbl_sin(float x) { return __sinf(x); }
RBL_FN float rbl_cos(float x) { return __cosf(x); }
RBL_FN float rbl_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
RBL_FN float rbl_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RBL_FN float rbl_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
RBL_FN float rbl_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
RBL_FN float rbl_max(float a, float b) { return fmaxf(a, b); }
RBL_FN float rbl_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Pair values (tree_lane_gen.hpp: Val::pair - a subtree and its structurally identical mate, e.g. the two arms, written
// as ONE instruction stream): arithmetic on rbl_f2 becomes v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, a plain float
// operand is broadcast through op_sel (no move); the transcendentals and min / max have no packed form and run per half.
typedef float rbl_f2 __attribute__((ext_vector_type(2)));
#define RBL_K2(a, b) (rbl_f2{a, b})
#define RBL_MK2(a, b) (rbl_f2{a, b})
RBL_FN float rbl_lo(rbl_f2 v) { return v.x; }
RBL_FN float rbl_hi(rbl_f2 v) { return v.y; }
RBL_FN float rbl_hsum(rbl_f2 v) { return v.x + v.y; }
RBL_FN rbl_f2 rbl_sin(rbl_f2 v) { return rbl_f2{rbl_sin(v.x), rbl_sin(v.y)}; }
RBL_FN rbl_f2 rbl_cos(rbl_f2 v) { return rbl_f2{rbl_cos(v.x), rbl_cos(v.y)}; }
RBL_FN rbl_f2 rbl_rsq(rbl_f2 v) { return rbl_f2{rbl_rsq(v.x), rbl_rsq(v.y)}; }
RBL_FN rbl_f2 rbl_rcp(rbl_f2 v) { return rbl_f2{rbl_rcp(v.x), rbl_rcp(v.y)}; }
RBL_FN rbl_f2 rbl_exp2(rbl_f2 v) { return rbl_f2{rbl_exp2(v.x), rbl_exp2(v.y)}; }
RBL_FN rbl_f2 rbl_med3(rbl_f2 v, float lo, float hi) { return rbl_f2{rbl_med3(v.x, lo, hi), rbl_med3(v.y, lo, hi)}; }
RBL_FN rbl_f2 rbl_max(rbl_f2 a, float b) { return rbl_f2{rbl_max(a.x, b), rbl_max(a.y, b)}; }
RBL_FN rbl_f2 rbl_fma(rbl_f2 a, rbl_f2 b, rbl_f2 c) { return __builtin_elementwise_fma(a, b, c); }
