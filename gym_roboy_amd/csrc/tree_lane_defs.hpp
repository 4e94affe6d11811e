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
// A pair CONSTANT (two different literals) cannot ride in one packed instruction: as an rbl_f2 operand the compiler builds it in a scalar
// register pair with two s_mov_b32 in front of the v_pk_* that reads it (271 of them in the upper body's step; on a lone wave the pair
// costs up to 3.3 ns on top of the instruction's 3.8 - tools/pk_const_probe.hip).  RBL_K2_SPLIT = 1 makes the constant a type of its own
// whose products and fused multiply-adds are written per half - two plain instructions with a literal operand each, the same vector-pipe
// time as the packed one (8 cycles), no scalar instructions - with the same arithmetic per component (bit-identical results).
#ifndef RBL_K2_SPLIT
#define RBL_K2_SPLIT 0
#endif
#if RBL_K2_SPLIT
struct rbl_k2 {
    float a, b;
    RBL_FN operator rbl_f2() const { return rbl_f2{a, b}; }
};
#define RBL_K2(a, b) (rbl_k2{a, b})
RBL_FN rbl_f2 operator*(rbl_f2 v, rbl_k2 k) { return rbl_f2{v.x * k.a, v.y * k.b}; }
RBL_FN rbl_f2 operator*(rbl_k2 k, rbl_f2 v) { return rbl_f2{k.a * v.x, k.b * v.y}; }
RBL_FN rbl_f2 operator+(rbl_f2 v, rbl_k2 k) { return rbl_f2{v.x + k.a, v.y + k.b}; }
RBL_FN rbl_f2 rbl_fma(rbl_f2 x, rbl_k2 k, rbl_f2 c) { return rbl_f2{__builtin_fmaf(x.x, k.a, c.x), __builtin_fmaf(x.y, k.b, c.y)}; }
RBL_FN rbl_f2 rbl_fma(rbl_k2 k, rbl_f2 x, rbl_f2 c) { return rbl_f2{__builtin_fmaf(k.a, x.x, c.x), __builtin_fmaf(k.b, x.y, c.y)}; }
RBL_FN rbl_f2 rbl_fma(rbl_f2 x, rbl_k2 k, rbl_k2 c) { return rbl_f2{__builtin_fmaf(x.x, k.a, c.a), __builtin_fmaf(x.y, k.b, c.b)}; }
#else
#define RBL_K2(a, b) (rbl_f2{a, b})
#endif
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
