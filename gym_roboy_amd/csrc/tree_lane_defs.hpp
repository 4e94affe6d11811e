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
// register pair with two s_mov_b32 in front of the v_pk_* that reads it (271 of them in the upper body's one-wave step; on a lone wave the
// pair costs up to 3.3 ns on top of the instruction's 3.8 - tools/pk_const_probe.hip).  The alternative: products and fused multiply-adds
// with a pair constant written PER HALF - two plain instructions with a literal operand each, the same vector-pipe time as the packed
// one (8 cycles), no scalar instructions, the same arithmetic per component (bit-identical results).  Which of the two pays depends on
// the code around it (profiles/r6_a/k2split_ab.log, k2split_euler_ab.log): straight-line Euler steps of the split forms gain 1-2 %
// (configs[3]: 8.78 -> 8.60 us); inside RK4's stage loop the packed form wins (the compiler keeps loop-invariant pairs resident in scalar
// registers: +4.5 % per half), and the one-wave form does not care.  So the constant is a type of its own whose mode is a property of the
// accessor type the generated function is instantiated with (its template parameter RBL_L: `static constexpr bool k2_split`); accessors
// without the member take RBL_K2_SPLIT (default 0: packed).
#ifndef RBL_K2_SPLIT
#define RBL_K2_SPLIT 0
#endif
template <class T, class = void> struct rbl_k2_split_of { static constexpr bool value = RBL_K2_SPLIT != 0; };
template <class T> struct rbl_k2_split_of<T, decltype(void(T::k2_split))> { static constexpr bool value = T::k2_split; };
template <bool SPLIT> struct rbl_k2c {
    float a, b;
    RBL_FN operator rbl_f2() const { return rbl_f2{a, b}; }      // (wherever the generated text hands a pair constant to a function)
};
#define RBL_K2(a, b) (rbl_k2c<rbl_k2_split_of<RBL_L>::value>{a, b})
// what an operand of a pair expression is: a pair constant (and its mode), a pair value, or a plain float that stands for both halves
template <class T> struct rbl_k2_info { static constexpr bool is = false, split = false; };
template <bool S> struct rbl_k2_info<rbl_k2c<S>> { static constexpr bool is = true, split = S; };
template <bool C, class T = void> struct rbl_enable_if {};
template <class T> struct rbl_enable_if<true, T> { typedef T type; };
RBL_FN rbl_f2 rbl_both(rbl_f2 v) { return v; }
RBL_FN rbl_f2 rbl_both(float s) { return rbl_f2{s, s}; }
template <bool S> RBL_FN rbl_f2 rbl_both(rbl_k2c<S> k) { return rbl_f2{k.a, k.b}; }
RBL_FN float rbl_h0(rbl_f2 v) { return v.x; }
RBL_FN float rbl_h1(rbl_f2 v) { return v.y; }
RBL_FN float rbl_h0(float s) { return s; }
RBL_FN float rbl_h1(float s) { return s; }
template <bool S> RBL_FN float rbl_h0(rbl_k2c<S> k) { return k.a; }
template <bool S> RBL_FN float rbl_h1(rbl_k2c<S> k) { return k.b; }
// sums, differences and products with a pair constant on either side: packed = ONE instruction on the pair (the constant in a scalar
// register pair), per half = two plain instructions with a literal each
#define RBL_K2_BINOP(OP)                                                                                                          \
    template <class A, class B, class = typename rbl_enable_if<rbl_k2_info<A>::is || rbl_k2_info<B>::is>::type>                   \
    RBL_FN rbl_f2 operator OP(A x, B y) {                                                                                         \
        if constexpr (rbl_k2_info<A>::split || rbl_k2_info<B>::split) return rbl_f2{rbl_h0(x) OP rbl_h0(y), rbl_h1(x) OP rbl_h1(y)}; \
        else return rbl_both(x) OP rbl_both(y);                                                                                   \
    }
RBL_K2_BINOP(+)
RBL_K2_BINOP(-)
RBL_K2_BINOP(*)
#undef RBL_K2_BINOP
template <bool S> RBL_FN rbl_k2c<S> operator-(rbl_k2c<S> k) { return rbl_k2c<S>{-k.a, -k.b}; }
// fused multiply-add with a pair constant in any position(s)
template <class A, class B, class C, class = typename rbl_enable_if<rbl_k2_info<A>::is || rbl_k2_info<B>::is || rbl_k2_info<C>::is>::type>
RBL_FN rbl_f2 rbl_fma(A x, B y, C z) {
    if constexpr (rbl_k2_info<A>::split || rbl_k2_info<B>::split || rbl_k2_info<C>::split)
        return rbl_f2{__builtin_fmaf(rbl_h0(x), rbl_h0(y), rbl_h0(z)), __builtin_fmaf(rbl_h1(x), rbl_h1(y), rbl_h1(z))};
    else
        return __builtin_elementwise_fma(rbl_both(x), rbl_both(y), rbl_both(z));
}
#define RBL_MK2(a, b) (rbl_f2{a, b})
RBL_FN float rbl_lo(rbl_f2 v) { return v.x; }
RBL_FN float rbl_hi(rbl_f2 v) { return v.y; }
RBL_FN float rbl_hsum(rbl_f2 v) { return v.x + v.y; }
template <bool S> RBL_FN float rbl_lo(rbl_k2c<S> k) { return k.a; }
template <bool S> RBL_FN float rbl_hi(rbl_k2c<S> k) { return k.b; }
template <bool S> RBL_FN float rbl_hsum(rbl_k2c<S> k) { return k.a + k.b; }
RBL_FN rbl_f2 rbl_sin(rbl_f2 v) { return rbl_f2{rbl_sin(v.x), rbl_sin(v.y)}; }
RBL_FN rbl_f2 rbl_cos(rbl_f2 v) { return rbl_f2{rbl_cos(v.x), rbl_cos(v.y)}; }
RBL_FN rbl_f2 rbl_rsq(rbl_f2 v) { return rbl_f2{rbl_rsq(v.x), rbl_rsq(v.y)}; }
RBL_FN rbl_f2 rbl_rcp(rbl_f2 v) { return rbl_f2{rbl_rcp(v.x), rbl_rcp(v.y)}; }
RBL_FN rbl_f2 rbl_exp2(rbl_f2 v) { return rbl_f2{rbl_exp2(v.x), rbl_exp2(v.y)}; }
RBL_FN rbl_f2 rbl_med3(rbl_f2 v, float lo, float hi) { return rbl_f2{rbl_med3(v.x, lo, hi), rbl_med3(v.y, lo, hi)}; }
RBL_FN rbl_f2 rbl_max(rbl_f2 a, float b) { return rbl_f2{rbl_max(a.x, b), rbl_max(a.y, b)}; }
RBL_FN rbl_f2 rbl_fma(rbl_f2 a, rbl_f2 b, rbl_f2 c) { return __builtin_elementwise_fma(a, b, c); }
