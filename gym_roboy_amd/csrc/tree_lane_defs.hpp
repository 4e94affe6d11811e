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
#define RBL_PART_BARRIER __syncthreads()

RBL_FN float rbl_sin(float x) { return __sinf(x); }
RBL_FN float rbl_cos(float x) { return __cosf(x); }
RBL_FN float rbl_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
RBL_FN float rbl_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RBL_FN float rbl_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
RBL_FN float rbl_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
RBL_FN float rbl_max(float a, float b) { return fmaxf(a, b); }
