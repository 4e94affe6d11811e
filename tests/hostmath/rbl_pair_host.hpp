// TEST HARNESS: host forms of the pair type the generated joint-tree code uses (tree_lane_defs.hpp has the device forms).
#pragma once
// pair values (a subtree and its mate as one stream): two floats with component-wise arithmetic, floats broadcast
struct rbl_f2 { float x, y; };
#define RBL_K2(a, b) (rbl_f2{a, b})
#define RBL_MK2(a, b) (rbl_f2{a, b})
inline rbl_f2 rbl_b(float a) { return rbl_f2{a, a}; }
inline rbl_f2 operator+(rbl_f2 a, rbl_f2 b) { return rbl_f2{a.x + b.x, a.y + b.y}; }
inline rbl_f2 operator-(rbl_f2 a, rbl_f2 b) { return rbl_f2{a.x - b.x, a.y - b.y}; }
inline rbl_f2 operator*(rbl_f2 a, rbl_f2 b) { return rbl_f2{a.x * b.x, a.y * b.y}; }
inline rbl_f2 operator-(rbl_f2 a) { return rbl_f2{-a.x, -a.y}; }
inline rbl_f2 operator+(rbl_f2 a, float b) { return a + rbl_b(b); }
inline rbl_f2 operator+(float a, rbl_f2 b) { return rbl_b(a) + b; }
inline rbl_f2 operator-(rbl_f2 a, float b) { return a - rbl_b(b); }
inline rbl_f2 operator-(float a, rbl_f2 b) { return rbl_b(a) - b; }
inline rbl_f2 operator*(rbl_f2 a, float b) { return a * rbl_b(b); }
inline rbl_f2 operator*(float a, rbl_f2 b) { return rbl_b(a) * b; }
inline float rbl_lo(rbl_f2 v) { return v.x; }
inline float rbl_hi(rbl_f2 v) { return v.y; }
inline float rbl_lo(float v) { return v; }
inline float rbl_hi(float v) { return v; }
inline float rbl_hsum(rbl_f2 v) { return v.x + v.y; }
inline rbl_f2 rbl_fma(rbl_f2 a, rbl_f2 b, rbl_f2 c) { return rbl_f2{std::fma(a.x, b.x, c.x), std::fma(a.y, b.y, c.y)}; }
inline rbl_f2 rbl_sin(rbl_f2 v) { return rbl_f2{rbl_sin(v.x), rbl_sin(v.y)}; }
inline rbl_f2 rbl_cos(rbl_f2 v) { return rbl_f2{rbl_cos(v.x), rbl_cos(v.y)}; }
inline rbl_f2 rbl_rsq(rbl_f2 v) { return rbl_f2{rbl_rsq(v.x), rbl_rsq(v.y)}; }
inline rbl_f2 rbl_rcp(rbl_f2 v) { return rbl_f2{rbl_rcp(v.x), rbl_rcp(v.y)}; }
inline rbl_f2 rbl_exp2(rbl_f2 v) { return rbl_f2{rbl_exp2(v.x), rbl_exp2(v.y)}; }
inline rbl_f2 rbl_med3(rbl_f2 v, float lo, float hi) { return rbl_f2{rbl_med3(v.x, lo, hi), rbl_med3(v.y, lo, hi)}; }
inline rbl_f2 rbl_max(rbl_f2 a, float b) { return rbl_f2{rbl_max(a.x, b), rbl_max(a.y, b)}; }

