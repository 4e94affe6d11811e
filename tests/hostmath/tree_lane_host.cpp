// TEST HARNESS (not product code): the text tree_lane_gen.hpp generates for one robot, compiled with g++ so that
// tests/test_tree_lane_gen.py can check the generated acceleration against the fp64 oracle without a GPU.
// Compile with -DRBL_GENERATED='"path/to/generated.hpp"'.
#include <cmath>

#define RBL_FN inline
#define RBL_TABLE(name, n) constexpr float name[n]
#define RBL_SCHED_BARRIER
#define RBL_LDS(slot) rbl_lds[slot]
#define RBL_NS rbl_host

inline float rbl_sin(float x) { return std::sin(x); }
inline float rbl_cos(float x) { return std::cos(x); }
inline float rbl_rsq(float x) { return 1.0f / std::sqrt(x); }
inline float rbl_rcp(float x) { return 1.0f / x; }
inline float rbl_exp2(float x) { return std::exp2(x); }
inline float rbl_med3(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
inline float rbl_max(float a, float b) { return std::fmax(a, b); }

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
inline rbl_f2 rbl_sin(rbl_f2 v) { return rbl_f2{rbl_sin(v.x), rbl_sin(v.y)}; }
inline rbl_f2 rbl_cos(rbl_f2 v) { return rbl_f2{rbl_cos(v.x), rbl_cos(v.y)}; }
inline rbl_f2 rbl_rsq(rbl_f2 v) { return rbl_f2{rbl_rsq(v.x), rbl_rsq(v.y)}; }
inline rbl_f2 rbl_rcp(rbl_f2 v) { return rbl_f2{rbl_rcp(v.x), rbl_rcp(v.y)}; }
inline rbl_f2 rbl_exp2(rbl_f2 v) { return rbl_f2{rbl_exp2(v.x), rbl_exp2(v.y)}; }
inline rbl_f2 rbl_med3(rbl_f2 v, float lo, float hi) { return rbl_f2{rbl_med3(v.x, lo, hi), rbl_med3(v.y, lo, hi)}; }
inline rbl_f2 rbl_max(rbl_f2 a, float b) { return rbl_f2{rbl_max(a.x, b), rbl_max(a.y, b)}; }

#include RBL_GENERATED

extern "C" int tl_dims(int *out) { out[0] = RBL_NQ; out[1] = RBL_NT; out[2] = RBL_ACCEL_LDS; return 0; }
// qdd of n envs: rows q[n][nq], qd[n][nq], set-points sp[n][nt] (already in set-point units: the activation
// offset is sp * KSG, as the kernels form it)
extern "C" int tl_accel(const float *q, const float *qd, const float *sp, float *qdd, int n) {
    for (int e = 0; e < n; ++e) {
        float qq[RBL_NQ], vv[RBL_NQ], spu[RBL_NT], a[RBL_NQ];
        float lds[RBL_ACCEL_LDS + 1];
        for (int j = 0; j < RBL_NQ; ++j) { qq[j] = q[e * RBL_NQ + j]; vv[j] = qd[e * RBL_NQ + j]; }
        for (int k = 0; k < RBL_NT; ++k) spu[k] = sp[e * RBL_NT + k] * rbl_host::KSG[k];
        rbl_host::rbl_accel(qq, vv, spu, a, lds);
        for (int j = 0; j < RBL_NQ; ++j) qdd[e * RBL_NQ + j] = a[j];
    }
    return 0;
}
