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
inline float rbl_fma(float a, float b, float c) { return std::fma(a, b, c); }

#include "rbl_pair_host.hpp"

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
