// TEST HARNESS (not product code): the split-form text tree_lane_gen.hpp generates for one robot (one function per
// wave, an exchange area and one barrier between them), compiled with g++: every part runs in a thread of its own, the
// barrier is a pthread barrier.  tests/test_tree_lane_gen.py checks the result against the fp64 oracle and that the
// trunk's accelerations come out bit-identical in every part.  Compile with -DRBL_GENERATED='"path/to/generated.hpp"' -pthread.
#include <pthread.h>

#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#define RBL_FN inline
#define RBL_TABLE(name, n) constexpr float name[n]
#define RBL_ITABLE(name, n) constexpr int name[n]
#define RBL_SCHED_BARRIER
#define RBL_LDS(slot) rbl_lds[slot]
#define RBL_X(slot) rbl_x[slot]
#define RBL_PART_BARRIER rbl_host_barrier()
#define RBL_NS rbl_host

static pthread_barrier_t g_barrier;
inline void rbl_host_barrier() { pthread_barrier_wait(&g_barrier); }
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
#ifndef RBL_NHELPERS
#define RBL_NHELPERS 0
#endif

extern "C" int tl_dims(int *out) { out[0] = RBL_NQ; out[1] = RBL_NT; out[2] = RBL_NPARTS; out[3] = RBL_PART_LDS; out[4] = RBL_X_SLOTS; out[5] = RBL_NHELPERS; return 0; }
// qdd of n envs; returns the number of trunk accelerations that differ between two parts (must be 0)
extern "C" int tl_accel(const float *q, const float *qd, const float *sp, float *qdd, int n) {
    int mismatches = 0;
    for (int e = 0; e < n; ++e) {
        float qq[RBL_NQ], vv[RBL_NQ], spu[RBL_NT];
        for (int j = 0; j < RBL_NQ; ++j) { qq[j] = q[e * RBL_NQ + j]; vv[j] = qd[e * RBL_NQ + j]; }
        for (int k = 0; k < RBL_NT; ++k) spu[k] = sp[e * RBL_NT + k] * rbl_host::KSG[k];
        float x[RBL_X_SLOTS + 1];
        float a[RBL_NPARTS][RBL_NQ];
        std::memset(a, 0, sizeof a);
        // (helper waves - "parts" RBL_NPARTS .. - take their state from the exchange area: they get zeros for q / qd)
        pthread_barrier_init(&g_barrier, nullptr, RBL_NPARTS + RBL_NHELPERS);
        std::vector<std::thread> th;
        for (int p = 0; p < RBL_NPARTS + RBL_NHELPERS; ++p)
            th.emplace_back([&, p] {
                float lds[RBL_PART_LDS + 1];
                if (p < RBL_NPARTS) { rbl_host::rbl_part(p, qq, vv, spu, a[p], lds, x); return; }
                float zq[RBL_NQ] = {}, za[RBL_NQ] = {};
                rbl_host::rbl_part(p, zq, zq, spu, za, lds, x);
            });
        for (auto &t : th) t.join();
        pthread_barrier_destroy(&g_barrier);
        for (int j = 0; j < RBL_NQ; ++j) {
            const int owner = rbl_host::PART_OF_JOINT[j];
            if (owner < 0) {
                for (int p = 1; p < RBL_NPARTS; ++p) mismatches += std::memcmp(&a[p][j], &a[0][j], sizeof(float)) != 0;
                qdd[e * RBL_NQ + j] = a[0][j];
            } else {
                qdd[e * RBL_NQ + j] = a[owner][j];
            }
        }
    }
    return mismatches;
}
