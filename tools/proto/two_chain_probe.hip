// Probe: the headline step as ONE launch over 262 144 envs per step, or as TWO independent chains of half-batch launches (two
// streams / two parallel branches of a hipGraph): do the chains overlap each other's launch gaps and load / store phases?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../gym_roboy_amd/csrc/msj_kernels.hpp"
using namespace rbk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int INTEG, int U>
static void launch(hipStream_t st, long n, float *q, float *qd, uint32_t *feas, const float *act, const Scale8 &us) {
    const Const8 c = rbk::BAKED_HOST;
    hipLaunchKernelGGL((msj_step_env_per_lane<INTEG, 256, U, true>), dim3(unsigned(n / 256)), dim3(256), 0, st, c, q, qd, feas, act, us, n, n);
}
template <int INTEG, int U>
int run(const char *name, long n, int C) {
    float *q[8], *qd[8], *act[8]; uint32_t *feas[8];
    for (int h = 0; h < C; ++h) {
        CK(hipMalloc(&q[h], 12 * n)); CK(hipMalloc(&qd[h], 12 * n)); CK(hipMalloc(&act[h], 32 * n)); CK(hipMalloc(&feas[h], 4 * n));
        CK(hipMemset(q[h], 0, 12 * n)); CK(hipMemset(qd[h], 0, 12 * n));
        std::vector<float> ha(8 * n);
        for (long i = 0; i < 8 * n; ++i) ha[i] = float((i * 2654435761u + h) % 2000) / 1000.f - 1.f;
        CK(hipMemcpy(act[h], ha.data(), 32 * n, hipMemcpyHostToDevice));
    }
    Scale8 us;
    for (int k = 0; k < 8; ++k) us.v[k] = 0.3f * rbk::BAKED_HOST.ten[k].ksg;
    hipStream_t s0, sx[8]; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); for (int h = 1; h < C; ++h) CK(hipStreamCreateWithFlags(&sx[h], hipStreamNonBlocking)); sx[0] = s0;
    hipEvent_t joins[8]; for (int h = 1; h < C; ++h) CK(hipEventCreateWithFlags(&joins[h], hipEventDisableTiming));
    hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int K = 100, reps = 20;
    // graphs: (a) one chain of K launches over both halves as ONE array is emulated by two back-to-back half launches? no: a true
    // single launch needs one array - use half 0 with 2 n envs?  Keep it simple: "single" = one launch of n envs on arrays of size n;
    // "two chains" = two launches of n / 2 envs each (first halves of the two array sets).
    hipGraph_t g1, g2; hipGraphExec_t x1, x2;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int t = 0; t < K; ++t) launch<INTEG, U>(s0, n, q[0], qd[0], feas[0], act[0], us);
    CK(hipStreamEndCapture(s0, &g1)); CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(fork, s0)); for (int h = 1; h < C; ++h) CK(hipStreamWaitEvent(sx[h], fork, 0));
    for (int t = 0; t < K; ++t) for (int h = 0; h < C; ++h) launch<INTEG, U>(sx[h], n / C, q[h], qd[h], feas[h], act[h], us);
    for (int h = 1; h < C; ++h) { CK(hipEventRecord(joins[h], sx[h])); CK(hipStreamWaitEvent(s0, joins[h], 0)); }
    CK(hipStreamEndCapture(s0, &g2)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
    for (int w = 0; w < 60; ++w) { CK(hipGraphLaunch(x1, s0)); }                       // settle the clocks
    CK(hipStreamSynchronize(s0));
    float ms1 = 0, ms2 = 0;
    CK(hipEventRecord(e0, s0)); for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(x1, s0)); CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0)); CK(hipEventElapsedTime(&ms1, e0, e1));
    CK(hipGraphLaunch(x2, s0)); CK(hipStreamSynchronize(s0));
    CK(hipEventRecord(e0, s0)); for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(x2, s0)); CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0)); CK(hipEventElapsedTime(&ms2, e0, e1));
    printf("%s, %ld envs per step: one launch per step %.2f us; %d chains %.2f us per step\n", name, n, ms1 * 1e3 / (K * reps), C, ms2 * 1e3 / (K * reps));
    for (int h = 0; h < C; ++h) { hipFree(q[h]); hipFree(qd[h]); hipFree(act[h]); hipFree(feas[h]); }
    return 0;
}
int main() {
    for (int rep = 0; rep < 3; ++rep) {
        if (run<0, 8>("Euler", 262144, 2)) return 1;
        if (run<1, 4>("RK4", 262144, 2)) return 1;
        if (run<0, 8>("Euler", 524288, 2)) return 1;
    }
    return 0;
}
