// Probe: two chains of half-batch launches, replayed as SHORT graphs (K steps, a synchronisation between graphs) - does a delay at the
// head of the second chain (so that the chains start out of phase instead of drifting apart over tens of steps) recover the steady state?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../gym_roboy_amd/csrc/msj_kernels.hpp"
using namespace rbk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void delay_kernel(unsigned ticks) {              // one wave; s_memrealtime ticks are 10 ns
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    do { __builtin_amdgcn_s_sleep(8); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); } while (t - t0 < ticks);
}
template <int INTEG, int U>
static void launch(hipStream_t st, long n, long cnt, float *q, float *qd, uint32_t *feas, const float *act, const Scale8 &us) {
    const Const8 c = rbk::BAKED_HOST;
    hipLaunchKernelGGL((msj_step_env_per_lane<INTEG, 256, U, true>), dim3(unsigned(cnt / 256)), dim3(256), 0, st, c, q, qd, feas, act, us, n, cnt);
}
int main() {
    const long n = 262144, h = n / 2;
    float *q, *qd, *act; uint32_t *feas;
    CK(hipMalloc(&q, 12 * n)); CK(hipMalloc(&qd, 12 * n)); CK(hipMalloc(&act, 32 * n)); CK(hipMalloc(&feas, 4 * n));
    CK(hipMemset(q, 0, 12 * n)); CK(hipMemset(qd, 0, 12 * n));
    std::vector<float> ha(8 * n);
    for (long i = 0; i < 8 * n; ++i) ha[i] = float((i * 2654435761u) % 2000) / 1000.f - 1.f;
    CK(hipMemcpy(act, ha.data(), 32 * n, hipMemcpyHostToDevice));
    Scale8 us;
    for (int k = 0; k < 8; ++k) us.v[k] = 0.3f * rbk::BAKED_HOST.ten[k].ksg;
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    for (int K : {20, 40, 100}) {                               // two LINEAR graphs, one per chain, launched on two streams
        hipGraph_t ga, gb; hipGraphExec_t xa, xb;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < K; ++t) launch<1, 4>(s0, n, h, q, qd, feas, act, us);
        CK(hipStreamEndCapture(s0, &ga)); CK(hipGraphInstantiate(&xa, ga, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < K; ++t) launch<1, 4>(s1, n, h, q + h, qd + h, feas + h, act + 8 * h, us);
        CK(hipStreamEndCapture(s1, &gb)); CK(hipGraphInstantiate(&xb, gb, nullptr, nullptr, 0));
        auto region = [&]() -> int {
            CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
            CK(hipGraphLaunch(xa, s0)); CK(hipGraphLaunch(xb, s1));
            CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
            return 0;
        };
        for (int w = 0; w < 30; ++w) { if (region()) return 1; CK(hipStreamSynchronize(s0)); }
        const int reps = 60;
        double total = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, s0)); if (region()) return 1; CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); total += ms;
        }
        printf("K = %3d, two linear graphs on two streams: %.2f us per step\n", K, total * 1e3 / (reps * K));
    }
    for (int K : {20, 40, 100}) for (int delay_us : {-1, 0}) {
        hipGraph_t g; hipGraphExec_t x;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        if (delay_us < 0) { for (int t = 0; t < K; ++t) launch<1, 4>(s0, n, n, q, qd, feas, act, us); }
        else {
            CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
            if (delay_us > 0) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s1, unsigned(delay_us * 100));
            for (int t = 0; t < K; ++t) { launch<1, 4>(s0, n, h, q, qd, feas, act, us); launch<1, 4>(s1, n, h, q + h, qd + h, feas + h, act + 8 * h, us); }
            CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
        }
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
        for (int w = 0; w < 30; ++w) { CK(hipGraphLaunch(x, s0)); CK(hipStreamSynchronize(s0)); }
        const int reps = 60;
        double total = 0;
        for (int r = 0; r < reps; ++r) {                      // a synchronisation between graphs, as a timed bench region has
            CK(hipEventRecord(e0, s0)); CK(hipGraphLaunch(x, s0)); CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); total += ms;
        }
        printf("K = %3d, %s: %.2f us per step\n", K, delay_us < 0 ? "one launch per step" : (std::string("two chains, second delayed by ") + std::to_string(delay_us) + " us").c_str(), total * 1e3 / (reps * K));
        hipGraphExecDestroy(x); hipGraphDestroy(g);
    }
    return 0;
}
