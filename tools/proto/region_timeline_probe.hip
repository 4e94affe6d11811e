// Probe (round 4): where do the ~30 us go that a K-step two-chain rollout costs beyond K x the steady-state time per step?
// Stamp kernels (s_memrealtime, 10 ns ticks) in stream order: before the fork, at the head and the tail of each chain, after the
// join.  Variants: two linear graphs / eager launches taking turns / eager head + graphs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../../gym_roboy_amd/csrc/msj_kernels.hpp"
using namespace rbk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void stamp(unsigned long long *slot) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    *slot = t;
}
// join without events: the chain's last kernel publishes a sequence number, the handle's stream spins on it (bounded)
__global__ void flag_set(uint32_t *flag, uint32_t v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void flag_wait(uint32_t *flag, uint32_t v) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if (int(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - v) >= 0) return;
        __builtin_amdgcn_s_sleep(4);
    }
}
static void launch(hipStream_t st, long n, long cnt, float *q, float *qd, uint32_t *feas, const float *act, const Scale8 &us) {
    const Const8 c = rbk::BAKED_HOST;
    hipLaunchKernelGGL((msj_step_env_per_lane_rs<1, 256, true, true>), dim3(unsigned(cnt / 256)), dim3(256), 0, st, c, q, qd, feas, act, us, n, cnt);
}
int main() {
    const long n = 262144, h = n / 2;
    float *q, *qd, *act; uint32_t *feas; unsigned long long *st_d;
    CK(hipMalloc(&q, 12 * n)); CK(hipMalloc(&qd, 12 * n)); CK(hipMalloc(&act, 32 * n)); CK(hipMalloc(&feas, 4 * n));
    CK(hipMalloc(&st_d, 8 * 8)); CK(hipMemset(q, 0, 12 * n)); CK(hipMemset(qd, 0, 12 * n));
    std::vector<float> ha(8 * n);
    for (long i = 0; i < 8 * n; ++i) ha[i] = float((i * 2654435761u) % 2000) / 1000.f - 1.f;
    CK(hipMemcpy(act, ha.data(), 32 * n, hipMemcpyHostToDevice));
    Scale8 us;
    for (int k = 0; k < 8; ++k) us.v[k] = 0.3f * rbk::BAKED_HOST.ten[k].ksg;
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int K = 20;
    hipGraph_t ga, gb; hipGraphExec_t xa[2], xb[2];
    for (int v = 0; v < 2; ++v) {              // graphs of K and of K - 4 steps
        const int kk = v ? K - 4 : K;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < kk; ++t) launch(s0, n, h, q, qd, feas, act, us);
        CK(hipStreamEndCapture(s0, &ga)); CK(hipGraphInstantiate(&xa[v], ga, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < kk; ++t) launch(s1, n, h, q + h, qd + h, feas + h, act + 8 * h, us);
        CK(hipStreamEndCapture(s1, &gb)); CK(hipGraphInstantiate(&xb[v], gb, nullptr, nullptr, 0));
    }
    const char *names[] = {"two graphs", "eager, taking turns", "eager head of 4 + graphs", "two graphs, chain 1 launched first", "one chain (graph of whole-batch launches)",
                           "chain 1 first, fork / join by stream memory ops", "chain 1 first, join by a spinning kernel"};
    int can_wait = 0; CK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can_wait);
    uint32_t *sigs[3];
    for (int i = 0; i < 2; ++i) { CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&sigs[i]), 8, hipMallocSignalMemory)); CK(hipMemset(sigs[i], 0, 8)); }
    CK(hipMalloc(reinterpret_cast<void **>(&sigs[2]), 8)); CK(hipMemset(sigs[2], 0, 8));
    uint32_t seq = 0;
    hipGraphExec_t xw;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int t = 0; t < K; ++t) launch(s0, n, n, q, qd, feas, act, us);
    CK(hipStreamEndCapture(s0, &ga)); CK(hipGraphInstantiate(&xw, ga, nullptr, nullptr, 0));
    for (int variant = 0; variant < 7; ++variant) {
        if (variant == 5 && !can_wait) continue;
        std::vector<std::vector<double>> rows;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s0, st_d + 0);
            if (variant == 4) {
                CK(hipGraphLaunch(xw, s0));
                hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s0, st_d + 5);
                CK(hipDeviceSynchronize());
                unsigned long long hst[8]; CK(hipMemcpy(hst, st_d, 64, hipMemcpyDeviceToHost));
                rows.push_back({0, 0, 0, 0, (hst[5] - hst[0]) * 0.01});
                continue;
            }
            ++seq;
            if (variant == 5) { CK(hipStreamWriteValue32(s0, sigs[0], seq, 0)); CK(hipStreamWaitValue32(s1, sigs[0], seq, hipStreamWaitValueGte, 0xffffffffu)); }
            else { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s0, st_d + 1);
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s1, st_d + 2);
            if (variant == 0) { CK(hipGraphLaunch(xa[0], s0)); CK(hipGraphLaunch(xb[0], s1)); }
            else if (variant == 3 || variant >= 5) { CK(hipGraphLaunch(xb[0], s1)); CK(hipGraphLaunch(xa[0], s0)); }
            else {
                const int eager = variant == 1 ? K : 4;
                for (int t = 0; t < eager; ++t) { launch(s0, n, h, q, qd, feas, act, us); launch(s1, n, h, q + h, qd + h, feas + h, act + 8 * h, us); }
                if (variant == 2) { CK(hipGraphLaunch(xa[1], s0)); CK(hipGraphLaunch(xb[1], s1)); }
            }
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s0, st_d + 3);
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s1, st_d + 4);
            if (variant == 5) { CK(hipStreamWriteValue32(s1, sigs[1], seq, 0)); CK(hipStreamWaitValue32(s0, sigs[1], seq, hipStreamWaitValueGte, 0xffffffffu)); }
            else if (variant == 6) { hipLaunchKernelGGL(flag_set, dim3(1), dim3(1), 0, s1, sigs[2], seq); hipLaunchKernelGGL(flag_wait, dim3(1), dim3(1), 0, s0, sigs[2], seq); }
            else { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s0, st_d + 5);
            CK(hipDeviceSynchronize());
            unsigned long long hst[8]; CK(hipMemcpy(hst, st_d, 64, hipMemcpyDeviceToHost));
            std::vector<double> r;
            for (int i = 1; i <= 5; ++i) r.push_back((hst[i] - hst[0]) * 0.01);
            rows.push_back(r);
        }
        // median of each column over the repeats (the first two dropped)
        printf("%-45s", names[variant]);
        for (int c = 0; c < 5; ++c) {
            std::vector<double> col;
            for (size_t r = 2; r < rows.size(); ++r) col.push_back(rows[r][c]);
            std::sort(col.begin(), col.end());
            printf(" %8.1f", col[col.size() / 2]);
        }
        printf("   us after the first stamp: head of chain 0, head of chain 1, tail of chain 0, tail of chain 1, after the join (%d steps)\n", K);
    }
    return 0;
}
