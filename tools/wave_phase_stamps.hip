// Diagnostic: per-wave phase timeline of the env-per-lane Euler step (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>
#include "../gym_roboy_amd/csrc/msj_build.hpp"
#include "msj_const_baked.hpp"
using Const8 = rb::MsjConst<float, 8>;

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long rstamp() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int NSUB>
__global__ void __launch_bounds__(256) k(const Const8 c, float *q, float *qd, unsigned *feas, const float *act,
                                         float act_scale, long n, unsigned long long *dbg) {
    const unsigned long long t0 = stamp(), r0 = rstamp();
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    float qq[3], vv[3], sp[8];
    const float4 a0 = reinterpret_cast<const float4 *>(act)[2 * i];
    const float4 a1 = reinterpret_cast<const float4 *>(act)[2 * i + 1];
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + i]; vv[j] = qd[j * n + i]; }
    const unsigned long long t1 = stamp();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = stamp();
    sp[0] = a0.x * act_scale; sp[1] = a0.y * act_scale; sp[2] = a0.z * act_scale; sp[3] = a0.w * act_scale;
    sp[4] = a1.x * act_scale; sp[5] = a1.y * act_scale; sp[6] = a1.z * act_scale; sp[7] = a1.w * act_scale;
    bool ok = true;
    for (int s = 0; s < NSUB; ++s) ok = rb::MsjModel<float, 8>::template step<0>(c, qq, vv, sp) && ok;
    asm volatile("" ::"v"(qq[0]), "v"(qq[1]), "v"(qq[2]), "v"(vv[0]), "v"(vv[1]), "v"(vv[2]));
    const unsigned long long t3 = stamp();
    for (int j = 0; j < 3; ++j) { q[j * n + i] = qq[j]; qd[j * n + i] = vv[j]; }
    feas[i] = ok ? 1u : 0u;
    const unsigned long long t4 = stamp();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t5 = stamp(), r1 = rstamp();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *d = dbg + (long(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8;
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3; d[4] = t4; d[5] = t5; d[6] = r0; d[7] = r1;
    }
}

int main() {
    const long n = 2097152;
    float *q, *qd, *act; unsigned *feas; unsigned long long *dbg;
    hipMalloc(&q, 12 * n); hipMalloc(&qd, 12 * n); hipMalloc(&act, 32 * n); hipMalloc(&feas, 4 * n);
    const long waves = n / 64;
    hipMalloc(&dbg, waves * 64);
    hipMemset(q, 0, 12 * n); hipMemset(qd, 0, 12 * n);
    std::vector<float> ha(8 * n);
    for (long i = 0; i < 8 * n; ++i) ha[i] = float((i * 2654435761u) % 2000) / 1000.f - 1.f;
    hipMemcpy(act, ha.data(), 32 * n, hipMemcpyHostToDevice);
    Const8 c = rb::kBaked;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 20; ++rep) {
        if (rep == 19) hipEventRecord(e0);
        hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, c, q, qd, feas, act, 0.3f, n, dbg);
        if (rep == 19) hipEventRecord(e1);
    }
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves * 8);
    hipMemcpy(h.data(), dbg, waves * 64, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0, rmin = ~0ull, rmax = 0;
    double s_issue = 0, s_wait = 0, s_comp = 0, s_st = 0, s_stw = 0, s_tot = 0;
    for (long w = 0; w < waves; ++w) {
        unsigned long long *d = &h[w * 8];
        tmin = std::min(tmin, d[0]); tmax = std::max(tmax, d[5]); rmin = std::min(rmin, d[6]); rmax = std::max(rmax, d[7]);
        s_issue += d[1] - d[0]; s_wait += d[2] - d[1]; s_comp += d[3] - d[2]; s_st += d[4] - d[3]; s_stw += d[5] - d[4]; s_tot += d[5] - d[0];
    }
    double ratio = 0; for (long w = 0; w < waves; ++w) { unsigned long long *d = &h[w * 8]; ratio += double(d[5] - d[0]) / double(d[7] - d[6]); }
    printf("shader cycles per 10ns realtime tick (mean over waves): %.3f -> shader clock %.3f GHz\n", ratio / waves, ratio / waves * 0.1);
    const double span_cyc = double(tmax - tmin), span_us = double(rmax - rmin) / 100.0;
    printf("kernel event time %.2f us; stamp span %.0f shader cycles = %.2f us realtime -> clock %.3f GHz\n", ms * 1e3, span_cyc, span_us, span_cyc / span_us / 1e3);
    printf("per-wave mean cycles: load-issue %.0f, load-wait %.0f, compute %.0f, store-issue %.0f, store-wait %.0f, total %.0f\n",
           s_issue / waves, s_wait / waves, s_comp / waves, s_st / waves, s_stw / waves, s_tot / waves);
    printf("mean concurrency (sum of wave lifetimes / span): %.1f waves per SIMD\n", s_tot / span_cyc / 1024.0);
    printf("mean waves in compute per SIMD: %.2f; in load-wait: %.2f; in store-wait: %.2f\n", s_comp / span_cyc / 1024.0, s_wait / span_cyc / 1024.0, s_stw / span_cyc / 1024.0);
    return 0;
}
