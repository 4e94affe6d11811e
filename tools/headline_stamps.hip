// Diagnostic (not product code): the headline kernel msj_step_env_per_lane<RK4, 256, 4, baked> with s_memtime / s_memrealtime
// stamps around its phases - where do the 16.5 us of a 262 144-env launch go?  Same source (msj_kernels.hpp / msj_math.hpp), same
// launch configuration; the stamps add one 64-byte store per wave.  Prints, per batch size: event time, the spread of the
// waves' start times (dispatch), and per wave the load wait, the arithmetic phase and the store drain.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -o tools/bin/headline_stamps tools/headline_stamps.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "../gym_roboy_amd/csrc/msj_kernels.hpp"

using namespace rbk;

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

#ifdef WPE
#define RB_WPE __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
#else
#define RB_WPE
#endif
template <int INTEG, bool STAMPS>
__global__ void __launch_bounds__(256) RB_WPE
stamped(float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas, const float *__restrict__ act, const Scale8 us,
        long n, unsigned long long *dbg) {
    #ifndef UNR
#define UNR 4
#endif
    constexpr int BLOCK = 256, UNROLL = UNR;
    unsigned long long t0 = 0, r0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (STAMPS) { t0 = stamp(); r0 = rstamp(); }
    const Const8 &c = rbk::BAKED;
    const long i = long(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= n) return;
    float qq[3], vv[3], sp[NT8];
    const float4 a0 = reinterpret_cast<const float4 *>(act)[2 * i];
    const float4 a1 = reinterpret_cast<const float4 *>(act)[2 * i + 1];
#pragma unroll
    for (int j = 0; j < 3; ++j) { qq[j] = q[j * n + i]; vv[j] = qd[j * n + i]; }
    if (STAMPS) { t1 = stamp(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t2 = stamp(); }
    const float a[NT8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int k = 0; k < NT8; ++k) sp[k] = a[k] * us.v[k];
    __shared__ float lds_sp[NT8][BLOCK];
#pragma unroll
    for (int k = 0; k < NT8; ++k) lds_sp[k][threadIdx.x] = sp[k];
    const bool ok = rb::MsjModel<float, NT8>::template step_sp<INTEG, UNROLL>(c, qq, vv, SpLds{&lds_sp[0][threadIdx.x], BLOCK});
    if (STAMPS) { asm volatile("" ::"v"(qq[0]), "v"(qq[1]), "v"(qq[2]), "v"(vv[0]), "v"(vv[1]), "v"(vv[2])); t3 = stamp(); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { q[j * n + i] = qq[j]; qd[j * n + i] = vv[j]; }
    feas[i] = ok ? 1u : 0u;
    if (STAMPS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t4 = stamp();
        const unsigned long long r1 = rstamp();
        if ((threadIdx.x & 63) == 0) {
            unsigned long long *d = dbg + (long(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8;
            d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3; d[4] = t4; d[5] = r0; d[6] = r1; d[7] = 0;
        }
    }
}

template <int INTEG, bool STAMPS>
float run(long n, float *q, float *qd, uint32_t *feas, const float *act, const Scale8 &us, unsigned long long *dbg) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((stamped<INTEG, STAMPS>), dim3(unsigned(n / 256)), dim3(256), 0, 0, q, qd, feas, act, us, n, dbg);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stamped<INTEG, STAMPS>), dim3(unsigned(n / 256)), dim3(256), 0, 0, q, qd, feas, act, us, n, dbg);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

// the product template itself in the same harness (eager back-to-back launches, the same action slab every step)
float run_product(long n, float *q, float *qd, uint32_t *feas, const float *act, const Scale8 &us) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    const Const8 c = rbk::BAKED_HOST;
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((msj_step_env_per_lane<1, 256, 4, true>), dim3(unsigned(n / 256)), dim3(256), 0, 0, c, q, qd, feas, act, us, n, n);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((msj_step_env_per_lane<1, 256, 4, true>), dim3(unsigned(n / 256)), dim3(256), 0, 0, c, q, qd, feas, act, us, n, n);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const long nmax = 2097152;
    float *q, *qd, *act;
    uint32_t *feas;
    unsigned long long *dbg;
    hipMalloc(&q, 12 * nmax); hipMalloc(&qd, 12 * nmax); hipMalloc(&act, 32 * nmax); hipMalloc(&feas, 4 * nmax);
    hipMalloc(&dbg, nmax / 64 * 64);
    std::vector<float> ha(8 * nmax);
    for (long i = 0; i < 8 * nmax; ++i) ha[i] = float((i * 2654435761u) % 2000) / 1000.f - 1.f;
    hipMemcpy(act, ha.data(), 32 * nmax, hipMemcpyHostToDevice);
    Scale8 us;
    for (int k = 0; k < 8; ++k) us.v[k] = 0.3f * rbk::BAKED_HOST.ten[k].ksg;
    {   // a second of the product kernel first: the clocks of an idle GPU take that long to settle
        const Const8 c = rbk::BAKED_HOST;
        for (int r = 0; r < 10000; ++r) hipLaunchKernelGGL((msj_step_env_per_lane<1, 256, 4, true>), dim3(8192), dim3(256), 0, 0, c, q, qd, feas, act, us, nmax, nmax);
        hipDeviceSynchronize();
    }
    for (long n : {262144l, 2097152l}) {
        hipMemset(q, 0, 12 * nmax); hipMemset(qd, 0, 12 * nmax);
        const float t_prod = run_product(n, q, qd, feas, act, us);
        hipMemset(q, 0, 12 * nmax); hipMemset(qd, 0, 12 * nmax);
        const float t_plain = run<1, false>(n, q, qd, feas, act, us, dbg);
        hipMemset(q, 0, 12 * nmax); hipMemset(qd, 0, 12 * nmax);
        const float t_st = run<1, true>(n, q, qd, feas, act, us, dbg);
        const long waves = n / 64;
        std::vector<unsigned long long> h(waves * 8);
        hipMemcpy(h.data(), dbg, waves * 64, hipMemcpyDeviceToHost);
        // shader clock: cycles per 10 ns tick of the real-time counter
        double ratio = 0;
        unsigned long long rmin = ~0ull, rmax = 0;
        for (long w = 0; w < waves; ++w) {
            const unsigned long long *d = &h[w * 8];
            ratio += double(d[4] - d[0]) / double(d[6] - d[5]);
            rmin = std::min(rmin, d[5]); rmax = std::max(rmax, d[6]);
        }
        const double ghz = ratio / waves * 0.1;
        std::vector<double> start(waves), life(waves), lwait(waves), comp(waves), stw(waves);
        for (long w = 0; w < waves; ++w) {
            const unsigned long long *d = &h[w * 8];
            start[w] = double(d[5] - rmin) / 100.0;                   // us after the first wave's start (real-time counter)
            life[w] = double(d[4] - d[0]) / ghz / 1e3;                // us
            lwait[w] = double(d[2] - d[0]) / ghz / 1e3;
            comp[w] = double(d[3] - d[2]) / ghz / 1e3;
            stw[w] = double(d[4] - d[3]) / ghz / 1e3;
        }
        auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[size_t(p * (v.size() - 1))]; };
        auto mean = [](const std::vector<double> &v) { double s = 0; for (double x : v) s += x; return s / v.size(); };
        printf("n = %ld (%ld waves, %.1f per SIMD): events %.2f us (the product template in this harness: %.2f), %.2f us with stamps; first start -> last end %.2f us; shader clock %.2f GHz\n",
               n, waves, waves / 1024.0, t_plain, t_prod, t_st, double(rmax - rmin) / 100.0, ghz);
        printf("  wave start after the first wave's: median %.2f, p90 %.2f, p99 %.2f, max %.2f us\n", pct(start, 0.5), pct(start, 0.9), pct(start, 0.99), pct(start, 1.0));
        printf("  per wave (mean / p10 / p90, us): lifetime %.2f / %.2f / %.2f;  loads issued + landed %.2f / %.2f / %.2f;  arithmetic %.2f / %.2f / %.2f;  stores drained %.2f / %.2f / %.2f\n",
               mean(life), pct(life, 0.1), pct(life, 0.9), mean(lwait), pct(lwait, 0.1), pct(lwait, 0.9), mean(comp), pct(comp, 0.1), pct(comp, 0.9),
               mean(stw), pct(stw, 0.1), pct(stw, 0.9));
        // arithmetic phase of the waves that START in the first 0.5 us vs the rest (later waves share the SIMD with fewer partners)
        double c_early = 0, c_late = 0; long n_early = 0, n_late = 0;
        for (long w = 0; w < waves; ++w) (start[w] < 0.5 ? (c_early += comp[w], ++n_early) : (c_late += comp[w], ++n_late));
        printf("  arithmetic phase: waves starting within 0.5 us of the first: %.2f us (%ld waves); later ones: %.2f us (%ld waves)\n",
               n_early ? c_early / n_early : 0.0, n_early, n_late ? c_late / n_late : 0.0, n_late);
    }
    return 0;
}
