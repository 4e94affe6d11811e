// Probe: does a wave64 whose upper 32 lanes are inactive issue its vector instructions faster on gfx950?
// 1024 SIMDs x W waves, each wave: a loop of independent + dependent v_fma chains; lanes >= ACTIVE return at once.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void __launch_bounds__(64) k(float *out, int active, int iters) {
    const int lane = threadIdx.x;
    if (lane >= active) return;
    float a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = float(lane + i) * 1e-3f;
    const float b = 1.0001f, c = 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < ILP; ++i) a[i] = __builtin_fmaf(a[i], b, c);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i];
    if (s == 12345.0f) out[blockIdx.x * 64 + lane] = s;
}
template <int ILP>
void run(float *out, int waves_per_simd, int active) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 1024 * waves_per_simd;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<ILP>, dim3(grid), dim3(64), 0, 0, out, active, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ILP>, dim3(grid), dim3(64), 0, 0, out, active, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = double(waves_per_simd) * iters * 16 * ILP;
    printf("ILP %d, %d waves/SIMD, %2d active lanes: %.1f us, %.2f ns per vector instruction per SIMD (%.2f cycles at 2.4 GHz)\n", ILP, waves_per_simd, active,
           ms * 1e3, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
    float *out; hipMalloc(&out, 1024 * 16 * 64 * 4);
    for (int rep = 0; rep < 300; ++rep) hipLaunchKernelGGL(k<4>, dim3(8192), dim3(64), 0, 0, out, 64, 2000);   // settle the clocks
    hipDeviceSynchronize();
    for (int w : {1, 2, 4, 8}) for (int act : {64, 32, 16}) run<1>(out, w, act);
    for (int w : {1, 2, 4, 8}) for (int act : {64, 32}) run<4>(out, w, act);
    return 0;
}
