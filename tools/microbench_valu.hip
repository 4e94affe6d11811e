// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs transcendentals on gfx950,
// at 1..8 waves per SIMD.  Decides whether packed fp32 math is worth hand-writing.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
    float a[8]; float2_ p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = float2_{a[i], a[i] + 1.f}; }
    const float m = 1.0001f + seed, c = 0.5f;
    const float2_ m2 = {m, m}, c2 = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
            else if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
            else if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);
            else if (MODE == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);
            else if (MODE == 4) a[i] = a[i] * m;
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *out; hipMalloc(&out, sizeof(float) * 256 * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_mul_f32"};
    for (int mode = 0; mode < 5; ++mode)
        for (int bpc = 1; bpc <= 8; bpc *= 2) {   // blocks of 4 waves per CU -> waves per SIMD
            dim3 grid(256 * bpc), block(256);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters, 0.f); break;
                    case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, 0.f); break;
                    case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, iters, 0.f); break;
                    case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, iters, 0.f); break;
                    case 4: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, out, iters, 0.f); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double winstr = double(iters) * 8 * bpc;            // wave-instructions per SIMD
            double ns_per = ms * 1e6 / winstr;
            printf("%-14s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n",
                   names[mode], bpc, ms, ns_per, ns_per * 2.4);
        }
    return 0;
}
