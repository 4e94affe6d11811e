// Does v_mfma_f32_32x32x2_f32 run beside f32 VALU work of OTHER waves on the same SIMD (gfx950)?
// 4 waves per SIMD (1024 blocks x 256 threads); per wave and trip: NV independent v_fma_f32 and/or NM MFMAs.
//   mode 0: VALU only   mode 1: MFMA only   mode 2: both in every wave   mode 3: even waves VALU, odd waves MFMA
// hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256, 4) k(float *out, int trips, float a, float b) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + float(threadIdx.x + i);
    f32x16 d0 = {0}, d1 = {0};
    const bool odd = (threadIdx.x >> 6) & 1;
    for (int t = 0; t < trips; ++t) {
        const bool do_v = MODE == 0 || MODE == 2 || (MODE == 3 && !odd);
        const bool do_m = MODE == 1 || MODE == 2 || (MODE == 3 && odd);
        if (do_m) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, d1, 0, 0, 0);
            }
        }
        if (do_v) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + d0[i] + d1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> float run(float *out, int trips) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, out, trips, 1.0001f, 0.5f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, out, trips, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float *out; hipMalloc(&out, 1024 * 256 * 4);
    const int trips = 2000;
    const float t0 = run<0>(out, trips), t1 = run<1>(out, trips), t2 = run<2>(out, trips), t3 = run<3>(out, trips);
    // per wave and trip: 64 v_fma (256 cycles at 4/instr) and 4 MFMAs (256 matrix-pipe cycles)
    printf("trips %d, 4 waves/SIMD; per wave-trip 64 v_fma_f32, 4 v_mfma_f32_32x32x2_f32\n", trips);
    printf("VALU only            %8.1f us  (%.2f ns per wave-trip and SIMD)\n", t0, t0 * 1e3 / trips / 4);
    printf("MFMA only            %8.1f us  (%.2f ns)\n", t1, t1 * 1e3 / trips / 4);
    printf("both in every wave   %8.1f us  (%.2f ns)   sum %.1f  max %.1f\n", t2, t2 * 1e3 / trips / 4, t0 + t1, t0 > t1 ? t0 : t1);
    printf("even VALU / odd MFMA %8.1f us  (half the work of each kind: sum/2 %.1f  max/2 %.1f)\n", t3, (t0 + t1) / 2, (t0 > t1 ? t0 : t1) / 2);
    return 0;
}
