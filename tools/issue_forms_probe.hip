// Probe: issue rate of vector instruction FORMS on gfx950 at W waves per SIMD, 8 independent chains per wave (inline asm pins the form).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS 8
#define REP16(X) X X X X X X X X X X X X X X X X
template <int FORM>
__global__ void __launch_bounds__(64) k(float *out, int iters, float sb, float sc) {
    float a[CHAINS], b[CHAINS], c[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { a[i] = float(threadIdx.x + i) * 1e-3f; b[i] = 1.0001f + i * 1e-6f; c[i] = 1e-6f * (i + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (FORM == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));                 // VOP2, 4 bytes
                if (FORM == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));              // VOP3, 8 bytes, 3 VGPR sources
                if (FORM == 2) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(a[i]) : "v"(b[i]));                // VOP2 + literal, 8 bytes
                if (FORM == 3) asm volatile("v_mul_f32 %0, 0x3f800347, %0" : "+v"(a[i]));                                  // VOP2 + literal
                if (FORM == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c[i]));                // VOP3 with an SGPR source
                if (FORM == 5) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));                              // VOP2, 4 bytes
                if (FORM == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[(i + 1) % CHAINS]), "v"(c[(i + 3) % CHAINS]));   // VOP3, sources from other chains' registers
                if (FORM == 7) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c[i]));
                if (FORM == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "s"(sb));                   // one SGPR twice
                if (FORM == 9) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c[i]));
                if (FORM == 10) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));                              // VOP2, SGPR src0
                if (FORM == 11) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c[i]));                  // VOP2, SGPR src0
                if (FORM == 12) asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a[i]) : "v"(c[i]));              // literal multiplier
                if (FORM == 13) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (FORM == 14) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a[i]));                                       // inline constant
                if (FORM == 15) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));
                if (FORM == 16) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a[i]) : "v"(b[i]));                        // VOP3 inline constant
                if (FORM == 17) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c[i]));
                if (FORM == 18) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));            // source modifier
                if (FORM == 19) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i];
    if (s == 12345.0f) out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int FORM>
void run(float *out, const char *name) {
    const int iters = 1000;
    printf("%-46s", name);
    for (int w : {1, 2, 4, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<FORM>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 1.0001f, 1e-6f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<FORM>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 1.0001f, 1e-6f);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw: %.2f cyc", w, ms * 1e6 / (double(w) * iters * 4 * CHAINS) * 2.4);
    }
    printf("   (cycles per instruction per SIMD at 2.4 GHz)\n");
}
int main() {
    float *out; hipMalloc(&out, 1024 * 16 * 64 * 4);
    for (int rep = 0; rep < 400; ++rep) hipLaunchKernelGGL(k<0>, dim3(8192), dim3(64), 0, 0, out, 1000, 1.0f, 0.0f);
    hipDeviceSynchronize();
    run<0>(out, "v_fmac_f32 (VOP2, 4 B)");
    run<5>(out, "v_mul_f32 (VOP2, 4 B)");
    run<7>(out, "v_add_f32 (VOP2, 4 B)");
    run<9>(out, "v_max_f32 (VOP2, 4 B)");
    run<1>(out, "v_fma_f32 3 VGPRs (VOP3, 8 B)");
    run<6>(out, "v_fma_f32 3 VGPRs of other chains (VOP3)");
    run<2>(out, "v_fmaak_f32 literal (8 B)");
    run<3>(out, "v_mul_f32 literal (8 B)");
    run<4>(out, "v_fma_f32 with one SGPR (VOP3)");
    run<8>(out, "v_fma_f32 the same SGPR twice (VOP3)");
    run<10>(out, "v_mul_f32 v, s, v (VOP2)");
    run<11>(out, "v_fmac_f32 v, s, v (VOP2)");
    run<15>(out, "v_add_f32 v, s, v (VOP2)");
    run<12>(out, "v_fmamk_f32 literal");
    run<14>(out, "v_mul_f32 inline constant 2.0");
    run<16>(out, "v_fma_f32 inline constant 1.0 (VOP3)");
    run<13>(out, "v_med3_f32");
    run<17>(out, "v_sub_f32");
    run<18>(out, "v_fma_f32 with a neg modifier");
    run<19>(out, "v_exp_f32");
    return 0;
}
