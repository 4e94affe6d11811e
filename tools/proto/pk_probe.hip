// Probe (lone wave and 2 waves per SIMD): cost of packed fp32 instructions and of scalar instructions interleaved with vector ones.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CH 8
template <int FORM>
__global__ void __launch_bounds__(64) k(float *out, int iters, float sb) {
    f2 a[CH], b[CH], c[CH];
    float x[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { a[i] = f2{float(threadIdx.x + i) * 1e-3f, 1e-3f}; b[i] = f2{1.0001f, 0.9999f}; c[i] = f2{1e-6f, 2e-6f}; x[i] = i * 1e-3f; }
    unsigned s0 = __float_as_uint(sb), s1 = s0 + 1;
    const unsigned long long sp = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(s1)) << 32) | __builtin_amdgcn_readfirstlane(s0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (FORM == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c[i]));
                if (FORM == 3) asm volatile("v_fmac_f32 %0, %2, %3\n\ts_mov_b32 %1, 0x3f800347" : "+v"(x[i]), "=s"(s0) : "v"(x[(i + 1) % CH]), "v"(x[(i + 2) % CH]));   // 1 vector + 1 scalar
                if (FORM == 4) asm volatile("v_fmac_f32 %0, %3, %4\n\ts_mov_b32 %1, 0x3f800347\n\ts_mov_b32 %2, 0x3f800348" : "+v"(x[i]), "=s"(s0), "=s"(s1) : "v"(x[(i + 1) % CH]), "v"(x[(i + 2) % CH]));   // 1 vector + 2 scalar
                if (FORM == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) % CH]), "v"(x[(i + 2) % CH]));
                if (FORM == 6) asm volatile("s_mov_b32 %1, 0x3f800347\n\ts_mov_b32 %2, 0x3f800348\n\tv_pk_fma_f32 %0, %0, %3, %4" : "+v"(a[i]), "=&s"(s0), "=&s"(s1) : "v"(b[i]), "v"(c[i]));
                if (FORM == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (FORM == 8) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (FORM == 9) asm volatile("v_accvgpr_write_b32 a0, %0" :: "v"(x[i]) : "a0");
                if (FORM == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sp), "v"(c[i]));        // one SGPR-pair source
                if (FORM == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(sp));
                if (FORM == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));   // broadcast of a low half
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += a[i].x + a[i].y + x[i];
    if (s == 12345.0f) out[blockIdx.x * 64 + threadIdx.x] = s + __uint_as_float(s0) + __uint_as_float(s1);
}
template <int FORM>
void run(float *out, const char *name) {
    const int iters = 1000;
    printf("%-58s", name);
    for (int w : {1, 2, 4, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<FORM>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 1.0001f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<FORM>, dim3(1024 * w), dim3(64), 0, 0, out, iters, 1.0001f);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw: %.2f cyc", w, ms * 1e6 / (double(w) * iters * 4 * CH) * 2.4);
    }
    printf("   (cycles per asm statement per SIMD at 2.4 GHz)\n");
}
int main() {
    float *out; hipMalloc(&out, 1024 * 16 * 64 * 4);
    for (int rep = 0; rep < 400; ++rep) hipLaunchKernelGGL(k<0>, dim3(8192), dim3(64), 0, 0, out, 1000, 1.0f);
    hipDeviceSynchronize();
    run<5>(out, "v_fmac_f32");
    run<0>(out, "v_pk_fma_f32 (3 VGPR pairs)");
    run<7>(out, "v_pk_fma_f32 with op_sel swizzles");
    run<1>(out, "v_pk_mul_f32");
    run<2>(out, "v_pk_add_f32");
    run<8>(out, "v_pk_mov_b32");
    run<9>(out, "v_accvgpr_write_b32");
    run<10>(out, "v_pk_fma_f32 with one SGPR-pair source");
    run<11>(out, "v_pk_mul_f32 with an SGPR-pair source");
    run<12>(out, "v_pk_fma_f32 with a broadcast source (op_sel_hi)");
    run<3>(out, "v_fmac_f32 + 1 s_mov_b32");
    run<4>(out, "v_fmac_f32 + 2 s_mov_b32");
    run<6>(out, "2 s_mov_b32 + v_pk_fma_f32");
    return 0;
}
