// Probe (round 6): what do the NON-arithmetic issue slots of the one-wave joint-tree kernel cost?  The generated
// upper-body step issues 5 569 vector instructions per wave-step of which 2 492 are v_pk_* on pair values, plus ~278 s_mov_b32 pairs
// that build the pair CONSTANTS (two different literals cannot ride in one v_pk instruction) and ~534 v_accvgpr moves.  Candidates:
// pair constants from a scalar-loaded table, long-lived values in AGPR-only roles.  This probe prices the forms on a lone wave per SIMD
// (and 2 / 4 waves), CHAINS dependent chains per wave (1 = fully dependent, as most of the generated trunk is; 2, 4):
//   A  v_pk_fma_f32 on VGPR pairs only                                  (the floor)
//   B  ... with the constant in an SGPR pair written by TWO s_mov_b32 in front of EVERY instruction   (what the kernel does today)
//   C  ... with the constant in an SGPR pair written ONCE outside the loop                             (an ideal resident table)
//   D  ... with the constant s_load_dwordx2'ed from memory in front of every instruction + s_waitcnt   (a naive table)
//   E  A + one v_accvgpr_read_b32 per instruction (a value parked in an AGPR and fetched for its use)
//   F  A + one ds_read_b32 per instruction + s_waitcnt (the same value parked in LDS instead)
// hipcc --offload-arch=gfx950 -O3 -o pk_const_probe tools/pk_const_probe.hip && ./pk_const_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __constant__ float ktab[64];

template <int FORM, int CHAINS>
__global__ void __launch_bounds__(64) k(float *out, int iters) {
    f2 a[CHAINS], b[CHAINS];
    __shared__ float lds[64 * 4];
    lds[threadIdx.x] = 1e-6f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { a[i] = f2{float(threadIdx.x + i) * 1e-3f, 1e-3f}; b[i] = f2{1.0001f + i * 1e-6f, 0.9999f}; }
    float park = 1e-6f;
    asm volatile("v_accvgpr_write_b32 a0, %0" ::"v"(park));
    const float *kt = ktab;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / CHAINS; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (FORM == 1) asm volatile("s_mov_b32 s20, 0x3f800347\n s_mov_b32 s21, 0x3f7ffcb9\n v_pk_fma_f32 %0, %0, s[20:21], %1" : "+v"(a[i]) : "v"(b[i]) : "s20", "s21");
                if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %0, s[22:23], %1" : "+v"(a[i]) : "v"(b[i]));
                if (FORM == 3) asm volatile("s_load_dwordx2 s[20:21], %2, 0x10\n s_waitcnt lgkmcnt(0)\n v_pk_fma_f32 %0, %0, s[20:21], %1" : "+v"(a[i]) : "v"(b[i]), "s"(kt) : "s20", "s21");
                if (FORM == 4) asm volatile("v_accvgpr_read_b32 %1, a0\n v_pk_fma_f32 %0, %0, %2, %2" : "+v"(a[i]), "=&v"(park) : "v"(b[i]));
                if (FORM == 5) asm volatile("ds_read_b32 %1, %3\n s_waitcnt lgkmcnt(0)\n v_pk_fma_f32 %0, %0, %2, %2" : "+v"(a[i]), "=&v"(park) : "v"(b[i]), "v"(int(threadIdx.x * 4)));
            }
    }
    float s = park;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i].x + a[i].y;
    if (s == 12345.0f) out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int FORM, int CHAINS>
void run(float *out, const char *name) {
    const int iters = 2000;
    printf("%-72s chains %d:", name, CHAINS);
    for (int w : {1, 2, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<FORM, CHAINS>), dim3(1024 * w), dim3(64), 0, 0, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<FORM, CHAINS>), dim3(1024 * w), dim3(64), 0, 0, out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw %.2f ns", w, ms * 1e6 / (double(w) * iters * 16));
    }
    printf("   (ns per v_pk_fma per SIMD)\n");
}
template <int CHAINS>
void all(float *out) {
    run<0, CHAINS>(out, "A  v_pk_fma_f32, VGPR pairs");
    run<1, CHAINS>(out, "B  + constant pair by two s_mov_b32 literals before every instruction");
    run<2, CHAINS>(out, "C  + constant pair resident in an SGPR pair");
    run<3, CHAINS>(out, "D  + constant pair s_load_dwordx2 + wait before every instruction");
    run<4, CHAINS>(out, "E  A + v_accvgpr_read_b32 per instruction");
    run<5, CHAINS>(out, "F  A + ds_read_b32 + wait per instruction");
}
int main() {
    float *out; hipMalloc(&out, 1024 * 8 * 64 * 4);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0f + 1e-6f * i;
    hipMemcpyToSymbol(HIP_SYMBOL(ktab), h, sizeof(h));
    for (int rep = 0; rep < 200; ++rep) hipLaunchKernelGGL((k<0, 4>), dim3(8192), dim3(64), 0, 0, out, 2000);
    hipDeviceSynchronize();
    all<1>(out); all<2>(out); all<4>(out);
    return 0;
}
