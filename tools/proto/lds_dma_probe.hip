// Probe: global_load_lds_dword into LDS above 64 KB (gfx950 has 160 KB), per-lane source addresses, wave-uniform M0 base.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma_dword(const float *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void probe(const float *src, const int *idx, float *out, int base_floats) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *dst = lds + base_floats + wave * 65 * 4;
    for (int k = 0; k < 4; ++k) {
        const unsigned a = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>(dst + k * 65)));
        dma_dword(src + idx[threadIdx.x] * 4 + k, a);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int k = 0; k < 4; ++k) out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + k] = dst[k * 65 + lane];
}
int main() {
    const int n = 256;
    std::vector<float> h(n * 4); std::vector<int> hi(n);
    for (int i = 0; i < n * 4; ++i) h[i] = float(i);
    for (int i = 0; i < n; ++i) hi[i] = (i * 37) % n;
    float *src, *out; int *idx;
    hipMalloc(&src, n * 16); hipMalloc(&out, n * 16); hipMalloc(&idx, n * 4);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int base : {0, 8192, 20000, 30000, 38000}) {
        hipMemset(out, 0, n * 16);
        hipLaunchKernelGGL(probe, dim3(1), dim3(n), 160 * 1024, 0, src, idx, out, base);
        std::vector<float> o(n * 4);
        hipError_t e = hipMemcpy(o.data(), out, n * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) bad += o[i * 4 + k] != h[hi[i] * 4 + k];
        printf("base %6d floats (%6d B): %s, %d mismatches\n", base, base * 4, hipGetErrorString(e), bad);
    }
    return 0;
}
