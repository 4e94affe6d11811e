// Floor of a per-step launch on this GPU: hipGraph replay of chains of dependent kernels.
//   empty      : kernel with no memory access (pure dispatch + kernel boundary)
//   touch      : every lane loads and stores what the 4 096-env tendon-per-lane step does
//                (one dword of q/qd planes + action + a 64-byte table record), no arithmetic
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/graph_floor tools/graph_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel(float *p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 0.f; }

__global__ void touch_kernel(float *q, float *qd, unsigned *feas, const float *act, const float4 *tab, long n) {
    const long g = long(blockIdx.x) * 64 + threadIdx.x, e = g >> 3; const int k = g & 7;
    if (e >= n) return;
    const float a = act[e * 8 + k];
    const float4 t0 = tab[4 * k], t1 = tab[4 * k + 1], t2 = tab[4 * k + 2], t3 = tab[4 * k + 3];
    float s = a + t0.x + t1.y + t2.z + t3.w;
    float v[6];
    for (int j = 0; j < 3; ++j) { v[j] = q[j * n + e]; v[3 + j] = qd[j * n + e]; }
    for (int j = 0; j < 6; ++j) s += v[j];
    if (k < 3) q[k * n + e] = v[k] + s * 1e-30f;
    else if (k < 6) qd[(k - 3) * n + e] = v[k] + s * 1e-30f;
    else if (k == 6) feas[e] = 1u;
}

int main() {
    const long n = 4096; const int nodes = 100, reps = 200;
    float *q, *qd, *act; unsigned *feas; float4 *tab;
    CK(hipMalloc(&q, 12 * n)); CK(hipMalloc(&qd, 12 * n)); CK(hipMalloc(&act, 32 * n)); CK(hipMalloc(&feas, 4 * n)); CK(hipMalloc(&tab, 512));
    CK(hipMemset(q, 0, 12 * n)); CK(hipMemset(qd, 0, 12 * n)); CK(hipMemset(act, 0, 32 * n)); CK(hipMemset(tab, 0, 512));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct Case { const char *name; int kind; int blocks; } cases[] = {
        {"empty kernel, 1 workgroup of 64", 0, 1}, {"empty kernel, 512 workgroups of 64", 0, 512},
        {"load/store only (the 4 096-env step's memory accesses), 512 workgroups", 1, 512}};
    for (auto &c : cases) {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < nodes; ++i) {
            if (c.kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(c.blocks), dim3(64), 0, st, q);
            else hipLaunchKernelGGL(touch_kernel, dim3(c.blocks), dim3(64), 0, st, q, qd, feas, act, tab, n);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-75s %.3f us per graph node\n", c.name, ms * 1e3 / (double(nodes) * reps));
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    }
    return 0;
}
