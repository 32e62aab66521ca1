// atomic_f64.hip — how long do N workgroups take to add their 2 x C per-channel partial sums into ONE set of fp64 accumulators
// (unsafe-fp-atomics: global_atomic_add_f64)?  The question behind folding the batch-norm finalize kernels into their producers:
// every workgroup of the statistics pass would end with C x 2 atomics on the same 2 C addresses.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_f64.hip -o /tmp/atomic_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void add_kernel(int C, double* __restrict__ acc, const float* __restrict__ x, int work)
{
    float s = 0.f, q = 0.f;
    for (int i = 0; i < work; i++) {                       // a little streaming work in front, like a statistics pass
        const float v = x[((size_t)blockIdx.x * work + i) * 256 + threadIdx.x];
        s += v; q += v * v;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(&acc[c], (double)s);
        atomicAdd(&acc[C + c], (double)q);
    }
}

__global__ __launch_bounds__(256) void plain_kernel(int C, float* __restrict__ partial, const float* __restrict__ x, int work)
{
    float s = 0.f, q = 0.f;
    for (int i = 0; i < work; i++) {
        const float v = x[((size_t)blockIdx.x * work + i) * 256 + threadIdx.x];
        s += v; q += v * v;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        partial[(size_t)blockIdx.x * 2 * C + c] = s;
        partial[(size_t)blockIdx.x * 2 * C + C + c] = q;
    }
}

int main()
{
    const int work = 32;
    for (int C : {128, 256, 512}) {
        for (int nwg : {256, 1024, 4096}) {
            double* acc; float* x; float* partial;
            hipMalloc(&acc, sizeof(double) * 2 * C);
            hipMalloc(&x, sizeof(float) * (size_t)nwg * work * 256);
            hipMalloc(&partial, sizeof(float) * (size_t)nwg * 2 * C);
            hipMemset(acc, 0, sizeof(double) * 2 * C);
            hipMemset(x, 0, sizeof(float) * (size_t)nwg * work * 256);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms[2];
            for (int mode = 0; mode < 2; mode++) {
                for (int w = 0; w < 3; w++) {
                    if (mode) hipLaunchKernelGGL(add_kernel, dim3(nwg), dim3(256), 0, 0, C, acc, x, work);
                    else hipLaunchKernelGGL(plain_kernel, dim3(nwg), dim3(256), 0, 0, C, partial, x, work);
                }
                hipEventRecord(e0, 0);
                for (int w = 0; w < 20; w++) {
                    if (mode) hipLaunchKernelGGL(add_kernel, dim3(nwg), dim3(256), 0, 0, C, acc, x, work);
                    else hipLaunchKernelGGL(plain_kernel, dim3(nwg), dim3(256), 0, 0, C, partial, x, work);
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[mode], e0, e1);
            }
            printf("C=%4d workgroups=%5d  partial stores %.2f us/launch   fp64 atomics %.2f us/launch\n", C, nwg, ms[0] * 50, ms[1] * 50);
            hipFree(acc); hipFree(x); hipFree(partial);
        }
    }
    return 0;
}
