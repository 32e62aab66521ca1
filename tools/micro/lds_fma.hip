// lds_fma.hip — what does a CU deliver when 16 waves mix ds_read_b128 of random 256-B rows with packed FMAs?
// Modes: 0 = reads only, 1 = FMAs only, 2 = both (the convlds inner loop: per step 3 reads + 2 address perms + 4 v_pk_fma_f32),
//        3 = both with 2 reads per step, 4 = both with 1 read per step, 5 = reads as ds_read_b64 (3 per step).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_fma.hip -o /tmp/lds_fma && /tmp/lds_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ T ld(unsigned a) { return *reinterpret_cast<const __attribute__((address_space(3))) T*>((size_t)a); }
template <int H>
__device__ __forceinline__ void pkfma(f32x2& acc, f32x2 x, f32x2 w)
{
    if constexpr (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(x), "v"(w));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(w));
}

template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k(int iters, float* out, long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 36 * 1024; i += 64 * NW) reinterpret_cast<float*>(lds)[i] = (float)(i & 7);
    // entries: [0, 16 KB): 4096 u32 "entries" = random slot | bin << 16
    unsigned seed = 12345u + blockIdx.x * 977u;
    for (int i = tid; i < 4096; i += 64 * NW) {
        unsigned h = (i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        reinterpret_cast<unsigned*>(lds + 144 * 1024)[i] = (h % 400u) | (((h >> 12) % 33u) << 16);
    }
    __syncthreads();
    const unsigned lb = (lane & 15) << 4;
    unsigned ra = 144 * 1024 + ((tid >> 4) & 63) * 256;          // a "record" per quarter-wave
    f32x2 acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        const u32x4 e0 = ld<u32x4>(ra + (it & 7) * 32), e1 = ld<u32x4>(ra + (it & 7) * 32 + 16);
        const unsigned ent[8] = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
        f32x4 x[8], w0[8], w1[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned arow = __builtin_amdgcn_perm(ent[u], lb, 0x0c050400u);
            const unsigned afil = __builtin_amdgcn_perm(ent[u], lb, 0x0c0c0600u);
            if (MODE == 1) {
                x[u] = f32x4{(float)arow, 1.f, 2.f, 3.f}; w0[u] = f32x4{(float)afil, 1.f, 2.f, 3.f}; w1[u] = w0[u];
            } else if (MODE == 5) {
                const f32x2 a = ld<f32x2>(arow + 33792), b = ld<f32x2>(afil), c = ld<f32x2>(afil + 8704);
                x[u] = f32x4{a[0], a[1], a[0], a[1]}; w0[u] = f32x4{b[0], b[1], b[0], b[1]}; w1[u] = f32x4{c[0], c[1], c[0], c[1]};
            } else {
                x[u] = ld<f32x4>(arow + 33792);
                w0[u] = (MODE == 4) ? x[u] : ld<f32x4>(afil);
                w1[u] = (MODE == 3 || MODE == 4) ? w0[u] : ld<f32x4>(afil + 8704);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const f32x2 x01 = {x[u][0], x[u][1]}, x23 = {x[u][2], x[u][3]};
            if (MODE == 0) {
                acc[0] += x01 + x23; acc[1] += f32x2{w0[u][0], w0[u][1]} + f32x2{w1[u][2], w1[u][3]};
            } else {
                pkfma<0>(acc[0], x01, f32x2{w0[u][0], w0[u][1]});
                pkfma<1>(acc[1], x01, f32x2{w0[u][2], w0[u][3]});
                pkfma<0>(acc[2], x23, f32x2{w1[u][0], w1[u][1]});
                pkfma<1>(acc[3], x23, f32x2{w1[u][2], w1[u][3]});
            }
        }
    }
    const long long t1 = clock64();
    float s = acc[0][0] + acc[0][1] + acc[1][0] + acc[1][1] + acc[2][0] + acc[2][1] + acc[3][0] + acc[3][1];
    if (s == 12345.678f) out[tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NW>
static void run(const char* what, float* out, long long* cyc, int wgs)
{
    const int iters = 2000, lds = 160 * 1024 / (16 / NW);
    hipFuncSetAttribute((const void*)k<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NW>), dim3(wgs), dim3(64 * NW), lds, 0, iters, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NW>), dim3(wgs), dim3(64 * NW), lds, 0, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double steps_per_cu = (double)iters * 8 * (wgs / 256.0) * NW;     // wave-steps per CU
    printf("%-44s NW=%2d: %.1f us, %.1f ns per wave-step per CU, wave clock %.0f ticks per 8-step block\n", what, NW, ms * 1e3,
           ms * 1e6 / steps_per_cu, (double)c / iters);
}

int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1 << 16);
    run<0, 16>("reads only (3 x b128 per step)", out, cyc, 256);
    run<1, 16>("FMAs only (4 pk_fma + 2 perm per step)", out, cyc, 256);
    run<2, 16>("3 reads + 4 pk_fma", out, cyc, 256);
    run<3, 16>("2 reads + 4 pk_fma", out, cyc, 256);
    run<4, 16>("1 read + 4 pk_fma", out, cyc, 256);
    run<5, 16>("3 x ds_read_b64 + 4 pk_fma", out, cyc, 256);
    run<2, 8>("3 reads + 4 pk_fma, 8 waves per CU", out, cyc, 256);
    run<0, 8>("reads only, 8 waves per CU", out, cyc, 256);
    return 0;
}
