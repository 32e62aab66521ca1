// scatter_reduce.hip — the two prices of the tile-wise SCATTER-REDUCE formulation of the depthwise-convolution gradient
// (per forward tile of <= 64 targets accumulate the grad_in rows of the tile's ~340-row union in LDS with
// ds_add_f32, write one partial row per (tile, row), then sum the ~6 partials per source), measured without any of the rest:
//   A. LDS float atomics: 16 waves per CU, every quarter wave adds 4 floats per lane (one 256-B row of a 64-channel slice per
//      "edge") into random rows of an 88-KB accumulator region — against the same number of ds_read_b128 of random rows
//      (what the gather formulation of the forward pays per edge: tools/micro/lds_fma.hip);
//   B. the partial rows themselves at S3DIS level 0 (16 x 8192 points, C = 128: 2273 tiles x 342 rows x 2 slices of 256 B =
//      398 MB): a streaming write of that many rows, then the second kernel: per source and slice gather its 6 partial rows
//      (random tiles) and write the sum.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/scatter_reduce.hip -o /tmp/scatter_reduce && /tmp/scatter_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: ds_add_f32 x4 per edge, 1: ds_read_b128 per edge, 2: ds_add_rtn? (not used)
__global__ __launch_bounds__(1024) void lds_kernel(int edges, float* out)
{
    extern __shared__ float acc[];                 // 352 rows x 64 floats = 88 KB
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15;
    for (int i = tid; i < 352 * 64; i += 1024) acc[i] = 0.f;
    __syncthreads();
    unsigned rng = (blockIdx.x * 1024u + (tid >> 4)) * 2654435761u + 12345u;      // one stream per quarter wave
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (int e = 0; e < edges; e++) {
        rng = rng * 1664525u + 1013904223u;
        const unsigned row = (rng >> 8) % 352u;
        float* p = acc + row * 64 + l16 * 4;
        if (MODE == 0) {
            __hip_atomic_fetch_add(p + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            sum += *reinterpret_cast<const f32x4*>(p);
        }
    }
    __syncthreads();
    float s = sum.x + sum.y + sum.z + sum.w;
    for (int i = tid; i < 352 * 64; i += 1024) s += acc[i];
    if (s == 12345.678f) out[blockIdx.x * 1024 + tid] = s;
}

__global__ __launch_bounds__(1024) void write_partials(int tiles, int rows_per_tile, float* __restrict__ part)
{
    // persistent: tile-slice t writes rows_per_tile rows of 64 floats, one quarter wave per row
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        float* base = part + (size_t)t * rows_per_tile * 64;
        for (int r = threadIdx.x >> 4; r < rows_per_tile; r += 64) {
            const f32x4 v = {(float)r, 1.f, 2.f, (float)t};
            *reinterpret_cast<f32x4*>(base + (size_t)r * 64 + (threadIdx.x & 15) * 4) = v;
        }
    }
}

template <int NP>
__global__ __launch_bounds__(256) void reduce_partials(int nsrc, const int* __restrict__ where, const float* __restrict__ part,
                                                       float* __restrict__ out)
{
    // one quarter wave per (source, slice): NP partial rows of 256 B at where[src * NP + j] (row index), summed
    const int q = (blockIdx.x * 256 + threadIdx.x) >> 4, l16 = threadIdx.x & 15;
    if (q >= nsrc) return;
    f32x4 x[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) x[j] = *reinterpret_cast<const f32x4*>(part + (size_t)where[q * NP + j] * 64 + l16 * 4);
    f32x4 s = x[0];
#pragma unroll
    for (int j = 1; j < NP; j++) s += x[j];
    *reinterpret_cast<f32x4*>(out + (size_t)q * 64 + l16 * 4) = s;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main()
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    // ---- A ----
    const int edges = 2000;      // per quarter wave; 256 CUs x 64 quarter waves x 2000 = 32.8 M "edges"
    hipFuncSetAttribute((const void*)lds_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 352 * 256);
    hipFuncSetAttribute((const void*)lds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 352 * 256);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(lds_kernel<0>, dim3(256), dim3(1024), 352 * 256, 0, edges, out);
            else hipLaunchKernelGGL(lds_kernel<1>, dim3(256), dim3(1024), 352 * 256, 0, edges, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        const double ms = time_ms(e0, e1);
        const double wave_ops = 256.0 * 16 * edges * (mode == 0 ? 4 : 1);
        printf("A %-34s %8.1f us   %.2f ns per CU and wave instruction   (%.1f M quarter-wave edges)\n",
               mode == 0 ? "4 x ds_add_f32 per edge (scatter)" : "1 x ds_read_b128 per edge (gather)", ms * 1e3,
               ms * 1e6 / (wave_ops / 256.0), 256.0 * 64 * edges / 1e6);
    }
    // ---- B ----
    const int tiles = 2273 * 2, rows = 342, nsrc = 131072 * 2, NP = 6;
    float* part; hipMalloc(&part, (size_t)tiles * rows * 64 * 4);
    float* red; hipMalloc(&red, (size_t)nsrc * 64 * 4);
    std::vector<int> h((size_t)nsrc * NP);
    srand(7);
    // a source's partials live in NEARBY tiles (spatially consecutive tiles share it): tile = base + small offset
    for (int s = 0; s < nsrc; s++) {
        const int base = (int)((long long)s * tiles / nsrc);
        for (int j = 0; j < NP; j++) {
            int t = base + (rand() % 9) - 4; t = t < 0 ? 0 : (t >= tiles ? tiles - 1 : t);
            h[(size_t)s * NP + j] = t * rows + rand() % rows;
        }
    }
    int* where; hipMalloc(&where, h.size() * 4); hipMemcpy(where, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(write_partials, dim3(512), dim3(1024), 0, 0, tiles, rows, part);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    const double wms = time_ms(e0, e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(reduce_partials<NP>, dim3((nsrc * 16 + 255) / 256), dim3(256), 0, 0, nsrc, where, part, red);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    const double rms = time_ms(e0, e1);
    const double mb = (double)tiles * rows * 256 / 1e6;
    printf("B write %.0f MB of partial rows: %.1f us (%.2f TB/s);  sum %d partials per (source, slice) -> %.0f MB: %.1f us;  together %.1f us\n",
           mb, wms * 1e3, mb / wms / 1e6, NP, (double)nsrc * 256 / 1e6, rms * 1e3, (wms + rms) * 1e3);
    printf("  (the gather gradient this would replace: 423 us isolated at this shape; the tile kernel's LDS phase comes on top)\n");
    return 0;
}
