// gather_bw.hip — what does the chip deliver for the conv gather's ACCESS PATTERN, with no arithmetic at all?
// Each wave reads `rows_per_wave` random rows of ROWB bytes from a table that lives in L2 (16 clouds x 8192 rows, the
// level-0 feature tensor), `inflight` wave loads issued before the first use, and adds the values into one register.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_bw.hip -o /tmp/gather_bw && /tmp/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// STRIDE = bytes between consecutive rows of the table (>= ROWB: the gather takes the first ROWB bytes of every row)
template <int ROWB, int INF, int STRIDE = ROWB>
__global__ __launch_bounds__(256) void gather(const float* __restrict__ tab, const int* __restrict__ idx, int per_wave, int nrows_cloud,
                                               float* __restrict__ out, int half = 0)
{
    constexpr int LPR = ROWB / 16;          // lanes per row (16 B per lane)
    constexpr int RPI = 64 / LPR;           // rows per wave load
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int cloud = blockIdx.x & 7;       // block b runs on XCD b % 8: one cloud pair per XCD, like the conv kernels
    const float* base = tab + (size_t)cloud * nrows_cloud * (STRIDE / 4) + (lane % LPR) * 4 + half * (ROWB / 4);
    const int* ix = idx + (size_t)wave * per_wave;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < per_wave; k += INF * RPI) {
        float4 x[INF];
#pragma unroll
        for (int u = 0; u < INF; u++) {
            const int r = ix[k + u * RPI + lane / LPR];
            x[u] = *reinterpret_cast<const float4*>(base + (size_t)r * (STRIDE / 4));
        }
#pragma unroll
        for (int u = 0; u < INF; u++) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[wave] = acc.x;
}

template <int ROWB, int INF>
static void run(const float* tab, const int* idx, float* out, int nrows_cloud, int waves, int per_wave)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((gather<ROWB, INF>), dim3(waves / 4), dim3(256), 0, 0, tab, idx, per_wave, nrows_cloud, out);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((gather<ROWB, INF>), dim3(waves / 4), dim3(256), 0, 0, tab, idx, per_wave, nrows_cloud, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)waves * per_wave * ROWB;
    printf("row %4d B, %d wave loads in flight: %.1f us  %.2f TB/s\n", ROWB, INF, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

// the 1-KB rows of the gradient's grad_out tensor taken as TWO passes of 512-B half rows (stride 1 KB): per XCD and pass the
// touched lines are 4 MB (= the L2) instead of 8 MB
static void run_halves(const float* tab, const int* idx, float* out, int nrows_cloud, int waves, int per_wave)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto both = [&]() {
        for (int h = 0; h < 2; h++)
            hipLaunchKernelGGL((gather<512, 4, 1024>), dim3(waves / 4), dim3(256), 0, 0, tab, idx, per_wave, nrows_cloud, out, h);
    };
    for (int i = 0; i < 3; i++) both();
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) both();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)waves * per_wave * 1024;
    printf("row 1024 B as two passes of 512-B halves: %.1f us  %.2f TB/s\n", ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    const int clouds = 16, nrows = 8192, waves = 131072, per_wave = 48;   // one wave per output point, 48 neighbours
    const size_t tab_floats = (size_t)clouds * nrows * 256;               // up to 1 KB rows
    float* tab; int* idx; float* out;
    hipMalloc(&tab, tab_floats * 4); hipMemset(tab, 0, tab_floats * 4);
    hipMalloc(&out, waves * 4);
    std::vector<int> h((size_t)waves * per_wave);
    srand(1);
    for (auto& v : h) v = rand() % nrows;
    hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<256, 4>(tab, idx, out, nrows, waves, per_wave);
    run<512, 4>(tab, idx, out, nrows, waves, per_wave);
    run<512, 8>(tab, idx, out, nrows, waves, per_wave);
    run<1024, 4>(tab, idx, out, nrows, waves, per_wave);
    run<1024, 8>(tab, idx, out, nrows, waves, per_wave);
    run_halves(tab, idx, out, nrows, waves, per_wave);
    return 0;
}
