// graph.hip — transposed neighbour graph ("in-edge lists") for the gradient kernels, gfx950.
//
// The reference's gradient kernels scatter with fp32 atomicAdd: one atomic per (point, neighbour, channel)
// (tf_conv3d_gpu.cu:51, tf_pool3d_gpu.cu:86, tf_unpool3d_gpu.cu:38,80).  On MI355X that is bound by the
// L2 atomic units (~130 G atomics/s measured in round 1: 9 ms for ONE conv gradient at S3DIS level 0).
// A scatter over the graph is a GATHER over its transpose, so every gradient kernel here walks, for each
// source point n, the list of (output point m, slot k) pairs that reference n:
//     grad_in[b,n,:] = sum over in-edges (m,k) of  grad_out[b,m,:] * (per-edge factor)
// -> no float atomics, each grad_in element written exactly once (no memset), coalesced row reads.
//
// Layout (all int32 / float32, device memory provided by the caller); F = number of bins (1 without bin_index):
//   offsets[B*(N*F+1)] in-edges of source (b,n) with bin f are entries [off[s], off[s+1]), s = b*(N*F+1) + n*F + f
//                      i.e. sorted by (cloud, source point, bin): the F segments of a source are contiguous, so a
//                      kernel can walk them in a fully unrolled loop with the bin id a compile-time constant
//                      ("gather / segment-sum"); cloud b owns the slab [b*M*K, (b+1)*M*K) of the entry arrays
//   ent_key[B*M*K]     m, the graph row (output point) of the edge
//   ent_scale[B*M*K]   1/nn_count[m]  (or weight[b,m,k] when a weight array is given)
// Built in three passes: in-degree count (integer atomics), per-cloud exclusive scan, fill (integer atomic
// cursor).  The fill order within one list is not deterministic, so float sums may differ in the last bits
// between runs — as with the reference's atomics, but without their cost.
#include "common.hpp"

namespace sph3d {

__global__ __launch_bounds__(256) void tg_count(int B, int N, int M, int K, int F, const int* __restrict__ nnIndex,
                                                const int* __restrict__ nnCount, const int* __restrict__ binIndex,
                                                int* __restrict__ deg)
{
    const long long total = (long long)B * M * K;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long row = e / K;
        const int k = (int)(e - row * K);
        if (k < nnCount[row]) {
            const int b = (int)(row / M);
            int f = binIndex ? binIndex[e] : 0;
            f = f < 0 ? 0 : (f >= F ? F - 1 : f);       // memory safety for out-of-range bin ids
            atomicAdd(&deg[((size_t)b * N + nnIndex[e]) * F + f], 1);
        }
    }
}

// one 1024-thread block per cloud: offsets = b*M*K + exclusive scan of deg; deg is cleared for reuse as cursor
__global__ __launch_bounds__(1024) void tg_scan(int N, int MK, int* __restrict__ deg, int* __restrict__ offsets)
{
    __shared__ int part[1024];
    const int b = (int)blockIdx.x;
    const int t = (int)threadIdx.x;
    int* d = deg + (size_t)b * N;
    int* off = offsets + (size_t)b * (N + 1);
    const int per = (N + 1023) / 1024;
    const int lo = t * per;
    const int hi = (lo + per) < N ? (lo + per) : N;
    int s = 0;
    for (int i = lo; i < hi; i++) s += d[i];
    part[t] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partial sums
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = b * MK + (t > 0 ? part[t - 1] : 0);
    for (int i = lo; i < hi; i++) {
        const int c = d[i];
        off[i] = run;
        run += c;
        d[i] = 0;
    }
    if (t == 1023) off[N] = b * MK + part[1023];
}

__global__ __launch_bounds__(256) void tg_fill(int B, int N, int M, int K, int F, const int* __restrict__ nnIndex,
                                               const int* __restrict__ nnCount, const int* __restrict__ binIndex,
                                               const float* __restrict__ weight, const int* __restrict__ offsets,
                                               int* __restrict__ cursor, int* __restrict__ entKey,
                                               float* __restrict__ entScale)
{
    const long long total = (long long)B * M * K;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long row = e / K;
        const int k = (int)(e - row * K);
        const int cnt = nnCount[row];
        if (k < cnt) {
            const int b = (int)(row / M);
            const int m = (int)(row - (long long)b * M);
            const int n = nnIndex[e];
            int f = binIndex ? binIndex[e] : 0;
            f = f < 0 ? 0 : (f >= F ? F - 1 : f);
            const size_t seg = (size_t)n * F + f;
            const int pos = atomicAdd(&cursor[(size_t)b * N * F + seg], 1);
            const int dst = offsets[(size_t)b * ((size_t)N * F + 1) + seg] + pos;
            entKey[dst] = m;
            entScale[dst] = weight ? weight[e] : 1.0f / (float)cnt;
        }
    }
}

}  // namespace sph3d

using namespace sph3d;

// bytes of scratch the build itself needs (the in-degree / cursor array)
extern "C" size_t sph3d_graph_transpose_workspace(int B, int N, int M, int K, int F)
{
    (void)M; (void)K;
    return sizeof(int) * (size_t)B * N * F;
}

extern "C" int sph3d_graph_transpose(int B, int N, int M, int K, int F,
                                     const int* nn_index, const int* nn_count, const int* bin_index,
                                     const float* weight, int* offsets, int* ent_key, float* ent_scale,
                                     void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0 && K > 0, "graph_transpose: bad dims B=%d N=%d M=%d K=%d", B, N, M, K);
    SPH3D_REQUIRE(F >= 1 && (bin_index != nullptr || F == 1), "graph_transpose: F=%d needs a bin_index", F);
    SPH3D_REQUIRE((long long)N * F < (1LL << 31), "graph_transpose: N*F overflows int32");
    SPH3D_REQUIRE((long long)B * M * K < (1LL << 31), "graph_transpose: B*M*K overflows int32");
    if (B == 0) return SPH3D_OK;
    const size_t need = sph3d_graph_transpose_workspace(B, N, M, K, F);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("graph_transpose: workspace %zu B < required %zu B", workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    int* deg = (int*)workspace;
    int rc = check_hip(hipMemsetAsync(deg, 0, need, st), "graph_transpose: memset");
    if (rc) return rc;
    const long long total = (long long)B * M * K;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (total > 0)
        hipLaunchKernelGGL(tg_count, dim3((unsigned)blocks), dim3(256), 0, st, B, N, M, K, F, nn_index, nn_count, bin_index, deg);
    hipLaunchKernelGGL(tg_scan, dim3(B), dim3(1024), 0, st, N * F, M * K, deg, offsets);
    if (total > 0)
        hipLaunchKernelGGL(tg_fill, dim3((unsigned)blocks), dim3(256), 0, st, B, N, M, K, F, nn_index, nn_count, bin_index,
                           weight, offsets, deg, ent_key, ent_scale);
    return check_launch("sph3d_graph_transpose");
}
