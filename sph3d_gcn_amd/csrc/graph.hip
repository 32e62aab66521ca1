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
// Built in three steps: segment count (integer atomics), per-cloud exclusive scan, fill (integer atomic
// cursor).  The fill order within one list is not deterministic, so float sums may differ in the last bits
// between runs — as with the reference's atomics, but without their cost.
#include "common.hpp"

namespace sph3d {

__global__ __launch_bounds__(256) void zero_fill_kernel(unsigned* __restrict__ p, size_t words)
{
    // the 16-byte aligned body as uint4 stores, the (at most 3 + 3) words around it one by one
    const size_t head = ((16 - (reinterpret_cast<size_t>(p) & 15)) & 15) >> 2;
    const size_t h = head < words ? head : words;
    uint4* q = reinterpret_cast<uint4*>(p + h);
    const size_t quads = (words - h) >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = t; i < quads; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (t < h) p[t] = 0u;
    const size_t tail0 = h + (quads << 2);
    if (t < words - tail0) p[tail0 + t] = 0u;
}

int zero_async(void* p, size_t bytes, hipStream_t stream, const char* what)
{
    if (bytes == 0) return SPH3D_OK;
    if (bytes < (256u << 10) || (bytes & 3) != 0 || (reinterpret_cast<size_t>(p) & 3) != 0)
        return check_hip(hipMemsetAsync(p, 0, bytes, stream), what);
    const size_t words = bytes >> 2;
    size_t blocks = (words / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned*)p, words);
    return check_launch(what);
}

__global__ __launch_bounds__(256) void tg_count(int B, int N, int M, int K, int F, const int* __restrict__ nnIndex,
                                                const int* __restrict__ nnCount, const int* __restrict__ binIndex,
                                                int* __restrict__ deg, int* __restrict__ slotPos, int* __restrict__ binUsed)
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
            // the value the atomic returns is the edge's position inside its segment: kept, so that the fill pass needs
            // no second round of atomics (measured: count 0.25 + fill 0.32 ms -> see DESIGN.md)
            slotPos[e] = atomicAdd(&deg[((size_t)b * N + nnIndex[e]) * F + f], 1);
            if (binUsed) binUsed[f] = 1;          // benign race: every writer stores 1
        }
    }
}

// Exclusive scan of the per-cloud counter arrays (L = N*F counters per cloud) in ONE pass (round 6; three kernels before:
// chunk sums, scan of the sums, apply — and a fourth for the active-bin list): a 256-thread block per chunk of kTgChunk counters
// scans its chunk, publishes the chunk's sum in a 64-bit status word (flag << 32 | value; flag 1 = the chunk's own sum, 2 = the
// inclusive prefix up to and including the chunk) and obtains its base by looking back over its predecessors' words — a whole
// wave at a time, 64 words per trip ("decoupled look-back").  Blocks are dispatched in index order and a block only ever waits for
// blocks of lower index, so the wait is bounded by work that is already running.  Clouds are independent scans (cloud b owns
// the entry slab starting at b*M*K).  The block behind the last chunk writes the active-bin list.
constexpr int kChunk = kTgChunk;

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* lds, int& total)
{
    // 256 threads; returns the exclusive prefix of v, total = block sum
    const int t = (int)threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int a = (t >= o) ? lds[t - o] : 0;
        __syncthreads();
        lds[t] += a;
        __syncthreads();
    }
    const int incl = lds[t];
    total = lds[255];
    __syncthreads();
    return incl - v;
}

// active_bins = [count, ascending list of the bins that occur anywhere in the graph] (conv gradient, compact variant)
__device__ __forceinline__ void active_bins_block(int F, const int* __restrict__ binUsed, int* __restrict__ active, int* lds)
{
    int run = 0;
    for (int base = 0; base < F; base += 256) {
        const int f = base + (int)threadIdx.x;
        const int u = (f < F && binUsed[f] != 0) ? 1 : 0;
        int total;
        const int ex = block_exclusive_scan_256(u, lds, total);
        if (u) active[1 + run + ex] = f;
        run += total;
    }
    if (threadIdx.x == 0) active[0] = run;
}

__global__ __launch_bounds__(256) void tg_scan(int B, int L, int chunks, int MK, int F, int* __restrict__ deg,
                                               unsigned long long* __restrict__ status, int* __restrict__ offsets,
                                               const int* __restrict__ binUsed, int* __restrict__ active)
{
    __shared__ int lds[256];
    __shared__ int s_base;
    if ((int)blockIdx.x == B * chunks) {             // the extra block: list of the bins that occur
        active_bins_block(F, binUsed, active, lds);
        return;
    }
    const int b = (int)blockIdx.x / chunks, c = (int)blockIdx.x % chunks;
    int* d = deg + (size_t)b * L;
    int* off = offsets + (size_t)b * ((size_t)L + 1);
    unsigned long long* st = status + (size_t)b * chunks;
    const int lo = c * kChunk;
    constexpr int PER = kChunk / 256;
    const int t0 = lo + (int)threadIdx.x * PER;
    int v[PER];
    int s = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        v[j] = (t0 + j) < L ? d[t0 + j] : 0;
        s += v[j];
    }
    int total;
    const int ex = block_exclusive_scan_256(s, lds, total);
    if (threadIdx.x < 64) {
        // wave 0: publish, look back, publish again
        const int lane = (int)threadIdx.x;
        int base = b * MK;
        if (c > 0) {
            if (lane == 0)
                __hip_atomic_store(&st[c], (1ull << 32) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int hi = c - 1;                          // nearest predecessor not yet accounted for
            for (;;) {
                const int i = hi - lane;
                unsigned long long w = 2ull << 32;   // in front of the cloud's first chunk: inclusive prefix 0
                if (i >= 0) {
                    do {
                        w = __hip_atomic_load(&st[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((w >> 32) == 0);
                }
                const unsigned long long incl = __ballot((w >> 32) == 2);
                const int first = incl ? (int)__builtin_ctzll(incl) : 64;       // nearest predecessor with an inclusive prefix
                int part = lane <= first ? (int)(unsigned)w : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
                base += part;
                if (incl) break;
                hi -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&st[c], (2ull << 32) | (unsigned)(base - b * MK + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = base;
        }
    }
    __syncthreads();
    int run = s_base + ex;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        if ((t0 + j) < L) {
            off[t0 + j] = run;
            d[t0 + j] = 0;
        }
        run += v[j];
    }
    if (c == chunks - 1 && threadIdx.x == 255) off[L] = run;     // end of the cloud's last segment
}

// one wave per graph row (b, m), lane = neighbour slot: no per-element 64-bit divisions, the row's count and 1/count
// are wave-uniform, the id / bin / slot reads are coalesced 256-byte rows.  (fb = first fill block, nfb = fill blocks of the launch)
__device__ __forceinline__ void tg_fill_rows(int fb, int nfb, int B, int N, int M, int K, int F, const int* __restrict__ nnIndex,
                                             const int* __restrict__ nnCount, const int* __restrict__ binIndex,
                                             const float* __restrict__ weight, const int* __restrict__ offsets,
                                             const int* __restrict__ slotPos, int* __restrict__ entKey,
                                             float* __restrict__ entScale)
{
    const int lane = (int)threadIdx.x & 63;
    const int wpb = (int)blockDim.x >> 6;                   // waves per block
    const long long nrows = (long long)B * M;
    const long long wstride = (long long)nfb * wpb;
    for (long long row = (long long)((int)blockIdx.x - fb) * wpb + ((int)threadIdx.x >> 6); row < nrows; row += wstride) {
        const int cnt = nnCount[row];
        if (cnt <= 0) continue;
        const int b = (int)(row / M);                       // once per row
        const int m = (int)(row - (long long)b * M);
        const float inv = 1.0f / (float)cnt;
        const int* __restrict__ ob = offsets + (size_t)b * ((size_t)N * F + 1);
        const int lim = cnt < K ? cnt : K;
        for (int k = lane; k < lim; k += 64) {
            const long long e = row * K + k;
            const int n = nnIndex[e];
            int f = binIndex ? binIndex[e] : 0;
            f = f < 0 ? 0 : (f >= F ? F - 1 : f);
            const int dst = ob[(size_t)n * F + f] + slotPos[e];
            if (entScale == nullptr) {
                entKey[dst] = m | (cnt << 24);              // packed entry (common.hpp): one scattered store per edge
            } else {
                entKey[dst] = m;
                entScale[dst] = weight ? weight[e] : inv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// spatial order: counting sort of a cloud's points by the Morton code of their cell in a 2^bpa-per-axis grid over the
// bounding box (cells isotropic, sized by the longest axis).  One 1024-thread workgroup per cloud, histogram in LDS.
// Order inside a cell = arrival order of an LDS atomic: it only decides which targets share a tile, never a result.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned spread3(unsigned v)      // 10 bits -> every third bit
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(1024) void spatial_order_kernel(int N, int bpa, const float* __restrict__ xyz,
                                                              int* __restrict__ order)
{
    extern __shared__ int hist[];                  // [1 << 3*bpa]
    __shared__ float red[6][16];
    __shared__ int wsum[16];
    const int b = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    const int NB = 1 << (3 * bpa);
    const float* p = xyz + (size_t)b * N * 3;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int n = tid; n < N; n += 1024)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = p[n * 3 + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; a++)
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
    if ((tid & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            red[a][tid >> 6] = lo[a];
            red[3 + a][tid >> 6] = hi[a];
        }
    for (int i = tid; i < NB; i += 1024) hist[i] = 0;
    __syncthreads();
    float ext = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float l = red[a][0], h = red[3 + a][0];
        for (int w = 1; w < 16; w++) {
            l = fminf(l, red[a][w]);
            h = fmaxf(h, red[3 + a][w]);
        }
        lo[a] = l;
        ext = fmaxf(ext, h - l);
    }
    const int G = 1 << bpa;
    const float inv = ext > 0.f ? (float)G / ext : 0.f;
    auto key_of = [&](int n) {
        unsigned k = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            int q = (int)((p[n * 3 + a] - lo[a]) * inv);
            q = q < 0 ? 0 : (q > G - 1 ? G - 1 : q);
            k |= spread3((unsigned)q) << a;
        }
        return (int)k;
    };
    for (int n = tid; n < N; n += 1024) atomicAdd(&hist[key_of(n)], 1);
    __syncthreads();
    // exclusive scan of the histogram: each thread owns NB/1024 consecutive buckets (NB >= 1024 by construction)
    const int per = NB >> 10;
    int s = 0;
    for (int j = 0; j < per; j++) s += hist[tid * per + j];
    int incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
    int run = base + incl - s;
    for (int j = 0; j < per; j++) {
        const int c = hist[tid * per + j];
        hist[tid * per + j] = run;
        run += c;
    }
    __syncthreads();
    for (int n = tid; n < N; n += 1024) order[(size_t)b * N + atomicAdd(&hist[key_of(n)], 1)] = n;
}

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_spatial_order(int B, int N, const float* xyz, int* order, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0, "spatial_order: bad dims B=%d N=%d", B, N);
    if (B == 0) return SPH3D_OK;
    int bpa = 4;                                   // buckets ~ 4 N, between 2^12 and 2^15
    while (bpa < 5 && (1 << (3 * bpa)) < 4 * N) bpa++;
    const size_t lds = sizeof(int) * ((size_t)1 << (3 * bpa));
    int rc = SPH3D_OK;
    if (lds > 64 * 1024) {
        rc = check_hip(hipFuncSetAttribute((const void*)spatial_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "spatial_order: hipFuncSetAttribute");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(spatial_order_kernel, dim3(B), dim3(1024), lds, as_stream(stream), N, bpa, xyz, order);
    return check_launch("sph3d_spatial_order");
}


// Degree-balanced processing order for the convolution gradient.  Its persistent workgroups deal a cloud's source points to
// their waves position by position (wave g takes positions g, g + stride, ...), and the in-degree of the sources is
// heavy-tailed (S3DIS level 0: mean 48, sigma 45, max 537: the first-K rule favours low indices), so in index order the
// busiest wave of an XCD gets 1.4x the mean number of edges.  Inside every window of 2048 consecutive sources this kernel
// sorts the sources by in-degree — descending in even windows, ascending in odd ones — so that the positions a wave visits
// run through all the degree quantiles: level-0 gradient 0.53 -> 0.46 ms (C = 128), 0.36 -> 0.31 ms (C = 64).
// One workgroup per (window, cloud): bitonic sort of unique keys (degree, local index) in LDS — deterministic.
constexpr int kOrderWindow = 2048;
__device__ __forceinline__ void balanced_order_window(int win, int b, int N, int F, const int* __restrict__ offsets,
                                                      int* __restrict__ order, unsigned* keys)
{
    const int nt = (int)blockDim.x;
    const int base = win * kOrderWindow;
    const int cnt = (N - base) < kOrderWindow ? (N - base) : kOrderWindow;
    const int* __restrict__ ob = offsets + (size_t)b * ((size_t)N * F + 1);
    for (int i = (int)threadIdx.x; i < kOrderWindow; i += nt) {
        unsigned k = 0xffffffffu;                                  // padding sorts to the end
        if (i < cnt) {
            const size_t n = (size_t)(base + i);
            int deg = ob[(n + 1) * F] - ob[n * F];
            deg = deg < (1 << 20) ? deg : (1 << 20);
            k = ((unsigned)deg << 11) | (unsigned)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= kOrderWindow; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = (int)threadIdx.x; i < kOrderWindow; i += nt) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned a = keys[i], c = keys[p];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = (int)threadIdx.x; i < cnt; i += nt) {
        const int src = (win & 1) ? i : (cnt - 1 - i);              // even windows: heaviest first
        order[(size_t)b * N + base + i] = base + (int)(keys[src] & 2047u);
    }
}

__global__ __launch_bounds__(1024) void tg_balanced_order(int N, int F, const int* __restrict__ offsets, int* __restrict__ order)
{
    __shared__ unsigned keys[kOrderWindow];
    balanced_order_window((int)blockIdx.x, (int)blockIdx.y, N, F, offsets, order, keys);
}

// fill pass and (order != NULL) the gradient's processing order in ONE launch: the first `nob` blocks sort one window each (the
// longer job: they start first), the rest fill the entry arrays; both only read the scan's offsets
__global__ __launch_bounds__(1024) void tg_fill_order(int nob, int B, int N, int M, int K, int F, const int* __restrict__ nnIndex,
                                                      const int* __restrict__ nnCount, const int* __restrict__ binIndex,
                                                      const float* __restrict__ weight, const int* __restrict__ offsets,
                                                      const int* __restrict__ slotPos, int* __restrict__ entKey,
                                                      float* __restrict__ entScale, int* __restrict__ order)
{
    __shared__ unsigned keys[kOrderWindow];
    if ((int)blockIdx.x < nob) {
        const int windows = (N + kOrderWindow - 1) / kOrderWindow;
        balanced_order_window((int)blockIdx.x % windows, (int)blockIdx.x / windows, N, F, offsets, order, keys);
        return;
    }
    tg_fill_rows(nob, (int)gridDim.x - nob, B, N, M, K, F, nnIndex, nnCount, binIndex, weight, offsets, slotPos, entKey, entScale);
}

// bytes of scratch the build itself needs (common.hpp: tg_ws)
extern "C" size_t sph3d_graph_transpose_workspace(int B, int N, int M, int K, int F)
{
    return sizeof(int) * tg_ws(nullptr, B, N, M, K, F).total_words;
}

static int tg_dims_ok(int B, int N, int M, int K, int F, const void* bin_index, const void* workspace, size_t workspace_bytes)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0 && K > 0, "graph_transpose: bad dims B=%d N=%d M=%d K=%d", B, N, M, K);
    SPH3D_REQUIRE(F >= 1 && (bin_index != nullptr || F == 1), "graph_transpose: F=%d needs a bin_index", F);
    SPH3D_REQUIRE((long long)N * F < (1LL << 31), "graph_transpose: N*F overflows int32");
    SPH3D_REQUIRE((long long)B * M * K < (1LL << 31), "graph_transpose: B*M*K overflows int32");
    const size_t need = sph3d_graph_transpose_workspace(B, N, M, K, F);
    if (B > 0 && (workspace == nullptr || workspace_bytes < need)) {
        set_error("graph_transpose: workspace %zu B < required %zu B", workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    return SPH3D_OK;
}

// phase 1: segment counts (+ the position of every edge in its segment, + which bins occur).  sph3d_build_sphere_graph
// produces the same three arrays from inside the neighbour search.
extern "C" int sph3d_graph_transpose_count(int B, int N, int M, int K, int F, const int* nn_index, const int* nn_count,
                                           const int* bin_index, int want_active, void* workspace, size_t workspace_bytes,
                                           sph3d_stream_t stream)
{
    int rc = tg_dims_ok(B, N, M, K, F, bin_index, workspace, workspace_bytes);
    if (rc || B == 0) return rc;
    hipStream_t st = as_stream(stream);
    const TgWs w = tg_ws(workspace, B, N, M, K, F);
    rc = zero_async(w.deg, sizeof(int) * w.zero_words, st, "graph_transpose: memset");       // counters, bin flags, scan status words
    if (rc) return rc;
    const long long total = (long long)B * M * K;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (total > 0)
        hipLaunchKernelGGL(tg_count, dim3((unsigned)blocks), dim3(256), 0, st, B, N, M, K, F, nn_index, nn_count, bin_index, w.deg,
                           w.slot_pos, want_active ? w.bin_used : nullptr);
    return check_launch("sph3d_graph_transpose_count");
}

// phase 2: scan of the counts -> offsets (+ the active-bin list), then the fill pass (no atomics: every edge knows its
// position) together with, when `order` is given, the gradient's degree-balanced processing order: two launches
static int tg_finish(int B, int N, int M, int K, int F, const int* nn_index, const int* nn_count, const int* bin_index,
                     const float* weight, int* offsets, int* ent_key, float* ent_scale, int* active_bins, int* order,
                     void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    int rc = tg_dims_ok(B, N, M, K, F, bin_index, workspace, workspace_bytes);
    if (rc || B == 0) return rc;
    SPH3D_REQUIRE(ent_scale != nullptr || tg_packable(M, K, weight),
                  "graph_transpose: ent_scale == NULL (packed entries) needs an un-weighted graph with M <= 2^24 and K <= 255 (M=%d K=%d)", M, K);
    hipStream_t st = as_stream(stream);
    const TgWs w = tg_ws(workspace, B, N, M, K, F);
    const long long total = (long long)B * M * K;
    hipLaunchKernelGGL(tg_scan, dim3(B * w.chunks + (active_bins ? 1 : 0)), dim3(256), 0, st, B, w.L, w.chunks, M * K, F, w.deg,
                       w.status, offsets, w.bin_used, active_bins);
    const int nob = order ? B * ((N + kOrderWindow - 1) / kOrderWindow) : 0;
    long long fb = total > 0 ? ((long long)B * M + 15) / 16 : 0;            // one wave per graph row, 16 per block
    if (fb > 16384) fb = 16384;
    if (nob + fb > 0)
        hipLaunchKernelGGL(tg_fill_order, dim3((unsigned)(nob + fb)), dim3(1024), 0, st, nob, B, N, M, K, F, nn_index, nn_count,
                           bin_index, weight, offsets, w.slot_pos, ent_key, ent_scale, order);
    return check_launch("sph3d_graph_transpose");
}

extern "C" int sph3d_graph_transpose_finish(int B, int N, int M, int K, int F,
                                            const int* nn_index, const int* nn_count, const int* bin_index,
                                            const float* weight, int* offsets, int* ent_key, float* ent_scale, int* active_bins,
                                            void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    return tg_finish(B, N, M, K, F, nn_index, nn_count, bin_index, weight, offsets, ent_key, ent_scale, active_bins, nullptr, workspace,
                     workspace_bytes, stream);
}

extern "C" int sph3d_graph_transpose_finish_ordered(int B, int N, int M, int K, int F,
                                                    const int* nn_index, const int* nn_count, const int* bin_index,
                                                    const float* weight, int* offsets, int* ent_key, float* ent_scale,
                                                    int* active_bins, int* order, void* workspace, size_t workspace_bytes,
                                                    sph3d_stream_t stream)
{
    return tg_finish(B, N, M, K, F, nn_index, nn_count, bin_index, weight, offsets, ent_key, ent_scale, active_bins, order, workspace,
                     workspace_bytes, stream);
}

extern "C" int sph3d_graph_balanced_order(int B, int N, int F, const int* offsets, int* order, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && F > 0, "graph_balanced_order: bad dims B=%d N=%d F=%d", B, N, F);
    if (B == 0) return SPH3D_OK;
    hipLaunchKernelGGL(tg_balanced_order, dim3((N + kOrderWindow - 1) / kOrderWindow, B), dim3(1024), 0, as_stream(stream), N, F,
                       offsets, order);
    return check_launch("sph3d_graph_balanced_order");
}

// tf.gather_nd for (cloud, point) index pairs: out[b, s, :] = params[pair.b, pair.p, :] for rows of `row` 4-byte elements
// (coordinates, neighbour lists, counts: models/SPH3D_s3dis.py:68-72 gathers the sampled points' rows this way).
// One launch instead of two int32 -> int64 conversions plus an advanced-indexing kernel per tensor.
__global__ __launch_bounds__(256) void gather_nd_rows(int B, int N, long long total, int row, const int* __restrict__ pairs,
                                                      const unsigned* __restrict__ params, unsigned* __restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long s = e / row;
        const int c = (int)(e - s * row);
        int b = pairs[s * 2], p = pairs[s * 2 + 1];
        b = b < 0 ? 0 : (b >= B ? B - 1 : b);          // out-of-range pairs are clamped, never read outside params
        p = p < 0 ? 0 : (p >= N ? N - 1 : p);
        out[e] = params[((size_t)b * N + p) * row + c];
    }
}

extern "C" int sph3d_gather_nd(int B, int N, long long S, int row, const int* pairs, const void* params, void* out,
                               sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B > 0 && N > 0 && S >= 0 && row > 0, "gather_nd: bad dims B=%d N=%d S=%lld row=%d", B, N, S, row);
    const long long total = S * row;
    if (total == 0) return SPH3D_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gather_nd_rows, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), B, N, total, row, pairs,
                       (const unsigned*)params, (unsigned*)out);
    return check_launch("sph3d_gather_nd");
}

// The pooling graph of a level is the rows of its intra-level graph at the sampled points (models/SPH3D_s3dis.py:68-72: two
// tf.gather_nd, one of nn_index and one of nn_count); its transposed graph serves the max-pool gradient.  One kernel: a wave per
// sampled row copies the row and its count and counts the in-edges of every neighbour it lists (the first pass of
// sph3d_graph_transpose, F = 1: same counters / slot positions in the same workspace layout) — three launches and a second read of
// the gathered rows fewer per pooling graph.
__global__ __launch_bounds__(256) void gather_rows_count_kernel(int B, int N, int S, int K, const int* __restrict__ pairs,
                                                                const int* __restrict__ nnIndex, const int* __restrict__ nnCount,
                                                                int* __restrict__ outIndex, int* __restrict__ outCount,
                                                                int* __restrict__ deg, int* __restrict__ slotPos)
{
    const int lane = (int)threadIdx.x & 63;
    const long long rows = (long long)B * S;
    for (long long s = (long long)blockIdx.x * 4 + ((int)threadIdx.x >> 6); s < rows; s += (long long)gridDim.x * 4) {
        int b = pairs[s * 2], p = pairs[s * 2 + 1];
        b = b < 0 ? 0 : (b >= B ? B - 1 : b);              // out-of-range pairs are clamped (as sph3d_gather_nd)
        p = p < 0 ? 0 : (p >= N ? N - 1 : p);
        const int ob = (int)(s / S);                        // the cloud the OUTPUT row belongs to (its transposed graph's cloud)
        const long long src = ((long long)b * N + p);
        const int cnt = nnCount[src];
        if (lane == 0) outCount[s] = cnt;
        for (int k = lane; k < K; k += 64) {
            const int id = nnIndex[src * K + k];
            outIndex[s * K + k] = id;
            if (deg != nullptr && k < cnt) slotPos[s * K + k] = atomicAdd(&deg[(size_t)ob * N + (id < 0 ? 0 : (id >= N ? N - 1 : id))], 1);
        }
    }
}

extern "C" int sph3d_gather_rows_count(int B, int N, int S, int K, const int* pairs, const int* nn_index, const int* nn_count,
                                       int* out_index, int* out_count, void* transpose_workspace, size_t transpose_workspace_bytes,
                                       sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B > 0 && N > 0 && S >= 0 && K > 0, "gather_rows_count: bad dims B=%d N=%d S=%d K=%d", B, N, S, K);
    if (S == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    int* deg = nullptr;
    int* slot = nullptr;
    if (transpose_workspace != nullptr) {
        int rc = tg_dims_ok(B, N, S, K, 1, nullptr, transpose_workspace, transpose_workspace_bytes);
        if (rc) return rc;
        const TgWs w = tg_ws(transpose_workspace, B, N, S, K, 1);
        rc = zero_async(w.deg, sizeof(int) * w.zero_words, st, "gather_rows_count: memset");
        if (rc) return rc;
        deg = w.deg;
        slot = w.slot_pos;
    }
    long long blocks = ((long long)B * S + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gather_rows_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, B, N, S, K, pairs, nn_index, nn_count, out_index,
                       out_count, deg, slot);
    return check_launch("sph3d_gather_rows_count");
}

extern "C" int sph3d_graph_transpose(int B, int N, int M, int K, int F,
                                     const int* nn_index, const int* nn_count, const int* bin_index,
                                     const float* weight, int* offsets, int* ent_key, float* ent_scale, int* active_bins,
                                     void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    int rc = sph3d_graph_transpose_count(B, N, M, K, F, nn_index, nn_count, bin_index, active_bins != nullptr, workspace,
                                         workspace_bytes, stream);
    if (rc) return rc;
    return sph3d_graph_transpose_finish(B, N, M, K, F, nn_index, nn_count, bin_index, weight, offsets, ent_key, ent_scale,
                                        active_bins, workspace, workspace_bytes, stream);
}
