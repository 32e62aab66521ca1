// nnquery.hpp — pieces shared by the two neighbour-search kernels (nnquery.hip: the chain walk over the whole cloud;
// nngrid.hip: the cell-grid search that replaces it whenever no query needs the reference's radius growth).
#pragma once
#include "common.hpp"

namespace sph3d {

// the reference's predicate on the euclidean distance s (tf_nnquery_gpu.cu:49)
__device__ __forceinline__ bool in_range(float s, float r)
{
    return s < r && (double)fabsf(s - r) > 1e-6;
}

// Wave-cooperative, all 64 lanes active, r wave-uniform.
// Returns the smallest non-negative float T with !in_range(sqrtf(T), r); then
// in_range(sqrtf(d2), r) == (d2 < T) for every d2 >= 0 (and false for NaN on both sides).
__device__ inline float range_threshold(float r)
{
    if (!in_range(0.0f, r)) return 0.0f;
    const unsigned lane = (unsigned)lane_id();
    unsigned lo = 0u;             // in_range holds at lo
    unsigned hi = 0x7f800000u;    // +inf: in_range fails
    while (hi - lo > 1u) {
        const unsigned span = hi - lo;
        const unsigned step = span / 65u + 1u;
        const unsigned long long c = (unsigned long long)lo + (unsigned long long)step * (lane + 1u);
        const bool p = (c < hi) && in_range(sqrtf(__uint_as_float((unsigned)c)), r);
        const int nt = __popcll(__ballot(p));   // p is a prefix of the lanes (monotone predicate)
        const unsigned long long nhi = (unsigned long long)lo + (unsigned long long)step * (unsigned)(nt + 1);
        if (nhi < hi) hi = (unsigned)nhi;
        lo = lo + step * (unsigned)nt;
    }
    return __uint_as_float(hi);
}

// exclusive prefix sum of v over the lanes of a wave (all lanes active)
__device__ __forceinline__ int wave_excl_scan(int v)
{
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if (lane_id() >= o) incl += u;
    }
    return incl - v;
}

// what the fused graph construction adds to the output pass of a search kernel (sph3d_build_sphere_graph): the spherical-
// kernel bin of every neighbour and, when a transposed graph will be needed, the edge's count in its (source point, bin)
// segment — the atomic's return value is the edge's position in the segment (graph.hip)
struct GraphFuse {
    int n, p, q, F;          // spherical kernel sizes, F = n*p*q + 1
    float radius;            // nominal radius of the binning (not the chain's grown radius)
    int* filt;               // [B,M,K] bin ids
    int* deg;                // [B*N*F] segment counters (zeroed by the caller) or nullptr
    int* slotPos;            // [B*M*K]
    int* binUsed;            // [F]
    int ocml;                // 1: the device library's atan2f (the reference as it builds here), 0: the shared correctly rounded one
};

// nngrid.hip.  Tries the cell-grid search: 0 = not applicable (nothing launched, *gate untouched); 1 = launched; < 0 = error
// status.  Launched, it has produced the rows of the queries at the first *grid_done positions of every chain (reference
// thread t visits queries t, t + 1024, ...: position = j / 1024; in `fixed` mode: of all queries) UNLESS the device flag *gate
// reads non-zero afterwards: a query without a neighbour inside its radius (the reference grows the radius there, and every
// later position of the chain with it), a grid too coarse to pay, non-finite coordinates.  The chain kernel runs behind it
// and reads the flag: it starts at position *grid_done, or at 0 when the flag is up.
// The grid's device memory (nngrid_workspace_bytes) is `workspace` when the caller gave one (too small: SPH3D_EWORKSPACE), else the
// library's per-(device, stream) buffer when library_scratch is set, else the grid is not used (0).
int nngrid_search(int B, int N, int M, int K, float radius, int fixed, const float* database, const float* query, int* nn_index,
                  int* nn_count, float* nn_dist, const GraphFuse* fuse, hipStream_t stream, const int** gate, int* grid_done,
                  void* workspace, size_t workspace_bytes, bool library_scratch);
size_t nngrid_workspace_bytes(int B, int N, int M);


}  // namespace sph3d
