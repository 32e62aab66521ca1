// tile.hip — tile plans for the LDS-tiled depthwise convolution (convtile.hip), gfx950.
//
// Why: the gather kernels of conv3d.hip fetch one feature row per edge through the CU's vector L1; the rows come from
// L2 (the cloud is L2-resident) but every edge pays a full L1 miss.  Round-1 counters: L1 83 % occupied, HBM idle,
// 10 % of the HBM roofline.  Spatially close output points share most of their neighbours (measured on the S3DIS-like
// level-0 graph: 16 Morton-consecutive points reference 190 distinct rows for 770 edges), so a workgroup that stages
// the UNION of a tile's rows in LDS once and gathers from LDS cuts the L1/L2 traffic 4x and turns every edge into one
// ds_read.  That needs, per graph (not per convolution):
//   1. sph3d_spatial_order   a processing order of the points in which consecutive points are close (Morton cells);
//   2. sph3d_rows_by_bin     the forward graph as a binned CSR: per output point its edges sorted by bin (the
//                            consumer sums the rows of one bin, then multiplies by the filter row ONCE per bin:
//                            6 edges per (point, bin) group on S3DIS-like data -> 6x fewer filter reads and FMAs);
//   3. sph3d_tile_plan       tiles of kTileP consecutive targets: the tile's distinct source rows (ulist) and, per
//                            edge, the slot of its row inside the tile.  A tile whose union exceeds the LDS capacity
//                            is split (16 -> 8 -> ... -> 1 targets).
// Everything here is integer work on the graph stream; results do not depend on the order (a tile only decides which
// rows are staged together; the summation order of a target is fixed by its CSR).
#include "common.hpp"

namespace sph3d {

// ---------------------------------------------------------------------------------------------------------------
// 1. spatial order: counting sort of a cloud's points by the Morton code of their cell in a 2^bpa-per-axis grid over
//    the bounding box (cells isotropic, sized by the longest axis).  One 1024-thread workgroup per cloud, histogram
//    in LDS (<= 32768 buckets).  Order inside a cell = arrival order of an LDS atomic (irrelevant for results).
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

// ---------------------------------------------------------------------------------------------------------------
// 2. forward graph as a binned CSR: one wave per output point; lane k holds slot k (K <= 64), a ballot per bin gives
//    the segment sizes and every edge's position.  Layout: bounds[(b*M+m)*(F+1) + f] = first entry of bin f,
//    bounds[..+F] = end; entries of point (b,m) live in [ (b*M+m)*K, +cnt ); key[] = neighbour (source row) id.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_by_bin_kernel(int rows_total, int K, int F, const int* __restrict__ nnIndex,
                                                          const int* __restrict__ nnCount, const int* __restrict__ binIndex,
                                                          int* __restrict__ bounds, int* __restrict__ key)
{
    const int row = (int)blockIdx.x * 4 + uniform((int)threadIdx.x >> 6);
    if (row >= rows_total) return;
    const int lane = lane_id();
    int cnt = uniform(nnCount[row]);
    cnt = cnt < 0 ? 0 : (cnt > K ? K : cnt);
    const bool valid = lane < cnt;
    const size_t base = (size_t)row * K;
    const int n = valid ? nnIndex[base + lane] : 0;
    int f = valid ? binIndex[base + lane] : -1;
    if (valid) f = f < 0 ? 0 : (f >= F ? F - 1 : f);
    int run = 0, mybound = 0, dest = 0;
    for (int b = 0; b < F; b++) {
        const unsigned long long mask = __ballot(f == b);
        if (lane == b) mybound = run;
        if (f == b) dest = run + prefix_popc(mask);
        run += __popcll(mask);
    }
    if (lane == F) mybound = run;
    if (lane <= F) bounds[(size_t)row * (F + 1) + lane] = (int)base + mybound;
    if (valid) key[base + dest] = n;
}

// ---------------------------------------------------------------------------------------------------------------
// 3. tile plan.  Candidate tile = kTileP consecutive positions of `order` (identity when null).  One 256-thread
//    workgroup per candidate:
//      * a target whose own edge list does not fit (more edges than `ucap` rows or than kSlotWords slot words: only
//        when a caller asks for a tiny capacity, K <= 64 <= ucap otherwise) leaves the tile and becomes a "direct" step
//        of its own; light targets come first in the tile's target list;
//      * bitmap of the source rows the light targets reference (LDS) -> union size; the light list is split in
//        halves until every sub-tile fits `ucap` rows;
//      * ranks by prefix popcounts -> row list and, per (target, bin) group, the slots of its rows: bytes, padded to whole words with the
//        zero row.
//    EVERYTHING the consumer needs for the first sub-tile of a candidate sits at an address computed from the
//    candidate index alone (so that it can be fetched two tiles ahead with no dependent load):
//      hdr[cand]            = { targets, rows } of the first light sub-tile
//      rows[cand*ucap + i]  its row list
//      tgt[cand*16 + i]     target ids, light first
//      pb[(cand*16+i)*(F+2)] word bounds of target i's bins inside its slot slab (F+1 values), then its edge count
//      slotw[(cand*16+i)*kSlotWords + j]
//    Further light sub-tiles of a split candidate and the direct steps are appended (atomically, any order) to the
//    cloud's extra-step list xsteps[b] = { cand, first target, targets, rows | -1, first row entry, 0, 0, 0 }, their
//    row lists to the pool behind the fixed part of `rows`.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ const int* target_bounds(const int* __restrict__ bounds, int b, int T, int F, int t)
{
    return bounds + ((size_t)b * T + t) * (F + 1);          // rows_by_bin layout: F+1 bounds per target
}

// marks the source rows of list entries [p0, p1) in the LDS bitmap, fills the exclusive prefix popcounts and returns the
// union size (workgroup-uniform); 256 threads
__device__ __forceinline__ int plan_build(unsigned* lbits, unsigned* lpre, int* wsum, int W, const int* __restrict__ key,
                                          const int* te0, const int* te1, int p0, int p1)
{
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < W; i += 256) lbits[i] = 0u;
    __syncthreads();
    for (int p = p0; p < p1; p++)
        for (int e = te0[p] + tid; e < te1[p]; e += 256) {
            const int n = key[e];
            atomicOr(&lbits[n >> 5], 1u << (n & 31));
        }
    __syncthreads();
    const int per = (W + 255) >> 8;
    int s = 0;
    for (int j = 0; j < per; j++) {
        const int i = tid * per + j;
        if (i < W) s += __popc(lbits[i]);
    }
    int incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += u;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int run = incl - s;
    for (int w = 0; w < (tid >> 6); w++) run += wsum[w];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    for (int j = 0; j < per; j++) {
        const int i = tid * per + j;
        if (i < W) {
            lpre[i] = (unsigned)run;
            run += __popc(lbits[i]);
        }
    }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(256) void tile_plan_kernel(int B, int T, int NS, int F, int cands, int ucap,
                                                        const int* __restrict__ order, const int* __restrict__ bounds,
                                                        const int* __restrict__ key,
                                                        int* __restrict__ hdr, int* __restrict__ tgtOut,
                                                        int* __restrict__ ulist,
                                                        int* __restrict__ pb, unsigned char* __restrict__ slot8,
                                                        int* __restrict__ xsteps, int* __restrict__ counters)
{
    extern __shared__ unsigned lbits[];           // [W] bitmap, then [W] exclusive prefix popcounts
    __shared__ int wsum[4];
    __shared__ int rawT[kTileP], rawE0[kTileP], rawE1[kTileP], rawW[kTileP];
    __shared__ int tgt[kTileP], te0[kTileP], te1[kTileP];
    __shared__ int subU[kTileP], subOff[kTileP];
    __shared__ int sh_nl, sh_alloc, sh_xbase;
    const int W = (NS + 31) >> 5;
    unsigned* lpre = lbits + W;
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();
    const int b = (int)blockIdx.x / cands, c = (int)blockIdx.x % cands;
    const size_t cand = (size_t)b * cands + c;
    const int pos0 = c * kTileP;
    const int npts = (T - pos0) < kTileP ? (T - pos0) : kTileP;
    const int hv = ucap;
    // every target of the candidate: edge range and number of slot words (one wave per target)
    for (int p = wave; p < kTileP; p += 4) {
        int e0 = 0, e1 = 0, t = 0, words = 0;
        if (p < npts) {
            t = order ? order[(size_t)b * T + pos0 + p] : pos0 + p;
            const int* o = target_bounds(bounds, b, T, F, t);
            const int ov = o[lane <= F ? lane : F];
            const int nx = __shfl_down(ov, 1);
            int pad = lane < F ? ((nx - ov + 3) >> 2) : 0;
            for (int q = 32; q > 0; q >>= 1) pad += __shfl_xor(pad, q);
            words = pad;
            e0 = __builtin_amdgcn_readfirstlane(ov);
            e1 = __builtin_amdgcn_readlane(ov, F);
        }
        if (lane == 0) {
            rawT[p] = t;
            rawE0[p] = e0;
            rawE1[p] = e1;
            rawW[p] = words;
        }
    }
    __syncthreads();
    if (tid == 0) {          // stable partition: light targets first
        int k = 0;
        for (int pass = 0; pass < 2; pass++) {
            for (int i = 0; i < npts; i++) {
                const bool h = (rawE1[i] - rawE0[i]) > hv || rawW[i] > kSlotWords;
                if (h == (pass == 1)) {
                    tgt[k] = rawT[i];
                    te0[k] = rawE0[i];
                    te1[k] = rawE1[i];
                    k++;
                }
            }
            if (pass == 0) sh_nl = k;
        }
    }
    __syncthreads();
    const int nl = sh_nl;
    if (tid < kTileP) tgtOut[cand * kTileP + tid] = tid < npts ? tgt[tid] : 0;

    // largest sub-tile size of the light list whose unions all fit
    int g = kTileP;
    for (;;) {
        bool ok = true;
        for (int s = 0; s * g < nl; s++) {
            const int p1 = ((s + 1) * g) < nl ? ((s + 1) * g) : nl;
            const int U = plan_build(lbits, lpre, wsum, W, key, te0, te1, s * g, p1);
            if (tid == 0) subU[s] = U;
            if (U > ucap) {
                ok = false;
                break;
            }
        }
        if (ok || g == 1) break;      // g == 1 always fits: a light target has at most min(heavy, ucap) edges
        g >>= 1;
    }
    __syncthreads();
    const int nsubL = (nl + g - 1) / g;
    const int nextra = (nsubL > 1 ? nsubL - 1 : 0) + (npts - nl);
    if (tid == 0) {
        int tot = 0;
        for (int s = 1; s < nsubL; s++) {
            subOff[s] = tot;
            tot += subU[s];
        }
        sh_alloc = tot ? atomicAdd(&counters[0], tot) : 0;
        sh_xbase = nextra ? atomicAdd(&counters[1 + b], nextra) : 0;
        hdr[cand * 2 + 0] = nsubL ? (g < nl ? g : nl) : 0;
        hdr[cand * 2 + 1] = nsubL ? subU[0] : 0;
    }
    __syncthreads();
    const size_t poolBase = (size_t)B * cands * ucap;
    int* xs = xsteps + ((size_t)b * cands * kTileP + sh_xbase) * 8;
    if (tid < nextra) {
        int q0, cnt, U, uoff;
        if (tid < nsubL - 1) {               // further light sub-tiles
            const int s = tid + 1;
            q0 = s * g;
            cnt = ((s + 1) * g < nl ? (s + 1) * g : nl) - q0;
            U = subU[s];
            uoff = (int)(poolBase + sh_alloc + subOff[s]);
        } else {                             // heavy targets
            q0 = nl + (tid - (nsubL > 1 ? nsubL - 1 : 0));
            cnt = 1;
            U = -1;
            uoff = 0;
        }
        int* r = xs + tid * 8;
        r[0] = c; r[1] = q0; r[2] = cnt; r[3] = U; r[4] = uoff; r[5] = 0; r[6] = 0; r[7] = 0;
    }
    for (int s = 0; s < nsubL; s++) {
        const int p0 = s * g, p1 = ((s + 1) * g) < nl ? ((s + 1) * g) : nl;
        if (nsubL > 1) plan_build(lbits, lpre, wsum, W, key, te0, te1, p0, p1);      // else: still in LDS from the search
        const size_t uoff = s == 0 ? cand * ucap : poolBase + sh_alloc + subOff[s];
        for (int i = tid; i < W; i += 256) {
            unsigned bits = lbits[i];
            int r = (int)lpre[i];
            while (bits) {
                const int bit = __builtin_ctz(bits);
                bits &= bits - 1;
                const int n = (i << 5) + bit;
                ulist[uoff + r] = n;
                r++;
            }
        }
        for (int p = p0 + wave; p < p1; p += 4) {
            const int* o = target_bounds(bounds, b, T, F, tgt[p]);
            const int ov = o[lane <= F ? lane : F];
            const int nx = __shfl_down(ov, 1);
            const int len = lane < F ? nx - ov : 0;
            const int pad = (len + 3) >> 2;
            int incl = pad;
            for (int q = 1; q < 64; q <<= 1) {
                const int u = __shfl_up(incl, q);
                if (lane >= q) incl += u;
            }
            const int dstart = incl - pad;                       // first slot word of this lane's group, inside the slab
            const int e_begin = __builtin_amdgcn_readfirstlane(ov);
            const int e_end = __builtin_amdgcn_readlane(ov, F);
            const size_t tp = cand * kTileP + p;
            if (lane <= F + 1) pb[tp * (F + 2) + lane] = lane <= F ? dstart : (e_end - e_begin);
            unsigned char* slab = slot8 + tp * (kSlotWords * 4);
            for (int eb = e_begin; eb < e_end; eb += 64) {
                const int e = eb + lane;
                const bool valid = e < e_end;
                const int n = valid ? key[e] : 0;
                int f = 0;                                   // bin of edge e: the last group whose start is <= e
                for (int j = 1; j < F; j++) {
                    const int bj = __builtin_amdgcn_readlane(ov, j);
                    f = e >= bj ? j : f;
                }
                const int gs = __shfl(ov, f);
                const int ds = __shfl(dstart, f);
                const int v = (int)lpre[n >> 5] + __popc(lbits[n >> 5] & ((1u << (n & 31)) - 1u));
                if (valid) slab[4 * ds + (e - gs)] = (unsigned char)v;
            }
            if (lane < F)
                for (int j = len; j < 4 * pad; j++) slab[4 * dstart + j] = (unsigned char)ucap;
        }
        __syncthreads();
    }
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

extern "C" int sph3d_rows_by_bin(int B, int M, int K, int F, const int* nn_index, const int* nn_count, const int* bin_index,
                                 int* bounds, int* key, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && M >= 0 && K > 0 && F > 0, "rows_by_bin: bad dims B=%d M=%d K=%d F=%d", B, M, K, F);
    SPH3D_REQUIRE(K <= 64 && F <= 63, "rows_by_bin: needs K <= 64 and F <= 63 (got K=%d F=%d)", K, F);
    SPH3D_REQUIRE((long long)B * M * K < (1LL << 31), "rows_by_bin: B*M*K overflows int32");
    const int rows = B * M;
    if (rows == 0) return SPH3D_OK;
    hipLaunchKernelGGL(rows_by_bin_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), rows, K, F, nn_index, nn_count,
                       bin_index, bounds, key);
    return check_launch("sph3d_rows_by_bin");
}

extern "C" int sph3d_tile_plan_sizes(int B, int T, int F, int ucap, long long E, int* n_cands, size_t* hdr_ints,
                                     size_t* tgt_ints, size_t* rows_ints, size_t* pb_ints, size_t* slot_words,
                                     size_t* xstep_ints, size_t* counter_ints)
{
    const int cands = (T + kTileP - 1) / kTileP;
    if (n_cands) *n_cands = cands;
    if (hdr_ints) *hdr_ints = (size_t)B * cands * 2;
    if (tgt_ints) *tgt_ints = (size_t)B * cands * kTileP;
    // fixed part (ucap per candidate) + pool for the further sub-tiles (sum of the unions <= number of edges) + read-ahead slack
    if (rows_ints) *rows_ints = (size_t)B * cands * ucap + (size_t)(E > 0 ? E : 1) + 256;
    if (pb_ints) *pb_ints = (size_t)B * cands * kTileP * (F + 2);
    if (slot_words) *slot_words = (size_t)B * cands * kTileP * kSlotWords;
    if (xstep_ints) *xstep_ints = (size_t)B * cands * kTileP * 8;
    if (counter_ints) *counter_ints = (size_t)B + 1;
    return SPH3D_OK;
}

extern "C" int sph3d_tile_plan(int B, int T, int NS, int F, int ucap,
                               const int* order, const int* bounds, const int* key,
                               int* tile_hdr, int* tile_targets, int* tile_rows, int* tile_pb,
                               int* slot_words, int* extra_steps, int* counters, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && T > 0 && NS > 0 && F > 0 && F <= 62, "tile_plan: bad dims B=%d T=%d NS=%d F=%d", B, T, NS, F);
    SPH3D_REQUIRE(ucap >= 4 && ucap <= 252 && ucap % 4 == 0, "tile_plan: ucap=%d must be a multiple of 4 in [4, 252]", ucap);
    if (B == 0) return SPH3D_OK;
    const int cands = (T + kTileP - 1) / kTileP;
    const size_t lds = sizeof(unsigned) * 2 * (size_t)((NS + 31) >> 5);
    SPH3D_REQUIRE(lds <= 150 * 1024, "tile_plan: %d source rows do not fit the LDS bitmap", NS);
    hipStream_t st = as_stream(stream);
    int rc = check_hip(hipMemsetAsync(counters, 0, ((size_t)B + 1) * sizeof(int), st), "tile_plan: memset");
    if (rc) return rc;
    if (lds > 60 * 1024) {
        rc = check_hip(hipFuncSetAttribute((const void*)tile_plan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "tile_plan: hipFuncSetAttribute");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(tile_plan_kernel, dim3(B * cands), dim3(256), lds, st, B, T, NS, F, cands, ucap, order, bounds, key,
                       tile_hdr, tile_targets, tile_rows, tile_pb, (unsigned char*)slot_words, extra_steps, counters);
    return check_launch("sph3d_tile_plan");
}
