// nngrid.hip — the range search over a cell grid, gfx950.
//
// cal_nn_binidx (tf_ops/nnquery/tf_nnquery_gpu.cu:15-65) tests every query against the whole cloud, and its per-thread
// `radius += 0.05` after EVERY pass (:59) makes the radius of a query depend on its position in the thread's chain of queries
// (thread t visits j = t, t + 1024, ...): position k searches with r_k = r_0 + 0.05 k — if no earlier query of the chain
// needed a second pass.  nnquery.hip walks the chains wave-cooperatively over the whole cloud, 64 points per step, and stops a
// query at nn_sample hits.  That early stop makes the LATE positions cheap (r = 0.45 at S3DIS level 0: a quarter of the cloud
// is in range, 64 hits after ~250 points) and leaves the EARLY ones as full scans of 8192 points for ~14-55 hits: positions
// 0-2 are 58 % of the kernel's work there.  Those queries are what a spatial grid is for:
//
//   nngrid_build_kernel   one workgroup per cloud: bounding box, cell size h >= 1.001 * r_0 (coarser if the box would
//                         need too many cells), counting sort of the points by cell in LDS -> cell_start[], points as
//                         (x, y, z, index) in cell order; the visiting order of the queries at positions <= kg
//                         (r_kg <= 2 r_0), sorted by (position, cell): neighbouring waves read the same cells.
//   nngrid_search_kernel  a QUARTER WAVE per query: its 16 lanes walk the (2c+1)^2 z-columns of the query's neighbourhood,
//                         c = ceil(r_k / h) <= 2 (cells are z-fastest, so a column's 2c+1 cells are ONE contiguous run of
//                         the sorted points), 32 candidates per trip, test  (dx*dx + dy*dy) + dz*dz < T(r_k)  with the
//                         reference's roundings and T the exact threshold of its predicate (nnquery.hpp), and set bit
//                         `index` of the query's bitmap in LDS for a hit.  The bitmap read in ascending order IS the
//                         reference's scan order: its first nn_sample set bits are the row; distance, spherical-kernel bin
//                         and the transposed graph's segment count are produced from them as in the chain kernel.
//
// Later positions (nndense_kernel): their spheres hold hundreds of points, so the reference's ascending scan meets nn_sample
// hits after a fraction of the cloud — a wave per query walks the cloud in index order, 256 points per trip, and stops there
// (measured: a grid reach of 4-5 r_0 costs as much as that partial scan).  With the position's radius tabulated, those queries
// are independent too: tens of thousands of waves where the chain kernel walks 1024 chains per cloud one query after the other.
//
// The rows are the chain kernel's, bit for bit, unless some query has no neighbour inside its radius (the reference then grows
// the radius, for this query and for the rest of its chain): the kernels raise a device flag, and the chain kernel — launched
// behind them, returning at once otherwise — computes the call from position 0.  The same flag is raised for clouds the grid
// cannot help (fewer than 512 cells: radius comparable to the extent) or cannot index (non-finite coordinates).
//
// Measured (MI355X, S3DIS level 0, 16 x 8192 points, r_0 = 0.1, K = 64, fused with bins and segment counts;
// tools/exp_nngrid.py): chain kernel alone 472 us; grid build 16 + query order + search of positions 0-4 147 + dense scan of
// positions 5-7 155 = 335 us (both kernels VALU-bound; counters of the fused call: search 60.7 M VALU + 26.6 M SALU over
// 20 480 waves, dense scan 69.3 M + 38.1 M over 12 288 waves of four queries, about half of it the output pass: two atan2f and
// the reference's double-precision bin arithmetic per neighbour).  Level 1 (16 x 2048, two
// positions, all from the grid): 84 -> 60 us.  In the training step (graph stream, beside the feature kernels) the fused graph
// construction drops from 1.69 to 1.23 ms per step: 1767 -> 1788 blocks/s.
#include <atomic>
#include <cstdlib>
#include "common.hpp"
#include "sphere_bin.hpp"
#include "nnquery.hpp"

namespace sph3d {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kGridMaxCells = 8192;      // LDS histogram of the build kernels (32 KB: they must fit beside the step's other kernels)
constexpr int kGridMinCells = 512;       // below this 27 cells are too large a share of the cloud
constexpr int kGridMaxK = 256;           // sorted hit lists of a wave's four queries: 4 * K * 2 B
constexpr int kGridMaxPos = 8;           // chain positions the grid may take (r_k <= kGridReach * r_0 bounds it further)
constexpr int kGridReach = 3;            // ... in units of r_0 = cells a query looks in every direction: (2 * 3 + 1)^2 = 49 columns

struct GridHdr {
    float minx, miny, minz, invh;
    int nx, ny, nz, nq;                  // nx == 0: cloud not indexed (the flag is raised); nq: queries in the visiting order
};
constexpr int kMaxPositions = 64;        // positions of a chain at M <= 65536 queries
struct GridRadii {
    float thr[kMaxPositions], rk[kMaxPositions];      // exact threshold T(r_k) and radius of chain position k
};

// cell coordinate along one axis; NaN / huge values land in a valid cell (the distance test rejects them)
__device__ __forceinline__ int cell_of(float v, float lo, float invh, int n)
{
    const float t = floorf((v - lo) * invh);
    return t >= 0.0f ? (t < (float)n ? (int)t : n - 1) : 0;       // false for NaN -> 0
}

// exclusive scan of one int per thread over a 1024-thread workgroup; `tmp` = 16 ints of LDS; all threads call
__device__ __forceinline__ int block_excl_scan_1024(int v, int* tmp, int& total)
{
    const int lane = lane_id(), w = (int)threadIdx.x >> 6;
    const int ex = wave_excl_scan(v);
    __syncthreads();
    if (lane == 63) tmp[w] = ex + v;
    __syncthreads();
    int base = 0, tot = 0;
    for (int k = 0; k < 16; k++) {
        const int t = tmp[k];
        if (k < w) base += t;
        tot += t;
    }
    total = tot;
    return base + ex;
}

// counting sort of `n` items by cell into LDS-resident offsets.  hist: ncell ints (LDS).  cellStartOut (may be null):
// [ncell + 1] global.  emit(item, position)
template <class CellOf, class Emit>
__device__ __forceinline__ void sort_by_cell(int n, int ncell, int* hist, int* tmp, int* cellStartOut, CellOf cellOf, Emit emit)
{
    const int tid = (int)threadIdx.x;
    for (int c = tid; c < ncell; c += 1024) hist[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[cellOf(i)], 1);
    __syncthreads();
    const int per = (ncell + 1023) / 1024;
    const int c0 = tid * per;
    int local = 0;
    for (int k = 0; k < per; k++)
        if (c0 + k < ncell) local += hist[c0 + k];
    int total;
    int run = block_excl_scan_1024(local, tmp, total);
    for (int k = 0; k < per; k++) {
        if (c0 + k < ncell) {
            const int c = hist[c0 + k];
            hist[c0 + k] = run;
            if (cellStartOut != nullptr) cellStartOut[c0 + k] = run;
            run += c;
        }
    }
    if (cellStartOut != nullptr && tid == 0) cellStartOut[ncell] = n;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) emit(i, atomicAdd(&hist[cellOf(i)], 1));
    __syncthreads();
}

// npos: chain positions the grid takes (queries j < npos * 1024; `fixed` mode: every query is position 0, npos = 1 and all M)
__global__ __launch_bounds__(1024) void nngrid_build_kernel(
    int N, int M, int npos, int fixed, float radius, const float* __restrict__ database,
    int* __restrict__ flag, GridRadii* __restrict__ radii, GridHdr* __restrict__ hdr, int* __restrict__ cellStart,
    float4* __restrict__ pts)
{
    extern __shared__ int hist[];                 // [kGridMaxCells]
    __shared__ float redf[6][16];
    __shared__ int tmp[16];
    __shared__ GridHdr H;
    __shared__ int bad;
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6;
    const float* db = database + (size_t)b * N * 3;
    if (tid == 0) bad = 0;
    // ---- bounding box of the cloud ----
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool nonfinite = false;
    for (int i = tid; i < N; i += 1024) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = db[(size_t)i * 3 + a];
            nonfinite = nonfinite || !(fabsf(v) < INFINITY);
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
        if (lane == 0) { redf[a][w] = lo[a]; redf[3 + a][w] = hi[a]; }
    }
    __syncthreads();
    if (nonfinite) bad = 1;
    __syncthreads();
    // the radius of position k and the exact threshold of its predicate: wave k computes T(r_k) (wave-cooperative search)
    if (b == 0) {
        // positions in use (<= kMaxPositions): the clouds b, b + 32, ... share a reference block, whose chains go on from cloud
        // to cloud (position = (b / 32) * slices per cloud + j / 1024)
        const int have = fixed ? 1 : (((int)gridDim.x + kRefGrid - 1) / kRefGrid) * ((M + kRefBlock - 1) / kRefBlock);
        for (int k0 = w; k0 < have; k0 += 16) {
            float rk = radius;
            for (int k = 0; k < k0; k++) rk = (float)((double)rk + 0.05);  // tf_nnquery_gpu.cu:59, the chain kernel's sequence
            const float T = range_threshold(rk);
            if (lane == 0) { radii->thr[k0] = T; radii->rk[k0] = rk; }
        }
    }
    if (tid == 0) {
        float l3[3], h3[3];
        for (int a = 0; a < 3; a++) {
            l3[a] = redf[a][0]; h3[a] = redf[3 + a][0];
            for (int k = 1; k < 16; k++) { l3[a] = fminf(l3[a], redf[a][k]); h3[a] = fmaxf(h3[a], redf[3 + a][k]); }
        }
        // cell size: two points closer than r_0 must sit in the same or in adjacent cells (a query of radius r looks
        // ceil(r / h) cells far).  h = 1.001 * r_0 leaves a margin of 1e-3 cells, three orders of magnitude above the rounding
        // of (v - lo) * invh for grids of this size
        const int maxCells = kGridMaxCells;
        float h = radius * 1.001f;
        int n3[3] = {0, 0, 0};
        bool ok = !bad && h > 0.0f && h < INFINITY;
        for (int it = 0; ok && it < 64; it++) {
            long long cells = 1;
            for (int a = 0; a < 3; a++) {
                const float e = (h3[a] - l3[a]) / h;
                if (!(e < 1.0e6f)) { cells = 1LL << 60; break; }
                n3[a] = (int)e + 1;
                cells *= n3[a];
            }
            if (cells <= maxCells) break;
            h *= cells > 8LL * maxCells ? 2.0f : 1.26f;
            if (it == 63) ok = false;
        }
        if (ok && (long long)n3[0] * n3[1] * n3[2] < kGridMinCells) ok = false;
        H.minx = l3[0]; H.miny = l3[1]; H.minz = l3[2];
        H.invh = 1.0f / h;
        H.nx = ok ? n3[0] : 0; H.ny = n3[1]; H.nz = n3[2];
        {
            // the queries of this cloud the grid takes: its slices whose chain position is below npos
            const int S = (M + kRefBlock - 1) / kRefBlock;
            int sl = npos - (b / kRefGrid) * S;
            sl = sl < 0 ? 0 : (sl > S ? S : sl);
            H.nq = fixed ? M : (M < sl * kRefBlock ? M : sl * kRefBlock);
        }
        hdr[b] = H;
        if (!ok) *flag = 1;
    }
    __syncthreads();
    if (H.nx == 0) return;
    const int nx = H.nx, ny = H.ny, nz = H.nz, ncell = nx * ny * nz;
    const float mx = H.minx, my = H.miny, mz = H.minz, invh = H.invh;
    auto cell3 = [&](const float* p) {
        return (cell_of(p[0], mx, invh, nx) * ny + cell_of(p[1], my, invh, ny)) * nz + cell_of(p[2], mz, invh, nz);
    };
    float4* P = pts + (size_t)b * N;
    sort_by_cell(N, ncell, hist, tmp, cellStart + (size_t)b * (kGridMaxCells + 1),
                 [&](int i) { return cell3(db + (size_t)i * 3); },
                 [&](int i, int pos) {
                     P[pos] = make_float4(db[(size_t)i * 3], db[(size_t)i * 3 + 1], db[(size_t)i * 3 + 2], __int_as_float(i));
                 });
}

// The visiting order of the queries the grid takes: by (position, cell) — query j sits at position j / 1024 of its chain, and a
// wave's four queries should share radius and cells.  One workgroup per (cloud, position) sorts that position's <= 1024
// queries by cell (`fixed` mode: consecutive slices of 1024 queries, all at the nominal radius).
__global__ __launch_bounds__(1024) void nngrid_qorder_kernel(int M, int nslices, const float* __restrict__ query,
                                                             const GridHdr* __restrict__ hdr, int* __restrict__ qorder)
{
    extern __shared__ int hist[];                 // [kGridMaxCells]
    __shared__ int tmp[16];
    const int b = (int)blockIdx.x / nslices, k = (int)blockIdx.x % nslices;
    const GridHdr H = hdr[b];
    const int j0 = k * kRefBlock;
    if (H.nx == 0 || j0 >= H.nq) return;
    const int n = H.nq - j0 < kRefBlock ? H.nq - j0 : kRefBlock;
    const int nx = H.nx, ny = H.ny, nz = H.nz;
    const float mx = H.minx, my = H.miny, mz = H.minz, invh = H.invh;
    const float* q = query + ((size_t)b * M + j0) * 3;
    int* qo = qorder + (size_t)b * M + j0;
    sort_by_cell(n, nx * ny * nz, hist, tmp, nullptr,
                 [&](int j) {
                     const float* p = q + (size_t)j * 3;
                     return (cell_of(p[0], mx, invh, nx) * ny + cell_of(p[1], my, invh, ny)) * nz + cell_of(p[2], mz, invh, nz);
                 },
                 [&](int j, int pos) { qo[pos] = j0 + j; });
}

// exclusive prefix sum over the 16 lanes of a quarter wave (all lanes active)
__device__ __forceinline__ int quarter_excl_scan(int v)
{
    const int l16 = lane_id() & 15;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const int u = __shfl_up(incl, o, 16);
        if (l16 >= o) incl += u;
    }
    return incl - v;
}

// A wave per four queries (consecutive in the visiting order), a QUARTER WAVE per query: its 16 lanes take 16 candidates of a
// column run per step (one 256-B load), and a hit sets bit `index` of the query's bitmap in LDS.  The bitmap is the whole
// bookkeeping: no slots, no overflow, and reading its bits in ascending order yields the hits in the reference's scan order —
// the first nn_sample set bits are the row.  W: 32-bit words of a bitmap (>= N / 32, a multiple of 16).
template <bool FUSE>
__global__ __launch_bounds__(64) void nngrid_search_kernel(
    int B, int N, int M, int K, int W, int parts, int fixed, GraphFuse fx, int* __restrict__ flag,
    const GridRadii* __restrict__ radii, const GridHdr* __restrict__ hdr, const int* __restrict__ cellStart,
    const float4* __restrict__ pts, const int* __restrict__ qorder, const float* __restrict__ database,
    const float* __restrict__ query, int* __restrict__ nnIndex, int* __restrict__ nnCount, float* __restrict__ nnDist)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int b, part;
    xcd_decode((int)blockIdx.x, B, parts, b, part);
    if (b < 0) return;
    const GridHdr H = hdr[b];
    if (H.nx == 0 || part * 4 >= H.nq) return;         // uniform per workgroup
    const int lane = lane_id(), g = lane >> 4, l16 = lane & 15;
    // LDS: bitmaps u32 [4][W]; column bounds int2 [4][64]; sorted hits u16 [4][K]
    unsigned* bm = reinterpret_cast<unsigned*>(smem) + (size_t)g * W;
    int2* bounds = reinterpret_cast<int2*>(smem + (size_t)4 * W * 4) + g * 64;
    unsigned short* list = reinterpret_cast<unsigned short*>(smem + (size_t)4 * W * 4 + 4 * 64 * 8) + (size_t)g * K;

    const int qpos = part * 4 + g;                         // the quarter's position in the visiting order
    const bool valid = qpos < H.nq;
    const float4* P = pts + (size_t)b * N;
    const int* cs = cellStart + (size_t)b * (kGridMaxCells + 1);
    const float* db = database + (size_t)b * N * 3;
    int qid = 0, kq = 0;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (valid) {
        qid = qorder[(size_t)b * M + qpos];
        const float* q = query + ((size_t)b * M + qid) * 3;
        qx = q[0]; qy = q[1]; qz = q[2];
        kq = fixed ? 0 : (b / kRefGrid) * ((M + kRefBlock - 1) / kRefBlock) + qid / kRefBlock;
    }
    const int nx = H.nx, ny = H.ny, nz = H.nz;
    const int cx = cell_of(qx, H.minx, H.invh, nx), cy = cell_of(qy, H.miny, H.invh, ny), cz = cell_of(qz, H.minz, H.invh, nz);
    const float T = radii->thr[kq];
    // cells to look at in every direction: ceil(r / h), one more than the floor to be on the safe side of the rounding
    // (r_k <= 3 r_0 < 3 h: at most 3)
    int c = (int)(radii->rk[kq] * H.invh) + 1;
    c = c > kGridReach ? kGridReach : c;
    const int side = 2 * c + 1;
    const int ncols = valid ? side * side : 0;
    const int zlo = cz - c > 0 ? cz - c : 0, zhi = cz + c < nz - 1 ? cz + c : nz - 1;

    for (int i = l16; i < W; i += 16) bm[i] = 0u;
    // the runs of all columns first (their loads in flight together), then the walk over them
    for (int ci = l16; ci < ncols; ci += 16) {
        const int X = cx + ci / side - c, Y = cy + ci % side - c;
        int2 be = make_int2(0, 0);
        if (X >= 0 && X < nx && Y >= 0 && Y < ny) {
            const int cb = (X * ny + Y) * nz;
            be.x = cs[cb + zlo];
            be.y = cs[cb + zhi + 1];
        }
        bounds[ci] = be;
    }
    __syncthreads();
    // ---- scan: the runs of the columns, one after the other ----
    int col = 0, p = 0, e = 0;
    if (ncols > 0) { const int2 be = bounds[0]; p = be.x; e = be.y; }
    for (;;) {
        while (p >= e && col + 1 < ncols) {
            col++;
            const int2 be = bounds[col];
            p = be.x; e = be.y;
        }
        if (__builtin_amdgcn_ballot_w64(p < e) == 0ull) break;
        // 64 candidates of the run per trip: four 256-B loads per quarter in flight
        float4 cd[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = p + u * 16 + l16;
            in[u] = i < e;
            cd[u] = P[in[u] ? i : 0];
        }
        p += 64;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float dx = cd[u].x - qx, dy = cd[u].y - qy, dz = cd[u].z - qz;
            const float d2 = (dx * dx + dy * dy) + dz * dz;        // tf_nnquery_gpu.cu:45-46
            if (in[u] && d2 < T) {
                const unsigned id = (unsigned)__float_as_int(cd[u].w);
                atomicOr(&bm[id >> 5], 1u << (id & 31u));
            }
        }
    }
    __syncthreads();
    // ---- the set bits in ascending order = the hits in the reference's scan order; the first K are the row ----
    const int wpl = W / 16;                                 // words per lane
    int mine = 0;
    for (int i = 0; i < wpl; i++) mine += __popc(bm[l16 * wpl + i]);
    int rank = quarter_excl_scan(mine);
    const int found = __shfl(rank + mine, 15, 16);          // hits of the query
    for (int i = 0; i < wpl && rank < K; i++) {
        unsigned word = bm[l16 * wpl + i];
        while (word != 0u && rank < K) {
            const int bit = __builtin_ctz(word);
            word &= word - 1u;
            list[rank++] = (unsigned short)((l16 * wpl + i) * 32 + bit);
        }
    }
    __syncthreads();
    const int cnt = found < K ? found : K;
    if (valid) {
        const size_t row = (size_t)b * M + qid;
        if (l16 == 0) {
            nnCount[row] = cnt;
            if (found == 0) *flag = 1;          // this query takes a second pass in the reference: the chain kernel redoes the call
        }
        for (int slot = l16; slot < K; slot += 16) {
            int id = 0, bin = 0;
            float dist = 0.0f;
            if (slot < cnt) {
                id = list[slot];
                const float dx = db[(size_t)id * 3] - qx;
                const float dy = db[(size_t)id * 3 + 1] - qy;
                const float dz = db[(size_t)id * 3 + 2] - qz;
                const float d2 = (dx * dx + dy * dy) + dz * dz;   // tf_nnquery_gpu.cu:45-46
                dist = sqrtf(sqrtf(d2));                          // :47 then :54 — sqrt of the distance
                if (FUSE) {
                    if (fx.filt != nullptr)
                        bin = fx.ocml ? sphere_bin<true>(dx, dy, dz, dist, fx.radius, fx.n, fx.p, fx.q)
                                      : sphere_bin<false>(dx, dy, dz, dist, fx.radius, fx.n, fx.p, fx.q);
                    if (fx.deg != nullptr) {
                        fx.slotPos[row * K + slot] = atomicAdd(&fx.deg[((size_t)b * N + id) * fx.F + bin], 1);
                        fx.binUsed[bin] = 1;          // benign race: every writer stores 1
                    }
                }
            }
            nnIndex[row * K + slot] = id;                          // unused slots read 0
            nnDist[row * K + slot] = dist;
            if (FUSE && fx.filt != nullptr) fx.filt[row * K + slot] = bin;
        }
    }
}

// The positions beyond the grid's reach: their spheres hold hundreds of points, so the reference's ascending scan meets
// nn_sample hits after a fraction of the cloud — a WAVE per query walks the cloud 128 points per trip in index order and stops
// there.  No chain state (the radius of position k is tabulated), every query independent: 49 152 waves at S3DIS level 0
// where the chain kernel walks 1024 chains per cloud one query after the other.  K <= 256 slots in LDS per wave.
constexpr int kDenseQ = 4;              // queries per wave: they share every trip's points
template <bool FUSE>
__global__ __launch_bounds__(256) void nndense_kernel(
    int B, int N, int M, int K, int j0, int parts, GraphFuse fx, int* __restrict__ flag, const GridRadii* __restrict__ radii,
    const GridHdr* __restrict__ hdr, const float* __restrict__ database, const float* __restrict__ query,
    int* __restrict__ nnIndex, int* __restrict__ nnCount, float* __restrict__ nnDist)
{
    extern __shared__ int lhits[];                    // [4 waves][kDenseQ][K]
    int b, part;
    xcd_decode((int)blockIdx.x, B, parts, b, part);
    if (b < 0 || hdr[b].nx == 0) return;              // (a cloud the grid gave up on: the chain kernel recomputes the call)
    const int lane = lane_id(), w = uniform((int)threadIdx.x >> 6);
    const int jw = j0 + (part * 4 + w) * kDenseQ;     // the wave's first query; its queries share a chain position (1024 | j0)
    if (jw >= M) return;
    int* h = lhits + w * kDenseQ * K;
    const float* db = database + (size_t)b * N * 3;
    if (jw < hdr[b].nq) return;                       // this cloud's early positions: the grid's
    const float T = radii->thr[(b / kRefGrid) * ((M + kRefBlock - 1) / kRefBlock) + jw / kRefBlock];
    float qx[kDenseQ], qy[kDenseQ], qz[kDenseQ];
    int s[kDenseQ];
    bool live[kDenseQ];
#pragma unroll
    for (int c = 0; c < kDenseQ; c++) {
        live[c] = jw + c < M;
        const size_t row = (size_t)b * M + (live[c] ? jw + c : jw);
        qx[c] = query[row * 3]; qy[c] = query[row * 3 + 1]; qz[c] = query[row * 3 + 2];
        s[c] = live[c] ? 0 : K;                       // a query past the end is closed from the start
    }
    // four strips of 64 points per trip, and the next trip's twelve loads are issued before this trip's points are tested
    constexpr int S = 4;
    float px[S], py[S], pz[S];
    auto fetch = [&](int base, float* x, float* y, float* z) {
#pragma unroll
        for (int u = 0; u < S; u++) {
            const int i = base + u * 64 + lane;
            const int cl = i < N ? i : N - 1;                   // points past the end: clamped here, masked at the test
            x[u] = db[(size_t)cl * 3]; y[u] = db[(size_t)cl * 3 + 1]; z[u] = db[(size_t)cl * 3 + 2];
        }
    };
    fetch(0, px, py, pz);
    for (int base = 0; base < N; base += 64 * S) {
        bool open = false;
#pragma unroll
        for (int c = 0; c < kDenseQ; c++) open = open || s[c] < K;
        if (!open) break;
        float nx[S], ny[S], nz[S];
        const bool more = base + 64 * S < N;
        if (more) fetch(base + 64 * S, nx, ny, nz);
#pragma unroll
        for (int c = 0; c < kDenseQ; c++) {
            if (s[c] < K) {                                       // wave-uniform
                unsigned long long m[S];
                bool hit[S];
                unsigned long long any = 0ull;
                // two strips per packed instruction (v_pk_add / v_pk_mul_f32: IEEE per component, no contraction — the
                // same roundings as the scalar form)
#pragma unroll
                for (int u = 0; u < S; u += 2) {
                    const f32x2 dx = f32x2{px[u], px[u + 1]} - qx[c];
                    const f32x2 dy = f32x2{py[u], py[u + 1]} - qy[c];
                    const f32x2 dz = f32x2{pz[u], pz[u + 1]} - qz[c];
                    const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;        // tf_nnquery_gpu.cu:45-46
                    hit[u] = base + u * 64 + lane < N && d2.x < T;
                    hit[u + 1] = base + (u + 1) * 64 + lane < N && d2.y < T;
                    m[u] = __builtin_amdgcn_ballot_w64(hit[u]);
                    m[u + 1] = __builtin_amdgcn_ballot_w64(hit[u + 1]);
                    any |= m[u] | m[u + 1];
                }
                if (any != 0ull) {
                    // ascending index: a strip's hits take their slots before the next strip's
                    int sc = s[c];
#pragma unroll
                    for (int u = 0; u < S; u++) {
                        const int pos = sc + prefix_popc(m[u]);
                        if (hit[u] && pos < K) h[c * K + pos] = base + u * 64 + lane;
                        sc += __popcll(m[u]);
                    }
                    s[c] = sc;
                }
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < S; u++) { px[u] = nx[u]; py[u] = ny[u]; pz[u] = nz[u]; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int c = 0; c < kDenseQ; c++) {
        if (!live[c]) break;
        const size_t row = (size_t)b * M + jw + c;
        const int cnt = s[c] < K ? s[c] : K;
        if (lane == 0) {
            nnCount[row] = cnt;
            if (s[c] == 0) *flag = 1;       // this query takes a second pass in the reference: the chain kernel redoes the call
        }
        for (int slot = lane; slot < K; slot += 64) {
            int id = 0, bin = 0;
            float dist = 0.0f;
            if (slot < cnt) {
                id = h[c * K + slot];
                const float dx = db[(size_t)id * 3] - qx[c];
                const float dy = db[(size_t)id * 3 + 1] - qy[c];
                const float dz = db[(size_t)id * 3 + 2] - qz[c];
                const float d2 = (dx * dx + dy * dy) + dz * dz;   // tf_nnquery_gpu.cu:45-46
                dist = sqrtf(sqrtf(d2));                          // :47 then :54 — sqrt of the distance
                if (FUSE) {
                    if (fx.filt != nullptr)
                        bin = fx.ocml ? sphere_bin<true>(dx, dy, dz, dist, fx.radius, fx.n, fx.p, fx.q)
                                      : sphere_bin<false>(dx, dy, dz, dist, fx.radius, fx.n, fx.p, fx.q);
                    if (fx.deg != nullptr) {
                        fx.slotPos[row * K + slot] = atomicAdd(&fx.deg[((size_t)b * N + id) * fx.F + bin], 1);
                        fx.binUsed[bin] = 1;          // benign race: every writer stores 1
                    }
                }
            }
            nnIndex[row * K + slot] = id;                          // unused slots read 0
            nnDist[row * K + slot] = dist;
            if (FUSE && fx.filt != nullptr) fx.filt[row * K + slot] = bin;
        }
    }
}

static std::atomic<long long> g_grid_launches{0};

// bytes of the cell grid of one call: header + flag, cell starts, the cloud in cell order, the query orders
size_t nngrid_workspace_bytes(int B, int N, int M)
{
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    // shapes the grid never takes (the same test as nngrid_search, minus what depends on K and the radius mode)
    if (N < 1024 || N > 65536 || M < 64 || M > kMaxPositions * kRefBlock || (long long)N * M < (1LL << 22)) return 0;
    const size_t hdrBytes = 1024 + sizeof(GridHdr) * (size_t)B;
    const size_t csBytes = sizeof(int) * (size_t)B * (kGridMaxCells + 1);
    const size_t ptBytes = sizeof(float4) * (size_t)B * N;
    const size_t qoBytes = sizeof(int) * (size_t)B * M;
    const size_t a16 = 15;
    return ((hdrBytes + a16) & ~a16) + ((csBytes + a16) & ~a16) + ptBytes + qoBytes;
}

int nngrid_search(int B, int N, int M, int K, float radius, int fixed, const float* database, const float* query, int* nn_index,
                  int* nn_count, float* nn_dist, const GraphFuse* fuse, hipStream_t st, const int** gate, int* grid_done,
                  void* workspace, size_t workspace_bytes, bool library_scratch)
{
    // worth it from ~4 M point pairs per cloud (below, the chain kernel's scan of the whole cloud from LDS is as fast as the
    // grid's build + search: 2048 x 512 measured 60 vs 68 us); the sorted hit lists hold 16-bit indices; with more than 32 clouds
    // (two clouds per reference block: its chains go on from one to the next) every thread must visit the same number of
    // queries per cloud, so that a query's position is a function of (cloud, j) alone
    if (N < 1024 || N > 65536 || M < 64 || M > kMaxPositions * kRefBlock || (long long)N * M < (1LL << 22) || K > kGridMaxK ||
        (!fixed && B > kRefGrid && ((M >= kRefBlock && M % kRefBlock != 0) ||
                                    ((B + kRefGrid - 1) / kRefGrid) * ((M + kRefBlock - 1) / kRefBlock) > kMaxPositions)))
        return 0;
    // chain positions whose radius stays <= 3 r_0 go through the grid; the later ones find nn_sample hits early in an ascending
    // scan (nndense_kernel)
    int npos = 1;
    if (!fixed) {
        float rk = radius;
        while (npos < kGridMaxPos) {
            rk = (float)((double)rk + 0.05);
            if (!(rk <= kGridReach * radius)) break;
            npos++;
        }
        const int have = ((B + kRefGrid - 1) / kRefGrid) * ((M + kRefBlock - 1) / kRefBlock);      // positions in use
        if (npos > have) npos = have;
    }
    const size_t hdrBytes = 1024 + sizeof(GridHdr) * (size_t)B;
    const size_t csBytes = sizeof(int) * (size_t)B * (kGridMaxCells + 1);
    const size_t ptBytes = sizeof(float4) * (size_t)B * N;
    const size_t qoBytes = sizeof(int) * (size_t)B * M;
    const size_t a16 = 15;
    const size_t total = ((hdrBytes + a16) & ~a16) + ((csBytes + a16) & ~a16) + ptBytes + qoBytes;
    // the grid's memory: the caller's workspace (the `_ws` entry points: no allocation, legal under stream capture), or — the
    // convenience entry points with the reference's signature — the library's buffer of this (device, stream); none: no grid
    unsigned char* ws = (unsigned char*)workspace;
    if (ws != nullptr) {
        if (workspace_bytes < total || (reinterpret_cast<size_t>(ws) & 15) != 0) {
            set_error("BuildSphereNeighbor: workspace %zu B < required %zu B (or not 16-byte aligned)", workspace_bytes, total);
            return SPH3D_EWORKSPACE;
        }
    } else if (library_scratch) {
        ws = (unsigned char*)stream_scratch(st, total);
    }
    if (ws == nullptr) return 0;
    int* flag = (int*)ws;
    GridRadii* radii = (GridRadii*)(ws + 64);
    GridHdr* hdr = (GridHdr*)(ws + 1024);
    int* cellStart = (int*)(ws + ((hdrBytes + a16) & ~a16));
    float4* pts = (float4*)((unsigned char*)cellStart + ((csBytes + a16) & ~a16));
    int* qorder = (int*)((unsigned char*)pts + ptBytes);
    int rc = check_hip(hipMemsetAsync(flag, 0, 16, st), "nngrid: memset");
    if (rc) return rc;
    const size_t ldsBuild = sizeof(int) * (size_t)kGridMaxCells;
    hipLaunchKernelGGL(nngrid_build_kernel, dim3(B), dim3(1024), ldsBuild, st, N, M, npos, fixed, radius, database, flag,
                       radii, hdr, cellStart, pts);
    const int nq = fixed ? M : (M < npos * kRefBlock ? M : npos * kRefBlock);
    const int nslices = (nq + kRefBlock - 1) / kRefBlock;
    hipLaunchKernelGGL(nngrid_qorder_kernel, dim3(B * nslices), dim3(1024), ldsBuild, st, M, nslices, query, hdr, qorder);
    const int parts = (nq + 3) / 4;
    const int W = ((N + 31) / 32 + 15) & ~15;
    const size_t lds = (size_t)4 * W * 4 + 4 * 64 * 8 + (size_t)4 * K * 2;
    GraphFuse fx{};
    if (fuse != nullptr) fx = *fuse;
#define SPH3D_GRID(FU)                                                                                                      \
    do {                                                                                                                    \
        auto kern = nngrid_search_kernel<FU>;                                                                               \
        hipLaunchKernelGGL(kern, dim3(xcd_grid(B, parts)), dim3(64), lds, st, B, N, M, K, W, parts, fixed, fx, flag, radii, \
                           hdr, cellStart, pts, qorder, database, query, nn_index, nn_count, nn_dist);                      \
    } while (0)
    if (fuse != nullptr) SPH3D_GRID(true); else SPH3D_GRID(false);
#undef SPH3D_GRID
    // the later positions: early-stopping scans.  The first query any cloud leaves to them (the clouds of the last block round
    // start at the highest positions): the kernel skips the part of a cloud that its grid share covers
    int j0 = M;
    if (!fixed) {
        const int S = (M + kRefBlock - 1) / kRefBlock;
        int sl = npos - ((B - 1) / kRefGrid) * S;
        sl = sl < 0 ? 0 : (sl > S ? S : sl);
        j0 = sl * kRefBlock < M ? sl * kRefBlock : M;
    }
    if (j0 < M) {
        const int dparts = (M - j0 + 4 * kDenseQ - 1) / (4 * kDenseQ);
        const size_t dlds = sizeof(int) * 4 * kDenseQ * (size_t)K;
        if (fuse != nullptr)
            hipLaunchKernelGGL(nndense_kernel<true>, dim3(xcd_grid(B, dparts)), dim3(256), dlds, st, B, N, M, K, j0, dparts, fx, flag,
                               radii, hdr, database, query, nn_index, nn_count, nn_dist);
        else
            hipLaunchKernelGGL(nndense_kernel<false>, dim3(xcd_grid(B, dparts)), dim3(256), dlds, st, B, N, M, K, j0, dparts, fx, flag,
                               radii, hdr, database, query, nn_index, nn_count, nn_dist);
    }
    rc = check_launch("nngrid_search");
    if (rc) return rc;
    g_grid_launches.fetch_add(1, std::memory_order_relaxed);
    *gate = flag;
    *grid_done = 1 << 20;          // every query is done unless the flag is up
    return 1;
}

}  // namespace sph3d

extern "C" long long sph3d_nngrid_launches(void) { return sph3d::g_grid_launches.load(std::memory_order_relaxed); }
