// nnquery.hip — range / cube neighbour search for gfx950.
//
// Replaces cal_nn_binidx / cal_nn_binidx_cube (tf_ops/nnquery/tf_nnquery_gpu.cu:15-113).
//
// MI355X design (not the reference's thread-per-query serial scan):
//   * The reference mutates its `radius` parameter inside each thread and never resets it, so
//     the queries a reference thread (block i%32, thread j%1024) visits form a CHAIN whose
//     search radius depends on everything before it.  A chain is therefore the unit of
//     sequential work here: one wavefront owns CPW chains and walks them query by query,
//     while its 64 lanes scan the database 64 points at a time ("strip").
//   * Ascending-index, first-K-win slotting inside a strip is a ballot plus a prefix popcount
//     (v_mbcnt), so no atomics and no sorting.
//   * The cloud's xyz are staged once per workgroup in LDS as SoA (<= 144 KB of the 160 KB),
//     16 waves per workgroup, so the O(M*N) scan never touches HBM/L2 again.
//   * The in-range predicate  sqrtf(d2) < r && |sqrtf(d2)-r| > 1e-6  is monotone in d2, so per
//     (chain, pass) the wave finds the exact float threshold T(r) with a 64-ary search over float
//     bit patterns (6 rounds) and the inner loop is sub/mul/add/compare only: no sqrt per pair,
//     bit-identical decisions.
//   * -ffp-contract=off: d2 = (dx*dx + dy*dy) + dz*dz must round exactly like the oracle.
#include <cstdlib>
#include "common.hpp"
#include "sphere_bin.hpp"
#include "nnquery.hpp"

namespace sph3d {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWavesPerWG = 16;
constexpr int kMaxChunk = 12288;   // points per LDS chunk (3 * 12288 * 4 B = 144 KB)
constexpr int kThrTable = 64;      // positions of the radius sequence with a precomputed threshold

// LDS image of a cloud chunk: blocks of 128 points (one trip of the scan = two strips of 64), each block
// x[strip 0][64] x[strip 1][64] y[0][64] y[1][64] z[0][64] z[1][64] (384 floats).  A lane's operands for the two strips of
// a trip are then sp[0], sp[64] (x), sp[128], sp[192] (y), sp[256], sp[320] (z) from ONE address register: three
// ds_read2st64_b32, each filling the register PAIR the packed subtract wants.  The chunk is padded to whole trips with a
// sentinel coordinate whose squared distance to any finite query overflows to +inf: d2 < T is false for every threshold,
// so the scan needs no bounds masks.
constexpr float kPadCoord = 3.0e38f;
__device__ __forceinline__ int strip_off(int k, int comp) { return (k >> 7) * 384 + comp * 128 + (k & 127); }

// stage `cn` points starting at point c0 of one cloud
__device__ __forceinline__ void stage_cloud(const float* __restrict__ dbi, int c0, int cn, float* lds)
{
    const float* src = dbi + (size_t)c0 * 3;
    for (int e = threadIdx.x; e < cn * 3; e += blockDim.x) {
        const float v = src[e];
        const int pnt = e / 3;
        const int comp = e - pnt * 3;
        lds[strip_off(pnt, comp)] = v;
    }
    const int cpad = (cn + 127) & ~127;
    for (int e = cn * 3 + (int)threadIdx.x; e < cpad * 3; e += blockDim.x) {
        const int pnt = e / 3;
        const int comp = e - pnt * 3;
        lds[strip_off(pnt, comp)] = kPadCoord;
    }
}

// old with lane `lane` replaced by `val` (both wave-uniform): v_writelane_b32 through the compiler's own intrinsic (it
// routes the lane select through M0: one SGPR operand per VALU instruction on gfx9)
extern "C" __device__ int sph3d_writelane_i32(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int write_lane(int old, int val, int lane) { return sph3d_writelane_i32(val, lane, old); }

// DEFER: the hits of a query are collected as bare indices in a per-chain LDS list during the scan, and the outputs of
// the finished query (indices, sqrt(sqrt(d2)), zero fill) are produced by ONE coalesced pass over its <= K slots.  The
// first version wrote nn_index / nn_dist from inside the strip loop: every strip with at least one hit ran two
// correctly rounded sqrtf and two scattered stores on the whole wave (~40 such strips per query at S3DIS level 0).
// FUSE (sph3d_build_sphere_graph): the output pass of a finished query also computes the spherical-kernel bin of every
// neighbour (the operands dx, dy, dz, sqrt-distance are in registers there) and, when a transposed graph will be
// needed, counts the edge into its (source point, bin) segment — the atomic's return value is the edge's position in
// the segment (graph.hip).  Replaces one launch + one pass over [B,M,K] for the bins and one for the segment counts.

template <int CPW, bool MULTI, bool DEFER, bool FUSE>
__global__ __launch_bounds__(kWavesPerWG * 64) void nnquery_sphere_kernel(
    int B, int N, int M, int K, float radius0, int chunkN, int groups, int fixed, GraphFuse fx,
    const float* __restrict__ database, const float* __restrict__ query,
    int* __restrict__ nnIndex, int* __restrict__ nnCount, float* __restrict__ nnDist, const int* __restrict__ gate, int gridDone)
{
    // gate != nullptr: the cell-grid search (nngrid.hip) ran in front and — unless it raised its flag — has produced the rows of
    // the first `gridDone` queries of every chain (the chain positions whose radius is small): the chains start behind them, at
    // the radius of that position.  With the flag raised this kernel computes the whole call.
    const int kstart = (gate != nullptr && *gate == 0) ? gridDone : 0;
    if ((long long)kstart * kRefBlock >= M) return;       // everything came from the grid
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int* lhits = reinterpret_cast<int*>(lds + 3 * ((chunkN + 127) & ~127));      // [wave][CPW][K] when DEFER
    // Every chain's radius walks the SAME sequence r_0 = radius0, r_{k+1} = float(double(r_k) + 0.05) (tf_nnquery_gpu.cu:59);
    // only the position k differs.  The exact thresholds T(r_k) of the first kThrTable positions are found once per workgroup
    // (four 6-round searches per wave) instead of once per chain and pass: the search was a third of the kernel's
    // instructions at S3DIS level 0 (32 queries per wave, ~400 instructions each).
    float* lthr = reinterpret_cast<float*>(lhits + (DEFER ? kWavesPerWG * CPW * K : 0));   // [kThrTable]
    {
        const int w = (int)threadIdx.x >> 6;
        float rk = radius0;
        for (int k = 0; k < kThrTable; k++) {
            if ((k & (kWavesPerWG - 1)) == w) {
                const float T = range_threshold(rk);
                if (lane_id() == 0) lthr[k] = T;
            }
            rk = (float)((double)rk + 0.05);
        }
        __syncthreads();
    }

    const int nt = M < kRefBlock ? M : kRefBlock;
    const int bb = (int)blockIdx.x / groups;     // reference block id  (= cloud index mod 32)
    const int g = (int)blockIdx.x % groups;
    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();

    int t[CPW];        // reference thread id of each chain
    float r[CPW];      // the chain's running radius (carried across queries AND clouds)
    int kpos[CPW];     // its position in the radius sequence: r[c] = r_kpos
#pragma unroll
    for (int c = 0; c < CPW; c++) {
        t[c] = (g * kWavesPerWG + wave) * CPW + c;
        r[c] = radius0;
        kpos[c] = kstart;
    }
    for (int k = 0; k < kstart; k++) {                      // (kstart is 0 or past the last query)
#pragma unroll
        for (int c = 0; c < CPW; c++) r[c] = (float)((double)r[c] + 0.05);
    }

    auto threshold_of = [&](int c) -> float {
        return kpos[c] < kThrTable ? uniformf(lthr[kpos[c]]) : range_threshold(r[c]);
    };

    for (int i = bb; i < B; i += kRefGrid) {
        const float* dbi = database + (size_t)i * N * 3;
        const float* qi = query + (size_t)i * M * 3;

        int j[CPW], s[CPW], passes[CPW];
        bool has[CPW];
        float qx[CPW], qy[CPW], qz[CPW], thr[CPW];
#pragma unroll
        for (int c = 0; c < CPW; c++) {
            j[c] = t[c] + kstart * kRefBlock;
            has[c] = (t[c] < nt) && (j[c] < M);
            s[c] = 0;
            passes[c] = 0;
            qx[c] = qy[c] = qz[c] = 0.0f;
            thr[c] = 0.0f;
            if (fixed) { r[c] = radius0; kpos[c] = 0; }         // fixed-radius mode: nothing is carried from cloud to cloud
            if (has[c]) {
                qx[c] = qi[j[c] * 3];
                qy[c] = qi[j[c] * 3 + 1];
                qz[c] = qi[j[c] * 3 + 2];
                thr[c] = threshold_of(c);
            }
        }

        if (!MULTI) {
            __syncthreads();   // previous cloud's scans are finished
            stage_cloud(dbi, 0, N, lds);
            __syncthreads();
        }

        // DEFER: hit-mask records of the current window of trips and the expansion into the chain's LDS slot list
        int rec[CPW][4];
        int written[CPW];       // hits of the chain's current pass that are already in its slot list (wave-uniform)
#pragma unroll
        for (int c = 0; c < CPW; c++) {
            rec[c][0] = rec[c][1] = rec[c][2] = rec[c][3] = 0;
            written[c] = 0;
        }
        int wtrip = 0, wbase = 0;       // trips recorded in the window; index of the window's first point
        auto flush_hits = [&](int c) {
            if (s[c] == written[c]) return;            // nothing recorded (wave-uniform)
            unsigned long long m0 = (unsigned long long)(unsigned)rec[c][0] | ((unsigned long long)(unsigned)rec[c][1] << 32);
            unsigned long long m1 = (unsigned long long)(unsigned)rec[c][2] | ((unsigned long long)(unsigned)rec[c][3] << 32);
            int pos = written[c] + wave_excl_scan(__popcll(m0) + __popcll(m1));
            int* h = lhits + (wave * CPW + c) * K;
            const int id0 = wbase + lane * 128;
            while (m0 != 0ull && pos < K) {            // ascending index: trip by trip (lane), first strip, second strip
                h[pos++] = id0 + (int)__builtin_ctzll(m0);
                m0 &= m0 - 1ull;
            }
            while (m1 != 0ull && pos < K) {
                h[pos++] = id0 + 64 + (int)__builtin_ctzll(m1);
                m1 &= m1 - 1ull;
            }
            written[c] = s[c];
            rec[c][0] = rec[c][1] = rec[c][2] = rec[c][3] = 0;
        };

        while (true) {
            bool any = false;
#pragma unroll
            for (int c = 0; c < CPW; c++) any = any || has[c];
            if (MULTI) any = __syncthreads_or(any ? 1 : 0) != 0;
            if (!any) break;

            // ---- one sweep over the database for every active chain of this wave ----
            wtrip = 0;
            wbase = 0;
            bool act[CPW];       // the chain has a query with free slots (wave-uniform)
#pragma unroll
            for (int c = 0; c < CPW; c++) act[c] = has[c] && s[c] < K;
            for (int c0 = 0; c0 < N; c0 += chunkN) {
                const int cn = (N - c0) < chunkN ? (N - c0) : chunkN;
                if (MULTI) {
                    __syncthreads();
                    stage_cloud(dbi, c0, cn, lds);
                    __syncthreads();
                }
                // two strips (128 database points) per trip, the pair of distances in one packed register: the subtractions,
                // products and sums are v_pk_{add,mul}_f32 (IEEE per component, no contraction: the same roundings as the scalar
                // form).
                if constexpr (DEFER) {
                    // A trip only RECORDS its hit masks: lane t of four registers per chain holds the two 64-bit masks of trip t
                    // of the current window of 64 trips (v_writelane; scalar popcounts keep the slot count), and the
                    // ascending-index slot list is expanded from the masks once per window (flush_hits).  The arithmetic of the
                    // CPW chains is unconditional and interleaved (no branch, no packed-math hazard nops); a chain without free
                    // slots is masked out on the scalar side.  Round 3 counters at S3DIS level 0: 44 k VALU + 33 k SALU and
                    // 13 branches per trip and wave before (per-trip slot computation, predicated LDS stores, bounds masks,
                    // one branch nest per chain) -> DESIGN.md section 0.
                    unsigned long long am[CPW];        // all ones while the chain's query has free slots
                    bool open = false;
#pragma unroll
                    for (int c = 0; c < CPW; c++) {
                        am[c] = act[c] ? ~0ull : 0ull;
                        open = open || act[c];
                    }
                    for (int base = 0; open && base < cn; base += 128) {
                        const float* sp = lds + (base >> 7) * 384 + lane;
                        const f32x2 x = {sp[0], sp[64]}, y = {sp[128], sp[192]}, z = {sp[256], sp[320]};
                        unsigned long long m0[CPW], m1[CPW], any = 0ull;
#pragma unroll
                        for (int c = 0; c < CPW; c++) {
                            const f32x2 dx = x - qx[c];
                            const f32x2 dy = y - qy[c];
                            const f32x2 dz = z - qz[c];
                            const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;   // tf_nnquery_gpu.cu:45-46
                            m0[c] = __builtin_amdgcn_ballot_w64(d2.x < thr[c]) & am[c];
                            m1[c] = __builtin_amdgcn_ballot_w64(d2.y < thr[c]) & am[c];
                            any |= m0[c] | m1[c];
                        }
                        if (any != 0ull) {
#pragma unroll
                            for (int c = 0; c < CPW; c++) {
                                if ((m0[c] | m1[c]) != 0ull) {
                                    rec[c][0] = write_lane(rec[c][0], (int)(unsigned)m0[c], wtrip);
                                    rec[c][1] = write_lane(rec[c][1], (int)(unsigned)(m0[c] >> 32), wtrip);
                                    rec[c][2] = write_lane(rec[c][2], (int)(unsigned)m1[c], wtrip);
                                    rec[c][3] = write_lane(rec[c][3], (int)(unsigned)(m1[c] >> 32), wtrip);
                                    s[c] += __popcll(m0[c]) + __popcll(m1[c]);
                                    if (s[c] >= K) am[c] = 0ull;
                                }
                            }
                            unsigned long long o = 0ull;
#pragma unroll
                            for (int c = 0; c < CPW; c++) o |= am[c];
                            open = o != 0ull;
                        }
                        if (++wtrip == 64) {
#pragma unroll
                            for (int c = 0; c < CPW; c++) flush_hits(c);
                            wtrip = 0;
                            wbase = c0 + base + 128;
                        }
                    }
                    // the window ends with the chunk
#pragma unroll
                    for (int c = 0; c < CPW; c++) {
                        flush_hits(c);
                        act[c] = has[c] && s[c] < K;
                    }
                    wtrip = 0;
                    wbase = c0 + cn;
                } else {
                    // K too large for LDS slot lists: slots are computed and written from inside the scan
                    for (int base = 0; base < cn; base += 128) {
                        bool open = false;   // some chain still has free slots
#pragma unroll
                        for (int c = 0; c < CPW; c++) open = open || act[c];
                        if (!open) break;
                        const float* sp = lds + (base >> 7) * 384 + lane;
                        const f32x2 x = {sp[0], sp[64]}, y = {sp[128], sp[192]}, z = {sp[256], sp[320]};
#pragma unroll
                        for (int c = 0; c < CPW; c++) {
                            if (act[c]) {   // wave-uniform
                                const f32x2 dx = x - qx[c];
                                const f32x2 dy = y - qy[c];
                                const f32x2 dz = z - qz[c];
                                const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;   // tf_nnquery_gpu.cu:45-46
                                const unsigned long long m0 = __builtin_amdgcn_ballot_w64(d2.x < thr[c]);
                                const unsigned long long m1 = __builtin_amdgcn_ballot_w64(d2.y < thr[c]);
                                if ((m0 | m1) != 0ull) {
                                    // ascending index: the first strip's hits take their slots before the second strip's
                                    const bool hit0 = (m0 >> lane) & 1ull, hit1 = (m1 >> lane) & 1ull;
                                    const int pos0 = s[c] + prefix_popc(m0);
                                    const int pos1 = s[c] + __popcll(m0) + prefix_popc(m1);
                                    const size_t o = ((size_t)i * M + j[c]) * K;
                                    if (hit0 && pos0 < K) {
                                        nnIndex[o + pos0] = c0 + base + lane;
                                        nnDist[o + pos0] = sqrtf(sqrtf(d2.x));   // :47 then :54 — sqrt of the distance
                                    }
                                    if (hit1 && pos1 < K) {
                                        nnIndex[o + pos1] = c0 + base + 64 + lane;
                                        nnDist[o + pos1] = sqrtf(sqrtf(d2.y));
                                    }
                                    s[c] += __popcll(m0) + __popcll(m1);
                                    act[c] = s[c] < K;
                                }
                            }
                        }
                    }
                }
            }

            // ---- end of pass: grow the radius (tf_nnquery_gpu.cu:59), maybe finish the query ----
#pragma unroll
            for (int c = 0; c < CPW; c++) {
                if (has[c]) {
                    r[c] = (float)((double)r[c] + 0.05);
                    kpos[c]++;
                    passes[c]++;
                    if (s[c] > 0 || passes[c] >= SPH3D_MAX_GROWTH_PASSES) {
                        const int cnt = s[c] < K ? s[c] : K;
                        const size_t row = (size_t)i * M + j[c];
                        if (lane == 0) nnCount[row] = cnt;
                        if (DEFER) {
                            // the query's K output slots in one coalesced pass: distance recomputed from the same operands
                            const int* h = lhits + (wave * CPW + c) * K;
                            for (int slot = lane; slot < K; slot += 64) {
                                int id = 0, bin = 0;
                                float dist = 0.0f;
                                if (slot < cnt) {
                                    id = h[slot];
                                    const float dx = dbi[(size_t)id * 3] - qx[c];
                                    const float dy = dbi[(size_t)id * 3 + 1] - qy[c];
                                    const float dz = dbi[(size_t)id * 3 + 2] - qz[c];
                                    const float d2 = (dx * dx + dy * dy) + dz * dz;   // tf_nnquery_gpu.cu:45-46
                                    dist = sqrtf(sqrtf(d2));                          // :47 then :54 — sqrt of the distance
                                    if (FUSE) {
                                        if (fx.filt != nullptr)
                                            bin = fx.ocml ? sphere_bin<true>(dx, dy, dz, dist, fx.radius, fx.n, fx.p, fx.q)
                                                          : sphere_bin<false>(dx, dy, dz, dist, fx.radius, fx.n, fx.p, fx.q);
                                        if (fx.deg != nullptr) {
                                            fx.slotPos[row * K + slot] = atomicAdd(&fx.deg[((size_t)i * N + id) * fx.F + bin], 1);
                                            fx.binUsed[bin] = 1;          // benign race: every writer stores 1
                                        }
                                    }
                                }
                                nnIndex[row * K + slot] = id;                          // unused slots read 0
                                nnDist[row * K + slot] = dist;
                                if (FUSE && fx.filt != nullptr) fx.filt[row * K + slot] = bin;
                            }
                        } else {
                            for (int slot = cnt + lane; slot < K; slot += 64) {   // unused slots read 0
                                nnIndex[row * K + slot] = 0;
                                nnDist[row * K + slot] = 0.0f;
                            }
                        }
                        j[c] += kRefBlock;
                        has[c] = j[c] < M;
                        s[c] = 0;
                        written[c] = 0;
                        passes[c] = 0;
                        if (fixed) { r[c] = radius0; kpos[c] = 0; }         // fixed-radius mode: every query starts from the nominal radius
                        if (has[c]) {
                            qx[c] = qi[j[c] * 3];
                            qy[c] = qi[j[c] * 3 + 1];
                            qz[c] = qi[j[c] * 3 + 2];
                        }
                    }
                    if (has[c]) thr[c] = threshold_of(c);
                }
            }
        }
    }
}

// Cube search: queries are independent (no growth loop), one wave per query, database from L2.
__global__ __launch_bounds__(256) void nnquery_cube_kernel(
    int B, int N, int M, int G, int K, float length,
    const float* __restrict__ database, const float* __restrict__ query,
    int* __restrict__ nnIndex, int* __restrict__ nnCount)
{
    const int lane = lane_id();
    const int wid = uniform((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    const float half = length / 2;            // tf_nnquery_gpu.cu:96  (float / int)
    const float cell = length / (float)G;     // :99
    for (long long qid = wid; qid < (long long)B * M; qid += nwaves) {
        const int i = (int)(qid / M);
        const float* dbi = database + (size_t)i * N * 3;
        const float qx = query[qid * 3], qy = query[qid * 3 + 1], qz = query[qid * 3 + 2];
        int* row = nnIndex + (size_t)qid * K * 2;
        int s = 0;
        for (int base = 0; base < N && s < K; base += 64) {
            const int k = base + lane;
            const bool inb = k < N;
            const int kk = inb ? k : N - 1;
            const float dx = dbi[kk * 3] - qx;
            const float dy = dbi[kk * 3 + 1] - qy;
            const float dz = dbi[kk * 3 + 2] - qz;
            const bool hit = inb && fabsf(dx) < half && fabsf(dy) < half && fabsf(dz) < half;
            const unsigned long long mask = __ballot(hit);
            if (mask != 0ull) {
                const int pos = s + prefix_popc(mask);
                if (hit && pos < K) {
                    const int xId = (int)((dx + half) / cell);
                    const int yId = (int)((dy + half) / cell);
                    const int zId = (int)((dz + half) / cell);
                    row[pos * 2] = k;
                    row[pos * 2 + 1] = xId * G * G + yId * G + zId;
                }
                s += __popcll(mask);
            }
        }
        const int cnt = s < K ? s : K;
        if (lane == 0) nnCount[qid] = cnt;
        for (int slot = cnt * 2 + lane; slot < K * 2; slot += 64) row[slot] = 0;
    }
}

// clears the fused counters again before the chain kernel recomputes a call the grid search gave up on
__global__ void gated_zero_kernel(const int* __restrict__ gate, int* __restrict__ p, long long n)
{
    if (*gate == 0) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0;
}

// bytes of LDS for the deferred hit lists (0 = write hits from inside the scan: K too large for LDS lists)
static size_t hits_bytes(int CPW, int K)
{
    const size_t b = sizeof(int) * (size_t)kWavesPerWG * CPW * K;
    return b <= 40 * 1024 ? b : 0;
}

template <int CPW, bool MULTI, bool DEFER>
static int launch_sphere(int B, int N, int M, int K, float radius, int chunkN,
                         const float* database, const float* query,
                         int* nn_index, int* nn_count, float* nn_dist, hipStream_t stream, int fixed,
                         const GraphFuse* fuse = nullptr, const int* gate = nullptr, int grid_done = 0)
{
    const int nb = B < kRefGrid ? B : kRefGrid;
    const int nt = M < kRefBlock ? M : kRefBlock;
    const int groups = (nt + kWavesPerWG * CPW - 1) / (kWavesPerWG * CPW);
    const size_t lds = (size_t)3 * ((chunkN + 127) & ~127) * sizeof(float) + (DEFER ? hits_bytes(CPW, K) : 0) + kThrTable * sizeof(float);
    if constexpr (DEFER) {
        if (fuse != nullptr) {
            auto kernf = nnquery_sphere_kernel<CPW, MULTI, DEFER, true>;
            if (lds > 64 * 1024) {
                int rc = check_hip(hipFuncSetAttribute((const void*)kernf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                                   "nnquery: hipFuncSetAttribute");
                if (rc) return rc;
            }
            hipLaunchKernelGGL(kernf, dim3(nb * groups), dim3(kWavesPerWG * 64), lds, stream,
                               B, N, M, K, radius, chunkN, groups, fixed, *fuse, database, query, nn_index, nn_count, nn_dist, gate, grid_done);
            return check_launch("sph3d_build_sphere_graph");
        }
    }
    auto kern = nnquery_sphere_kernel<CPW, MULTI, DEFER, false>;
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "nnquery: hipFuncSetAttribute");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(nb * groups), dim3(kWavesPerWG * 64), lds, stream,
                       B, N, M, K, radius, chunkN, groups, fixed, GraphFuse{}, database, query, nn_index, nn_count, nn_dist, gate, grid_done);
    return check_launch("sph3d_build_sphere_neighbor");
}

}  // namespace sph3d

using namespace sph3d;

static int sphere_neighbor(int fixed, int B, int N, int M, int nn_sample, float radius,
                           const float* database, const float* query,
                           int* nn_index, int* nn_count, float* nn_dist,
                           sph3d_stream_t stream, const GraphFuse* fuse = nullptr, bool* fused = nullptr,
                           void* search_ws = nullptr, size_t search_ws_bytes = 0, bool library_scratch = true)
{
    if (fused) *fused = false;
    SPH3D_REQUIRE(radius > 0, "Range search requires radius>0, got %g", (double)radius);          // tf_nnquery.cpp:60
    SPH3D_REQUIRE(nn_sample > 0, "BuildSphereNeighbor requires nn_sample>0, got %d", nn_sample);  // :63
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0, "BuildSphereNeighbor: bad dims B=%d N=%d M=%d", B, N, M);
    if (B == 0 || M == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    const long long chains = (long long)(B < kRefGrid ? B : kRefGrid) * (M < kRefBlock ? M : kRefBlock);
    const int cpw = chains >= 256LL * kWavesPerWG * 4 ? 4 : (chains >= 256LL * kWavesPerWG * 2 ? 2 : 1);
    // LDS: the cloud chunk (12 B per point) next to the deferred hit lists
    size_t hb = hits_bytes(cpw, nn_sample);
    int maxChunk = (int)((160 * 1024 - kThrTable * 4 - hb) / 12) & ~127;      // whole trips of 128 points
    if (maxChunk > kMaxChunk) maxChunk = kMaxChunk;
    int chunkN = N < maxChunk ? N : maxChunk;
    const bool multi = N > chunkN;
    if (multi) {
        hb = hits_bytes(1, nn_sample);
        maxChunk = (int)((160 * 1024 - kThrTable * 4 - hb) / 12) & ~127;
        if (maxChunk > kMaxChunk) maxChunk = kMaxChunk;
        chunkN = maxChunk;
    }
    if (fused) *fused = (fuse != nullptr) && hb != 0;       // the bins come from the deferred output pass
    if (hb == 0) fuse = nullptr;
    // The cell-grid search first (nngrid.hip): complete unless some query has no neighbour inside the nominal radius — then
    // (device flag `gate`) the chain kernel below recomputes the call, after the fused counters have been cleared again.
    const int* gate = nullptr;
    int grid_done = 0;
    static const bool grid_on = !(getenv("SPH3D_NNGRID") && atoi(getenv("SPH3D_NNGRID")) == 0);
    if (grid_on) {
        const int g = nngrid_search(B, N, M, nn_sample, radius, fixed, database, query, nn_index, nn_count, nn_dist, fuse, st, &gate,
                                    &grid_done, search_ws, search_ws_bytes, library_scratch);
        if (g < 0) return g;
        if (g > 0 && fuse != nullptr && fuse->deg != nullptr) {
            const long long cnt = (long long)B * N * fuse->F + fuse->F;
            hipLaunchKernelGGL(gated_zero_kernel, dim3(256), dim3(256), 0, st, gate, fuse->deg, cnt);
        }
    }
#define SPH3D_NN(CP, MU)                                                                                          \
    return hb ? launch_sphere<CP, MU, true>(B, N, M, nn_sample, radius, chunkN, database, query, nn_index, nn_count, nn_dist, st, fixed, fuse, gate, grid_done) \
              : launch_sphere<CP, MU, false>(B, N, M, nn_sample, radius, chunkN, database, query, nn_index, nn_count, nn_dist, st, fixed, nullptr, gate, grid_done)
    if (multi) { SPH3D_NN(1, true); }
    if (cpw == 4) { SPH3D_NN(4, false); }
    if (cpw == 2) { SPH3D_NN(2, false); }
    SPH3D_NN(1, false);
#undef SPH3D_NN
}

extern "C" int sph3d_build_sphere_neighbor(int B, int N, int M, int nn_sample, float radius,
                                           const float* database, const float* query,
                                           int* nn_index, int* nn_count, float* nn_dist,
                                           sph3d_stream_t stream)
{
    return sphere_neighbor(0, B, N, M, nn_sample, radius, database, query, nn_index, nn_count, nn_dist, stream);
}

extern "C" int sph3d_build_sphere_neighbor_fixed(int B, int N, int M, int nn_sample, float radius,
                                                 const float* database, const float* query,
                                                 int* nn_index, int* nn_count, float* nn_dist,
                                                 sph3d_stream_t stream)
{
    return sphere_neighbor(1, B, N, M, nn_sample, radius, database, query, nn_index, nn_count, nn_dist, stream);
}

// ---- the same searches with the cell grid's memory from the caller: nothing is allocated, so the calls are legal under stream
// capture and independent of any library state (kernels never allocate).  workspace == NULL: no grid, the chain
// kernel computes the call (same rows).
extern "C" size_t sph3d_build_sphere_neighbor_workspace(int B, int N, int M) { return nngrid_workspace_bytes(B, N, M); }

extern "C" int sph3d_build_sphere_neighbor_ws(int B, int N, int M, int nn_sample, float radius,
                                              const float* database, const float* query,
                                              int* nn_index, int* nn_count, float* nn_dist,
                                              void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    return sphere_neighbor(0, B, N, M, nn_sample, radius, database, query, nn_index, nn_count, nn_dist, stream, nullptr, nullptr,
                           workspace, workspace_bytes, false);
}

extern "C" int sph3d_build_sphere_neighbor_fixed_ws(int B, int N, int M, int nn_sample, float radius,
                                                    const float* database, const float* query,
                                                    int* nn_index, int* nn_count, float* nn_dist,
                                                    void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    return sphere_neighbor(1, B, N, M, nn_sample, radius, database, query, nn_index, nn_count, nn_dist, stream, nullptr, nullptr,
                           workspace, workspace_bytes, false);
}

// Fused graph construction of one level: neighbour search + spherical-kernel bins (+ the segment counts of
// the transposed graph) in ONE kernel.  Same outputs, bit for bit, as sph3d_build_sphere_neighbor followed by
// sph3d_spherical_kernel with the same database / query (and sph3d_graph_transpose's counting pass).
extern "C" int sph3d_spherical_kernel(int B, int N, int M, int K, int n, int p, int q, float radius,
                                      const float* database, const float* query,
                                      const int* nn_index, const int* nn_count, const float* nn_dist,
                                      int* filt_index, sph3d_stream_t stream);
extern "C" int sph3d_graph_transpose_count(int B, int N, int M, int K, int F, const int* nn_index, const int* nn_count,
                                           const int* bin_index, int want_active, void* workspace, size_t workspace_bytes,
                                           sph3d_stream_t stream);

static int build_sphere_graph_impl(bool ocml, int B, int N, int M, int nn_sample, float radius, int n, int p, int q,
                                   const float* database, const float* query,
                                   int* nn_index, int* nn_count, float* nn_dist, int* filt_index,
                                   void* transpose_workspace, size_t transpose_workspace_bytes, sph3d_stream_t stream,
                                   void* search_ws = nullptr, size_t search_ws_bytes = 0, bool library_scratch = true)
{
    // filt_index == NULL: no bins (an inter-level graph: one segment per source point), only the search and the counts
    const bool binned = filt_index != nullptr;
    if (binned) {
        SPH3D_REQUIRE(n > 2 && n % 2 == 0, "Need n_>2 and n_%%2==0, got %d", n);               // tf_buildkernel.cpp:43
        SPH3D_REQUIRE(p > 0 && p % 2 == 0, "Need p_>0 and p_%%2==0, got %d", p);               // :46
        SPH3D_REQUIRE(q > 0, "Need q_>0, got %d", q);                                          // :49
    } else {
        SPH3D_REQUIRE(transpose_workspace != nullptr, "build_sphere_graph: without bins the call must count (workspace)");
    }
    const int F = binned ? n * p * q + 1 : 1;
    GraphFuse fx{};
    fx.n = n; fx.p = p; fx.q = q; fx.F = F;
    fx.radius = radius;
    fx.filt = filt_index;
    fx.ocml = ocml ? 1 : 0;
    if (transpose_workspace != nullptr) {
        const size_t need = sph3d_graph_transpose_workspace(B, N, M, nn_sample, F);
        if (transpose_workspace_bytes < need) {
            set_error("build_sphere_graph: transpose workspace %zu B < required %zu B", transpose_workspace_bytes, need);
            return SPH3D_EWORKSPACE;
        }
        // layout of graph.hip (common.hpp: tg_ws): counters, bin flags and scan status words (one zero fill), slot positions
        const TgWs w = tg_ws(transpose_workspace, B, N, M, nn_sample, F);
        fx.deg = w.deg;
        fx.binUsed = w.bin_used;
        fx.slotPos = w.slot_pos;
        int rc = zero_async(w.deg, sizeof(int) * w.zero_words, as_stream(stream), "build_sphere_graph: memset");
        if (rc) return rc;
    }
    bool fused = false;
    int rc = sphere_neighbor(0, B, N, M, nn_sample, radius, database, query, nn_index, nn_count, nn_dist, stream, &fx, &fused,
                             search_ws, search_ws_bytes, library_scratch);
    if (rc || fused) return rc;
    // shapes whose hit lists do not fit LDS: the same results from the separate kernels
    if (binned)
        rc = ocml ? sph3d_spherical_kernel_ocml(B, N, M, nn_sample, n, p, q, radius, database, query, nn_index, nn_count, nn_dist, filt_index, stream)
                  : sph3d_spherical_kernel(B, N, M, nn_sample, n, p, q, radius, database, query, nn_index, nn_count, nn_dist, filt_index, stream);
    if (rc || transpose_workspace == nullptr) return rc;
    return sph3d_graph_transpose_count(B, N, M, nn_sample, F, nn_index, nn_count, filt_index, binned ? 1 : 0, transpose_workspace,
                                       transpose_workspace_bytes, stream);
}

extern "C" int sph3d_build_sphere_graph(int B, int N, int M, int nn_sample, float radius, int n, int p, int q,
                                        const float* database, const float* query,
                                        int* nn_index, int* nn_count, float* nn_dist, int* filt_index,
                                        void* transpose_workspace, size_t transpose_workspace_bytes, sph3d_stream_t stream)
{
    return build_sphere_graph_impl(false, B, N, M, nn_sample, radius, n, p, q, database, query, nn_index, nn_count, nn_dist, filt_index,
                                   transpose_workspace, transpose_workspace_bytes, stream);
}

extern "C" int sph3d_build_sphere_graph_ocml(int B, int N, int M, int nn_sample, float radius, int n, int p, int q,
                                             const float* database, const float* query,
                                             int* nn_index, int* nn_count, float* nn_dist, int* filt_index,
                                             void* transpose_workspace, size_t transpose_workspace_bytes, sph3d_stream_t stream)
{
    return build_sphere_graph_impl(true, B, N, M, nn_sample, radius, n, p, q, database, query, nn_index, nn_count, nn_dist, filt_index,
                                   transpose_workspace, transpose_workspace_bytes, stream);
}

extern "C" int sph3d_build_sphere_graph_ws(int B, int N, int M, int nn_sample, float radius, int n, int p, int q, int ocml,
                                           const float* database, const float* query,
                                           int* nn_index, int* nn_count, float* nn_dist, int* filt_index,
                                           void* transpose_workspace, size_t transpose_workspace_bytes,
                                           void* search_workspace, size_t search_workspace_bytes, sph3d_stream_t stream)
{
    return build_sphere_graph_impl(ocml != 0, B, N, M, nn_sample, radius, n, p, q, database, query, nn_index, nn_count, nn_dist,
                                   filt_index, transpose_workspace, transpose_workspace_bytes, stream, search_workspace,
                                   search_workspace_bytes, false);
}

extern "C" int sph3d_build_cube_neighbor(int B, int N, int M, int grid_size, int nn_sample, float length,
                                         const float* database, const float* query,
                                         int* nn_index, int* nn_count, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(length > 0, "Cube search requires length>0, got %g", (double)length);
    SPH3D_REQUIRE(nn_sample > 0, "BuildCubeNeighbor requires nn_sample>0, got %d", nn_sample);
    SPH3D_REQUIRE(grid_size > 0, "BuildCubeNeighbor requires grid_size>0, got %d", grid_size);
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0, "BuildCubeNeighbor: bad dims B=%d N=%d M=%d", B, N, M);
    if (B == 0 || M == 0) return SPH3D_OK;
    const long long q = (long long)B * M;
    long long blocks = (q + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nnquery_cube_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       B, N, M, grid_size, nn_sample, length, database, query, nn_index, nn_count);
    return check_launch("sph3d_build_cube_neighbor");
}
