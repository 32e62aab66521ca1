// tile2.hip — LDS-tiled depthwise spherical convolution, forward, second design (round 3), gfx950.
//
// Same arithmetic as depthwise_conv3d_forward (tf_ops/convolution/tf_conv3d_gpu.cu:7-29); what changes is where the
// neighbour rows are read from.  The gather kernels of conv3d.hip fetch one 512-B feature row per edge through the CU's
// vector L1 (64 B/clk) from L2 (~56 B/clk/CU): 6.3 M edges x 512 B = 3.2 GB is a 90-us floor at level 0 of the S3DIS plan
// before a single FMA.  The LDS moves 256 B/clk/CU.  Spatially close output points share neighbours (8 Morton-consecutive
// points of an S3DIS-like block reference ~140 distinct rows for ~385 edges), so a workgroup that stages the UNION of a
// tile's rows in LDS once and gathers from LDS cuts the L2 traffic 2.7-4x and turns every edge into one ds_read_b64.
//
// Differences from the round-2 tiled kernel (tile.hip / convtile.hip: 0.21 ms, never the default because its plan cost
// 0.32 ms per graph and its skeleton 82 of 207 us):
//   * tiles are built GREEDILY (as many consecutive targets as fit `ucap` rows, at most 16), so a tile always fits and the
//     consumer has no split / overflow path; one plan kernel, one pass over the graph;
//   * rows travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write), so there is no software
//     pipeline to maintain: two or more independent 512-thread workgroups per CU overlap staging with gathering;
//   * a target's edges are sorted by bin in the plan; the consumer sums the rows of a (target, bin) group and multiplies
//     by the filter row once per group.  Edge slots and group ends are lane-indexed VGPRs read with v_readlane (two vector
//     instructions per edge, ~2 scalar): the round-2 kernel spent 7 scalar instructions per edge on slot extraction and
//     the CU issues one scalar instruction per cycle;
//   * the filter row of a group comes straight from global memory (L1/L2 hit, one 1-KB wave load per group, prefetched one
//     group ahead): the 34-KB LDS filter table is what limited the round-2 kernel to one workgroup per CU.
//
// Plan layout (per graph, shared by every convolution on it; built by tile2_plan_kernel):
//   chunk   = 32 consecutive positions of `order` (a spatial order of the cloud's output points, or the identity)
//   chdr    [B*nchunks][kHdrInts]  int : [0] tiles of the chunk; per tile t: [1+2t] = first target | targets<<8 | rows<<16,
//                                        [2+2t] = offset of the tile's row list inside the chunk's ulist slab
//   rec     [B*nchunks*32][64]     u32 : one 256-B record per target (position order): bytes 0..63 = LDS slot of each edge,
//                                        edges sorted by bin group; u16[32..95] = (bin << 8 | end of group) per group;
//                                        word 48 = edge count, 49 = groups, 50 = target id m
//   ulist   [B*nchunks][32*64]     u16 : the tiles' row lists (source point ids), one after the other
#include <cstdlib>
#include "common.hpp"

namespace sph3d {

constexpr int kT2Chunk = 64;          // targets per chunk: tiles never span chunks
constexpr int kT2MaxT = 32;           // targets per tile (two per wave of the 16-wave consumer)
constexpr int kT2HdrInts = 136;       // 1 + 2*64, padded
constexpr int kT2RecWords = 64;
constexpr int kT2UlistPerChunk = kT2Chunk * 64;

// ---------------------------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile2_plan_kernel(
    int B, int M, int N, int K, int F, int ucap, int nchunks, int W,
    const int* __restrict__ order, const int* __restrict__ nnIndex, const int* __restrict__ nnCount,
    const int* __restrict__ binIndex, int* __restrict__ chdr, unsigned* __restrict__ rec,
    unsigned short* __restrict__ ulist)
{
    extern __shared__ unsigned dyn[];                 // bitmaps: [5][W] (greedy pass + one per wave), prefix counts: [4][W]
    __shared__ int sIdx[kT2Chunk][64];
    __shared__ unsigned char sBin[kT2Chunk][64];
    __shared__ int sTm[kT2Chunk], sCnt[kT2Chunk];
    __shared__ int sTile[kT2Chunk][2];                // a word, ulist offset
    __shared__ int sNt;
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();
    const int b = (int)blockIdx.x / nchunks, c = (int)blockIdx.x % nchunks;
    const int pos0 = c * kT2Chunk;
    const int npts = (M - pos0) < kT2Chunk ? (M - pos0) : kT2Chunk;
    unsigned* bmA = dyn;
    unsigned* bmW = dyn + (size_t)(1 + wave) * W;
    unsigned* preW = dyn + (size_t)(5 + wave) * W;

    for (int i = tid; i < W; i += 256) bmA[i] = 0u;
    // the chunk's neighbour rows -> LDS (one wave per target row, lane = slot)
    for (int j = wave; j < kT2Chunk; j += 4) {
        int m = 0, cnt = 0, n = -1, f = 0;
        if (j < npts) {
            m = order ? order[(size_t)b * M + pos0 + j] : pos0 + j;
            const size_t row = (size_t)b * M + m;
            cnt = nnCount[row];
            cnt = cnt < 0 ? 0 : (cnt > K ? K : cnt);
            if (lane < cnt) {
                n = nnIndex[row * K + lane];
                f = binIndex[row * K + lane];
                f = f < 0 ? 0 : (f >= F ? F - 1 : f);      // same clamp as the gather kernels / the transposed graph
                n = n < 0 ? 0 : (n >= N ? N - 1 : n);
            }
        }
        sIdx[j][lane] = n;
        sBin[j][lane] = (unsigned char)f;
        if (lane == 0) {
            sTm[j] = m;
            sCnt[j] = cnt;
        }
    }
    __syncthreads();

    // greedy tiling (wave 0): consecutive targets while the union of their rows fits `ucap` and the tile has < 16 targets
    if (wave == 0) {
        int U = 0, T = 0, tstart = 0, nt = 0, uoff = 0;
        for (int j = 0; j < npts; j++) {
            const int n = sIdx[j][lane];
            const bool valid = n >= 0;
            unsigned word = valid ? bmA[n >> 5] : 0u;
            bool isnew = valid && !((word >> (n & 31)) & 1u);
            int cnew = __popcll(__ballot(isnew));
            if (T > 0 && (U + cnew > ucap || T == kT2MaxT)) {
                if (lane == 0) {
                    sTile[nt][0] = tstart | (T << 8) | (U << 16);
                    sTile[nt][1] = uoff;
                }
                nt++;
                uoff += U;
                for (int q = tstart; q < j; q++) {
                    const int nq = sIdx[q][lane];
                    if (nq >= 0) bmA[nq >> 5] = 0u;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                U = 0;
                T = 0;
                tstart = j;
                isnew = valid;
                cnew = __popcll(__ballot(isnew));
            }
            if (isnew) atomicOr(&bmA[n >> 5], 1u << (n & 31));
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            U += cnew;
            T++;
        }
        if (T > 0) {
            if (lane == 0) {
                sTile[nt][0] = tstart | (T << 8) | (U << 16);
                sTile[nt][1] = uoff;
            }
            nt++;
        }
        if (lane == 0) sNt = nt;
    }
    __syncthreads();
    const int nt = sNt;
    const size_t chunk = (size_t)b * nchunks + c;
    if (tid < kT2HdrInts) {
        int v = 0;
        if (tid == 0) v = nt;
        else if (tid <= 2 * kT2Chunk && (tid - 1) / 2 < nt) v = sTile[(tid - 1) / 2][(tid - 1) & 1];
        chdr[chunk * kT2HdrInts + tid] = v;
    }

    // per tile (one wave each): ranks of the union's rows = LDS slots, row list, bin-sorted slot bytes + groups
    const int WPL = (W + 63) >> 6;                    // bitmap words per lane
    for (int t = wave; t < nt; t += 4) {
        const int a = sTile[t][0], uoff = sTile[t][1];
        const int tstart = a & 0xff, T = (a >> 8) & 0xff;
        for (int i = lane; i < W; i += 64) bmW[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int q = tstart; q < tstart + T; q++) {
            const int n = sIdx[q][lane];
            if (n >= 0) atomicOr(&bmW[n >> 5], 1u << (n & 31));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // exclusive prefix popcounts: lane owns words [lane*WPL, +WPL)
        int s = 0;
        for (int i = 0; i < WPL; i++) {
            const int wi = lane * WPL + i;
            if (wi < W) s += __popc(bmW[wi]);
        }
        int incl = s;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        int run = incl - s;
        unsigned short* ul = ulist + chunk * kT2UlistPerChunk + uoff;
        for (int i = 0; i < WPL; i++) {
            const int wi = lane * WPL + i;
            if (wi < W) {
                unsigned bits = bmW[wi];
                preW[wi] = (unsigned)run;
                while (bits) {
                    const int bit = __builtin_ctz(bits);
                    bits &= bits - 1;
                    ul[run++] = (unsigned short)((wi << 5) + bit);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int q = tstart; q < tstart + T; q++) {
            const int n = sIdx[q][lane];
            const bool valid = n >= 0;
            const int f = sBin[q][lane];
            int slot = 0;
            if (valid) slot = (int)preW[n >> 5] + __popc(bmW[n >> 5] & ((1u << (n & 31)) - 1u));
            // counting sort by bin, groups in order of first occurrence, slot order inside a group = neighbour order
            unsigned long long rem = __ballot(valid);
            int pos = 0, base = 0, g = 0;
            unsigned mygrp = 0u;
            while (rem) {
                const int lead = (int)__builtin_ctzll(rem);
                const int f0 = __builtin_amdgcn_readlane(f, lead);
                const unsigned long long mk = __ballot(valid && f == f0);
                if (valid && f == f0) pos = base + prefix_popc(mk);
                base += __popcll(mk);
                if (lane == g) mygrp = ((unsigned)f0 << 8) | (unsigned)base;
                g++;
                rem &= ~mk;
            }
            unsigned* r = rec + (chunk * kT2Chunk + q) * kT2RecWords;
            unsigned char* r8 = reinterpret_cast<unsigned char*>(r);
            unsigned short* r16 = reinterpret_cast<unsigned short*>(r);
            if (valid) r8[pos] = (unsigned char)slot;
            r16[32 + lane] = (unsigned short)mygrp;                 // groups beyond g: 0
            if (lane == 0) {
                r[48] = (unsigned)sCnt[q];
                r[49] = (unsigned)g;
                r[50] = (unsigned)sTm[q];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward.  ONE persistent 16-wave workgroup per CU walks a contiguous range of chunks (XCD-affine: the clouds of an XCD
// stay in its L2).  Per tile: LDS-DMA of the tile's rows (ids fetched one tile ahead), barrier, every wave gathers up to two
// whole targets from LDS, barrier.  Measured dead ends of this round (kept in git history):
//   * 8-wave workgroups, two per CU, tiles of <= 16 targets, one chunk per workgroup, two edges per trip: 258 us at the
//     north-star shape (skeleton 36 + staging 39 + gather 190: a wave's dependent readlane -> add -> ds_read -> add chain per
//     group, four waves per SIMD, nothing overlapping);
//   * double-buffered rows + LDS filter + 2-4 waves sharing a target (partials through LDS): 372 us.  Small tiles (ucap 112
//     -> 4 targets) make the per-tile bookkeeping (~600 instructions per wave) the whole cost.
// What this version does instead: BIG tiles (up to 32 targets, ~250 rows) so that the per-tile work is amortised over
// ~1500 edges, eight LDS reads in flight per wave whatever the group structure, no divisions in the loop.
// ---------------------------------------------------------------------------------------------------------------
template <int NV>
struct Vec;
template <>
struct Vec<1> { using type = float; };
template <>
struct Vec<2> { using type = float __attribute__((ext_vector_type(2))); };
template <>
struct Vec<4> { using type = float __attribute__((ext_vector_type(4))); };

constexpr int kT2Waves = 16;

template <int R, int VEC, typename WV>
__device__ __forceinline__ void t2_flush(float (&acc)[VEC * R], typename Vec<VEC>::type s, WV wf)
{
    constexpr int NO = VEC * R;
    if constexpr (VEC == 1) {
        if constexpr (R == 1) acc[0] = fmaf(s, wf, acc[0]);
        else {
            acc[0] = fmaf(s, wf[0], acc[0]);
            acc[1] = fmaf(s, wf[1], acc[1]);
        }
    } else {
#pragma unroll
        for (int v = 0; v < NO; v++) acc[v] = fmaf(s[v / R], wf[v], acc[v]);
    }
}

// R = depth multiplier (1, 2); VEC = channels per lane: 1 (64-channel slice, 256-B rows) or 2 (128-channel slice, 512-B rows)
template <int R, int VEC>
__global__ __launch_bounds__(1024) void dwconv_tile2_fwd(
    int B, int N, int M, int F, int C, int nchunks, int nslices, int ucap,
    const int* __restrict__ chdr, const unsigned* __restrict__ rec, const unsigned short* __restrict__ ulist,
    const float* __restrict__ input, const float* __restrict__ filter, float* __restrict__ output, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];      // [ucap][SLC] rows, then [F][SLC * R] filter slice
    constexpr int SLC = 64 * VEC;
    constexpr int ROWB = SLC * 4;
    constexpr int RPI = 1024 / ROWB;              // rows per 1-KB DMA instruction (2 or 4)
    constexpr int LPR = 64 / RPI;
    constexpr int NO = VEC * R;
    using xv_t = typename Vec<VEC>::type;
    using wv_t = typename Vec<NO>::type;
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();
    const int CR = C * R;

    // ---- this workgroup's slice and range of chunks (flat index over the chunks of the XCD's clouds) ----
    const int WPX = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7;
    const int wi = (int)blockIdx.x >> 3;
    const int slice = wi % nslices;
    const int nparts = WPX / nslices;
    const int part = wi / nslices;
    const bool affine = (B & 7) == 0;
    long long total, gpart, gparts;
    if (affine) {
        total = (long long)(B >> 3) * nchunks;
        gpart = part;
        gparts = nparts;
    } else {
        total = (long long)B * nchunks;
        gpart = (long long)xcd * nparts + part;
        gparts = 8LL * nparts;
    }
    const int f_begin = (int)(total * gpart / gparts);
    const int f_end = (int)(total * (gpart + 1) / gparts);
    if (f_begin >= f_end) return;
    const int c0 = slice * SLC;
    // ---- filter slice -> LDS, once per workgroup.  (From global memory the row of the next group arrives ~800 cycles after
    // it is asked for, a group is ~6 edges = ~150 cycles of work: measured 195 us of gather time, most of it these waits.)
    float* lfilt = rows + (size_t)ucap * SLC;
    {
        const int SLo = SLC * R;
        for (int e = tid * 4; e < F * SLo; e += kT2Waves * 64 * 4) {
            const int f = e / SLo, j = e - f * SLo;
            *reinterpret_cast<float4*>(&lfilt[e]) = *reinterpret_cast<const float4*>(&filter[(size_t)f * CR + (size_t)c0 * R + j]);
        }
    }

    // ---- tile cursor: header of the current chunk in lanes (tile t -> lane t), the next chunk's prefetched ----
    int fH = f_begin;
    int clH = f_begin / nchunks, chH = f_begin - clH * nchunks;          // cloud slot, chunk inside the cloud
    auto chunk_index = [&](int cl, int ch) -> int { return (affine ? xcd + 8 * cl : cl) * nchunks + ch; };
    auto load_hdr = [&](int cl, int ch, int& hA, int& hO, int& hN) {
        const int* hp = chdr + (size_t)chunk_index(cl, ch) * kT2HdrInts;
        hN = hp[0];
        hA = hp[1 + 2 * lane];
        hO = hp[2 + 2 * lane];
    };
    int hA, hO, hN, hA2 = 0, hO2 = 0, hN2 = 0;
    load_hdr(clH, chH, hA, hO, hN);
    {
        int ch2 = chH + 1, cl2 = clH;
        if (ch2 == nchunks) { ch2 = 0; cl2++; }
        if (fH + 1 < f_end) load_hdr(cl2, ch2, hA2, hO2, hN2);
    }
    int ntH = uniform(hN);
    int tH = -1;
    struct Tile { int valid, b, chunk, a, uoff; };
    auto next_tile = [&]() -> Tile {
        Tile t;
        t.valid = 0; t.b = 0; t.chunk = 0; t.a = 0; t.uoff = 0;
        for (;;) {
            if (tH + 1 < ntH) {
                tH++;
                t.valid = 1;
                t.b = affine ? xcd + 8 * clH : clH;
                t.chunk = t.b * nchunks + chH;
                t.a = __builtin_amdgcn_readlane(hA, tH);
                t.uoff = __builtin_amdgcn_readlane(hO, tH);
                return t;
            }
            if (fH + 1 >= f_end) return t;
            fH++;
            chH++;
            if (chH == nchunks) { chH = 0; clH++; }
            hA = hA2;
            hO = hO2;
            ntH = uniform(hN2);
            tH = -1;
            int ch2 = chH + 1, cl2 = clH;
            if (ch2 == nchunks) { ch2 = 0; cl2++; }
            if (fH + 1 < f_end) load_hdr(cl2, ch2, hA2, hO2, hN2);
        }
    };
    // a word of a tile: first target | targets << 8 | rows << 16
    auto rows_per_wave = [&](int U) -> int { return ((U + kT2Waves * RPI - 1) / (kT2Waves * RPI)) * RPI; };
    auto ids_of = [&](const Tile& t) -> int {
        const int U = t.a >> 16;
        const int RW = rows_per_wave(U);
        const int r = wave * RW + lane;
        int v = 0;
        if (t.valid && lane < RW && r < U) v = ulist[(size_t)t.chunk * kT2UlistPerChunk + t.uoff + r];
        return v;
    };
    auto rec_of = [&](const Tile& t, int h) -> unsigned {
        const int tstart = t.a & 0xff, T = (t.a >> 8) & 0xff;
        const int jj = wave + kT2Waves * h;
        unsigned v = 0u;
        if (t.valid && jj < T) v = rec[((size_t)t.chunk * kT2Chunk + tstart + jj) * kT2RecWords + lane];
        return v;
    };

    Tile t0 = next_tile();
    int ids0 = ids_of(t0);
    unsigned rc0[2] = {rec_of(t0, 0), rec_of(t0, 1)};
    const char* lrow = reinterpret_cast<const char*>(rows) + lane * (VEC * 4);
    const float* fl = lfilt + lane * NO;
    constexpr int FST = SLC * R;                  // floats per filter row in LDS

    while (t0.valid) {
        const int T = (t0.a >> 8) & 0xff, U = t0.a >> 16;
        // ---- stage the tile's rows: LDS-DMA, RPI rows per wave instruction ----
        if (!(dbg & 1)) {
            const int RW = rows_per_wave(U);
            const float* inb = input + (size_t)t0.b * N * C + c0 + (lane % LPR) * 4;
            for (int j = 0; j < RW; j += RPI) {
                const int i0 = wave * RW + j;
                if (i0 >= U) break;
                int rid = __builtin_amdgcn_readlane(ids0, j);
#pragma unroll
                for (int q = 1; q < RPI; q++) {
                    const int rq = __builtin_amdgcn_readlane(ids0, j + q);     // rows past U: id 0 (a valid row, never read)
                    rid = (lane / LPR) == q ? rq : rid;
                }
                const float* gp = inb + (size_t)rid * C;
                float* lp = rows + (size_t)i0 * SLC;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
            }
        }
        // ---- in flight under the DMA: next tile's row ids and records, the first filter rows of this tile's targets ----
        Tile t1 = next_tile();
        const int ids1 = ids_of(t1);
        const unsigned rn0 = rec_of(t1, 0), rn1 = rec_of(t1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- gather from LDS: wave w takes targets w and w + 16 of the tile ----
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int jj = wave + kT2Waves * h;
            if (jj >= T || (dbg & 2)) break;
            const unsigned r0 = rc0[h];
            const int cnt = __builtin_amdgcn_readlane((int)r0, 48);
            const int m = __builtin_amdgcn_readlane((int)r0, 50);
            // lane e <- slot byte of edge e (words 0..15), lane g <- group g (u16 in words 16..47)
            const unsigned ew = (unsigned)__builtin_amdgcn_ds_bpermute((lane >> 2) << 2, (int)r0);
            const int eoff = (int)((ew >> ((lane & 3) * 8)) & 0xffu) * ROWB;
            const unsigned gw = (unsigned)__builtin_amdgcn_ds_bpermute((16 + (lane >> 1)) << 2, (int)r0);
            const int grp = (int)((gw >> ((lane & 1) * 16)) & 0xffffu);
            float acc[NO];
#pragma unroll
            for (int v = 0; v < NO; v++) acc[v] = 0.f;
            int g = 0;
            int gd = __builtin_amdgcn_readlane(grp, 0);
            int gend = gd & 0xff;
            wv_t wf = *reinterpret_cast<const wv_t*>(fl + (gd >> 8) * FST);
            int gd1 = __builtin_amdgcn_readlane(grp, 1);
            wv_t wn = *reinterpret_cast<const wv_t*>(fl + (gd1 >> 8) * FST);
            xv_t s = {};
            auto flush = [&]() {
                t2_flush<R, VEC>(acc, s, wf);
                s = xv_t{};
                wf = wn;
                gd = gd1;
                gend = gd & 0xff;
                g++;
                gd1 = __builtin_amdgcn_readlane(grp, (g + 1) & 63);          // groups past the last: 0 -> bin 0, never used
                wn = *reinterpret_cast<const wv_t*>(fl + (gd1 >> 8) * FST);
            };
            int eb = 0;
            for (; eb + 8 <= cnt; eb += 8) {                                   // full batches: no bound tests
                xv_t x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int o = __builtin_amdgcn_readlane(eoff, eb + u);
                    x[u] = *reinterpret_cast<const xv_t*>(lrow + o);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    s += x[u];
                    if (eb + u + 1 == gend) flush();
                }
            }
            if (eb < cnt) {                                                    // tail batch
                xv_t x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = (eb + u) < cnt ? (eb + u) : (cnt - 1);
                    const int o = __builtin_amdgcn_readlane(eoff, k);
                    x[u] = *reinterpret_cast<const xv_t*>(lrow + o);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (eb + u < cnt) {
                        s += x[u];
                        if (eb + u + 1 == gend) flush();
                    }
                }
            }
            const float fc = (float)cnt;
            wv_t o;
            if constexpr (NO == 1) o = cnt > 0 ? acc[0] / fc : 0.f;
            else {
#pragma unroll
                for (int v = 0; v < NO; v++) o[v] = cnt > 0 ? acc[v] / fc : 0.f;
            }
            *reinterpret_cast<wv_t*>(output + ((size_t)t0.b * M + m) * CR + (size_t)(c0 + lane * VEC) * R) = o;
        }
        __syncthreads();
        t0 = t1;
        ids0 = ids1;
        rc0[0] = rn0;
        rc0[1] = rn1;
    }
}

static int t2_dbg()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SPH3D_T2_DBG");
        v = e ? atoi(e) : 0;
    }
    return v;
}

static bool t2_shape_ok(int F, int C, int r, int K)
{
    return (r == 1 || r == 2) && K <= 64 && F <= 255 && (C == 64 || (C >= 128 && C % 128 == 0));
}

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_tile2_plan_sizes(int B, int M, size_t* hdr_ints, size_t* rec_words, size_t* ulist_shorts)
{
    const size_t nchunks = (size_t)((M + kT2Chunk - 1) / kT2Chunk);
    if (hdr_ints) *hdr_ints = (size_t)B * nchunks * kT2HdrInts;
    if (rec_words) *rec_words = (size_t)B * nchunks * kT2Chunk * kT2RecWords;
    if (ulist_shorts) *ulist_shorts = (size_t)B * nchunks * kT2UlistPerChunk + 64;
    return SPH3D_OK;
}

extern "C" int sph3d_tile2_plan(int B, int N, int M, int K, int F, int ucap, const int* order, const int* nn_index,
                                const int* nn_count, const int* bin_index, int* chunk_hdr, unsigned* records,
                                unsigned short* row_lists, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M > 0 && K > 0 && F > 0, "tile2_plan: bad dims B=%d N=%d M=%d K=%d F=%d", B, N, M, K, F);
    SPH3D_REQUIRE(K <= 64 && F <= 255 && N <= 65536, "tile2_plan: needs K <= 64, F <= 255, N <= 65536 (got K=%d F=%d N=%d)", K, F, N);
    SPH3D_REQUIRE(ucap >= 64 && ucap <= 256 && ucap % 4 == 0, "tile2_plan: ucap=%d must be a multiple of 4 in [64, 256] (>= K rows so that one target always fits; slots are bytes)", ucap);
    if (B == 0) return SPH3D_OK;
    const int nchunks = (M + kT2Chunk - 1) / kT2Chunk;
    const int W = (N + 31) >> 5;
    const size_t lds = sizeof(unsigned) * 9 * (size_t)W;
    hipStream_t st = as_stream(stream);
    if (lds > 48 * 1024) {
        int rc = check_hip(hipFuncSetAttribute((const void*)tile2_plan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "tile2_plan: hipFuncSetAttribute");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(tile2_plan_kernel, dim3(B * nchunks), dim3(256), lds, st, B, M, N, K, F, ucap, nchunks, W, order, nn_index,
                       nn_count, bin_index, chunk_hdr, records, row_lists);
    return check_launch("sph3d_tile2_plan");
}

extern "C" int sph3d_depthwise_conv3d_tiled2_supported(int F, int C, int r, int K) { return t2_shape_ok(F, C, r, K) ? 1 : 0; }

static size_t t2_lds(int F, int C, int r, int ucap)
{
    const int SLC = C == 64 ? 64 : 128;
    return sizeof(float) * ((size_t)ucap * SLC + (size_t)F * SLC * r);
}

template <int R, int VEC>
static int launch_t2(int B, int N, int M, int F, int C, int ucap, const int* chdr, const unsigned* rec, const unsigned short* ulist,
                     const float* input, const float* filter, float* output, hipStream_t st)
{
    const size_t lds = t2_lds(F, C, R, ucap);
    auto kern = dwconv_tile2_fwd<R, VEC>;
    if (lds > 48 * 1024) {
        int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "DepthwiseConv3dTiled2: hipFuncSetAttribute");
        if (rc) return rc;
    }
    const int nchunks = (M + kT2Chunk - 1) / kT2Chunk;
    const int nslices = C / (64 * VEC);
    // one persistent workgroup per CU (rows of 512 B: ucap = 288 fills 144 KB of LDS); narrower rows leave room for two
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    hipLaunchKernelGGL(kern, dim3(256 * per_cu), dim3(1024), lds, st, B, N, M, F, C, nchunks, nslices, ucap, chdr, rec, ulist, input,
                       filter, output, t2_dbg());
    return check_launch("sph3d_depthwise_conv3d_tiled2");
}

extern "C" int sph3d_depthwise_conv3d_tiled2(int B, int N, int M, int F, int C, int r, int ucap, const int* chunk_hdr,
                                             const unsigned* records, const unsigned short* row_lists, const float* input,
                                             const float* filter, float* output, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M > 0 && F > 0 && C > 0, "DepthwiseConv3dTiled2: bad dims B=%d N=%d M=%d F=%d C=%d", B, N, M, F, C);
    SPH3D_REQUIRE(t2_shape_ok(F, C, r, 64), "DepthwiseConv3dTiled2: needs r in {1,2}, C = 64 or a multiple of 128, F <= 255 (got r=%d C=%d F=%d)",
                  r, C, F);
    SPH3D_REQUIRE(ucap >= 64 && ucap <= 256 && ucap % 4 == 0, "DepthwiseConv3dTiled2: ucap=%d out of range", ucap);
    SPH3D_REQUIRE((long long)N * C < (1LL << 31), "DepthwiseConv3dTiled2: N*C overflows int32");
    if (B == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    const int nsl = C == 64 ? 1 : C / 128;
    SPH3D_REQUIRE(nsl <= 32 && (32 % nsl) == 0, "DepthwiseConv3dTiled2: %d channel slices do not divide the 32 workgroups of an XCD", nsl);
    SPH3D_REQUIRE(t2_lds(F, C, r, ucap) <= 160 * 1024, "DepthwiseConv3dTiled2: ucap=%d rows + the filter slice do not fit the LDS", ucap);
    if (C == 64)
        return r == 2 ? launch_t2<2, 1>(B, N, M, F, C, ucap, chunk_hdr, records, row_lists, input, filter, output, st)
                      : launch_t2<1, 1>(B, N, M, F, C, ucap, chunk_hdr, records, row_lists, input, filter, output, st);
    return r == 2 ? launch_t2<2, 2>(B, N, M, F, C, ucap, chunk_hdr, records, row_lists, input, filter, output, st)
                  : launch_t2<1, 2>(B, N, M, F, C, ucap, chunk_hdr, records, row_lists, input, filter, output, st);
}
