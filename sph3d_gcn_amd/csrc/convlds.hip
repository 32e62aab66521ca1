// convlds.hip — depthwise spherical convolution gathered from LDS tiles (forward), gfx950.
//
// Replaces depthwise_conv3d_forward (tf_ops/convolution/tf_conv3d_gpu.cu:7-29): out = 1/cnt * sum_k in[nn_k] * filt[bin_k], fp32,
// summed bin group by bin group ((x_a + x_b) * w per pair of edges of a bin; the reference: edge by edge in neighbour order):
// ~1e-7 relative to the oracle, bound 1e-5.  What changes against the gather kernels of conv3d.hip is where the rows come from.
//
// Why.  The gather kernels fetch one feature row per edge through the CU's vector L1: 6.3 M edges x 512 B = 3.2 GB at
// level 0 of the S3DIS plan, served L2 -> L1 at <= 25 TB/s (tools/micro/gather_bw.hip): 129 us before a single FMA, 196 us
// measured.  The LDS moves 256 B/clk/CU (~150 TB/s over the chip).  Spatially close output points share neighbours: 32
// Morton-consecutive points of an S3DIS-like block reference ~250 distinct rows for ~1540 edges (6.2x reuse), so a workgroup
// that stages the UNION of a tile's rows once and gathers from LDS reads 0.5 GB through the L1 instead of 3.2.
//
// Rounds 2-3 built five LDS-tiled kernels on that observation (convtile.hip, tile2.hip: 207-372 us, all slower than the
// gather kernel end to end).  Their common shape was one wave per target, edges grouped by bin, slot bytes extracted on the
// scalar unit, a flush branch per group: 15.5 wave instructions per edge, most of them bookkeeping, and a dependent
// scalar -> vector -> LDS chain per edge that 16 waves per CU could not hide.  This kernel has NO per-edge scalar work and no
// branches inside a target:
//   * a channel SLICE is 64 input channels = 256-B rows; a lane owns 4 input channels (one ds_read_b128 per edge) and their
//     4r outputs, so a target needs 16 lanes and a wave carries FOUR targets in lockstep (targets of a tile are dealt to
//     waves in order of neighbour count, so the four of a wave finish together);
//   * per target the plan holds a RECORD: its edges sorted by bin and PAIRED inside a bin group (an odd group's last edge
//     pairs with an all-zero LDS row), 48 pair entries {slot A | slot B << 16, bin}.  Records of a tile travel to LDS with
//     the rows; a lane reads 4 entries with two ds_read_b128 and turns an entry into two row addresses and the filter address
//     with one v_perm_b32 each ((slot << 8) | lane byte, (bin << 8) | lane byte: rows and filter planes are 256 B apart by
//     construction, region bases are immediate offsets of the ds_read);
//   * per pair and wave: 3 v_perm + 2 row reads + r filter reads + 2 packed adds + 2r packed FMAs, nothing else: the filter
//     row — two thirds of the LDS bytes of an edge at r = 2 — is read once per pair (measured with tools/micro/lds_fma.hip:
//     a b128 read costs 2.1 ns per CU and overlaps with the FMAs, so the reads per edge ARE the kernel's floor);
//   * rows travel global -> LDS by LDS-DMA (global_load_lds_dwordx4), no registers; two 8-wave workgroups per CU (80 KB of
//     LDS each), so one stages while the other gathers — no software pipeline inside a workgroup.
// The plan is per GRAPH (every convolution on the graph and its channel slices share it), built by one kernel after the
// neighbour search: greedy tiles of <= 64 spatially consecutive targets whose row union fits the LDS, union ranks by bitmap +
// prefix popcounts, bin-sorted pair records.
#include <cstdlib>
#include "common.hpp"

namespace sph3d {

constexpr int kLcChunk = 128;           // consecutive positions of the spatial order handled by one plan workgroup; tiles never span chunks

constexpr int kLcHdrInts = 132;         // per chunk: [0] tiles, [1 + 2t] first | targets << 8 | rows << 16, [2 + 2t] row-list offset
constexpr int kLcPairs = 48;            // pair entries per target: <= (64 edges + 32 odd bin groups) / 2
constexpr int kLcRecWords = 2 * kLcPairs;   // a pair entry = two words: slot A | slot B << 16, bin
constexpr int kLcRowBytes = 256;        // one 64-channel slice of a feature row
constexpr int kLcRowsPerChunk = kLcChunk * 64;

// LDS map of the consumer (bytes): filter planes | records | rows.  PR = filter rows per plane (F + 1 <= PR), NW = waves per
// workgroup (16: one workgroup per CU with 160 KB, tiles of <= 64 targets; 8: two per CU with 80 KB each, <= 32 targets).
// Region bases are immediate offsets of the ds_read (< 64 KB); the address register holds (slot or bin) << 8 | lane byte.
template <int R, int PR, int NW>
struct LcMap {
    static constexpr int kLds = NW == 16 ? 160 * 1024 : 80 * 1024;
    static constexpr int kRecBytes = 4 * NW * kLcRecWords * 4;
    static constexpr int RECB = R * PR * 256;            // filter plane q at q * PR * 256
    static constexpr int RB = RECB + kRecBytes;          // rows
    static constexpr int kFit = (((kLds - RB) / 256) - 1) & ~3;
    static constexpr int UCAP = kFit > 1020 ? 1020 : kFit;     // slot UCAP is the zero row
    static constexpr int TOTAL = RB + (UCAP + 1) * 256;
    static_assert(UCAP >= 64 && TOTAL <= kLds && RB < 65536, "LDS map");
};

// rows a tile may stage for a graph with F bins: the smaller of the r = 1, 2 capacities (a plan serves both)
static int lc_waves()
{
    static int v = 0;
    if (!v) {
        const char* e = getenv("SPH3D_LC_WAVES");
        v = (e && atoi(e) == 8) ? 8 : 16;
    }
    return v;
}
static int lc_plan_ucap(int F)
{
    const bool w16 = lc_waves() == 16;
    if (F + 1 <= 34) return w16 ? LcMap<2, 34, 16>::UCAP : LcMap<2, 34, 8>::UCAP;
    if (F + 1 <= 66) return w16 ? LcMap<2, 66, 16>::UCAP : LcMap<2, 66, 8>::UCAP;
    return 0;
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

// ---------------------------------------------------------------------------------------------------------------
// plan.  One 256-thread workgroup per chunk of 128 consecutive positions of `order` (identity when null).
//   1. the chunk's neighbour rows -> LDS (one wave per target row, lane = slot);
//   2. wave 0 walks the targets once and cuts tiles greedily: a tile takes consecutive targets while the union of
//      their source rows fits `ucap` and it has < maxT targets (an LDS bitmap of the cloud tells new rows from known ones);
//   3. one wave per tile: bitmap of the tile's rows -> exclusive prefix popcounts -> rank of a row = its LDS slot;
//      row list (ascending row id); per target the RECORD: its edges sorted by bin and paired inside a bin group, 48 pair
//      entries {slot A | slot B << 16, bin} (an odd group's last edge pairs with the zero row = slot ucap; unused entries:
//      zero row twice and the zero filter row F) and meta = {target id, count | pairs << 8}; targets ranked by pair count
//      (descending, ties by position).
// Outputs (all addressed from the chunk index):
//   chdr    [B*nchunks][132]     int : see kLcHdrInts
//   rec     [B*nchunks*128][96]  u32 : records, a tile's targets contiguous from `first`, in rank order
//   tmeta   [B*nchunks*128][2]   int
//   rowlist [B*nchunks][128*64]  u16 : the tiles' row lists one after the other
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lc_plan_kernel(
    int B, int M, int N, int K, int F, int ucap, int maxT, int nchunks, int W,
    const int* __restrict__ order, const int* __restrict__ nnIndex, const int* __restrict__ nnCount,
    const int* __restrict__ binIndex, int* __restrict__ chdr, unsigned* __restrict__ rec, int* __restrict__ tmeta,
    unsigned short* __restrict__ rowlist)
{
    extern __shared__ unsigned dyn[];                 // bitmaps: [5][W] (greedy pass + one per wave), prefix counts: [4][W]
    __shared__ unsigned short sIdx[kLcChunk][64];
    __shared__ unsigned char sBin[kLcChunk][64];
    __shared__ int sTm[kLcChunk], sCnt[kLcChunk];
    __shared__ int sTile[kLcChunk];                   // first | targets << 8
    __shared__ int sNt;
    __shared__ unsigned short sPair[4][kLcPairs * 4]; // per wave: the record under construction
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();
    const int b = (int)blockIdx.x / nchunks, c = (int)blockIdx.x % nchunks;
    const int pos0 = c * kLcChunk;
    const int npts = (M - pos0) < kLcChunk ? (M - pos0) : kLcChunk;
    unsigned* bmA = dyn;
    unsigned* bmW = dyn + (size_t)(1 + wave) * W;
    unsigned* preW = dyn + (size_t)(5 + wave) * W;

    for (int i = tid; i < W; i += 256) bmA[i] = 0u;
    for (int j = wave; j < kLcChunk; j += 4) {
        int m = 0, cnt = 0, n = 0, f = 0;
        if (j < npts) {
            m = order ? order[(size_t)b * M + pos0 + j] : pos0 + j;
            const size_t row = (size_t)b * M + m;
            cnt = nnCount[row];
            cnt = cnt < 0 ? 0 : (cnt > K ? K : cnt);
            cnt = cnt > 64 ? 64 : cnt;
            if (lane < cnt) {
                n = nnIndex[row * K + lane];
                f = binIndex[row * K + lane];
                f = f < 0 ? 0 : (f >= F ? F - 1 : f);      // same clamp as the gather kernels / the transposed graph
                n = n < 0 ? 0 : (n >= N ? N - 1 : n);
            }
        }
        sIdx[j][lane] = (unsigned short)n;
        sBin[j][lane] = (unsigned char)f;
        if (lane == 0) {
            sTm[j] = m;
            sCnt[j] = cnt;
        }
    }
    __syncthreads();

    if (wave == 0) {
        int U = 0, T = 0, tstart = 0, nt = 0;
        for (int j = 0; j < npts; j++) {
            const int cj = sCnt[j];
            const bool valid = lane < cj;
            const int n = sIdx[j][lane];
            const unsigned word = valid ? bmA[n >> 5] : 0u;
            bool isnew = valid && !((word >> (n & 31)) & 1u);
            int cnew = __popcll(__ballot(isnew));          // an upper bound when a row lists a neighbour twice
            if (T > 0 && (U + cnew > ucap || T == maxT)) {
                if (lane == 0) sTile[nt] = tstart | (T << 8);
                nt++;
                for (int q = tstart; q < j; q++)
                    if (lane < sCnt[q]) bmA[sIdx[q][lane] >> 5] = 0u;
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
            if (lane == 0) sTile[nt] = tstart | (T << 8);
            nt++;
        }
        if (lane == 0) sNt = nt;
    }
    __syncthreads();
    const int nt = sNt;                               // <= 64: a tile holds >= 2 targets (ucap >= 128 rows, <= 64 rows per target)
    const size_t chunk = (size_t)b * nchunks + c;
    int* hp = chdr + chunk * kLcHdrInts;
    if (tid == 0) hp[0] = nt;
    if (tid > 2 * nt && tid < kLcHdrInts) hp[tid] = 0;

    const int WPL = (W + 63) >> 6;                    // bitmap words per lane
    for (int t = wave; t < nt; t += 4) {
        const int a = sTile[t];
        const int tstart = a & 0xff, T = (a >> 8) & 0xff;
        // slab offset of the tile's row list: a tile's rows never exceed the sum of its targets' counts, so the sum of the
        // counts of the targets before it is a valid (and deterministic) offset
        int uoff = 0;
        for (int q = lane; q < tstart; q += 64) uoff += sCnt[q];
        for (int o = 32; o > 0; o >>= 1) uoff += __shfl_xor(uoff, o);
        for (int i = lane; i < W; i += 64) bmW[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int q = tstart; q < tstart + T; q++)
            if (lane < sCnt[q]) {
                const int n = sIdx[q][lane];
                atomicOr(&bmW[n >> 5], 1u << (n & 31));
            }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // exclusive prefix popcounts: lane owns words [lane*WPL, +WPL)
        int s = 0;
        for (int i = 0; i < WPL; i++) {
            const int wi = lane * WPL + i;
            s += wi < W ? __popc(bmW[wi]) : 0;
        }
        int incl = s;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        const int U = __builtin_amdgcn_readlane(incl, 63);
        int run = incl - s;
        unsigned short* ul = rowlist + chunk * kLcRowsPerChunk + uoff;
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
        if (lane == 0) {
            hp[1 + 2 * t] = tstart | (T << 8) | (U << 16);
            hp[2 + 2 * t] = uoff;
        }
        // a target's edges are sorted by bin and PAIRED inside a bin group (an odd group's last edge pairs with the zero
        // row): the consumer reads one filter row per pair.  Pairs per target: sum over groups of ceil(n / 2) <= 48.
        // rank of target tstart + lane among the tile's targets: more PAIRS first, ties by position
        int mypairs = -1;
        for (int qq = 0; qq < T; qq++) {
            const int q = tstart + qq;
            const int cq = sCnt[q];
            const bool valid = lane < cq;
            const int f = sBin[q][lane];
            unsigned long long rem = __ballot(valid);
            int np = 0;
            while (rem) {
                const int f0 = __builtin_amdgcn_readlane(f, (int)__builtin_ctzll(rem));
                const unsigned long long mk = __ballot(valid && f == f0);
                np += (__popcll(mk) + 1) >> 1;
                rem &= ~mk;
            }
            if (lane == qq) mypairs = np;
        }
        int rank = 0;
        for (int i = 0; i < T; i++) {
            const int ci = __builtin_amdgcn_readlane(mypairs, i);
            rank += (ci > mypairs || (ci == mypairs && i < lane)) ? 1 : 0;
        }
        unsigned short* sp = sPair[wave];
        for (int qq = 0; qq < T; qq++) {
            const int q = tstart + qq;
            const int cq = sCnt[q];
            const int dst = tstart + __builtin_amdgcn_readlane(rank, qq);
            const bool valid = lane < cq;
            const int f = sBin[q][lane];
            int slot = ucap;
            if (valid) {
                const int n = sIdx[q][lane];
                slot = (int)preW[n >> 5] + __popc(bmW[n >> 5] & ((1u << (n & 31)) - 1u));
            }
            // default entries: zero row twice, zero filter row
            for (int i = lane; i < kLcPairs * 4; i += 64) sp[i] = (unsigned short)((i & 3) < 2 ? ucap : ((i & 3) == 2 ? F : 0));
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            unsigned long long rem = __ballot(valid);
            int pbase = 0;
            while (rem) {
                const int f0 = __builtin_amdgcn_readlane(f, (int)__builtin_ctzll(rem));
                const unsigned long long mk = __ballot(valid && f == f0);
                if (valid && f == f0) {
                    const int i = prefix_popc(mk);
                    const int pr = pbase + (i >> 1);
                    sp[pr * 4 + (i & 1)] = (unsigned short)slot;
                    if ((i & 1) == 0) sp[pr * 4 + 2] = (unsigned short)f0;
                }
                pbase += (__popcll(mk) + 1) >> 1;
                rem &= ~mk;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            unsigned* rp = rec + (chunk * kLcChunk + dst) * kLcRecWords;
            const unsigned* sw = reinterpret_cast<const unsigned*>(sp);
            rp[lane] = sw[lane];
            if (lane < kLcRecWords - 64) rp[64 + lane] = sw[64 + lane];
            if (lane == 0) {
                tmeta[(chunk * kLcChunk + dst) * 2] = sTm[q];
                tmeta[(chunk * kLcChunk + dst) * 2 + 1] = cq | (pbase << 8);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef SPH3D_LC_PERM
#define SPH3D_LC_PERM 1
#endif
// pair entry {slot A | slot B << 16, bin} -> (slot << 8) | lane byte, (bin << 8) | lane byte.  v_perm_b32: selector bytes
// 0-3 take bytes of the second operand, 4-7 of the first, 0x0c is a zero byte
__device__ __forceinline__ unsigned lc_arow_a(unsigned w0, unsigned lb)
{
#if SPH3D_LC_PERM
    return __builtin_amdgcn_perm(w0, lb, 0x0c050400u);
#else
    return ((w0 & 0xffffu) << 8) | lb;
#endif
}
__device__ __forceinline__ unsigned lc_arow_b(unsigned w0, unsigned lb)
{
#if SPH3D_LC_PERM
    return __builtin_amdgcn_perm(w0, lb, 0x0c070600u);
#else
    return ((w0 >> 16) << 8) | lb;
#endif
}
__device__ __forceinline__ unsigned lc_afil(unsigned w1, unsigned lb)
{
#if SPH3D_LC_PERM
    return __builtin_amdgcn_perm(w1, lb, 0x0c0c0400u);
#else
    return ((w1 & 0xffu) << 8) | lb;
#endif
}

// acc += {x[H], x[H]} * w  (H = 0: low half of x broadcast, 1: high half): one v_pk_fma_f32
template <int H>
__device__ __forceinline__ void lc_fma(f32x2& acc, f32x2 x, f32x2 w)
{
    if constexpr (H == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(x), "v"(w));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(w));
}

// LDS read at a raw byte address (the kernel has no static LDS: the dynamic block starts at 0), so that region bases fold into
// the instruction's immediate offset instead of an add per read
template <typename T>
__device__ __forceinline__ T lc_ld(unsigned addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) T*>((size_t)addr);
}

// workgroup barrier WITHOUT the release fence of __syncthreads(): that fence waits for the wave's global stores (vmcnt(0)), i.e.
// every tile's output rows would have to reach the L2 before its LDS rows may be replaced (measured: 4000 of 14700 cycles per
// tile).  Only LDS traffic is ordered here: the wave's own LDS operations are complete, then the barrier.
__device__ __forceinline__ void lc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct LcTile {
    int valid, b, chunk, slice, a, uoff;
};

// R = depth multiplier, PR = filter rows per LDS plane (>= F + 1), NW = waves per workgroup.  The input may be the channel
// concatenation of two tensors [input (Ca channels) | input2 (C - Ca)] that was never materialised (input2 == nullptr: one
// tensor); Ca % 64 == 0, so a slice lies inside one of them.
template <int R, int PR, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void dwconv_fwd_lds(
    int B, int N, int M, int F, int C, int nchunks, int nslices, int ucap,
    const int* __restrict__ chdr, const unsigned* __restrict__ rec, const int* __restrict__ tmeta,
    const unsigned short* __restrict__ rowlist,
    const float* __restrict__ input, const float* __restrict__ input2, int Ca,
    const float* __restrict__ filter, float* __restrict__ output, int dbg)
{
    using Map = LcMap<R, PR, NW>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();
    const int l16 = lane & 15;
    const int CR = C * R;

    // ---- work of this workgroup: one channel slice and, of every cloud of its XCD in turn, a range of chunks (the workgroups
    // of an XCD walk the XCD's clouds one after the other).  Measured alternative: all workgroups of an XCD on ONE slice at a
    // time (2 MB of rows per XCD and phase instead of 4 MB, the filter slice re-staged per phase): 166-172 us instead of 154 at
    // level 0 of the S3DIS plan, C = 128 — the staging is not bound by L2 misses (an explicit L2 warm-up of the next tile's rows
    // did not help either: 165 us) but by the LDS-DMA path, ~12 B/clk per CU. ----
    const int WPX = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7;
    const int wi = (int)blockIdx.x >> 3;
    const bool affine = (B & 7) == 0;
    const int nclouds = affine ? (B >> 3) : B;
    int sgroups = WPX < nslices ? WPX : nslices;                      // workgroup groups of the XCD, each on its own slices
    const int nparts_x = WPX / sgroups;                               // workgroups of this XCD sharing a slice
    const int sg = wi % sgroups, part_x = wi / sgroups;
    if (part_x >= nparts_x) return;
    const long long gpart = affine ? part_x : (long long)xcd * nparts_x + part_x;
    const long long gparts = affine ? nparts_x : 8LL * nparts_x;
    const int nsg = sgroups;
    const int ch_begin = (int)((long long)nchunks * gpart / gparts);
    const int span = (int)((long long)nchunks * (gpart + 1) / gparts) - ch_begin;
    const int nsl = (nslices - sg + nsg - 1) / nsg;                    // slices of this group: sg, sg + nsg, ...
    const int f_end = nclouds * nsl * span;
    if (f_end <= 0) return;

    // ---- tile cursor: header of the current chunk in lanes (tile t -> lane t), the next chunk's prefetched ----
    int fH = 0;
    auto chunk_of = [&](int f, int& b, int& chunk, int& slice) {
        const int si = f / (nclouds * span), rem = f - si * (nclouds * span);
        const int ci = rem / span, ch = ch_begin + (rem - ci * span);
        b = affine ? xcd + 8 * ci : ci;
        chunk = b * nchunks + ch;
        slice = sg + si * nsg;
    };
    auto load_hdr = [&](int f, int& hA, int& hO, int& hN) {
        int b, chunk, slice;
        chunk_of(f, b, chunk, slice);
        const int* hp = chdr + (size_t)chunk * kLcHdrInts;
        hN = hp[0];
        hA = hp[1 + 2 * lane];
        hO = hp[2 + 2 * lane];
    };
    int hA, hO, hN, hA2 = 0, hO2 = 0, hN2 = 0;
    load_hdr(0, hA, hO, hN);
    if (1 < f_end) load_hdr(1, hA2, hO2, hN2);
    int ntH = uniform(hN);
    int tH = -1;
    auto next_tile = [&]() -> LcTile {
        LcTile t;
        t.valid = 0; t.b = 0; t.chunk = 0; t.slice = 0; t.a = 0; t.uoff = 0;
        for (;;) {
            if (tH + 1 < ntH) {
                tH++;
                t.valid = 1;
                chunk_of(fH, t.b, t.chunk, t.slice);
                t.a = __builtin_amdgcn_readlane(hA, tH);
                t.uoff = __builtin_amdgcn_readlane(hO, tH);
                return t;
            }
            if (fH + 1 >= f_end) return t;
            fH++;
            hA = hA2;
            hO = hO2;
            ntH = uniform(hN2);
            tH = -1;
            if (fH + 1 < f_end) load_hdr(fH + 1, hA2, hO2, hN2);
        }
    };
    // rows a wave stages: a contiguous block of RW = 4 * ceil(U / (4 NW)) rows, four per DMA instruction
    auto ids_of = [&](const LcTile& t) -> int {
        const int U = t.a >> 16;
        const int RW = ((U + 4 * NW - 1) / (4 * NW)) << 2;
        int r = wave * RW + lane;
        r = r < U ? r : U - 1;
        int v = 0;
        if (t.valid && lane < RW && U > 0) v = rowlist[(size_t)t.chunk * kLcRowsPerChunk + t.uoff + r];
        return v;
    };
    // lanes of quarter q: {target id, count | pairs << 8} of target 4 * wave + q of the tile
    auto meta_of = [&](const LcTile& t, int& m, int& cp) {
        const int tstart = t.a & 0xff, T = (t.a >> 8) & 0xff;
        const int rho = 4 * wave + (lane >> 4);
        m = 0;
        cp = 0;
        if (t.valid && rho < T) {
            const int2 v = *reinterpret_cast<const int2*>(tmeta + ((size_t)t.chunk * kLcChunk + tstart + rho) * 2);
            m = v.x;
            cp = v.y;
        }
    };
    // the filter slice -> LDS planes (plane q: outputs 4q..4q+3 of every lane) and the zero rows
    auto stage_filter = [&](int slice) {
        const int c0 = slice * 64;
        float* lf = reinterpret_cast<float*>(lds);
        for (int e = tid; e < F * 16 * R; e += 64 * NW) {
            const int f = e / (16 * R), rem = e - f * (16 * R);
            const int l = rem / R, q = rem - l * R;
            const f32x4 v = *reinterpret_cast<const f32x4*>(filter + (size_t)f * CR + (size_t)(c0 + l * 4) * R + q * 4);
            *reinterpret_cast<f32x4*>(lf + (size_t)q * PR * 64 + f * 64 + l * 4) = v;
        }
        for (int e = tid; e < 64 * R; e += 64 * NW) {
            const int q = e >> 6, j = e & 63;
            lf[(size_t)q * PR * 64 + F * 64 + j] = 0.f;
        }
        for (int e = tid; e < 64; e += 64 * NW) reinterpret_cast<float*>(lds + Map::RB)[(size_t)ucap * 64 + e] = 0.f;
    };
    // a tile's rows (LDS-DMA, four 256-B rows per wave instruction) and every wave's own four records
    auto stage_tile = [&](const LcTile& t, int ids) {
        const int tstart = t.a & 0xff, T = (t.a >> 8) & 0xff, U = t.a >> 16;
        const int RW = ((U + 4 * NW - 1) / (4 * NW)) << 2;
        const int c0 = t.slice * 64;
        const bool second = input2 != nullptr && c0 >= Ca;
        const int Cs = input2 == nullptr ? C : (second ? C - Ca : Ca);        // row stride of the source tensor
        const float* inb = (second ? input2 : input) + (c0 - (second ? Ca : 0)) + l16 * 4 + (size_t)t.b * N * Cs;
        for (int j = 0; j < RW; j += 4) {
            const int i0 = wave * RW + j;
            if (i0 >= U) break;
            const int rid = __shfl(ids, j + (lane >> 4));
            const float* gp = inb + (size_t)rid * Cs;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(lds + Map::RB + (size_t)i0 * kLcRowBytes), 16, 0, 0);
        }
        if (wave * 4 < T) {
            // four records of 384 B: one full DMA instruction and one of 32 lanes
            const unsigned* gp = rec + ((size_t)t.chunk * kLcChunk + tstart + wave * 4) * kLcRecWords + lane * 4;
            char* lp = lds + Map::RECB + wave * (4 * kLcRecWords * 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
            if (lane < 32)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + 256),
                                                 (__attribute__((address_space(3))) void*)(lp + 1024), 16, 0, 0);
        }
    };

    LcTile t0 = next_tile();
    int ids0 = ids_of(t0);
    int m0, cp0;
    meta_of(t0, m0, cp0);
    const unsigned lb = (unsigned)l16 << 4;
    const int rho = 4 * wave + (lane >> 4);
    const unsigned recaddr0 = (unsigned)(Map::RECB + rho * (kLcRecWords * 4));
    int cur_slice = t0.slice;
    stage_filter(cur_slice);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(dbg & 1)) stage_tile(t0, ids0);

    while (t0.valid) {
        const int T = (t0.a >> 8) & 0xff;
        // ---- in flight under the DMA and the gather: the next tile's row ids and meta ----
        LcTile t1 = next_tile();
        const int ids1 = ids_of(t1);
        int m1, cp1;
        meta_of(t1, m1, cp1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lc_barrier();
        // ---- gather: quarter-wave q of wave w takes target 4w + q of the tile (rank order: the four finish together) ----
        if (4 * wave < T && !(dbg & 2)) {
            const int cnt = cp0 & 0xff;
            int nmax = cp0 >> 8;                         // pairs of this quarter's target
            nmax = max(nmax, __shfl_xor(nmax, 16));
            nmax = max(nmax, __shfl_xor(nmax, 32));
            nmax = uniform(nmax);
            f32x2 acc[2 * R];
#pragma unroll
            for (int v = 0; v < 2 * R; v++) acc[v] = f32x2{0.f, 0.f};
            unsigned ra = recaddr0;
            // software pipeline over the pairs: two pairs (8 reads) in flight, the reads of pair p + 2 are issued right after
            // the FMAs of pair p; the entries of the next four pairs are requested a trip ahead.  (Reads past the record's
            // 48 entries fetch the next record or row bytes and form addresses nobody uses: pb + 4 >= nmax ends the loop.)
            f32x4 xa[2], xb[2], w0[2], w1[2];
            auto load_pair = [&](int i, unsigned e_slots, unsigned e_bin) {
                const unsigned aa = lc_arow_a(e_slots, lb);
                const unsigned ab = lc_arow_b(e_slots, lb);
                const unsigned af = lc_afil(e_bin, lb);
                xa[i] = lc_ld<f32x4>(aa + Map::RB);
                xb[i] = lc_ld<f32x4>(ab + Map::RB);
                w0[i] = lc_ld<f32x4>(af);
                if constexpr (R == 2) w1[i] = lc_ld<f32x4>(af + PR * 256);
            };
            auto fma_pair = [&](int i) {
                const f32x4 sx = xa[i] + xb[i];
                const f32x2 x01 = {sx[0], sx[1]}, x23 = {sx[2], sx[3]};
                if constexpr (R == 2) {
                    lc_fma<0>(acc[0], x01, f32x2{w0[i][0], w0[i][1]});
                    lc_fma<1>(acc[1], x01, f32x2{w0[i][2], w0[i][3]});
                    lc_fma<0>(acc[2], x23, f32x2{w1[i][0], w1[i][1]});
                    lc_fma<1>(acc[3], x23, f32x2{w1[i][2], w1[i][3]});
                } else {
                    acc[0] = __builtin_elementwise_fma(x01, f32x2{w0[i][0], w0[i][1]}, acc[0]);
                    acc[1] = __builtin_elementwise_fma(x23, f32x2{w0[i][2], w0[i][3]}, acc[1]);
                }
            };
            u32x4 ea = lc_ld<u32x4>(ra), eb = lc_ld<u32x4>(ra + 16);
            load_pair(0, ea[0], ea[1]);
            load_pair(1, ea[2], ea[3]);
            for (int pb = 0; pb < nmax; pb += 4) {
                ra += 32;
                const u32x4 na = lc_ld<u32x4>(ra), nb = lc_ld<u32x4>(ra + 16);
                __builtin_amdgcn_sched_barrier(0);
                fma_pair(0);
                load_pair(0, eb[0], eb[1]);
                __builtin_amdgcn_sched_barrier(0);
                fma_pair(1);
                load_pair(1, eb[2], eb[3]);
                __builtin_amdgcn_sched_barrier(0);
                fma_pair(0);
                load_pair(0, na[0], na[1]);
                __builtin_amdgcn_sched_barrier(0);
                fma_pair(1);
                load_pair(1, na[2], na[3]);
                __builtin_amdgcn_sched_barrier(0);
                ea = na;
                eb = nb;
            }
            if (rho < T) {
                const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
                float* op = output + ((size_t)t0.b * M + m0) * CR + (size_t)(t0.slice * 64 + l16 * 4) * R;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    f32x4 o = {acc[2 * q][0] * inv, acc[2 * q][1] * inv, acc[2 * q + 1][0] * inv, acc[2 * q + 1][1] * inv};
                    *reinterpret_cast<f32x4*>(op + 4 * q) = o;
                }
            }
        }
        lc_barrier();
        // ---- the next tile's rows and records (and filter slice, when the slice changes) ----
        if (t1.valid) {
            if (t1.slice != cur_slice) {
                cur_slice = t1.slice;
                stage_filter(cur_slice);
            }
            if (!(dbg & 1)) stage_tile(t1, ids1);
        }
        t0 = t1;
        m0 = m1;
        cp0 = cp1;
    }
}

static int lc_dbg()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SPH3D_LC_DBG");
        v = e ? atoi(e) : 0;
    }
    return v;
}

static bool lc_shape_ok(int F, int C, int r, int K)
{
    return (r == 1 || r == 2) && K >= 1 && K <= 64 && F >= 1 && F + 1 <= 66 && C >= 64 && C % 64 == 0;
}

template <int R, int PR, int NW>
static int launch_lc(int B, int N, int M, int F, int C, int ucap, const int* chdr, const unsigned* rec, const int* tmeta,
                     const unsigned short* rowlist, const float* input, const float* input2, int Ca, const float* filter,
                     float* output, hipStream_t st)
{
    using Map = LcMap<R, PR, NW>;
    SPH3D_REQUIRE(ucap <= Map::UCAP, "DepthwiseConv3dLds: the plan stages %d rows, the kernel holds %d", ucap, Map::UCAP);
    auto kern = dwconv_fwd_lds<R, PR, NW>;
    static bool attr_done = false;
    if (!attr_done) {
        int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Map::TOTAL),
                           "DepthwiseConv3dLds: hipFuncSetAttribute");
        if (rc) return rc;
        attr_done = true;
    }
    const int nchunks = (M + kLcChunk - 1) / kLcChunk;
    const int nslices = C / 64;
    // one (NW = 16) or two (NW = 8) workgroups per CU
    const int wpx = NW == 16 ? 32 : 64;
    hipLaunchKernelGGL(kern, dim3(8 * wpx), dim3(64 * NW), Map::TOTAL, st, B, N, M, F, C, nchunks, nslices, ucap, chdr, rec,
                       tmeta, rowlist, input, input2, Ca, filter, output, lc_dbg());
    return check_launch("sph3d_depthwise_conv3d_lds");
}

template <int NW>
static int lc_forward_nw(int B, int N, int M, int F, int C, int r, int ucap, const int* chdr, const unsigned* rec, const int* tmeta,
                         const unsigned short* rowlist, const float* input, const float* input2, int Ca, const float* filter,
                         float* output, hipStream_t st)
{
    if (F + 1 <= 34)
        return r == 2 ? launch_lc<2, 34, NW>(B, N, M, F, C, ucap, chdr, rec, tmeta, rowlist, input, input2, Ca, filter, output, st)
                      : launch_lc<1, 34, NW>(B, N, M, F, C, ucap, chdr, rec, tmeta, rowlist, input, input2, Ca, filter, output, st);
    return r == 2 ? launch_lc<2, 66, NW>(B, N, M, F, C, ucap, chdr, rec, tmeta, rowlist, input, input2, Ca, filter, output, st)
                  : launch_lc<1, 66, NW>(B, N, M, F, C, ucap, chdr, rec, tmeta, rowlist, input, input2, Ca, filter, output, st);
}

static int lc_forward(int B, int N, int M, int F, int C, int r, const int* chdr, const unsigned* rec, const int* tmeta,
                      const unsigned short* rowlist, const float* input, const float* input2, int Ca, const float* filter,
                      float* output, hipStream_t st)
{
    const int ucap = lc_plan_ucap(F);
    if (lc_waves() == 16)
        return lc_forward_nw<16>(B, N, M, F, C, r, ucap, chdr, rec, tmeta, rowlist, input, input2, Ca, filter, output, st);
    return lc_forward_nw<8>(B, N, M, F, C, r, ucap, chdr, rec, tmeta, rowlist, input, input2, Ca, filter, output, st);
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

extern "C" int sph3d_conv_plan_ucap(int F) { return lc_plan_ucap(F); }


extern "C" int sph3d_conv_plan_sizes(int B, int M, size_t* hdr_ints, size_t* rec_words, size_t* meta_ints, size_t* rowlist_shorts)
{
    const size_t nchunks = (size_t)((M + kLcChunk - 1) / kLcChunk);
    if (hdr_ints) *hdr_ints = (size_t)B * nchunks * kLcHdrInts;
    if (rec_words) *rec_words = (size_t)B * nchunks * kLcChunk * kLcRecWords + 512;     // + read-ahead slack of the record prefetch
    if (meta_ints) *meta_ints = (size_t)B * nchunks * kLcChunk * 2;
    if (rowlist_shorts) *rowlist_shorts = (size_t)B * nchunks * kLcRowsPerChunk + 64;
    return SPH3D_OK;
}

extern "C" int sph3d_conv_plan(int B, int N, int M, int K, int F, const int* order, const int* nn_index, const int* nn_count,
                               const int* bin_index, int* chunk_hdr, unsigned* records, int* target_meta,
                               unsigned short* row_lists, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M > 0 && K > 0 && F > 0, "conv_plan: bad dims B=%d N=%d M=%d K=%d F=%d", B, N, M, K, F);
    SPH3D_REQUIRE(K <= 64 && N <= 65536 && M < (1 << 24), "conv_plan: needs K <= 64, N <= 65536, M < 2^24 (got K=%d N=%d M=%d)", K, N, M);
    const int ucap = lc_plan_ucap(F);
    SPH3D_REQUIRE(ucap >= 128, "conv_plan: F=%d bins do not leave room for a tile's rows (F <= 65)", F);
    if (B == 0) return SPH3D_OK;
    const int nchunks = (M + kLcChunk - 1) / kLcChunk;
    const int W = (N + 31) >> 5;
    const size_t lds = sizeof(unsigned) * 9 * (size_t)W;
    hipStream_t st = as_stream(stream);
    if (lds > 32 * 1024) {
        int rc = check_hip(hipFuncSetAttribute((const void*)lc_plan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "conv_plan: hipFuncSetAttribute");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(lc_plan_kernel, dim3(B * nchunks), dim3(256), lds, st, B, M, N, K, F, ucap, 4 * lc_waves(), nchunks, W, order, nn_index,
                       nn_count, bin_index, chunk_hdr, records, target_meta, row_lists);
    return check_launch("sph3d_conv_plan");
}

extern "C" int sph3d_depthwise_conv3d_lds_supported(int F, int C, int r, int K) { return lc_shape_ok(F, C, r, K) ? 1 : 0; }

extern "C" int sph3d_depthwise_conv3d_lds(int B, int N, int M, int F, int C, int r, const int* chunk_hdr, const unsigned* records,
                                          const int* target_meta, const unsigned short* row_lists, const float* input,
                                          const float* filter, float* output, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M > 0 && F > 0 && C > 0, "DepthwiseConv3dLds: bad dims B=%d N=%d M=%d F=%d C=%d", B, N, M, F, C);
    SPH3D_REQUIRE(lc_shape_ok(F, C, r, 64), "DepthwiseConv3dLds: needs r in {1,2}, C %% 64 == 0, F <= 65 (got r=%d C=%d F=%d)", r, C, F);
    SPH3D_REQUIRE((long long)N * C < (1LL << 31), "DepthwiseConv3dLds: N*C overflows int32");
    if (B == 0) return SPH3D_OK;
    return lc_forward(B, N, M, F, C, r, chunk_hdr, records, target_meta, row_lists, input, nullptr, 0, filter, output, as_stream(stream));
}

extern "C" int sph3d_depthwise_conv3d_lds_cat(int B, int N, int M, int F, int Ca, int Cb, int r, const int* chunk_hdr,
                                              const unsigned* records, const int* target_meta, const unsigned short* row_lists,
                                              const float* input_a, const float* input_b, const float* filter, float* output,
                                              sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M > 0 && F > 0 && Ca > 0 && Cb > 0, "DepthwiseConv3dLdsCat: bad dims B=%d N=%d M=%d F=%d Ca=%d Cb=%d",
                  B, N, M, F, Ca, Cb);
    SPH3D_REQUIRE(lc_shape_ok(F, Ca + Cb, r, 64) && Ca % 64 == 0, "DepthwiseConv3dLdsCat: needs r in {1,2}, Ca %% 64 == 0, Cb %% 64 == 0, F <= 65");
    SPH3D_REQUIRE((long long)N * (Ca + Cb) < (1LL << 31), "DepthwiseConv3dLdsCat: N*C overflows int32");
    if (B == 0) return SPH3D_OK;
    return lc_forward(B, N, M, F, Ca + Cb, r, chunk_hdr, records, target_meta, row_lists, input_a, input_b, Ca, filter, output,
                      as_stream(stream));
}
