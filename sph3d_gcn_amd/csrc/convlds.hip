// convlds.hip — depthwise spherical convolution gathered from LDS tiles (forward), gfx950.
//
// Same arithmetic, same summation order (k = 0..cnt-1, one fmaf per edge and output, one reciprocal per point) and
// therefore the same bits as dwconv_fwd_multi of conv3d.hip; replaces depthwise_conv3d_forward
// (tf_ops/convolution/tf_conv3d_gpu.cu:7-29).  What changes is where the neighbour rows come from.
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
//   * per target the plan holds a RECORD of 64 u16 entries in neighbour order, entry = LDS slot of the edge's row | bin << 8
//     (padding entries: the zero row and the zero filter row).  Records of a tile travel to LDS with the rows; a lane reads 8
//     entries with one ds_read_b128 and turns an entry into the row address AND the filter address with one v_perm_b32 each
//     ((slot << 8) | lane byte, (bin << 8) | lane byte: rows and filter planes are 256 B apart by construction, region bases
//     are immediate offsets of the ds_read);
//   * per edge and wave: 2 v_perm + 1 row read + r filter reads + 2r packed FMAs, nothing else; eight edges are unrolled;
//   * rows travel global -> LDS by LDS-DMA (global_load_lds_dwordx4), no registers; two 8-wave workgroups per CU (80 KB of
//     LDS each), so one stages while the other gathers — no software pipeline inside a workgroup.
// The plan is per GRAPH (every convolution on the graph and its channel slices share it), built by one kernel after the
// neighbour search: greedy tiles of <= 32 spatially consecutive targets whose row union fits the LDS, union ranks by bitmap +
// prefix popcounts, records.  Nothing is sorted by bin: the summation order of a target is its neighbour order.
#include <cstdlib>
#include "common.hpp"

namespace sph3d {

constexpr int kLcChunk = 128;           // consecutive positions of the spatial order handled by one plan workgroup; tiles never span chunks

constexpr int kLcHdrInts = 132;         // per chunk: [0] tiles, [1 + 2t] first | targets << 8 | rows << 16, [2 + 2t] row-list offset
constexpr int kLcRecWords = 64;         // 64 u32 entries per target: slot | bin << 16
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
//      row list (ascending row id); targets ranked by neighbour count (descending, ties by position);
//      per target the record of 64 entries (slot | bin << 16 in neighbour order, padding = ucap | F << 16) and
//      meta = target id | count << 24.
// Outputs (all addressed from the chunk index):
//   chdr    [B*nchunks][132]     int : see kLcHdrInts
//   rec     [B*nchunks*128][64]  u32 : records, a tile's targets contiguous from `first`, in rank order
//   tmeta   [B*nchunks*128]      int
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
        // rank of target tstart + lane among the tile's targets: more neighbours first, ties by position
        const int myc = lane < T ? sCnt[tstart + lane] : -1;
        int rank = 0;
        for (int i = 0; i < T; i++) {
            const int ci = __builtin_amdgcn_readlane(myc, i);
            rank += (ci > myc || (ci == myc && i < lane)) ? 1 : 0;
        }
        for (int qq = 0; qq < T; qq++) {
            const int q = tstart + qq;
            const int cq = sCnt[q];
            const int dst = tstart + __builtin_amdgcn_readlane(rank, qq);
            unsigned ent = (unsigned)ucap | ((unsigned)F << 16);
            if (lane < cq) {
                const int n = sIdx[q][lane];
                const int slot = (int)preW[n >> 5] + __popc(bmW[n >> 5] & ((1u << (n & 31)) - 1u));
                ent = (unsigned)slot | ((unsigned)sBin[q][lane] << 16);
            }
            rec[(chunk * kLcChunk + dst) * kLcRecWords + lane] = ent;
            if (lane == 0) tmeta[chunk * kLcChunk + dst] = sTm[q] | (cq << 24);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef SPH3D_LC_PERM
#define SPH3D_LC_PERM 1
#endif
// entry -> (slot << 8) | lane byte and (bin << 8) | lane byte.  v_perm_b32: selector bytes 0-3 take bytes of the second
// operand, 4-7 of the first, 0x0c is a zero byte
__device__ __forceinline__ unsigned lc_arow(unsigned ent, unsigned lb)
{
#if SPH3D_LC_PERM
    return __builtin_amdgcn_perm(ent, lb, 0x0c050400u);
#else
    return ((ent & 0xffffu) << 8) | lb;
#endif
}
__device__ __forceinline__ unsigned lc_afil(unsigned ent, unsigned lb)
{
#if SPH3D_LC_PERM
    return __builtin_amdgcn_perm(ent, lb, 0x0c0c0600u);
#else
    return (((ent >> 16) & 0xffu) << 8) | lb;
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

struct LcTile {
    int valid, b, chunk, a, uoff;
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

    // ---- this workgroup's slice and its share of the chunks: the workgroups of an XCD walk the XCD's clouds one after the
    // other (every part takes its range of chunks of cloud 0, then of cloud 1, ...), so one cloud's rows stay in the L2 ----
    const int WPX = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7;
    const int wi = (int)blockIdx.x >> 3;
    const int slice = wi % nslices;
    const int nparts = WPX / nslices;
    const int part = wi / nslices;
    if (part >= nparts) return;
    const bool affine = (B & 7) == 0;
    const int nclouds = affine ? (B >> 3) : B;
    const long long gpart = affine ? part : (long long)xcd * nparts + part;
    const long long gparts = affine ? nparts : 8LL * nparts;
    const int ch_begin = (int)((long long)nchunks * gpart / gparts);
    const int span = (int)((long long)nchunks * (gpart + 1) / gparts) - ch_begin;
    const int f_end = nclouds * span;
    if (f_end <= 0) return;
    const int c0 = slice * 64;                       // first input channel of the slice
    const bool second = input2 != nullptr && c0 >= Ca;
    const int Cs = input2 == nullptr ? C : (second ? C - Ca : Ca);        // row stride of the source tensor
    const float* src = (second ? input2 : input) + (c0 - (second ? Ca : 0)) + l16 * 4;

    // ---- filter slice -> LDS planes (plane q: outputs 4q..4q+3 of every lane), zero rows ----
    {
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
    }

    // ---- tile cursor: header of the current chunk in lanes (tile t -> lane t), the next chunk's prefetched ----
    int fH = 0;
    auto chunk_of = [&](int f, int& b, int& chunk) {
        const int ci = f / span, ch = ch_begin + (f - ci * span);
        b = affine ? xcd + 8 * ci : ci;
        chunk = b * nchunks + ch;
    };
    auto load_hdr = [&](int f, int& hA, int& hO, int& hN) {
        int b, chunk;
        chunk_of(f, b, chunk);
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
        t.valid = 0; t.b = 0; t.chunk = 0; t.a = 0; t.uoff = 0;
        for (;;) {
            if (tH + 1 < ntH) {
                tH++;
                t.valid = 1;
                chunk_of(fH, t.b, t.chunk);
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
    auto meta_of = [&](const LcTile& t) -> int {
        const int tstart = t.a & 0xff, T = (t.a >> 8) & 0xff;
        const int rho = 4 * wave + (lane >> 4);
        int v = 0;
        if (t.valid && rho < T) v = tmeta[(size_t)t.chunk * kLcChunk + tstart + rho];
        return v;
    };

    LcTile t0 = next_tile();
    int ids0 = ids_of(t0);
    int meta0 = meta_of(t0);
    const unsigned lb = (unsigned)l16 << 4;
    const int rho = 4 * wave + (lane >> 4);
    const unsigned recaddr0 = (unsigned)(Map::RECB + rho * (kLcRecWords * 4));
    __syncthreads();

    while (t0.valid) {
        const int tstart = t0.a & 0xff, T = (t0.a >> 8) & 0xff, U = t0.a >> 16;
        // ---- stage the tile: rows (LDS-DMA, four 256-B rows per wave instruction); every wave its own four records ----
        if (!(dbg & 1)) {
            const int RW = ((U + 4 * NW - 1) / (4 * NW)) << 2;
            const float* inb = src + (size_t)t0.b * N * Cs;
            for (int j = 0; j < RW; j += 4) {
                const int i0 = wave * RW + j;
                if (i0 >= U) break;
                const int rid = __shfl(ids0, j + (lane >> 4));
                const float* gp = inb + (size_t)rid * Cs;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(lds + Map::RB + (size_t)i0 * kLcRowBytes), 16, 0, 0);
            }
            if (wave * 4 < T) {
                const unsigned* gp = rec + ((size_t)t0.chunk * kLcChunk + tstart + wave * 4) * kLcRecWords + lane * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(lds + Map::RECB + wave * 1024), 16, 0, 0);
            }
        }
        // ---- in flight under the DMA: the next tile's row ids and meta ----
        LcTile t1 = next_tile();
        const int ids1 = ids_of(t1);
        const int meta1 = meta_of(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- gather: quarter-wave q of wave w takes target 4w + q of the tile (rank order: the four finish together) ----
        if (4 * wave < T && !(dbg & 2)) {
            const int cnt = (int)((unsigned)meta0 >> 24);
            int nmax = cnt;
            nmax = max(nmax, __shfl_xor(nmax, 16));
            nmax = max(nmax, __shfl_xor(nmax, 32));
            nmax = uniform(nmax);
            f32x2 acc[2 * R];
#pragma unroll
            for (int v = 0; v < 2 * R; v++) acc[v] = f32x2{0.f, 0.f};
            unsigned ra = recaddr0;
            // eight edges per trip; the reads of four edges are issued before the first of their FMAs, the next four edges'
            // reads before the FMAs of these (the compiler's own schedule waited for every edge's three reads); the entries
            // of the next trip are requested a trip ahead
            u32x4 e0 = lc_ld<u32x4>(ra), e1 = lc_ld<u32x4>(ra + 16);
            for (int eb = 0; eb < nmax; eb += 8) {
                ra += 32;
                const unsigned ent[8] = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
                f32x4 x[8], w0[8], w1[8];
#pragma unroll
                for (int h = 0; h < 2; h++) {
#pragma unroll
                    for (int uu = 0; uu < 4; uu++) {
                        const int u = 4 * h + uu;
                        const unsigned arow = lc_arow(ent[u], lb);
                        const unsigned afil = lc_afil(ent[u], lb);
                        x[u] = lc_ld<f32x4>(arow + Map::RB);
                        w0[u] = lc_ld<f32x4>(afil);
                        if constexpr (R == 2) w1[u] = lc_ld<f32x4>(afil + PR * 256);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (entries past the record's 64 read the next record or the rows region: never used, eb + 8 >= nmax ends the loop)
                e0 = lc_ld<u32x4>(ra);
                e1 = lc_ld<u32x4>(ra + 16);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const f32x2 x01 = {x[u][0], x[u][1]}, x23 = {x[u][2], x[u][3]};
                    if constexpr (R == 2) {
                        lc_fma<0>(acc[0], x01, f32x2{w0[u][0], w0[u][1]});
                        lc_fma<1>(acc[1], x01, f32x2{w0[u][2], w0[u][3]});
                        lc_fma<0>(acc[2], x23, f32x2{w1[u][0], w1[u][1]});
                        lc_fma<1>(acc[3], x23, f32x2{w1[u][2], w1[u][3]});
                    } else {
                        acc[0] = __builtin_elementwise_fma(x01, f32x2{w0[u][0], w0[u][1]}, acc[0]);
                        acc[1] = __builtin_elementwise_fma(x23, f32x2{w0[u][2], w0[u][3]}, acc[1]);
                    }
                }
            }
            if (rho < T) {
                const int m = meta0 & 0xffffff;
                const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
                float* op = output + ((size_t)t0.b * M + m) * CR + (size_t)(c0 + l16 * 4) * R;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    f32x4 o = {acc[2 * q][0] * inv, acc[2 * q][1] * inv, acc[2 * q + 1][0] * inv, acc[2 * q + 1][1] * inv};
                    *reinterpret_cast<f32x4*>(op + 4 * q) = o;
                }
            }
        }
        __syncthreads();
        t0 = t1;
        ids0 = ids1;
        meta0 = meta1;
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
    // one (NW = 16) or two (NW = 8) workgroups per CU; the parts of a slice divide the chunks of every cloud of the XCD
    int wpx = NW == 16 ? 32 : 64;
    if (wpx < nslices) wpx = nslices;
    wpx = (wpx / nslices) * nslices;
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
    if (rec_words) *rec_words = (size_t)B * nchunks * kLcChunk * kLcRecWords + 256;     // + read-ahead slack of the record DMA
    if (meta_ints) *meta_ints = (size_t)B * nchunks * kLcChunk;
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
