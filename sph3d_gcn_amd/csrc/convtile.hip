// convtile.hip — LDS-tiled depthwise spherical convolution, forward, for gfx950 (layers with C >= 128).
//
// Same arithmetic as dwconv_fwd_row of conv3d.hip (tf_ops/convolution/tf_conv3d_gpu.cu:7-29), different data movement;
// needs the tile plan of tile.hip.  Per tile of <= 16 spatially consecutive output points a 16-wave workgroup
//   1. STAGES the union of the tile's neighbour rows (a 128-channel slice of each, 512 B) in LDS: global loads issued one
//      tile ahead into registers, written to LDS between two barriers;
//   2. gives every wave one output point and walks its edges GROUPED BY BIN: the rows of a (point, bin) group are summed
//      with one ds_read_b64 + one packed add per edge (a lane owns two channels, the wave a whole row slice; the slot of
//      an edge's row is a byte of the plan's slot words, extracted on the scalar unit), then multiplied by the filter
//      row ONCE per group (6 edges on S3DIS-like data).
// The gather kernels pay one L1-miss row gather + one 1-KB LDS filter read + 4 FMA per edge.
//
// Measured (round 2, B = 16 x 8192 points, K = 64, F = 33; profiles/r02_conv_tiled_*.csv):
//   C = 128, r = 2:  0.207 ms vs 0.275 ms for dwconv_fwd_row (16.2 % vs 12.2 % of the 8 TB/s roofline on 269 MB);
//   faster on every C >= 128 level of the S3DIS plan (0.038/0.048, 0.061/0.072, 0.110/0.125, 0.030/0.034, ...).
// Counters: 100 M instructions instead of 140 M, but the waves still wait 53 % of their cycles — 82 us of the 207 are the
// skeleton (per-tile metadata loads through the vector L1 at 16 cycles per wave instruction, two barriers per tile, the
// split tiles' slow path), 39 us staging, 92 us the gather.  Variants measured and dropped: 64-channel slices with two
// workgroups per CU (two edges per ds_read_b64, half-waves: 13.9 vector instructions per edge, 0.28-0.33 ms); LDS-DMA
// staging without the register pipeline (0.33-0.37 ms: descriptor -> row ids -> slot words -> rows as four dependent
// round trips per tile); the same scheme for the gradients over the transposed graph (source points with hundreds of
// in-edges blow up the tile unions: 2.6 ms vs 0.53 ms for dwconv_bwd_t_vec).
// The plan costs 0.32 ms per level-0 graph, so the tiled forward pays only when a graph is reused by many convolutions
// (inference on a fixed cloud, several steps on one batch); a training step that rebuilds its graphs every step is
// faster on the gather kernels, which therefore stay the default (sph3d_gcn_amd/_plan.py).
#include "common.hpp"

namespace sph3d {

// VEC consecutive floats (16-B aligned for VEC >= 4, 8-B for VEC = 2) -> registers
template <int VEC>
__device__ __forceinline__ void ld_vec(const float* p, float (&x)[VEC])
{
    if constexpr (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        x[0] = t.x;
        x[1] = t.y;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const float4 t = *reinterpret_cast<const float4*>(p + j);
            x[j] = t.x;
            x[j + 1] = t.y;
            x[j + 2] = t.z;
            x[j + 3] = t.w;
        }
    }
}

// work decode shared by both kernels: item -> (cloud, candidate tile); clouds XCD-affine when B is a multiple of 8
__device__ __forceinline__ int tile_items(int B, int cands, int xcd)
{
    return ((B & 7) == 0) ? (B >> 3) * cands : (B * cands - xcd + 7) / 8;
}
__device__ __forceinline__ void tile_item(int B, int cands, int xcd, int it, int& b, int& c)
{
    if ((B & 7) == 0) {
        b = xcd + 8 * (it / cands);
        c = it % cands;
    } else {
        const int id = xcd + 8 * it;
        b = id / cands;
        c = id % cands;
    }
}

// what a wave prefetches for one tile: the ids of the rows it stages and, for its target, bounds + slot words
template <int VEC, int NW>
struct TileRegs {
    static constexpr int SLC = 32 * VEC, LPR = SLC / 4, RPI = 64 / LPR;
    static constexpr int TPW = (kTileP + NW - 1) / NW;                      // targets per wave and tile
    static constexpr int MAXIT = ((252 + RPI - 1) / RPI + NW - 1) / NW;     // wave loads per wave and tile (ucap <= 252)
    int hv;              // lane 0: targets, lane 1: rows of the tile
    int rid[MAXIT];      // row ids this wave stages
    int tm[TPW];         // target ids (broadcast)
    int ov[TPW];         // lane f <= F: first slot word of bin f; lane F + 1: edge count
    int wv[TPW];         // the target's slot words
};

template <int VEC, int NW>
__device__ __forceinline__ void tile_fetch(TileRegs<VEC, NW>& t, size_t cand, int ucap, int F, int wave, int lane,
                                           const int* __restrict__ hdr, const int* __restrict__ tgt,
                                           const int* __restrict__ ulist,
                                           const int* __restrict__ pb, const int* __restrict__ slotw)
{
    using T = TileRegs<VEC, NW>;
    t.hv = hdr[cand * 2 + (lane & 1)];
#pragma unroll
    for (int k = 0; k < T::MAXIT; k++) {
        int r = (wave + k * NW) * T::RPI + lane / T::LPR;
        r = r < ucap ? r : ucap - 1;
        t.rid[k] = ulist[cand * ucap + r];
    }
#pragma unroll
    for (int j = 0; j < T::TPW; j++) {
        const int idx = (wave + j * NW) & (kTileP - 1);
        const size_t tp = cand * kTileP + idx;
        t.tm[j] = tgt[tp];
        t.ov[j] = pb[tp * (F + 2) + (lane <= F + 1 ? lane : F + 1)];
        t.wv[j] = slotw[tp * kSlotWords + lane];
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward.  Software pipeline over the candidate tiles of a workgroup: the plan keeps everything of a tile at addresses
// computed from the tile index, so while tile k is gathered from LDS the registers already hold the ROWS of tile k+1
// (global loads issued one tile earlier, written to LDS after the barrier) and the loads of the row ids / bounds / slot
// words of tile k+2 are in flight.
// ------------------------------------------------------------------------------------------------------------
// The target's slot words form ONE stream (every bin group padded to whole words), read two words = eight rows per
// trip so that eight ds_reads are in flight per wait; a group ends on a word boundary, where its sum is multiplied by the
// filter row — which was requested when the previous group ended.  (Walking group by group, 4 reads then a wait, the
// waves spent 54 % of their cycles in s_waitcnt: a group is 6 edges on average, a one-target-per-wave tile only as fast
// as one wave's dependent chain.)
template <int R>
__device__ __forceinline__ void fwd_target_whole(const float* rows_lane, const float* lfilt_lane, int F, int ov, int wv,
                                                 unsigned zword, bool store, float* op)
{
    constexpr int NO = 2 * R;
    const int lane = lane_id();
    const int nx = __shfl_down(ov, 1);
    unsigned long long mask = __ballot(lane < F && nx > ov);
    const int cnt = __builtin_amdgcn_readlane(ov, F + 1);
    const int dend = __builtin_amdgcn_readlane(ov, F);
    const char* rb = reinterpret_cast<const char*>(rows_lane);
    float acc[NO];
#pragma unroll
    for (int jj = 0; jj < NO; jj++) acc[jj] = 0.f;
    int f = mask ? (int)__builtin_ctzll(mask) : 0;
    mask &= mask - 1;
    int gend = __builtin_amdgcn_readlane(ov, f + 1);
    float Wf[NO];
    ld_vec<NO>(lfilt_lane + f * (128 * R), Wf);
    float S0 = 0.f, S1 = 0.f;
#define SPH3D_FLUSH()                                                                                   \
    {                                                                                                   \
        _Pragma("unroll") for (int jj = 0; jj < NO; jj++) acc[jj] = fmaf(jj / R == 0 ? S0 : S1, Wf[jj], acc[jj]); \
        S0 = 0.f;                                                                                       \
        S1 = 0.f;                                                                                       \
        f = mask ? (int)__builtin_ctzll(mask) : f;                                                      \
        mask &= mask - 1;                                                                               \
        gend = __builtin_amdgcn_readlane(ov, f + 1);                                                    \
        ld_vec<NO>(lfilt_lane + f * (128 * R), Wf);                                                     \
    }
    for (int d = 0; d < dend; d += 2) {
        const unsigned w0 = (unsigned)__builtin_amdgcn_readlane(wv, d);
        const unsigned w1r = (unsigned)__builtin_amdgcn_readlane(wv, (d + 1) & 63);
        const unsigned w1 = (d + 1) < dend ? w1r : zword;
        float2 x[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            x[i] = *reinterpret_cast<const float2*>(rb + (((w0 >> (8 * i)) & 0xffu) << 9));          // scalar: slot * 512 B
            x[4 + i] = *reinterpret_cast<const float2*>(rb + (((w1 >> (8 * i)) & 0xffu) << 9));
        }
        S0 += ((x[0].x + x[1].x) + x[2].x) + x[3].x;
        S1 += ((x[0].y + x[1].y) + x[2].y) + x[3].y;
        if (d + 1 == gend) SPH3D_FLUSH()
        S0 += ((x[4].x + x[5].x) + x[6].x) + x[7].x;
        S1 += ((x[4].y + x[5].y) + x[6].y) + x[7].y;
        if (d + 2 == gend) SPH3D_FLUSH()
    }
#undef SPH3D_FLUSH
    const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
    for (int jj = 0; jj < NO; jj++) acc[jj] *= inv;
    if (store) {
        if constexpr (NO == 2) *reinterpret_cast<float2*>(op) = make_float2(acc[0], acc[1]);
        else *reinterpret_cast<float4*>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

template <int R>
__global__ __launch_bounds__(1024) void dwconv_tile_fwd_whole(
    int B, int N, int M, int F, int C, int cands, int nslices, int W, int ucap,
    const int* __restrict__ hdr, const int* __restrict__ tgt, const int* __restrict__ ulist,
    const int* __restrict__ pb, const int* __restrict__ slotw, const int* __restrict__ xsteps,
    const int* __restrict__ counters,
    const float* __restrict__ input, const float* __restrict__ filter, float* __restrict__ output)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = 16;
    using T = TileRegs<4, NW>;             // staging geometry of 128-channel rows: 32 lanes per row, 2 rows per wave load
    constexpr int SLC = 128, LPR = T::LPR, RPI = T::RPI, MAXIT = T::MAXIT, NO = 2 * R;
    static_assert(T::TPW == 1 && RPI == 2, "one target per wave and tile");
    const int capR = (ucap + RPI - 1) / RPI * RPI;
    const unsigned zword = 0x01010101u * (unsigned)capR;
    float* rows = lds;                                   // [capR + RPI][128]
    float* lfilt = lds + (size_t)(capR + RPI) * SLC;     // [F][128 * R]
    const int CR = C * R;
    const int xcd = (int)blockIdx.x & 7;
    const int q = (int)blockIdx.x >> 3;
    const int slice = q % nslices;
    const int w = q / nslices;
    const int c0 = slice * SLC;
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();
    {
        const int SLr = ((C - c0) < SLC ? (C - c0) : SLC) * R;
        for (int e = tid * 4; e < F * SLC * R; e += NW * 64 * 4) {
            const int f = e / (SLC * R);
            const int i = e - f * (SLC * R);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < SLr) v = *reinterpret_cast<const float4*>(&filter[(size_t)f * CR + (size_t)c0 * R + i]);
            *reinterpret_cast<float4*>(&lfilt[e]) = v;
        }
        for (int i = tid; i < RPI * SLC; i += NW * 64) rows[(size_t)capR * SLC + i] = 0.f;
    }
    __syncthreads();
    const bool lane_ok = (c0 + lane * 2) < C;
    int stc = c0 + (lane % LPR) * 4;
    if (stc + 4 > C) stc = c0;
    const int nitems = tile_items(B, cands, xcd);

#define SPH3D_ROWS_LOAD(t, b)                                                                 \
    {                                                                                         \
        const int U_ = __builtin_amdgcn_readlane((t).hv, 1);                                  \
        const float* inb_ = input + (size_t)(b) * N * C + stc;                                \
        _Pragma("unroll") for (int k = 0; k < MAXIT; k++) {                                   \
            const int i = wave + k * NW;                                                      \
            const int r = i * RPI + lane / LPR;                                               \
            const int id = r < U_ ? (t).rid[k] : 0;                                           \
            float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                      \
            if (i * RPI < U_) v_ = *reinterpret_cast<const float4*>(inb_ + (size_t)id * C);   \
            ra[k] = v_;                                                                       \
        }                                                                                     \
    }
    T ta, tb, tc;
    float4 ra[MAXIT];
    int ba = 0, ca = 0, bb = 0, cb = 0, bc = 0, cc = 0;
    {
        const int i0 = w < nitems ? w : 0, i1 = (w + W) < nitems ? (w + W) : 0;
        tile_item(B, cands, xcd, i0, ba, ca);
        tile_fetch<4, NW>(ta, (size_t)ba * cands + ca, ucap, F, wave, lane, hdr, tgt, ulist, pb, slotw);
        tile_item(B, cands, xcd, i1, bb, cb);
        tile_fetch<4, NW>(tb, (size_t)bb * cands + cb, ucap, F, wave, lane, hdr, tgt, ulist, pb, slotw);
        SPH3D_ROWS_LOAD(ta, ba)
    }
    for (int it = w; it < nitems; it += W) {
        const int cntT = __builtin_amdgcn_readlane(ta.hv, 0);
        const int U = __builtin_amdgcn_readlane(ta.hv, 1);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXIT; k++) {
            const int i = wave + k * NW;
            if (i * RPI < U) *reinterpret_cast<float4*>(rows + (size_t)i * (RPI * SLC) + lane * 4) = ra[k];
        }
        __syncthreads();
        SPH3D_ROWS_LOAD(tb, bb)
        {
            const int i2 = (it + 2 * W) < nitems ? (it + 2 * W) : it;
            tile_item(B, cands, xcd, i2, bc, cc);
            tile_fetch<4, NW>(tc, (size_t)bc * cands + cc, ucap, F, wave, lane, hdr, tgt, ulist, pb, slotw);
        }
        if (wave < cntT) {
            const int m = __builtin_amdgcn_readfirstlane(ta.tm[0]);
            float* op = &output[((size_t)ba * M + m) * CR + (size_t)(c0 + lane * 2) * R];
            fwd_target_whole<R>(rows + lane * 2, lfilt + lane * NO, F, ta.ov[0], ta.wv[0], zword, lane_ok, op);
        }
        ta = tb;
        tb = tc;
        ba = bb; bb = bc;
    }
#undef SPH3D_ROWS_LOAD
    // ---- extra steps (further sub-tiles of split candidates; the plan never makes a forward target direct here) ----
    const bool affine = (B & 7) == 0;
    for (int b = affine ? xcd : 0; b < B; b += affine ? 8 : 1) {
        const int nx = uniform(counters[1 + b]);
        const int* __restrict__ xs = xsteps + (size_t)b * cands * kTileP * 8;
        const float* inb = input + (size_t)b * N * C;
        for (int i = affine ? w : (xcd + 8 * w); i < nx; i += affine ? W : 8 * W) {
            const int c = uniform(xs[i * 8 + 0]), q0 = uniform(xs[i * 8 + 1]), cntT = uniform(xs[i * 8 + 2]);
            const int U = uniform(xs[i * 8 + 3]), uoff = uniform(xs[i * 8 + 4]);
            const size_t cand = (size_t)b * cands + c;
            __syncthreads();
            for (int k = wave; k * RPI < U; k += NW) {
                int r = k * RPI + lane / LPR;
                r = r < U ? r : U - 1;
                const int id = ulist[uoff + r];
                *reinterpret_cast<float4*>(rows + (size_t)k * (RPI * SLC) + lane * 4) =
                    *reinterpret_cast<const float4*>(inb + (size_t)id * C + stc);
            }
            __syncthreads();
            if (wave < cntT && U >= 0) {
                const size_t tp = cand * kTileP + q0 + wave;
                const int m = uniform(tgt[tp]);
                float* op = &output[((size_t)b * M + m) * CR + (size_t)(c0 + lane * 2) * R];
                const int ov = pb[tp * (F + 2) + (lane <= F + 1 ? lane : F + 1)];
                const int wv = slotw[tp * kSlotWords + lane];
                fwd_target_whole<R>(rows + lane * 2, lfilt + lane * NO, F, ov, wv, zword, lane_ok, op);
            }
        }
    }
}


static int tile_dims_ok(int B, int N, int M, int F, int C, int r, int ucap, const char* who)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M > 0 && F > 0 && C > 0 && ucap >= 4 && ucap <= 252 && ucap % 4 == 0,
                  "%s: bad dims B=%d N=%d M=%d F=%d C=%d ucap=%d", who, B, N, M, F, C, ucap);
    SPH3D_REQUIRE((r == 1 || r == 2) && C % 4 == 0 && F <= 62,
                  "%s: the tiled kernels need r in {1,2}, C %% 4 == 0, F <= 62 (got r=%d C=%d F=%d)", who, r, C, F);
    return SPH3D_OK;
}

static int set_lds(const void* kern, size_t lds, const char* who)
{
    SPH3D_REQUIRE(lds <= 160 * 1024, "%s: %zu B of LDS needed (capacity 160 KiB): lower ucap", who, lds);
    if (lds > 64 * 1024)
        return check_hip(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), who);
    return SPH3D_OK;
}

// workgroups per XCD and channel slice: every CU of the XCD holds `per_cu` resident workgroups
static int tile_W(int nslices, int per_cu)
{
    int W = 32 * per_cu / (nslices < 1 ? 1 : nslices);
    return W < 1 ? 1 : W;
}

template <int R>
static int launch_tile_fwd_whole(int B, int N, int M, int F, int C, int ucap, const int* hdr, const int* tgt, const int* ulist,
                                 const int* pb, const int* slotw, const int* xsteps, const int* counters,
                                 const float* input, const float* filter, float* output, hipStream_t st)
{
    const size_t lds = sizeof(float) * ((size_t)(ucap + 2) * 128 + (size_t)F * 128 * R);
    auto kern = dwconv_tile_fwd_whole<R>;
    int rc = set_lds((const void*)kern, lds, "DepthwiseConv3dTiled");
    if (rc) return rc;
    const int nslices = (C + 127) / 128;
    const int cands = (M + kTileP - 1) / kTileP;
    const int W = tile_W(nslices, 1);
    hipLaunchKernelGGL(kern, dim3(8 * W * nslices), dim3(1024), lds, st, B, N, M, F, C, cands, nslices, W, ucap, hdr, tgt, ulist,
                       pb, slotw, xsteps, counters, input, filter, output);
    return check_launch("sph3d_depthwise_conv3d_tiled");
}

// does the whole-wave forward kernel apply?  (rows + filter slice must fit one CU's LDS; the plan must not hold direct targets)
static bool whole_ok(int F, int C, int r, int ucap, int K)
{
    return C >= 128 && K <= ucap && sizeof(float) * ((size_t)(ucap + 2) * 128 + (size_t)F * 128 * r) <= 160 * 1024;
}

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_depthwise_conv3d_tiled_supported(int F, int C, int r, int K, int ucap)
{
    return (r == 1 || r == 2) && C % 4 == 0 && F <= 62 && whole_ok(F, C, r, ucap, K) ? 1 : 0;
}

extern "C" int sph3d_depthwise_conv3d_tiled(int B, int N, int M, int F, int C, int r, int ucap,
                                            const int* tile_hdr, const int* tile_targets, const int* tile_rows,
                                            const int* tile_pb, const int* slot_words, const int* extra_steps,
                                            const int* counters,
                                            const float* input, const float* filter, float* output, sph3d_stream_t stream)
{
    int rc = tile_dims_ok(B, N, M, F, C, r, ucap, "DepthwiseConv3dTiled");
    if (rc) return rc;
    SPH3D_REQUIRE(whole_ok(F, C, r, ucap, 0), "DepthwiseConv3dTiled: needs C >= 128 and F*%d*4 + %d*512 B <= 160 KiB of LDS",
                  128 * r, ucap + 2);
    if (B == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    if (r == 2)
        return launch_tile_fwd_whole<2>(B, N, M, F, C, ucap, tile_hdr, tile_targets, tile_rows, tile_pb, slot_words, extra_steps,
                                        counters, input, filter, output, st);
    return launch_tile_fwd_whole<1>(B, N, M, F, C, ucap, tile_hdr, tile_targets, tile_rows, tile_pb, slot_words, extra_steps,
                                    counters, input, filter, output, st);
}
