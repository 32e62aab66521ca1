// convtile.hip — LDS-tiled depthwise spherical convolution (forward + both gradients) for gfx950.
//
// Same arithmetic as conv3d.hip (tf_ops/convolution/tf_conv3d_gpu.cu:7-101), different data movement; needs the tile
// plan of tile.hip.  Per tile of <= 16 spatially consecutive targets a workgroup
//   1. STAGES the union of the tile's source rows (one 64- or 128-channel slice of each) in LDS — forward: LDS-DMA
//      (global_load_lds_dwordx4, 4 row slices per wave instruction, no registers); gradient: through registers, each row
//      multiplied by 1/nn_count of its output point on the way in, so that every edge below is a pure add;
//   2. walks each target's edges GROUPED BY BIN: the rows of one (target, bin) group are summed with one ds_read per
//      TWO edges (the two half-waves take alternate edges of the group; a lane owns VEC channels), then multiplied by
//      the filter row once per group (forward), or multiplied by the filter row and accumulated into the filter
//      gradient once per group (backward).  Round-1 kernels paid one L1-miss row gather + one 1-KB LDS filter read + 4 FMA
//      per EDGE; here an edge costs half a ds_read_b64 and half a packed add, a group (6 edges on S3DIS-like data) one
//      filter read + 4 FMA.
// Workgroups are persistent, one channel slice each (filter slice staged once; one partial filter-gradient table per
// workgroup), two per CU so that one stages while the other gathers; clouds stay XCD-affine (rows come from that L2).
// Sub-tiles whose union does not fit (marked by the plan) are gathered straight from memory by the same loop.
// Summation order of one output: bins ascending, inside a bin the even-position edges then the odd-position edges
// (two partial sums), halves added at the end; division by nn_count last (the reference divides every term;
// agreement with the oracle ~1e-7 relative, bound 1e-5).
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

// Sum of the staged rows of ONE (target, bin) group.  The group's slots are bytes, four per word, padded with the zero
// row; words [p0, p1) of the plan's slot array.  `wv` caches 64 consecutive words of the target (lane j holds word
// cbase + j; one coalesced load per target, re-filled when a long list runs past it), so a word reaches the scalar unit
// with v_readlane and each half-wave picks its byte with one v_bfe_u32: bytes 0 and 2 -> lanes 0-31, bytes 1 and 3 ->
// lanes 32-63.  Two words (8 edges, 4 ds_reads) per trip.
template <int VEC, int SL>
__device__ __forceinline__ void group_sum(const float* rows_lane, const int* __restrict__ slotw, int p0, int p1, int dlast,
                                          int& cbase, int& wv, int lane, int hs, unsigned zword, float (&S)[VEC])
{
    for (int d = p0; d < p1; d += 2) {
        if (d + 1 - cbase >= 64) {           // wave-uniform
            cbase = d;
            const int j = (d + lane) < dlast ? (d + lane) : dlast;
            wv = slotw[j];
        }
        const unsigned w0 = (unsigned)__builtin_amdgcn_readlane(wv, d - cbase);
        const unsigned w1r = (unsigned)__builtin_amdgcn_readlane(wv, d + 1 - cbase);
        const unsigned w1 = (d + 1) < p1 ? w1r : zword;
        // slot byte -> LDS byte offset of the row: v_bfe_u32 (scalar word, per-lane bit offset) + v_lshl_add_u32
        const char* rb = reinterpret_cast<const char*>(rows_lane);
        const unsigned a0 = __builtin_amdgcn_ubfe(w0, (unsigned)hs, 8u) * (unsigned)(SL * 4);
        const unsigned a1 = __builtin_amdgcn_ubfe(w0, (unsigned)hs + 16u, 8u) * (unsigned)(SL * 4);
        const unsigned a2 = __builtin_amdgcn_ubfe(w1, (unsigned)hs, 8u) * (unsigned)(SL * 4);
        const unsigned a3 = __builtin_amdgcn_ubfe(w1, (unsigned)hs + 16u, 8u) * (unsigned)(SL * 4);
        float x0[VEC], x1[VEC], x2[VEC], x3[VEC];
        ld_vec<VEC>(reinterpret_cast<const float*>(rb + a0), x0);
        ld_vec<VEC>(reinterpret_cast<const float*>(rb + a1), x1);
        ld_vec<VEC>(reinterpret_cast<const float*>(rb + a2), x2);
        ld_vec<VEC>(reinterpret_cast<const float*>(rb + a3), x3);
#pragma unroll
        for (int v = 0; v < VEC; v++) S[v] = (((S[v] + x0[v]) + x1[v]) + x2[v]) + x3[v];
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward.  R = depth multiplier, VEC = input channels per lane (slice = 32*VEC channels), NW waves per workgroup.
// ------------------------------------------------------------------------------------------------------------
template <int R, int VEC, int NW>
__global__ __launch_bounds__(NW * 64) void dwconv_tile_fwd(
    int B, int N, int M, int F, int C, int cands, int nslices, int W, int ucap,
    const int* __restrict__ order, const int* __restrict__ desc, const int* __restrict__ ulist,
    const int* __restrict__ pbounds, const int* __restrict__ slotw, const int* __restrict__ nnCount,
    const int* __restrict__ bounds, const int* __restrict__ key,
    const float* __restrict__ input, const float* __restrict__ filter, float* __restrict__ output)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SLC = 32 * VEC;          // input channels per slice
    constexpr int LPR = SLC / 4;           // staging lanes per row slice (16 B each)
    constexpr int RPI = 64 / LPR;          // row slices per wave load
    constexpr int NO = VEC * R;            // output channels per lane
    const int capR = (ucap + RPI - 1) / RPI * RPI;
    const unsigned zword = 0x01010101u * (unsigned)capR;     // four padding slots (the all-zero row, slot capR = ucap)
    float* rows = lds;                                   // [capR + RPI][SLC]
    float* lfilt = lds + (size_t)(capR + RPI) * SLC;     // [F][32][NO]
    const int CR = C * R;
    const int xcd = (int)blockIdx.x & 7;
    const int q = (int)blockIdx.x >> 3;
    const int slice = q % nslices;
    const int w = q / nslices;
    const int c0 = slice * SLC;
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();
    const int half = lane >> 5, li = lane & 31;

    {   // filter slice: bin f's SLC*R floats are contiguous in the filter; columns beyond C read as zero
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

    const bool lane_ok = (c0 + li * VEC) < C;
    const int cl = lane_ok ? (c0 + li * VEC) : c0;          // clamped channel for branch-free direct gathers
    const int lrow = li * VEC;                              // lane's float offset inside a staged row
    const int hs = half * 8;                                // this half-wave's byte of a slot word (and byte + 2)
    int stc = c0 + (lane % LPR) * 4;                        // staging: this lane's 4 channels
    if (stc + 4 > C) stc = c0;                              // partial last slice: the unused columns receive copies
    const int nitems = tile_items(B, cands, xcd);

    for (int it = w; it < nitems; it += W) {
        int b, c;
        tile_item(B, cands, xcd, it, b, c);
        const int* __restrict__ d = desc + ((size_t)b * cands + c) * kDescInts;
        const int g = uniform(d[0]);
        const int pos0 = c * kTileP;
        const int npts = (M - pos0) < kTileP ? (M - pos0) : kTileP;
        const float* inb = input + (size_t)b * N * C;
        for (int s = 0; s * g < npts; s++) {
            const int U = uniform(d[1 + 2 * s]);
            const int uoff = uniform(d[2 + 2 * s]);
            const int p0 = s * g;
            const int p1 = (p0 + g) < npts ? (p0 + g) : npts;
            // ---- stage the union rows (LDS-DMA: lane group j of wave load i fetches row i*RPI + j) ----
            for (int i = wave; i * RPI < U; i += NW) {
                int r = i * RPI + lane / LPR;
                r = r < U ? r : U - 1;
                const int rowid = ulist[uoff + r];
                const float* gp = inb + (size_t)rowid * C + stc;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(rows + (size_t)i * (RPI * SLC)), 16, 0, 0);
            }
            __syncthreads();
            // ---- gather ----
            for (int p = p0 + wave; p < p1; p += NW) {
                const int m = order ? uniform(order[(size_t)b * M + pos0 + p]) : (pos0 + p);
                const size_t row = (size_t)b * M + m;
                const int* __restrict__ bd = (U >= 0 ? pbounds : bounds) + row * (F + 1);
                const int ov = bd[lane <= F ? lane : F];
                const int nx = __shfl_down(ov, 1);
                unsigned long long mask = __ballot(lane < F && nx > ov);
                const int cnt = uniform(nnCount[row]);
                const int dlast = __builtin_amdgcn_readlane(ov, F) - 1;
                int cbase = __builtin_amdgcn_readfirstlane(ov);
                int wv = 0;
                if (U >= 0 && mask) wv = slotw[(cbase + lane) < dlast ? (cbase + lane) : dlast];
                float acc[NO];
#pragma unroll
                for (int j = 0; j < NO; j++) acc[j] = 0.f;
                while (mask) {
                    const int f = (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int e0 = __builtin_amdgcn_readlane(ov, f);
                    const int e1 = __builtin_amdgcn_readlane(ov, f + 1);
                    float S[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; v++) S[v] = 0.f;
                    if (U >= 0) {
                        group_sum<VEC, SLC>(rows + lrow, slotw, e0, e1, dlast, cbase, wv, lane, hs, zword, S);
                    } else {
                        for (int e = e0; e < e1; e += 2) {
                            const bool vb = (e + 1) < e1;
                            const int ka = key[e];
                            const int kb = key[vb ? (e + 1) : e];
                            const int kk = half ? kb : ka;
                            float x0[VEC];
                            ld_vec<VEC>(inb + (size_t)kk * C + cl, x0);
                            const float z = (half && !vb) ? 0.f : 1.f;
#pragma unroll
                            for (int v = 0; v < VEC; v++) S[v] = fmaf(x0[v], z, S[v]);
                        }
                    }
                    float Wf[NO];
                    ld_vec<NO>(lfilt + f * (SLC * R) + li * NO, Wf);
#pragma unroll
                    for (int j = 0; j < NO; j++) acc[j] = fmaf(S[j / R], Wf[j], acc[j]);
                }
#pragma unroll
                for (int j = 0; j < NO; j++) acc[j] += __shfl_xor(acc[j], 32);
                if (half == 0 && lane_ok) {
                    const float fc = (float)cnt;
                    float* op = &output[row * CR + (size_t)(c0 + li * VEC) * R];
#pragma unroll
                    for (int j = 0; j < NO; j++) acc[j] = cnt > 0 ? acc[j] / fc : 0.f;
                    if constexpr (NO == 2) {
                        *reinterpret_cast<float2*>(op) = make_float2(acc[0], acc[1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < NO; j += 4)
                            *reinterpret_cast<float4*>(op + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// both gradients.  Targets are SOURCE points n, rows are grad_output rows (slice = 32*VEC OUTPUT channels) pre-scaled
// by 1/nn_count[m] while staged.  grad_input: registers, one store per element.  grad_filter: per-lane register table
// acc[bin][VEC] reached through a switch on the (wave-uniform) bin of the finished group — only the groups that exist
// are visited (round 1 walked all F bins of every source with a compile-time bin index).
// ------------------------------------------------------------------------------------------------------------
#define SPH3D_ACC_CASE(i)                                                                  \
    case i:                                                                                \
        if constexpr (i < MAXF) {                                                          \
            _Pragma("unroll") for (int v = 0; v < VEC; v++) acc[i][v] = fmaf(S[v], xin[v], acc[i][v]); \
        }                                                                                  \
        break;
#define SPH3D_ACC_CASE8(b) SPH3D_ACC_CASE(b) SPH3D_ACC_CASE(b + 1) SPH3D_ACC_CASE(b + 2) SPH3D_ACC_CASE(b + 3) \
                           SPH3D_ACC_CASE(b + 4) SPH3D_ACC_CASE(b + 5) SPH3D_ACC_CASE(b + 6) SPH3D_ACC_CASE(b + 7)

template <int R, int VEC, int NW, int MAXF>
__global__ __launch_bounds__(NW * 64, (VEC == 2 && MAXF <= 33) ? 4 : 2) void dwconv_tile_bwd(
    int B, int N, int M, int F, int C, int cands, int nslices, int W, int ucap,
    const int* __restrict__ order, const int* __restrict__ desc, const int* __restrict__ ulist,
    const float* __restrict__ uscale, const int* __restrict__ pbounds, const int* __restrict__ slotw,
    const int* __restrict__ offsets, const int* __restrict__ entKey, const float* __restrict__ entScale,
    const float* __restrict__ input, const float* __restrict__ filter, const float* __restrict__ gradOutput,
    float* __restrict__ gradInput, float* __restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SLW = 32 * VEC;          // output channels per slice
    constexpr int LPR = SLW / 4;
    constexpr int RPI = 64 / LPR;
    constexpr int VI = VEC / R;            // input channels per lane (VEC = 2 or 4, R = 1 or 2)
    const int capR = (ucap + RPI - 1) / RPI * RPI;
    const unsigned zword = 0x01010101u * (unsigned)capR;
    const int rowFloats = (capR + RPI) * SLW;
    const int tabFloats = F * SLW;
    float* rows = lds;                                                   // also the reduction table at the end
    float* lfilt = lds + (rowFloats > tabFloats ? rowFloats : tabFloats); // [F][SLW]
    const int CR = C * R;
    const int xcd = (int)blockIdx.x & 7;
    const int q = (int)blockIdx.x >> 3;
    const int slice = q % nslices;
    const int w = q / nslices;
    const int o0 = slice * SLW;
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();
    const int half = lane >> 5, li = lane & 31;

    for (int e = tid; e < F * SLW; e += NW * 64) {
        const int f = e / SLW;
        const int j = e - f * SLW;
        lfilt[e] = (o0 + j) < CR ? filter[(size_t)f * CR + o0 + j] : 0.f;
    }
    for (int i = tid; i < RPI * SLW; i += NW * 64) rows[(size_t)capR * SLW + i] = 0.f;
    __syncthreads();

    const bool lane_ok = (o0 + li * VEC) < CR;
    const int ol = lane_ok ? (o0 + li * VEC) : o0;          // this lane's first output channel (clamped)
    const int lrow = li * VEC;
    const int hs = half * 8;
    int sto = o0 + (lane % LPR) * 4;
    if (sto + 4 > CR) sto = o0;
    float acc[MAXF][VEC];
#pragma unroll
    for (int i = 0; i < MAXF; i++)
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[i][v] = 0.f;
    const int nitems = tile_items(B, cands, xcd);

    for (int it = w; it < nitems; it += W) {
        int b, c;
        tile_item(B, cands, xcd, it, b, c);
        const int* __restrict__ d = desc + ((size_t)b * cands + c) * kDescInts;
        const int g = uniform(d[0]);
        const int pos0 = c * kTileP;
        const int npts = (N - pos0) < kTileP ? (N - pos0) : kTileP;
        const float* gob = gradOutput + (size_t)b * M * CR;
        const int* __restrict__ offb = offsets + (size_t)b * ((size_t)N * F + 1);
        for (int s = 0; s * g < npts; s++) {
            const int U = uniform(d[1 + 2 * s]);
            const int uoff = uniform(d[2 + 2 * s]);
            const int p0 = s * g;
            const int p1 = (p0 + g) < npts ? (p0 + g) : npts;
            // ---- stage the union of grad_output rows, scaled by 1/nn_count of their output point ----
#pragma unroll 2
            for (int i = wave; i * RPI < U; i += NW) {
                int r = i * RPI + lane / LPR;
                r = r < U ? r : U - 1;
                const int rowid = ulist[uoff + r];
                const float sc = uscale[uoff + r];
                float4 t = *reinterpret_cast<const float4*>(gob + (size_t)rowid * CR + sto);
                t.x *= sc; t.y *= sc; t.z *= sc; t.w *= sc;
                *reinterpret_cast<float4*>(rows + (size_t)i * (RPI * SLW) + lane * 4) = t;
            }
            __syncthreads();
            for (int p = p0 + wave; p < p1; p += NW) {
                const int n = order ? uniform(order[(size_t)b * N + pos0 + p]) : (pos0 + p);
                const int* __restrict__ bd = U >= 0 ? pbounds + ((size_t)b * N + n) * (F + 1) : offb + (size_t)n * F;
                const int ov = bd[lane <= F ? lane : F];
                const int nx = __shfl_down(ov, 1);
                unsigned long long mask = __ballot(lane < F && nx > ov);
                const int dlast = __builtin_amdgcn_readlane(ov, F) - 1;
                int cbase = __builtin_amdgcn_readfirstlane(ov);
                int wv = 0;
                if (U >= 0 && mask) wv = slotw[(cbase + lane) < dlast ? (cbase + lane) : dlast];
                float xin[VEC];
                {
                    const float* xp = input + ((size_t)b * N + n) * C + ol / R;
#pragma unroll
                    for (int v = 0; v < VEC; v++) xin[v] = xp[v / R];
                }
                float gi[VEC];
#pragma unroll
                for (int v = 0; v < VEC; v++) gi[v] = 0.f;
                while (mask) {
                    const int f = (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int e0 = __builtin_amdgcn_readlane(ov, f);
                    const int e1 = __builtin_amdgcn_readlane(ov, f + 1);
                    float S[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; v++) S[v] = 0.f;
                    if (U >= 0) {
                        group_sum<VEC, SLW>(rows + lrow, slotw, e0, e1, dlast, cbase, wv, lane, hs, zword, S);
                    } else {
                        for (int e = e0; e < e1; e += 2) {
                            const bool vb = (e + 1) < e1;
                            const int eb = vb ? (e + 1) : e;
                            const int kk = half ? entKey[eb] : entKey[e];
                            float sc = half ? entScale[eb] : entScale[e];
                            if (half && !vb) sc = 0.f;
                            float x0[VEC];
                            ld_vec<VEC>(gob + (size_t)kk * CR + ol, x0);
#pragma unroll
                            for (int v = 0; v < VEC; v++) S[v] = fmaf(x0[v], sc, S[v]);
                        }
                    }
                    float Wf[VEC];
                    ld_vec<VEC>(lfilt + f * SLW + lrow, Wf);
#pragma unroll
                    for (int v = 0; v < VEC; v++) gi[v] = fmaf(S[v], Wf[v], gi[v]);
                    switch (f) {
                        SPH3D_ACC_CASE8(0) SPH3D_ACC_CASE8(8) SPH3D_ACC_CASE8(16) SPH3D_ACC_CASE8(24)
                        SPH3D_ACC_CASE8(32) SPH3D_ACC_CASE8(40) SPH3D_ACC_CASE8(48) SPH3D_ACC_CASE8(56)
                        default: break;
                    }
                }
#pragma unroll
                for (int v = 0; v < VEC; v++) gi[v] += __shfl_xor(gi[v], 32);
                if (half == 0 && lane_ok) {
                    float* gp = &gradInput[((size_t)b * N + n) * C + ol / R];
                    float o[VI];
#pragma unroll
                    for (int u = 0; u < VI; u++) {
                        float t = 0.f;
#pragma unroll
                        for (int rr = 0; rr < R; rr++) t += gi[u * R + rr];
                        o[u] = t;
                    }
                    if constexpr (VI == 1) gp[0] = o[0];
                    else if constexpr (VI == 2) *reinterpret_cast<float2*>(gp) = make_float2(o[0], o[1]);
                    else *reinterpret_cast<float4*>(gp) = make_float4(o[0], o[1], o[2 % VI], o[3 % VI]);
                }
            }
            __syncthreads();
        }
    }

    // ---- one partial filter-gradient table per workgroup: halves, then the waves take turns on one LDS table ----
#pragma unroll
    for (int i = 0; i < MAXF; i++)
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[i][v] += __shfl_xor(acc[i][v], 32);
    float* tab = rows;               // [F][SLW]; everyone passed the last barrier of the loop
    for (int w2 = 0; w2 < NW; w2++) {
        if (wave == w2 && half == 0) {
#pragma unroll
            for (int i = 0; i < MAXF; i++) {
                if (i < F) {
#pragma unroll
                    for (int v = 0; v < VEC; v++) {
                        float* p = &tab[i * SLW + lrow + v];
                        *p = (w2 == 0) ? acc[i][v] : (*p + acc[i][v]);
                    }
                }
            }
        }
        __syncthreads();
    }
    float* out = partial + ((size_t)xcd * W + w) * ((size_t)F * CR);
    for (int e = tid; e < F * SLW; e += NW * 64) {
        const int f = e / SLW;
        const int j = e - f * SLW;
        if ((o0 + j) < CR) out[(size_t)f * CR + o0 + j] = tab[e];
    }
}
#undef SPH3D_ACC_CASE
#undef SPH3D_ACC_CASE8

// grad_filter[j] = sum of the partial tables in fixed order (1024 threads = 32 outputs x 32 partial-lanes)
__global__ __launch_bounds__(1024) void reduce_slabs(int nparts, int total, const float* __restrict__ partial,
                                                    float* __restrict__ gradFilter)
{
    __shared__ float red[32][32];
    const int cx = (int)threadIdx.x & 31, py = (int)threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + cx;
    float s = 0.f;
    if (j < total)
        for (int p = py; p < nparts; p += 32) s += partial[(size_t)p * total + j];
    red[py][cx] = s;
    __syncthreads();
    if (py == 0 && j < total) {
        for (int k = 1; k < 32; k++) s += red[k][cx];
        gradFilter[j] = s;
    }
}

static int tile_dims_ok(int B, int N, int M, int F, int C, int r, int ucap, const char* who)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M > 0 && F > 0 && C > 0 && ucap >= 4 && ucap <= 252 && ucap % 4 == 0, "%s: bad dims B=%d N=%d M=%d F=%d C=%d ucap=%d", who, B, N,
                  M, F, C, ucap);
    SPH3D_REQUIRE((r == 1 || r == 2) && C % 4 == 0 && F <= 63,
                  "%s: the tiled kernels need r in {1,2}, C %% 4 == 0, F <= 63 (got r=%d C=%d F=%d)", who, r, C, F);
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

template <int R, int VEC, int NW>
static int launch_tile_fwd(int B, int N, int M, int F, int C, int ucap, const int* order, const int* desc, const int* ulist,
                           const int* pbounds, const int* slotw, const int* nn_count, const int* bounds, const int* key,
                           const float* input, const float* filter, float* output, hipStream_t st)
{
    constexpr int SLC = 32 * VEC, RPI = 64 / (SLC / 4);
    const int capR = (ucap + RPI - 1) / RPI * RPI;
    const size_t lds = sizeof(float) * ((size_t)(capR + RPI) * SLC + (size_t)F * SLC * R);
    auto kern = dwconv_tile_fwd<R, VEC, NW>;
    int rc = set_lds((const void*)kern, lds, "DepthwiseConv3dTiled");
    if (rc) return rc;
    const int nslices = (C + SLC - 1) / SLC;
    const int cands = (M + kTileP - 1) / kTileP;
    const int per_cu = (int)((160 * 1024) / lds) < (2048 / (NW * 64)) ? (int)((160 * 1024) / lds) : (2048 / (NW * 64));
    const int W = tile_W(nslices, per_cu < 1 ? 1 : per_cu);
    hipLaunchKernelGGL(kern, dim3(8 * W * nslices), dim3(NW * 64), lds, st, B, N, M, F, C, cands, nslices, W, ucap, order, desc,
                       ulist, pbounds, slotw, nn_count, bounds, key, input, filter, output);
    return check_launch("sph3d_depthwise_conv3d_tiled");
}

template <int R, int VEC, int NW, int MAXF>
static int launch_tile_bwd(int B, int N, int M, int F, int C, int ucap, const int* order, const int* desc, const int* ulist,
                           const float* uscale, const int* pbounds, const int* slotw, const int* offsets, const int* ent_key,
                           const float* ent_scale, const float* input, const float* filter, const float* grad_output, float* grad_input,
                           float* grad_filter, float* partial, int W, hipStream_t st)
{
    constexpr int SLW = 32 * VEC, RPI = 64 / (SLW / 4);
    const int CR = C * R;
    const int capR = (ucap + RPI - 1) / RPI * RPI;
    const size_t rowF = (size_t)(capR + RPI) * SLW, tabF = (size_t)F * SLW;
    const size_t lds = sizeof(float) * ((rowF > tabF ? rowF : tabF) + tabF);
    auto kern = dwconv_tile_bwd<R, VEC, NW, MAXF>;
    int rc = set_lds((const void*)kern, lds, "DepthwiseConv3dGradTiled");
    if (rc) return rc;
    const int nslices = (CR + SLW - 1) / SLW;
    const int cands = (N + kTileP - 1) / kTileP;
    hipLaunchKernelGGL(kern, dim3(8 * W * nslices), dim3(NW * 64), lds, st, B, N, M, F, C, cands, nslices, W, ucap, order, desc,
                       ulist, uscale, pbounds, slotw, offsets, ent_key, ent_scale, input, filter, grad_output, grad_input, partial);
    const int total = F * CR;
    hipLaunchKernelGGL(reduce_slabs, dim3((total + 31) / 32), dim3(1024), 0, st, 8 * W, total, partial, grad_filter);
    return check_launch("sph3d_depthwise_conv3d_grad_tiled");
}

static int bwd_tile_W(int C, int r, int variant)
{
    const int VEC = (variant / 100 == 4) ? 4 : 2;
    const int nslices = (C * r + 32 * VEC - 1) / (32 * VEC);
    return tile_W(nslices, VEC == 4 ? 1 : 2);
}

}  // namespace sph3d

using namespace sph3d;

// variant = 100 * VEC + NW (channels per lane, waves per workgroup); 0 = default
extern "C" int sph3d_depthwise_conv3d_tiled(int B, int N, int M, int F, int C, int r, int ucap, int variant,
                                            const int* order, const int* tile_desc, const int* tile_rows,
                                            const int* pbounds, const int* slot_words, const int* nn_count,
                                            const int* bounds, const int* key,
                                            const float* input, const float* filter, float* output, sph3d_stream_t stream)
{
    int rc = tile_dims_ok(B, N, M, F, C, r, ucap, "DepthwiseConv3dTiled");
    if (rc) return rc;
    if (B == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    if (variant == 0) variant = 208;
#define SPH3D_FWD(RR, VV, WW)                                                                                              \
    if (r == RR && variant == (VV * 100 + WW))                                                                            \
        return launch_tile_fwd<RR, VV, WW>(B, N, M, F, C, ucap, order, tile_desc, tile_rows, pbounds, slot_words, nn_count, bounds, \
                                           key, input, filter, output, st);
    SPH3D_FWD(2, 2, 8) SPH3D_FWD(1, 2, 8) SPH3D_FWD(2, 2, 16) SPH3D_FWD(1, 2, 16)
    SPH3D_FWD(2, 4, 8) SPH3D_FWD(1, 4, 8) SPH3D_FWD(2, 4, 16) SPH3D_FWD(1, 4, 16)
#undef SPH3D_FWD
    set_error("DepthwiseConv3dTiled: unknown variant %d", variant);
    return SPH3D_EINVAL;
}

extern "C" size_t sph3d_depthwise_conv3d_grad_tiled_workspace(int F, int C, int r, int variant)
{
    if (variant == 0) variant = 208;
    return sizeof(float) * (size_t)8 * bwd_tile_W(C, r, variant) * F * C * r;
}

extern "C" int sph3d_depthwise_conv3d_grad_tiled(int B, int N, int M, int F, int C, int r, int ucap, int variant,
                                                 const int* order, const int* tile_desc, const int* tile_rows,
                                                 const float* tile_row_scale, const int* pbounds, const int* slot_words,
                                                 const int* offsets, const int* ent_key, const float* ent_scale,
                                                 const float* input, const float* filter, const float* grad_output,
                                                 float* grad_input, float* grad_filter,
                                                 void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    int rc = tile_dims_ok(B, N, M, F, C, r, ucap, "DepthwiseConv3dGradTiled");
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (B == 0) return check_hip(hipMemsetAsync(grad_filter, 0, sizeof(float) * (size_t)F * C * r, st), "conv3d grad: memset");
    if (variant == 0) variant = 208;
    const size_t need = sph3d_depthwise_conv3d_grad_tiled_workspace(F, C, r, variant);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("DepthwiseConv3dGradTiled: workspace %zu B < required %zu B", workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    const int W = bwd_tile_W(C, r, variant);
#define SPH3D_BWD(RR, VV, WW)                                                                                               \
    if (r == RR && variant == (VV * 100 + WW)) {                                                                           \
        if (F <= 33)                                                                                                       \
            return launch_tile_bwd<RR, VV, WW, 33>(B, N, M, F, C, ucap, order, tile_desc, tile_rows, tile_row_scale, pbounds, slot_words, offsets, \
                                                   ent_key, ent_scale, input, filter, grad_output, grad_input, grad_filter,  \
                                                   (float*)workspace, W, st);                                              \
        return launch_tile_bwd<RR, VV, WW, 63>(B, N, M, F, C, ucap, order, tile_desc, tile_rows, tile_row_scale, pbounds, slot_words, offsets,  \
                                               ent_key, ent_scale, input, filter, grad_output, grad_input, grad_filter,      \
                                               (float*)workspace, W, st);                                                  \
    }
    SPH3D_BWD(2, 2, 8) SPH3D_BWD(1, 2, 8) SPH3D_BWD(2, 2, 16) SPH3D_BWD(1, 2, 16)
    SPH3D_BWD(2, 4, 8) SPH3D_BWD(1, 4, 8)
#undef SPH3D_BWD
    set_error("DepthwiseConv3dGradTiled: unknown variant %d", variant);
    return SPH3D_EINVAL;
}
