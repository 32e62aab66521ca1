// conv3d.hip — depthwise spherical convolution (forward + gradients) for gfx950.
//
// Replaces depthwise_conv3d_forward / depthwise_input_backward / depthwise_filter_backward
// (tf_ops/convolution/tf_conv3d_gpu.cu:7-101) and their launchers (:107-140).
//
// MI355X design:
//   * One WAVEFRONT per output point (the reference: one thread per output CHANNEL, every
//     channel-thread re-reading the neighbour/bin lists and doing K read-modify-writes on global
//     memory).  The point's neighbour ids and bin ids are fetched ONCE as two coalesced 256-B reads (lane k
//     holds slot k) and broadcast lane->scalar with v_readlane eight at a time, so eight gather addresses are
//     scalar, eight feature-row gathers (each one coalesced wave read, float2/float4 per lane) and eight LDS
//     filter reads are in flight before the first FMA.
//   * Lanes span output channels, 4 consecutive channels per lane: accumulate in registers, one
//     coalesced float4 store per point.  A "slice" is 256 output channels; wider layers loop slices.
//   * The filter table slice (F x 256 floats = 33 KB at F=33) lives in LDS, read as ds_read_b128.
//   * Backward fuses both gradients in ONE pass over the TRANSPOSED graph (graph.hip): grad_input is a
//     gather (registers, one store per element, no float atomics to memory), grad_filter accumulates in
//     per-lane REGISTER tables (one row per bin, compile-time indexed: the transposed graph is sorted by
//     (source, bin)), summed across waves / workgroups at the end.  The reference scattered grad_input with
//     one global atomicAdd per (point, neighbour, channel) and re-ran the whole gather ceil(F*C*r/12288)
//     times for grad_filter (tf_conv3d_gpu.cu:51, 126-139).
//   * Workgroups of one cloud are dealt to one XCD (xcd_decode) so the cloud's feature rows
//     (N*C*4 B, 4 MiB at N=8192,C=128) stay in that XCD's 4 MiB L2.
//   * Numerics: sum_k in*filt in fp32 FMA order k = 0..cnt-1, one division by cnt at the end (the
//     reference divides every term); agreement with the oracle is ~1e-7 relative, bound 1e-5.
#include <cstdlib>
#include "common.hpp"

namespace sph3d {

constexpr int kSlice = 256;       // output channels per wave pass (64 lanes x 4)
constexpr int kFwdPointsPerWG = 32;
#ifndef SPH3D_FWD_SB
#define SPH3D_FWD_SB 4
#endif
constexpr int kFwdSB = SPH3D_FWD_SB;   // dwconv_fwd_multi: wave loads (each EPL neighbour rows) issued together
#ifndef SPH3D_FWD_WAVES
#define SPH3D_FWD_WAVES 4
#endif
constexpr int kMultiWaves = SPH3D_FWD_WAVES;          // dwconv_fwd_multi: waves per workgroup (they share one LDS filter table)
constexpr int kMultiPoints = 8 * kMultiWaves;         // output points per workgroup
constexpr int kBatch = 8;               // dwconv_fwd_row: neighbours whose gathers are issued together

// ------------------------------------------------------------------------------------------
// forward, vectorised: R = depth multiplier (1 or 2), C % 4 == 0.
// A lane owns FOUR INPUT channels (one 16-B gather per edge) and their 4R output channels; an edge therefore needs
// LPE = C/4 lanes (16 / 32 / 64 for C = 64 / 128 / >= 256) and one wave load instruction fetches EPL = 64/LPE
// different neighbour rows, lane group g taking the edges k = g (mod EPL) of the point.  Round-1 counters that
// forced this shape (C = 128, r = 2, one edge per load, float2 per lane): the CU's vector L1 was 83 % occupied
// (59 % processing + 24 % stalled on pending lines) — the L1 spends 16 cycles per wave load whether the lanes ask
// for 8 or 16 bytes and whether or not half of them repeat lane 0's address; the L2 and HBM were far from busy.
// The group partial sums are added across lane groups at the end (ds_swizzle-free: two ds_bpermute rounds at most).
// ------------------------------------------------------------------------------------------
template <int R, int LPE, int SB>
__global__ __launch_bounds__(64 * kMultiWaves) void dwconv_fwd_multi(
    int B, int N, int M, int F, int C, int K, int mblocks, int nslices,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount, const int* __restrict__ binIndex,
    const float* __restrict__ input, const float* __restrict__ filter, float* __restrict__ output)
{
    extern __shared__ __attribute__((aligned(16))) float lfilt[];   // [F + 1][R][SLI]: compile-time strides
    constexpr int EPL = 64 / LPE;               // edges per wave load
    constexpr int NO = 4 * R;                   // output channels per lane
    constexpr int SLI = 4 * LPE;                // input channels per slice
    constexpr int FSTB = SLI * R * 4;           // bytes per filter row in LDS
    const int CR = C * R;
    int b, part;
    xcd_decode((int)blockIdx.x, B, mblocks * nslices, b, part);
    if (b < 0) return;
    const int slice = part / mblocks;
    const int mb = part - slice * mblocks;
    const int ci0 = slice * SLI;                                     // first input channel of the slice
    const int SLi = (C - ci0) < SLI ? (C - ci0) : SLI;               // multiple of 4
    const int SLo = SLi * R;

    // stage the filter slice, de-interleaved so that the lanes' 16-B reads are contiguous (conflict-free ds_read_b128):
    // quad q of lane li (its output channels 4q..4q+3, i.e. slice columns li*4R + 4q + j) lives at [f][q][li*4 + j]
    for (int e = threadIdx.x * 4; e < F * SLo; e += blockDim.x * 4) {
        const int f = e / SLo;
        const int cl = e - f * SLo;                 // multiple of 4
        const int l4 = cl / (4 * R), q = (cl >> 2) % R;
        *reinterpret_cast<float4*>(&lfilt[f * (SLI * R) + q * SLI + l4 * 4]) =
            *reinterpret_cast<const float4*>(&filter[(size_t)f * CR + ci0 * R + cl]);
    }
    // row F: all zeros, the filter row of padding slots (see dwconv_fwd_row)
    for (int e = threadIdx.x; e < SLI * R; e += blockDim.x) lfilt[F * (SLI * R) + e] = 0.f;
    __syncthreads();

    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int g = lane / LPE;                   // edge group of this lane
    const int li = lane - g * LPE;
    const bool act = li * 4 < SLi;
    const int cic = act ? li * 4 : 0;           // clamped copy for branch-free loads
    const int m_begin = mb * kMultiPoints;
    const int m_end = (m_begin + kMultiPoints) < M ? (m_begin + kMultiPoints) : M;
    // Instruction diet (round 3; the kernel is bound by instruction ISSUE, ~4.4 SIMD cycles per wave instruction, not by
    // bytes: profiles/r03_pmc_sq_*): uniform row base + 32-bit per-lane byte offsets (no 64-bit vector adds), ONE cross-lane
    // hand-over per edge (neighbour id and bin id packed in a word: id in the low 24 bits, so v_mul_u32_u24 turns the word
    // into the row's byte offset without masking it), the second filter quad at an immediate LDS offset, one reciprocal per
    // point instead of a correctly rounded division per output.
    const char* inb = reinterpret_cast<const char*>(input + (size_t)b * N * C + ci0);
    const unsigned cicb = (unsigned)cic * 4u;
    const unsigned rowb = (unsigned)C * 4u;     // bytes per input row (< 2^24: launcher)
    const char* lfb = reinterpret_cast<const char*>(lfilt);

    for (int m = m_begin + wave; m < m_end; m += kMultiWaves) {
        const size_t row = (size_t)b * M + m;
        const int cnt = uniform(nnCount[row]);
        float acc[NO];
#pragma unroll
        for (int v = 0; v < NO; v++) acc[v] = 0.f;
        for (int kt = 0; kt < cnt; kt += 64) {
            // the row's neighbour ids and bin ids: ONE coalesced 256-B read each (lane k holds slot kt + k) ...
            const int myk = kt + lane;
            const int kn = (cnt - kt) < 64 ? (cnt - kt) : 64;
            const int mykc = myk < cnt ? myk : kt;                 // padding lanes: the chunk's first neighbour times the zero row
            const int idxv = nnIndex[row * K + mykc];
            int binv = binIndex[row * K + mykc];
            binv = binv < 0 ? 0 : (binv >= F ? F - 1 : binv);     // out-of-range bin ids: same clamp as the transposed graph
            binv = myk < cnt ? binv : F;
            const unsigned pk = ((unsigned)idxv & 0xffffffu) | ((unsigned)binv << 24);      // N <= 2^24, F <= 254: launcher; an id outside [0, 2^24) must not reach the bin field
            static_assert(64 % (SB * EPL) == 0, "batches must tile the 64-edge chunk (no index clamps)");
            // ... then consumed SB wave loads (SB*EPL edges) at a time: all their gathers are in flight before the
            // first FMA (the kernel is latency-bound otherwise)
            for (int k0 = 0; k0 < kn; k0 += SB * EPL) {
                float4 x[SB];
                unsigned fo[SB];
#pragma unroll
                for (int u = 0; u < SB; u++) {
                    // lane group g takes edge k0 + u*EPL + g.  Measured: ds_bpermute (0.145 ms at C = 64) beats EPL v_readlane
                    // broadcasts + per-lane selects (0.203 ms)
                    const int kq = k0 + u * EPL + g;               // <= 63: padding slots hold a real row and the zero filter row
                    const unsigned p = (unsigned)__shfl((int)pk, kq);
                    const unsigned off = __umul24(p, rowb) + cicb;                 // low 24 bits of p = neighbour id
                    fo[u] = __umul24(p >> 24, (unsigned)FSTB) + cicb;
                    x[u] = *reinterpret_cast<const float4*>(inb + off);
                }
#pragma unroll
                for (int u = 0; u < SB; u++) {
                    const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        const float4 w = *reinterpret_cast<const float4*>(lfb + fo[u] + q * (SLI * 4));
                        // outputs 4q..4q+3 of this lane belong to input channels (4q + j) / R
                        acc[4 * q + 0] = fmaf(xs[(4 * q + 0) / R], w.x, acc[4 * q + 0]);
                        acc[4 * q + 1] = fmaf(xs[(4 * q + 1) / R], w.y, acc[4 * q + 1]);
                        acc[4 * q + 2] = fmaf(xs[(4 * q + 2) / R], w.z, acc[4 * q + 2]);
                        acc[4 * q + 3] = fmaf(xs[(4 * q + 3) / R], w.w, acc[4 * q + 3]);
                    }
                }
            }
        }
        // add the partial sums of the EPL lane groups (lanes li, li + LPE, ...)
#pragma unroll
        for (int o = LPE; o < 64; o <<= 1)
#pragma unroll
            for (int v = 0; v < NO; v++) acc[v] += __shfl_xor(acc[v], o);
        if (act && g == 0) {
            // cnt == 0 only for rows the caller marked empty: output 0.  One reciprocal per point (1 ulp from the per-term
            // divisions of the reference; the bound on the op is 1e-5)
            const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
            float* op = &output[row * CR + (size_t)(ci0 + li * 4) * R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                float4 o;
                o.x = acc[4 * q + 0] * inv;
                o.y = acc[4 * q + 1] * inv;
                o.z = acc[4 * q + 2] * inv;
                o.w = acc[4 * q + 3] * inv;
                *reinterpret_cast<float4*>(&op[4 * q]) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward, one edge per wave load (C >= 128): R = depth multiplier (1 or 2), CR % 4 == 0
// ------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void dwconv_fwd_row(
    int B, int N, int M, int F, int C, int K, int mblocks, int nslices,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount, const int* __restrict__ binIndex,
    const float* __restrict__ input, const float* __restrict__ filter, float* __restrict__ output,
    const int* __restrict__ order, const float* __restrict__ input2 = nullptr, int Ca = 0)
{
    extern __shared__ __attribute__((aligned(16))) float lfilt[];   // [F][SL]
    const int CR = C * R;
    int b, part;
    xcd_decode((int)blockIdx.x, B, mblocks * nslices, b, part);
    if (b < 0) return;
    const int slice = part / mblocks;
    const int mb = part - slice * mblocks;
    const int slice0 = slice * kSlice;
    const int SL = (CR - slice0) < kSlice ? (CR - slice0) : kSlice;   // multiple of 4

    // stage the filter slice: lfilt[f][cl] = filter[f*CR + slice0 + cl]
    for (int e = threadIdx.x * 4; e < F * SL; e += blockDim.x * 4) {
        const int f = e / SL;
        const int cl = e - f * SL;
        *reinterpret_cast<float4*>(&lfilt[e]) = *reinterpret_cast<const float4*>(&filter[(size_t)f * CR + slice0 + cl]);
    }
    // row F: all zeros — the filter row of padding slots (batches of eight run past the neighbour count without clamps or
    // conditional FMAs: a padding slot multiplies the row's FIRST neighbour by exactly 0)
    for (int e = threadIdx.x; e < SL; e += blockDim.x) lfilt[F * SL + e] = 0.f;
    __syncthreads();

    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int cl0 = lane * 4;
    const bool act = cl0 < SL;
    const int cin0 = (slice0 + cl0) / R;      // first input channel of this lane
    const int clc = act ? cl0 : 0;            // clamped copies for branch-free loads
    const int cinc = act ? cin0 : slice0 / R;
    (void)cin0;
    const int m_begin = mb * kFwdPointsPerWG;
    const int m_end = (m_begin + kFwdPointsPerWG) < M ? (m_begin + kFwdPointsPerWG) : M;
    // input2 != nullptr: the input is the channel concatenation [input (Ca channels) | input2 (C - Ca)] of two tensors that were
    // never concatenated (sph3d_depthwise_conv3d_cat: a decoder level's un-pooled features and the encoder's skip features).
    // The launcher guarantees that a 256-output slice lies inside ONE of them, so the choice is per workgroup: a base
    // pointer, a row stride and a channel shift.
    const bool second = input2 != nullptr && (slice0 / R) >= Ca;
    const int Cs = input2 == nullptr ? C : (second ? C - Ca : Ca);       // row stride of the source tensor
    const int cshift = second ? Ca : 0;
    const float* inb = (second ? input2 : input) + (size_t)b * N * Cs - cshift;

    for (int mi = m_begin + wave; mi < m_end; mi += 4) {
        // optional processing order: measured in round 1 (Morton order of the output points): no gain, the rows
        // come from L2 either way (profiles/); kept as a hook for the LDS-tiled variant
        const int m = order ? uniform(order[(size_t)b * M + mi]) : mi;
        const size_t row = (size_t)b * M + m;
        const int cnt = uniform(nnCount[row]);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int kt = 0; kt < cnt; kt += 64) {
            // the row's neighbour ids and bin ids: ONE coalesced 256-B read each (lane k holds slot kt + k) ...
            const int myk = kt + lane;
            const int kn = (cnt - kt) < 64 ? (cnt - kt) : 64;
            const int mykc = myk < cnt ? myk : kt;                 // padding lanes: the chunk's first neighbour (a real one) ...
            const int idxv = nnIndex[row * K + mykc];
            int binv = binIndex[row * K + mykc];
            binv = binv < 0 ? 0 : (binv >= F ? F - 1 : binv);     // out-of-range bin ids: same clamp as the transposed graph
            binv = myk < cnt ? binv : F;                           // ... times the zero row
            // element offsets of the input row and of the filter row, once per 64 edges (N * C < 2^32: checked by the launcher)
            const unsigned noff = (unsigned)idxv * (unsigned)Cs;
            const int foff = binv * (SL >> 2);                     // in float4 units: keeps the LDS read a 16-byte aligned ds_read_b128
            // ... then consumed eight at a time: 8 lane->scalar broadcasts, 8 independent row gathers and 8 filter
            // reads are in flight before the first FMA (the kernel is latency-bound otherwise).  kBatch divides 64, so
            // k8 + u <= 63: no clamp
            for (int k8 = 0; k8 < kn; k8 += kBatch) {
                unsigned n[kBatch];
                int f[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    n[u] = (unsigned)__builtin_amdgcn_readlane((int)noff, k8 + u);
                    f[u] = __builtin_amdgcn_readlane(foff, k8 + u);
                }
                float4 w[kBatch];
                float4 x[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    // no per-lane branch here: inactive lanes (slices narrower than 256) read lane 0's columns, so the
                    // eight loads stay in one basic block and are all in flight together
                    w[u] = reinterpret_cast<const float4*>(lfilt)[f[u] + (clc >> 2)];
                    const float* __restrict__ rp = inb + n[u];          // wave-uniform row address (scalar registers)
                    if (R == 2) {
                        const float2 t = *reinterpret_cast<const float2*>(&rp[(unsigned)cinc]);
                        x[u] = make_float4(t.x, t.x, t.y, t.y);
                    } else {
                        x[u] = *reinterpret_cast<const float4*>(&rp[(unsigned)cinc]);
                    }
                }
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    acc.x = fmaf(x[u].x, w[u].x, acc.x);
                    acc.y = fmaf(x[u].y, w[u].y, acc.y);
                    acc.z = fmaf(x[u].z, w[u].z, acc.z);
                    acc.w = fmaf(x[u].w, w[u].w, acc.w);
                }
            }
        }
        if (act) {
            // cnt == 0 only for rows the caller marked empty: output 0.  One reciprocal per point, not a correctly rounded
            // division per output (4 x ~11 instructions per lane and point on an issue-bound kernel)
            const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
            float4 o;
            o.x = acc.x * inv;
            o.y = acc.y * inv;
            o.z = acc.z * inv;
            o.w = acc.w * inv;
            *reinterpret_cast<float4*>(&output[row * CR + slice0 + cl0]) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward, generic: any C, r (odd channel counts of the ModelNet plan: C = 35, 67, 131)
// lane owns output channels slice0 + lane + 64*t, t < 4
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_fwd_generic(
    int B, int N, int M, int F, int C, int r, int K, int mblocks, int nslices,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount, const int* __restrict__ binIndex,
    const float* __restrict__ input, const float* __restrict__ filter, float* __restrict__ output)
{
    extern __shared__ __attribute__((aligned(16))) float lfilt[];   // [F][SL]
    const int CR = C * r;
    int b, part;
    xcd_decode((int)blockIdx.x, B, mblocks * nslices, b, part);
    if (b < 0) return;
    const int slice = part / mblocks;
    const int mb = part - slice * mblocks;
    const int slice0 = slice * kSlice;
    const int SL = (CR - slice0) < kSlice ? (CR - slice0) : kSlice;

    for (int e = threadIdx.x; e < F * SL; e += blockDim.x) {
        const int f = e / SL;
        const int cl = e - f * SL;
        lfilt[e] = filter[(size_t)f * CR + slice0 + cl];
    }
    __syncthreads();

    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    int cl[4], cin[4];
    bool act[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        cl[t] = lane + 64 * t;
        act[t] = cl[t] < SL;
        cin[t] = act[t] ? (slice0 + cl[t]) / r : 0;
        if (!act[t]) cl[t] = 0;
    }
    const int m_begin = mb * kFwdPointsPerWG;
    const int m_end = (m_begin + kFwdPointsPerWG) < M ? (m_begin + kFwdPointsPerWG) : M;
    const float* inb = input + (size_t)b * N * C;

    for (int m = m_begin + wave; m < m_end; m += 4) {
        const size_t row = (size_t)b * M + m;
        const int cnt = uniform(nnCount[row]);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int* __restrict__ irow = nnIndex + row * K;   // wave-uniform address -> scalar loads
        const int* __restrict__ brow = binIndex + row * K;
        {
#pragma unroll 4
            for (int kk = 0; kk < cnt; kk++) {
                const int n = irow[kk];
                int f = brow[kk];
                f = f < 0 ? 0 : (f >= F ? F - 1 : f);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float x = inb[(size_t)n * C + cin[t]];
                    const float w = lfilt[f * SL + cl[t]];
                    acc[t] = fmaf(x, w, acc[t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (act[t]) output[row * CR + slice0 + cl[t]] = cnt > 0 ? acc[t] / (float)cnt : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward over the TRANSPOSED graph (graph.hip): one wave per source point n, NO float atomics in the loop.
// Per in-edge (m, scale) of segment (n, f): one coalesced gather of grad_out[b,m, V channels/lane]; per segment one
// LDS read of the filter row; with g = grad_out * scale (scale = 1/nn_count[m]):
//     grad_in[b,n,c]     += sum_rho g * filt[f,c,rho]     registers, stored once per element
//     grad_filt[f,c,rho] += g * in[b,n,c]                 REGISTERS: every lane keeps a private [MAXF][V]
//                                                         accumulator; the transposed graph is sorted by
//                                                         (n, bin), so the kernel walks the F segments of a
//                                                         source in a fully unrolled loop: acc[f] is a fixed
//                                                         register, the filter row is read once per segment
//                                                         ("gather / segment-sum").
// Round-1 measurement that forced this shape: the same loop with the table in LDS updated by ds_add_f32
// ran 12.4 ms at (B=16, N=8192, C=128, r=2); without those LDS float atomics 1.4 ms — ds_add_f32 retires
// roughly one LANE per 4 cycles on gfx950.  Global fp32 atomics (the reference's scheme) were 9.0 ms.
// At the end the 4 waves of a workgroup sum their accumulators through LDS with plain reads/writes (taking
// turns), and the workgroup writes ONE partial table to a workspace slab; reduce_filter_partials adds the slabs.
// V = channels per lane: 4 for F <= 33 (the [8,2,2] and [8,2,1] kernels), 2 for F <= 65 ([8,2,3] = 49 bins).
// Edges are consumed four at a time: four scalar key loads and four independent row gathers are issued before
// any is used (memory-level parallelism; the kernel is latency-bound at 2 waves/SIMD otherwise).
// ------------------------------------------------------------------------------------------
constexpr int kBwdTWaves = 4;
constexpr int kBwdTPointsPerWG = 64;     // measured: 256 -> 1.90 ms, 128 -> 1.15, 64 -> 0.90, 32 -> 0.91, 16 -> 1.22 (tail / balance vs per-block setup)

#ifndef SPH3D_BWD_NL2
#define SPH3D_BWD_NL2 3     // wave loads per batch of the half-wave form (2 edges per load)
#endif
#ifndef SPH3D_BWD_NL4
#define SPH3D_BWD_NL4 2     // ... of the quarter-wave form (4 edges per load)
#endif
// PARTS = 2 (CR <= 128, V == 4): a grad_out row is at most 32 lanes wide, so the two halves of the wave take ALTERNATE edges of
// a segment (one wave load = two rows) and keep separate partial sums, added across the halves once per source
// (grad_input) / once per launch (the filter accumulators).  PARTS = 4 (CR <= 64): four quarter waves, four rows per load
// (the ModelNet plan's 64-output layers: half of the lanes idle otherwise).
// COMPACT: the accumulator table has one row per ACTIVE bin of the graph (active_bins = [count, ascending list], written by
// sph3d_graph_transpose): with the reference's sqrt-distance quirk the inner radial shell is empty at small radii
// (SURVEY §0.5), so at S3DIS levels 0-2 only 17 of the 33 bins ever occur: 68 accumulator VGPRs instead of 132, four
// workgroups per CU instead of three, and a 17- instead of 33-iteration segment loop.  The host cannot know the count
// without a device->host sync, so BOTH variants are launched and each returns at once unless the count is in its range.
// HUB (round 6, clouds of 32 768 points and more): on one 65 536-point cloud the first-K rule gives a few hundred low-index sources
// thousands of in-edges each (p99 1235, maximum 11 146), and a source is one wave's sequential walk: the launch took as long as its
// biggest hub (1.4 ms at level 0 of the ScanNet-shape plan for 4.2 M edge slots; the 16 x 8192 batch with twice the edges: 0.42).
//   HUB = 1: this kernel, but a source with more than hubT in-edges is not walked: its grad_input row is zeroed and (slice 0)
//            its id appended to hubList = [count, b * N + n ...];
//   HUB = 2: the hub kernel, launched behind it on the same stream: hubW workgroups per slice; kHubGroup workgroups share a listed
//            source, their 4 x kHubGroup waves take every (4 kHubGroup)-th 64-edge chunk of it (the chunk loop clips segments to a
//            chunk anyway), add their grad_input partial sums with float atomics (order not fixed — as the order of a segment's
//            entries already is not) and keep the filter gradient in their accumulators like every other wave: slabs
//            slabBase .. slabBase + hubW - 1 of `partial`.
//   HUB = 0 (every other launch, the headline's among them): neither — the instantiations of rounds 1-5, unchanged.
constexpr int kHubGroup = 8;
template <int R, int V, int MAXF, int PARTS, bool COMPACT, int HUB = 0>
__global__ __launch_bounds__(kBwdTWaves * 64, COMPACT ? 4 : 3) void dwconv_bwd_t_vec(
    int B, int N, int M, int F, int C, int W, int parts, int nslices,
    const int* __restrict__ offsets, const int* __restrict__ entKey, const float* __restrict__ entScale,
    const int* __restrict__ order, const int* __restrict__ activeBins, int compactMax,
    const float* __restrict__ input, const float* __restrict__ filter, const float* __restrict__ gradOutput,
    float* __restrict__ gradInput, float* __restrict__ partial,
    const float* __restrict__ input2 = nullptr, float* __restrict__ gradInput2 = nullptr, int Ca = 0,
    int* __restrict__ hubList = nullptr, int hubT = 0, int slabBase = 0)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SLW = 64 * V;                 // slice width in output channels
    constexpr int VI = (V >= R) ? V / R : 1;    // input channels per lane
    const int CR = C * R;
    // which of the two launches does the work (wave-uniform, decided from device memory: no host sync)
    const int A = activeBins ? uniform(activeBins[0]) : F;
    if (COMPACT ? (A > MAXF) : (activeBins != nullptr && A <= compactMax)) return;
    // persistent workgroups: W per XCD and channel slice, all resident together (launch bounds: 3 or 4 per CU)
    // (HUB == 2: W workgroups per slice in all, numbered w = 0 .. W - 1; xcd is not used)
    const int xcd = HUB == 2 ? 0 : ((int)blockIdx.x & 7);
    const int q = HUB == 2 ? (int)blockIdx.x : ((int)blockIdx.x >> 3);
    const int slice = q / W;
    const int w = q - slice * W;
    const int slice0 = slice * SLW;
    const int SL = (CR - slice0) < SLW ? (CR - slice0) : SLW;     // multiple of V
    float* lfilt = lds;                                           // [F][SL]
    // input2 != nullptr: input / gradInput are the two halves [Ca | C - Ca] of a channel concatenation that was never made
    // (sph3d_depthwise_conv3d_grad_t_cat); a slice lies inside one of them (launcher), so the choice is per workgroup
    const bool second = input2 != nullptr && (slice0 / R) >= Ca;
    const int Cs = input2 == nullptr ? C : (second ? C - Ca : Ca);
    const float* __restrict__ xin = (second ? input2 : input) - (second ? Ca : 0);
    float* __restrict__ gin = (second ? gradInput2 : gradInput) - (second ? Ca : 0);

    for (int e = threadIdx.x * V; e < F * SL; e += blockDim.x * V) {
        const int f = e / SL;
        const int cl = e - f * SL;
#pragma unroll
        for (int v = 0; v < V; v++) lfilt[e + v] = filter[(size_t)f * CR + slice0 + cl + v];
    }
    __syncthreads();

    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int stride = W * kBwdTWaves;          // waves of this XCD that sweep a cloud side by side
    constexpr bool HALF = PARTS > 1;                              // the wave is split into PARTS groups of 64 / PARTS lanes
    constexpr int PL = 64 / PARTS;                                // lanes per group = row width / V
    const int half = HALF ? (lane / PL) : 0;                      // the lane's group
    const int cl0 = (HALF ? (lane % PL) : lane) * V;
    const bool act = cl0 < SL;
    const int cin0 = (slice0 + cl0) / R;
    // per-lane gradient-of-filter accumulators, one row per bin; every index below is a compile-time constant
    // (the bin loop is fully unrolled), so the table lives in VGPRs
    float acc[MAXF][V];
#pragma unroll
    for (int i = 0; i < MAXF; i++)
#pragma unroll
        for (int v = 0; v < V; v++) acc[i][v] = 0.f;

    // COMPACT: lane i holds the bin of accumulator row i (one load for the whole launch; the per-segment scalar load of
    // activeBins[1 + fi] was not hoisted by the compiler and put an s_load + lgkmcnt(0) in front of every segment)
    // (lanes >= A read entry 1, never past the A + 1 ints the list holds: the caller's tensor is F + 1 ints and F may be < MAXF)
    const int abv = COMPACT ? activeBins[1 + ((lane < MAXF && lane < A) ? lane : 0)] : 0;

    // Work items of this XCD: (cloud, part) pairs dealt round-robin; a part is a contiguous range of POSITIONS in the
    // processing order (source_order, or the point index).  Wave gw visits positions gw, gw + stride, ...; the
    // accumulators live across ALL items, so a workgroup writes one partial table for the whole launch.
    // (Measured round 2 and dropped: handing the positions out dynamically through one ticket counter per item.  The
    // static interleave leaves the busiest wave of an XCD with 1.4x the mean number of in-edges — the in-degree is
    // heavy-tailed: level 0 mean 48, sigma 45, max 537, tools/exp_bwd_balance.py — but with tickets of 4 or 8 positions
    // the level-0 kernels ran 0.66 / 0.55 ms against 0.53 / 0.35 ms, with or without waiting for the atomic at once.)
    // (Also dropped: the four waves of a workgroup taking the workgroup's positions from an LDS counter — the extra live
    // scalars of the flattened (item, position) loop pushed the kernel from 119 VGPRs to 128 + 232 B of scratch: 1.16 ms.)
    const int nhub = HUB == 2 ? uniform(hubList[0]) : 0;
    for (int item = HUB == 2 ? w / kHubGroup : xcd; HUB == 2 ? item < nhub : item < B * parts; item += HUB == 2 ? W / kHubGroup : 8) {
    const int hubSrc = HUB == 2 ? uniform(hubList[1 + item]) : 0;
    const int b = HUB == 2 ? hubSrc / N : item / parts;
    const int pi = HUB == 2 ? 0 : item - b * parts;
    // part pi = the positions pi, pi + parts, ... (parts > 1 only when B is not a multiple of 8).  Interleaved, not contiguous
    // ranges: on ONE 65 536-point cloud the first-K rule gives the low indices nearly all in-edges (median in-degree 4, p99
    // 1235, maximum 11 146), and a contiguous first eighth put them all on one XCD
    // wave-uniform base of this cloud's grad_out rows + the lane's column: row gathers are uniform base + uniform row
    // offset (an SGPR pair) + one loop-invariant 32-bit lane offset
    const float* __restrict__ gou = gradOutput + (size_t)b * M * CR + slice0;
    const unsigned cla = (unsigned)(act ? cl0 : 0);
    const int* __restrict__ offb = offsets + (size_t)b * ((size_t)N * F + 1);
    for (int p = HUB == 2 ? 0 : pi + parts * (w * kBwdTWaves + wave); HUB == 2 ? p < 1 : p < N; p += HUB == 2 ? 1 : parts * stride) {
        const int n = HUB == 2 ? hubSrc - b * N : (order ? uniform(order[(size_t)b * N + p]) : p);
        // the source's F+1 segment bounds: ONE coalesced read (lane f holds bound f), consumed with v_readlane at
        // compile-time lanes — not 33 dependent scalar loads (measured: they dominated the sparse levels)
        const int* __restrict__ o = offb + (size_t)n * F;
        const int ov0 = o[lane <= F ? lane : F];
        const int ov1 = (MAXF >= 64) ? o[(64 + lane) <= F ? (64 + lane) : F] : 0;
        float xi[VI], xv[V];
#pragma unroll
        for (int u = 0; u < VI; u++) xi[u] = xin[((size_t)b * N + n) * Cs + (act ? cin0 : slice0 / R) + u];
#pragma unroll
        for (int v = 0; v < V; v++) xv[v] = xi[(V >= R) ? v / R : 0];
        float gi[V];
#pragma unroll
        for (int v = 0; v < V; v++) gi[v] = 0.f;
        // The source's in-edges are one contiguous run [E0, E1) of the (source, bin)-sorted entry arrays.  Their keys
        // and scales are fetched 64 at a time with ONE coalesced vector load each (lane = edge) and handed to the
        // gathers with v_readlane.  Round 2 counters on the scalar version of this loop (s_load of key and scale per
        // edge, 64-bit scalar address arithmetic, clamps): 16.8 scalar-ALU + 2.7 scalar-memory instructions per edge =
        // half of the CU's one-per-cycle scalar issue for the whole kernel, and a dependent s_load -> gather round
        // trip per batch of four edges.
        const int E0 = __builtin_amdgcn_readlane(ov0, 0);
        const int E1 = uniform(o[F]);
        // COMPACT: which accumulator rows have a non-empty segment at this source (about half of the 17 do not): one mask per
        // source instead of two v_readlane + clamps + compare per empty segment
        unsigned long long nonempty = ~0ull;
        if (COMPACT) {
            const int fl = lane < MAXF ? abv : 0;
            const int s0 = __builtin_amdgcn_ds_bpermute(fl << 2, ov0);
            const int s1 = __builtin_amdgcn_ds_bpermute((fl + 1) << 2, ov0);
            nonempty = __ballot(s1 > s0);
        }
        if (HUB == 1 && (E1 - E0) > hubT) {
            // a hub: left to the hub kernel (every slice's wave decides the same from the degree alone; the list holds each source once)
            if (slice == 0 && lane == 0) hubList[1 + atomicAdd(&hubList[0], 1)] = b * N + n;
            if (act && half == 0 && V >= R) {
                float* gp = &gin[((size_t)b * N + n) * Cs + cin0];
#pragma unroll
                for (int u = 0; u < VI; u++) gp[u] = 0.f;
            }
            continue;
        }
        const int cfirst = HUB == 2 ? E0 + 64 * ((w % kHubGroup) * kBwdTWaves + wave) : E0;
        const int cstep = HUB == 2 ? 64 * kHubGroup * kBwdTWaves : 64;
        for (int cb = cfirst; cb < E1; cb += cstep) {
        const int cn = (E1 - cb) < 64 ? (E1 - cb) : 64;
        const int li = lane < cn ? lane : 0;
        // (entScale == NULL: packed entries, common.hpp: row and count in one word, 1 / count divided here: once per lane and chunk)
        const int wkey = entKey[cb + li];
        const unsigned kel = (unsigned)tg_key(wkey, entScale == nullptr) * (unsigned)CR;     // row offset in floats (M * CR < 2^32, checked by the launcher)
        const float svl = entScale == nullptr ? tg_packed_scale(wkey) : entScale[cb + li];
        const float sv = lane < cn ? svl : 0.f;
#pragma unroll
        for (int fi = 0; fi < MAXF; fi++) {
            // the mask exists for the compact table only (MAXF <= 64 there); the full table of the V = 2 plan has MAXF = 65 and a
            // shift by 64 would be undefined — and let the compiler drop bin 64 altogether
            if (fi < (COMPACT ? A : F) && (!COMPACT || ((nonempty >> (fi & 63)) & 1ull))) {
                // COMPACT: row fi of the accumulators belongs to bin activeBins[1 + fi] (wave-uniform, F <= 63)
                const int f = COMPACT ? __builtin_amdgcn_readlane(abv, fi) : fi;
                const int a0 = COMPACT ? __builtin_amdgcn_readlane(ov0, f)
                                       : __builtin_amdgcn_readlane(fi < 64 ? ov0 : ov1, fi & 63);
                const int a1 = COMPACT ? __builtin_amdgcn_readlane(ov0, f + 1)
                                       : __builtin_amdgcn_readlane((fi + 1) < 64 ? ov0 : ov1, (fi + 1) & 63);
                // the part of the segment inside this chunk, as lane numbers
                const int e0 = (a0 > cb ? a0 : cb) - cb;
                const int e1 = (a1 < cb + cn ? a1 : cb + cn) - cb;
                if (e0 < e1) {                               // wave-uniform: most (n, bin) segments are empty or short
                    float sg[V];
#pragma unroll
                    for (int v = 0; v < V; v++) sg[v] = 0.f;
                    // the segment's filter row is read from LDS now, under the gathers, not after them
                    float wr[V];
                    {
                        const float* wrow = &lfilt[f * SL + (act ? cl0 : 0)];
#pragma unroll
                        for (int v = 0; v < V; v++) wr[v] = wrow[v];
                    }
                    if (HALF) {
                        // the two half-waves take alternate edges: lane-half h reads edge e + 2u + h of the chunk through
                        // ds_bpermute (lane number modulo 64, scales masked to the segment: see the full-wave branch)
                        constexpr int NL = PARTS == 2 ? SPH3D_BWD_NL2 : SPH3D_BWD_NL4;          // wave loads per batch = PARTS * NL edges
                        const float svh = ((unsigned)(lane - e0) < (unsigned)(e1 - e0)) ? sv : 0.f;
                        // (exact-count last batches, which pay in the full-wave branch, measured no gain here: 0.305 vs 0.319 ms)
                        // a batch that runs past lane 63 wraps (ds_bpermute takes the lane modulo 64) onto lanes 0.. of the chunk, which
                        // belong to the segment when it is (nearly) the whole chunk: impossible while the batch divides 64 (4 and 8
                        // edges), masked by the edge number in that one batch otherwise (6 edges: segments of 61+ in-edges)
#define SPH3D_BWD_HALF_BATCH(WRAPS)                                                                                  \
    {                                                                                                                \
        const int a4 = ((e + half) << 2);                                                                            \
        unsigned ko[NL];                                                                                             \
        float sc[NL];                                                                                                \
        _Pragma("unroll") for (int u = 0; u < NL; u++) {                                                             \
            ko[u] = (unsigned)__builtin_amdgcn_ds_bpermute(a4 + 4 * PARTS * u, (int)kel);                            \
            sc[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(a4 + 4 * PARTS * u, __float_as_int(svh)));           \
            if (WRAPS) sc[u] = (e + half + PARTS * u) < 64 ? sc[u] : 0.f;                                            \
        }                                                                                                            \
        float g[NL][V];                                                                                              \
        _Pragma("unroll") for (int u = 0; u < NL; u++) {                                                             \
            const float4 t = *reinterpret_cast<const float4*>(&gou[ko[u] + cla]);                                    \
            g[u][0] = t.x; g[u][1 % V] = t.y; g[u][2 % V] = t.z; g[u][3 % V] = t.w;                                  \
        }                                                                                                            \
        _Pragma("unroll") for (int u = 0; u < NL; u++)                                                               \
            _Pragma("unroll") for (int v = 0; v < V; v++) sg[v] = fmaf(g[u][v], sc[u], sg[v]);                       \
    }
                        for (int e = e0; e < e1; e += PARTS * NL) {
                            if ((64 % (PARTS * NL)) != 0 && e + PARTS * NL > 64) SPH3D_BWD_HALF_BATCH(true)
                            else SPH3D_BWD_HALF_BATCH(false)
                        }
#undef SPH3D_BWD_HALF_BATCH
                    } else {
                    // scales of THIS segment's lanes, zero elsewhere: a batch may then run past the segment end (and, as
                    // v_readlane takes the lane number modulo 64, wrap to lanes below e0) without clamps or selects:
                    // the extra gathers hit valid rows (every lane holds a valid key) and are multiplied by exactly 0
                    const float svm = ((unsigned)(lane - e0) < (unsigned)(e1 - e0)) ? sv : 0.f;
                    // four edges at a time (their row gathers are independent and issued together), then the segment's last one
                    // to three edges with exactly that many gathers: a padded last batch made a third of all gathers padding
                    // (8.5 M wave loads for 6.3 M edges at level 0), and these kernels pay 16 L1 cycles per wave load
#define SPH3D_BWD_BATCH(NB)                                                                                          \
    {                                                                                                                \
        float sc[NB];                                                                                                \
        float g[NB][V];                                                                                              \
        _Pragma("unroll") for (int u = 0; u < NB; u++) {                                                             \
            const unsigned ko = (unsigned)__builtin_amdgcn_readlane((int)kel, e + u);                                \
            sc[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(svm), e + u));                           \
            const float* __restrict__ rp = gou + ko;                /* wave-uniform row address */                   \
            /* branch-free (inactive lanes read lane 0's columns): the gathers stay in one basic block */            \
            if (V == 4) {                                                                                            \
                const float4 t = *reinterpret_cast<const float4*>(&rp[cla]);                                         \
                g[u][0] = t.x; g[u][1 % V] = t.y; g[u][2 % V] = t.z; g[u][3 % V] = t.w;                              \
            } else if (V == 2) {                                                                                     \
                const float2 t = *reinterpret_cast<const float2*>(&rp[cla]);                                         \
                g[u][0] = t.x; g[u][1 % V] = t.y;                                                                    \
            } else {                                                                                                 \
                g[u][0] = rp[cla];                                                                                   \
            }                                                                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int u = 0; u < NB; u++)                                                               \
            _Pragma("unroll") for (int v = 0; v < V; v++) sg[v] = fmaf(g[u][v], sc[u], sg[v]);                       \
    }
                    int e = e0;
                    for (; e + 4 <= e1; e += 4) SPH3D_BWD_BATCH(4)
                    const int rem = e1 - e;
                    if (rem == 3) SPH3D_BWD_BATCH(3)
                    else if (rem == 2) SPH3D_BWD_BATCH(2)
                    else if (rem == 1) SPH3D_BWD_BATCH(1)
#undef SPH3D_BWD_BATCH
                    }
#pragma unroll
                    for (int v = 0; v < V; v++) {
                        gi[v] = fmaf(sg[v], wr[v], gi[v]);
                        acc[fi][v] = fmaf(sg[v], xv[v], acc[fi][v]);
                    }
                }
            }
        }
        }   // chunks of 64 in-edges
        if (HALF) {
#pragma unroll
            for (int v = 0; v < V; v++) {
                gi[v] += __shfl_xor(gi[v], 32);
                if (PARTS == 4) gi[v] += __shfl_xor(gi[v], 16);
            }
        }
        if (act && half == 0) {
            float* gp = &gin[((size_t)b * N + n) * Cs + cin0];
            if (V >= R) {
#pragma unroll
                for (int u = 0; u < VI; u++) {
                    float s = 0.f;
#pragma unroll
                    for (int rr = 0; rr < R; rr++) s += gi[u * R + rr];
                    if (HUB == 2) atomicAdd(&gp[u], s);          // the row was zeroed by the HUB == 1 launch in front of this one
                    else gp[u] = s;
                }
            }
        }
    }

    }   // items

    if (HALF) {
#pragma unroll
        for (int i = 0; i < MAXF; i++)
#pragma unroll
            for (int v = 0; v < V; v++) {
                acc[i][v] += __shfl_xor(acc[i][v], 32);
                if (PARTS == 4) acc[i][v] += __shfl_xor(acc[i][v], 16);
            }
    }
    // workgroup reduction of the per-wave accumulators: waves take turns on one [F][SL] LDS table
    __syncthreads();                 // everyone is done reading lfilt
    float* tab = lds;                // reuse: [F][SL]
    // COMPACT: only the rows of bins that occur are summed, written to the slab and read by reduce_filter_partials (which
    // writes zeros for the others): half of the slab traffic at S3DIS radii (17 of 33 bins)
    __shared__ unsigned long long activeMask;
    if (COMPACT) {
        if (threadIdx.x == 0) activeMask = 0ull;
        __syncthreads();
        if ((int)threadIdx.x < A) atomicOr(&activeMask, 1ull << activeBins[1 + threadIdx.x]);
        __syncthreads();
    }
    for (int w = 0; w < kBwdTWaves; w++) {
        if (wave == w && act && half == 0) {
#pragma unroll
            for (int i = 0; i < MAXF; i++) {
                if (i < (COMPACT ? A : F)) {
                    const int f = COMPACT ? uniform(activeBins[1 + i]) : i;
#pragma unroll
                    for (int v = 0; v < V; v++) {
                        float* p = &tab[f * SL + cl0 + v];
                        *p = (w == 0) ? acc[i][v] : (*p + acc[i][v]);
                    }
                }
            }
        }
        __syncthreads();
    }
    float* out = partial + (HUB == 2 ? (size_t)slabBase + w : (size_t)xcd * W + w) * ((size_t)F * CR);
    const unsigned long long am = COMPACT ? activeMask : ~0ull;
    for (int e = threadIdx.x; e < F * SL; e += blockDim.x) {
        const int f = e / SL;
        const int cl = e - f * SL;
        if (!COMPACT || ((am >> f) & 1ull)) out[(size_t)f * CR + slice0 + cl] = tab[e];
    }
}

// grad_filter[j] = sum over the B*nblocks partial tables, fixed order -> deterministic given the partials.
// 1024 threads = 32 outputs x 32 partial-lanes (a one-thread-per-output loop over ~1000 slabs is latency-bound; 8 lanes: 20-30 us).
__global__ __launch_bounds__(1024) void reduce_filter_partials(int nparts, int total, int CR, const float* __restrict__ partial,
                                                              float* __restrict__ gradFilter,
                                                              const int* __restrict__ activeBins, int compactMax, int npartsCompact)
{
    const bool compact = activeBins != nullptr && activeBins[0] <= compactMax;
    if (compact) nparts = npartsCompact;   // the compact launch wrote the slabs — only the rows of the bins that occur
    __shared__ float red[32][32];      // 32 outputs x 32 partial-lanes
    __shared__ unsigned long long activeMask;
    if (compact) {
        if (threadIdx.x == 0) activeMask = 0ull;
        __syncthreads();
        if ((int)threadIdx.x < activeBins[0]) atomicOr(&activeMask, 1ull << activeBins[1 + threadIdx.x]);
        __syncthreads();
    }
    const int cx = (int)threadIdx.x & 31, py = (int)threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + cx;
    float s = 0.f;
    if (j < total && (!compact || ((activeMask >> (j / CR)) & 1ull))) {
        // eight slabs per trip, loads issued together (one per trip = up to 32 dependent L2 round trips per launch)
        for (int p0 = py; p0 < nparts; p0 += 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int p = p0 + u * 32;
                v[u] = partial[(size_t)(p < nparts ? p : py) * total + j];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (p0 + u * 32 < nparts) s += v[u];
        }
    }
    red[py][cx] = s;
    __syncthreads();
    if (py == 0 && j < total) {
        for (int k = 1; k < 32; k++) s += red[k][cx];
        gradFilter[j] = s;
    }
}

// generic transposed backward (any C, r <= 256): lanes own INPUT channels c_base + lane + 64*t,
// a slice is SC = min(C - c_base, max(1, 256 / r)) input channels; LDS table [F][SC*r]
__global__ __launch_bounds__(256) void dwconv_bwd_t_generic(
    int B, int N, int M, int F, int C, int r, int nblocks, int nslices, int sliceC,
    const int* __restrict__ offsets, const int* __restrict__ entKey, const float* __restrict__ entScale,
    const float* __restrict__ input, const float* __restrict__ filter, const float* __restrict__ gradOutput,
    float* __restrict__ gradInput, float* __restrict__ gradFilter)
{
    extern __shared__ __attribute__((aligned(16))) float gtab[];
    const int CR = C * r;
    int b, part;
    xcd_decode((int)blockIdx.x, B, nblocks * nslices, b, part);
    if (b < 0) return;
    const int slice = part / nblocks;
    const int nb = part - slice * nblocks;
    const int c_base = slice * sliceC;
    const int SC = (C - c_base) < sliceC ? (C - c_base) : sliceC;
    const int SW = SC * r;
    for (int e = threadIdx.x; e < F * SW; e += blockDim.x) gtab[e] = 0.f;
    __syncthreads();

    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int n_begin = nb * 64;
    const int n_end = (n_begin + 64) < N ? (n_begin + 64) : N;
    const int* __restrict__ offb = offsets + (size_t)b * ((size_t)N * F + 1);
    for (int n = n_begin + wave; n < n_end; n += 4) {
        const int* __restrict__ o = offb + (size_t)n * F;
        for (int t = 0; t < 4; t++) {
            const int cl = lane + 64 * t;
            if (cl >= SC) break;
            const int cin = c_base + cl;
            const float x = input[((size_t)b * N + n) * C + cin];
            float gi = 0.f;
            for (int f = 0; f < F; f++) {
                const int e0 = o[f], e1 = o[f + 1];
                if (e0 >= e1) continue;
                for (int rr = 0; rr < r; rr++) {
                    const int cout = cin * r + rr;
                    float sg = 0.f;
                    for (int e = e0; e < e1; e++)
                        sg = fmaf(gradOutput[((size_t)b * M + tg_key(entKey[e], entScale == nullptr)) * CR + cout],
                                  entScale == nullptr ? tg_packed_scale(entKey[e]) : entScale[e], sg);
                    gi = fmaf(sg, filter[(size_t)f * CR + cout], gi);
                    unsafeAtomicAdd(&gtab[f * SW + cl * r + rr], sg * x);     // one LDS atomic per (segment, channel)
                }
            }
            gradInput[((size_t)b * N + n) * C + cin] = gi;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < F * SW; e += blockDim.x) {
        const int f = e / SW;
        const int j = e - f * SW;
        const float g = gtab[e];
        if (g != 0.f) unsafeAtomicAdd(&gradFilter[(size_t)f * CR + c_base * r + j], g);
    }
}

static int conv_dims_ok(int B, int N, int M, int F, int C, int r, int K, const char* who)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0 && F > 0 && C > 0 && r > 0 && K > 0,
                  "%s: bad dims B=%d N=%d M=%d F=%d C=%d r=%d K=%d", who, B, N, M, F, C, r, K);
    const size_t slice = (size_t)(C * r < kSlice ? C * r : kSlice);
    SPH3D_REQUIRE((size_t)F * slice * sizeof(float) <= 160 * 1024,
                  "%s: filter table slice F=%d does not fit LDS", who, F);
    return SPH3D_OK;
}

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_depthwise_conv3d(int B, int N, int M, int F, int C, int r, int K,
                                      const int* nn_index, const int* nn_count, const int* bin_index,
                                      const float* input, const float* filter, float* output,
                                      sph3d_stream_t stream)
{
    int rc = conv_dims_ok(B, N, M, F, C, r, K, "DepthwiseConv3d");
    if (rc) return rc;
    if (B == 0 || M == 0) return SPH3D_OK;
    const int CR = C * r;
    const int mblocks = (M + kFwdPointsPerWG - 1) / kFwdPointsPerWG;
    hipStream_t st = as_stream(stream);
    const bool vec = (C % 4 == 0) && (r == 1 || r == 2);
#define SPH3D_BIG_LDS(kern)                                                                                       \
    if (lds > 64 * 1024) {                                                                                        \
        rc = check_hip(hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), \
                       "conv3d: hipFuncSetAttribute");                                                            \
        if (rc) return rc;                                                                                        \
    }
    const int mmblocks = (M + kMultiPoints - 1) / kMultiPoints;
    const dim3 mgrid(xcd_grid(B, mmblocks));
    const bool multi_ok = vec && N <= (1 << 24) && F <= 254 && (unsigned long long)N * C * 4ull + 1024ull < (1ull << 32);
    if (multi_ok && C > 64 && C <= 128) {
        // 65..128 channels: 32 lanes per edge, TWO neighbour rows per wave load.  (Round 1 had measured this shape slower than one
        // row per load, 0.353 vs 0.297 ms at C = 128; with the zero-row padding and vector-side offsets of round 2 it is
        // 0.202 vs 0.240 ms = 16.6 % of the roofline, ahead of the LDS-tiled kernel and without a plan.  64 lanes per edge for
        // C >= 256 — one pass over the edges instead of one per 128-channel slice, but 70 KB of filter table per workgroup —
        // stays behind the row kernel: 0.069 vs 0.060 ms at 2048 x 256.)
        const int nslices = 1;
        const size_t lds = (size_t)(F + 1) * 128 * r * sizeof(float);    // + the zero row of the padding slots; rows of 128 * r
        const dim3 grid(xcd_grid(B, mblocks * nslices));
        if (r == 2) {
            SPH3D_BIG_LDS((dwconv_fwd_multi<2, 32, kFwdSB>))
            hipLaunchKernelGGL((dwconv_fwd_multi<2, 32, kFwdSB>), mgrid, dim3(64 * kMultiWaves), lds, st, B, N, M, F, C, K, mmblocks,
                               nslices, nn_index, nn_count, bin_index, input, filter, output);
        } else {
            SPH3D_BIG_LDS((dwconv_fwd_multi<1, 32, kFwdSB>))
            hipLaunchKernelGGL((dwconv_fwd_multi<1, 32, kFwdSB>), mgrid, dim3(64 * kMultiWaves), lds, st, B, N, M, F, C, K, mmblocks,
                               nslices, nn_index, nn_count, bin_index, input, filter, output);
        }
    } else
    if (multi_ok && C <= 64) {
        // narrow layers: 16 lanes per edge, four neighbour rows per wave load (measured at C = 64, r = 2: 0.254 -> 0.145 ms;
        // at C >= 128 the one-edge-per-load kernel below is faster: 0.297 vs 0.353 ms with two edges per load)
        const int nslices = 1;
        const size_t lds = (size_t)(F + 1) * 64 * r * sizeof(float);     // + the zero row of the padding slots; rows of 64 * r
        const dim3 grid(xcd_grid(B, mblocks * nslices));
        if (r == 2) {
            SPH3D_BIG_LDS((dwconv_fwd_multi<2, 16, kFwdSB>))
            hipLaunchKernelGGL((dwconv_fwd_multi<2, 16, kFwdSB>), mgrid, dim3(64 * kMultiWaves), lds, st, B, N, M, F, C, K, mmblocks,
                               nslices, nn_index, nn_count, bin_index, input, filter, output);
        } else {
            SPH3D_BIG_LDS((dwconv_fwd_multi<1, 16, kFwdSB>))
            hipLaunchKernelGGL((dwconv_fwd_multi<1, 16, kFwdSB>), mgrid, dim3(64 * kMultiWaves), lds, st, B, N, M, F, C, K, mmblocks,
                               nslices, nn_index, nn_count, bin_index, input, filter, output);
        }
    } else if (vec && (unsigned long long)N * C + 256ull < (1ull << 32)) {       // (32-bit row offsets in the kernel)
        const int nslices = (CR + kSlice - 1) / kSlice;
        const int SLmax = CR < kSlice ? CR : kSlice;
        const size_t lds = (size_t)(F + 1) * SLmax * sizeof(float);      // + the zero row of the padding slots
        const dim3 grid(xcd_grid(B, mblocks * nslices));
        if (r == 2) {
            SPH3D_BIG_LDS(dwconv_fwd_row<2>)
            hipLaunchKernelGGL(dwconv_fwd_row<2>, grid, dim3(256), lds, st, B, N, M, F, C, K, mblocks, nslices,
                               nn_index, nn_count, bin_index, input, filter, output, nullptr);
        } else {
            SPH3D_BIG_LDS(dwconv_fwd_row<1>)
            hipLaunchKernelGGL(dwconv_fwd_row<1>, grid, dim3(256), lds, st, B, N, M, F, C, K, mblocks, nslices,
                               nn_index, nn_count, bin_index, input, filter, output, nullptr);
        }
    } else {
        const int nslices = (CR + kSlice - 1) / kSlice;
        const int SLmax = CR < kSlice ? CR : kSlice;
        const size_t lds = (size_t)F * SLmax * sizeof(float);
        const dim3 grid(xcd_grid(B, mblocks * nslices));
        SPH3D_BIG_LDS(dwconv_fwd_generic)
        hipLaunchKernelGGL(dwconv_fwd_generic, grid, dim3(256), lds, st, B, N, M, F, C, r, K, mblocks, nslices,
                           nn_index, nn_count, bin_index, input, filter, output);
    }
    return check_launch("sph3d_depthwise_conv3d");
}

// layout of the transposed graph inside a caller-provided workspace
struct TGraphWs {
    int* offsets; int* key; float* scale; int* active; void* scratch; size_t scratch_bytes;
};
static size_t tgraph_ws_bytes(int B, int N, int M, int K, int F)
{
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return al(sizeof(int) * (size_t)B * ((size_t)N * F + 1)) + 2 * al(sizeof(int) * (size_t)B * M * K) +
           al(sizeof(int) * ((size_t)F + 1)) + al(sph3d_graph_transpose_workspace(B, N, M, K, F));
}
static TGraphWs tgraph_carve(void* ws, int B, int N, int M, int K, int F)
{
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char* p = (char*)ws;
    TGraphWs t;
    t.offsets = (int*)p; p += al(sizeof(int) * (size_t)B * ((size_t)N * F + 1));
    t.key = (int*)p; p += al(sizeof(int) * (size_t)B * M * K);
    t.scale = (float*)p; p += al(sizeof(int) * (size_t)B * M * K);
    t.active = (int*)p; p += al(sizeof(int) * ((size_t)F + 1));
    t.scratch = p; t.scratch_bytes = al(sph3d_graph_transpose_workspace(B, N, M, K, F));
    return t;
}

extern "C" size_t sph3d_scatter_grad_workspace(int B, int N, int M, int K) { return tgraph_ws_bytes(B, N, M, K, 1); }
extern "C" size_t sph3d_depthwise_conv3d_grad_t_workspace(int B, int N, int F, int C, int r);
extern "C" size_t sph3d_depthwise_conv3d_grad_workspace(int B, int N, int M, int F, int C, int r, int K)
{
    return tgraph_ws_bytes(B, N, M, K, F) + sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, C, r);
}

static int vec_plan(int F, int CR, int r, int& V)
{
    if (!(r == 1 || r == 2)) return 0;
    if (F <= 33 && CR % 4 == 0) { V = 4; return 1; }
    if (F <= 65 && CR % 2 == 0) { V = 2; return 1; }
    return 0;
}

// persistent-sweep plan of dwconv_bwd_t_vec: `parts` position ranges per cloud so that (cloud, part) items divide
// evenly over the 8 XCDs, and W workgroups per XCD and channel slice:
//   * large levels: ~64 sources per workgroup (per-workgroup cost: one F x CR partial table), at most 3 per CU;
//   * small levels (N = 128..768: the whole launch is a few hundred sources per XCD): the kernel is a chain of
//     dependent gathers per source, so the sources are spread over enough workgroups to put two on every CU
//     (down to 2 sources per wave) instead of leaving most CUs idle.
#ifndef SPH3D_BWD_FILL
#define SPH3D_BWD_FILL 128
#endif
constexpr int kBwdFillWG = SPH3D_BWD_FILL;   // workgroups per XCD that a small level is spread over (compact kernel: 4 fit a CU)
static void bwd_plan(int B, int N, int nslices, int wg_per_cu, int& parts, int& W)
{
    int g = B & 7;                         // gcd(B, 8)
    g = g == 0 ? 8 : (g & -g);
    parts = 8 / g;
    if (parts > N) parts = 1;
    const long long items_per_xcd = ((long long)B * parts + 7) / 8;
    const long long src = items_per_xcd * ((N + parts - 1) / parts);
    const long long w_work = (src + kBwdTPointsPerWG - 1) / kBwdTPointsPerWG;
    long long w_fill = (wg_per_cu >= 4 ? kBwdFillWG : 64) / (nslices < 1 ? 1 : nslices);
    if (w_fill < 1) w_fill = 1;
    const long long w_min = (src + 2 * kBwdTWaves - 1) / (2 * kBwdTWaves);
    long long w = w_fill < w_min ? w_fill : w_min;
    if (w < w_work) w = w_work;
    const long long cap = 32LL * wg_per_cu;
    W = (int)(w < 1 ? 1 : (w > cap ? cap : w));
}

static int bwd_slices(int F, int CR, int r)
{
    int V = 0;
    if (!vec_plan(F, CR, r, V)) return 0;
    return (CR + 64 * V - 1) / (64 * V);
}

// hub sources (dwconv_bwd_t_vec<..., HUB>): clouds of at least 32 768 points (measured: at 16 384 the two extra launches cost more
// than the few hubs of that level give back, profiles/r06_ab_conv_hub.log), sources with more than 1024 in-edges; kHubWG
// workgroups per channel slice in the hub launch.  SPH3D_BWD_HUB_MIN_N / SPH3D_BWD_HUB_T: read per call (tests, experiments)
constexpr int kHubWG = 1024;       // slabs reserved for the hub launch (its workgroup count: hub_wgs() <= kHubWG)
static int hub_wgs()
{
    const char* e = getenv("SPH3D_BWD_HUB_WG");
    int w = e ? atoi(e) : 1024;
    w = w < kHubGroup ? kHubGroup : (w > kHubWG ? kHubWG : w);
    return w - w % kHubGroup;
}
static int hub_min_n() { const char* e = getenv("SPH3D_BWD_HUB_MIN_N"); return e ? atoi(e) : 32768; }
static int hub_threshold() { const char* e = getenv("SPH3D_BWD_HUB_T"); const int t = e ? atoi(e) : 1024; return t < 1 ? 1 : t; }

extern "C" size_t sph3d_depthwise_conv3d_grad_t_workspace(int B, int N, int F, int C, int r)
{
    int V = 0;
    if (!vec_plan(F, C * r, r, V)) return 0;
    int parts, W;
    bwd_plan(B, N, bwd_slices(F, C * r, r), 4, parts, W);      // sized for the compact variant (4 workgroups per CU)
    // + the hub launch's slabs and the hub list (always: the size must not depend on the per-call switches)
    return sizeof(float) * ((size_t)8 * W + kHubWG) * F * C * r + sizeof(int) * ((size_t)B * N + 4);
}

constexpr int kCompactBins = 17;     // accumulator rows of the compact variant

template <int R, int V, int MAXF, int PARTS>
static int launch_bwd_t_vec(int B, int N, int M, int F, int C, const int* offsets, const int* ent_key,
                            const float* ent_scale, const int* order, const int* active_bins, const float* input,
                            const float* filter, const float* grad_output, float* grad_input, float* grad_filter,
                            float* partial, hipStream_t st, const float* input2 = nullptr, float* grad_input2 = nullptr, int Ca = 0)
{
    const int CR = C * R;
    const int SLW = 64 * V;
    const int nslices = (CR + SLW - 1) / SLW;
    int parts, W, Wc = 0;
    bwd_plan(B, N, nslices, 3, parts, W);
    const int SLmax = CR < SLW ? CR : SLW;
    const size_t lds = (size_t)F * SLmax * sizeof(float);
    // the compact launch exists for the 4-channels-per-lane plans with at most 63 bins (one register of segment bounds)
    const bool compact = active_bins != nullptr && V == 4 && MAXF > kCompactBins && F <= 63;
    auto kern = dwconv_bwd_t_vec<R, V, MAXF, PARTS, false>;
    auto kernc = dwconv_bwd_t_vec<R, V, (V == 4 ? kCompactBins : MAXF), PARTS, (V == 4)>;
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "conv3d: hipFuncSetAttribute");
        if (rc) return rc;
        rc = check_hip(hipFuncSetAttribute((const void*)kernc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "conv3d: hipFuncSetAttribute");
        if (rc) return rc;
    }
    const int* ab = compact ? active_bins : nullptr;
    const int total = F * CR;
    if constexpr (V == 4) {
        if (N >= hub_min_n()) {
            // big clouds: hub sources are left out of the sweep and shared among the waves of the hub launch (see the kernel)
            int pc = parts;
            if (compact) bwd_plan(B, N, nslices, 4, pc, Wc);
            const int wmax = W > Wc ? W : Wc;
            int* hub_list = reinterpret_cast<int*>(partial + ((size_t)8 * wmax + kHubWG) * total);
            int rc = check_hip(hipMemsetAsync(hub_list, 0, sizeof(int), st), "conv3d grad: hub list");
            if (rc) return rc;
            const int T = hub_threshold();
            const int HW = hub_wgs();
            auto kern1 = dwconv_bwd_t_vec<R, V, MAXF, PARTS, false, 1>;
            auto kern2 = dwconv_bwd_t_vec<R, V, MAXF, PARTS, false, 2>;
            auto kernc1 = dwconv_bwd_t_vec<R, V, kCompactBins, PARTS, true, 1>;
            auto kernc2 = dwconv_bwd_t_vec<R, V, kCompactBins, PARTS, true, 2>;
            if (lds > 64 * 1024) {
                for (const void* k : {(const void*)kern1, (const void*)kern2, (const void*)kernc1, (const void*)kernc2}) {
                    rc = check_hip(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "conv3d: hipFuncSetAttribute");
                    if (rc) return rc;
                }
            }
            if (compact) {
                hipLaunchKernelGGL(kernc1, dim3(8 * Wc * nslices), dim3(kBwdTWaves * 64), lds, st, B, N, M, F, C, Wc, pc, nslices, offsets, ent_key,
                                   ent_scale, order, ab, kCompactBins, input, filter, grad_output, grad_input, partial, input2, grad_input2, Ca,
                                   hub_list, T, 0);
                hipLaunchKernelGGL(kernc2, dim3(HW * nslices), dim3(kBwdTWaves * 64), lds, st, B, N, M, F, C, HW, 1, nslices, offsets,
                                   ent_key, ent_scale, order, ab, kCompactBins, input, filter, grad_output, grad_input, partial, input2,
                                   grad_input2, Ca, hub_list, T, 8 * Wc);
            }
            hipLaunchKernelGGL(kern1, dim3(8 * W * nslices), dim3(kBwdTWaves * 64), lds, st, B, N, M, F, C, W, parts, nslices, offsets, ent_key,
                               ent_scale, order, ab, kCompactBins, input, filter, grad_output, grad_input, partial, input2, grad_input2, Ca,
                               hub_list, T, 0);
            hipLaunchKernelGGL(kern2, dim3(HW * nslices), dim3(kBwdTWaves * 64), lds, st, B, N, M, F, C, HW, 1, nslices, offsets, ent_key,
                               ent_scale, order, ab, kCompactBins, input, filter, grad_output, grad_input, partial, input2, grad_input2, Ca,
                               hub_list, T, 8 * W);
            hipLaunchKernelGGL(reduce_filter_partials, dim3((total + 31) / 32), dim3(1024), 0, st, 8 * W + HW, total, CR, partial,
                               grad_filter, ab, kCompactBins, 8 * Wc + HW);
            return check_launch("sph3d_depthwise_conv3d_grad_t");
        }
    }
    if (compact) {
        int pc;
        bwd_plan(B, N, nslices, 4, pc, Wc);
        hipLaunchKernelGGL(kernc, dim3(8 * Wc * nslices), dim3(kBwdTWaves * 64), lds, st, B, N, M, F, C,
                           Wc, pc, nslices, offsets, ent_key, ent_scale, order, ab, kCompactBins, input, filter, grad_output,
                           grad_input, partial, input2, grad_input2, Ca, (int*)nullptr, 0, 0);
    }
    hipLaunchKernelGGL(kern, dim3(8 * W * nslices), dim3(kBwdTWaves * 64), lds, st, B, N, M, F, C,
                       W, parts, nslices, offsets, ent_key, ent_scale, order, ab, kCompactBins, input, filter, grad_output,
                       grad_input, partial, input2, grad_input2, Ca, (int*)nullptr, 0, 0);
    hipLaunchKernelGGL(reduce_filter_partials, dim3((total + 31) / 32), dim3(1024), 0, st, 8 * W, total, CR, partial,
                       grad_filter, ab, kCompactBins, 8 * Wc);
    return check_launch("sph3d_depthwise_conv3d_grad_t");
}

extern "C" int sph3d_depthwise_conv3d_grad_t(int B, int N, int M, int F, int C, int r,
                                             const int* offsets, const int* ent_key, const float* ent_scale,
                                             const int* source_order, const int* active_bins,
                                             const float* input, const float* filter, const float* grad_output,
                                             float* grad_input, float* grad_filter,
                                             void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    int rc = conv_dims_ok(B, N, M, F, C, r, 1, "DepthwiseConv3dGrad");
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const int CR = C * r;
    if (B == 0) return check_hip(hipMemsetAsync(grad_filter, 0, sizeof(float) * (size_t)F * CR, st), "conv3d grad: memset");
    int V = 0;
    // (the vector kernels address a cloud's grad_out rows with 32-bit element offsets)
    if (vec_plan(F, CR, r, V) && (unsigned long long)M * CR + 256ull < (1ull << 32)) {
        const size_t need = sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, C, r);
        if (workspace == nullptr || workspace_bytes < need) {
            set_error("DepthwiseConv3dGrad: workspace %zu B < required %zu B", workspace_bytes, need);
            return SPH3D_EWORKSPACE;
        }
        float* partial = (float*)workspace;
#define SPH3D_GO(RR, VV, MF) \
    return launch_bwd_t_vec<RR, VV, MF, 1>(B, N, M, F, C, offsets, ent_key, ent_scale, source_order, active_bins, input, \
                                        filter, grad_output, grad_input, grad_filter, partial, st)
        // rows of at most 128 (64) outputs: two half (four quarter) waves take alternate edges; a narrower row leaves lanes of
        // each part idle, but fewer than the full-wave form would (the ModelNet plan's 36-, 64- and 68-channel layers)
#define SPH3D_GO_PARTS(RR, PP) \
    return launch_bwd_t_vec<RR, 4, 33, PP>(B, N, M, F, C, offsets, ent_key, ent_scale, source_order, active_bins, input, \
                                           filter, grad_output, grad_input, grad_filter, partial, st)
        if (V == 4 && CR <= 64 && r == 2) SPH3D_GO_PARTS(2, 4);
        if (V == 4 && CR <= 64 && r == 1) SPH3D_GO_PARTS(1, 4);
        if (V == 4 && CR <= 128 && r == 2) SPH3D_GO_PARTS(2, 2);
        if (V == 4 && CR <= 128 && r == 1) SPH3D_GO_PARTS(1, 2);
#undef SPH3D_GO_PARTS
        if (V == 4 && r == 2) SPH3D_GO(2, 4, 33);
        if (V == 4 && r == 1) SPH3D_GO(1, 4, 33);
        if (V == 2 && r == 2) SPH3D_GO(2, 2, 65);
        SPH3D_GO(1, 2, 65);
#undef SPH3D_GO
    }
    // generic path (odd channel counts, other multipliers, F > 64): LDS float atomics, slow but general
    rc = check_hip(hipMemsetAsync(grad_filter, 0, sizeof(float) * (size_t)F * CR, st), "conv3d grad: memset");
    if (rc) return rc;
    SPH3D_REQUIRE(r <= 256, "DepthwiseConv3dGrad: depth multiplier %d > 256 unsupported", r);
    int sliceC = 256 / r;
    if (sliceC < 1) sliceC = 1;
    if (sliceC > C) sliceC = C;
    const int nslices = (C + sliceC - 1) / sliceC;
    const int nblocks = (N + 63) / 64;
    const size_t lds = (size_t)F * sliceC * r * sizeof(float);
    if (lds > 64 * 1024) {
        rc = check_hip(hipFuncSetAttribute((const void*)dwconv_bwd_t_generic, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds), "conv3d: hipFuncSetAttribute");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(dwconv_bwd_t_generic, dim3(xcd_grid(B, nblocks * nslices)), dim3(256), lds, st, B, N, M, F, C, r,
                       nblocks, nslices, sliceC, offsets, ent_key, ent_scale, input, filter, grad_output, grad_input,
                       grad_filter);
    return check_launch("sph3d_depthwise_conv3d_grad_t");
}

extern "C" int sph3d_depthwise_conv3d_grad(int B, int N, int M, int F, int C, int r, int K,
                                           const int* nn_index, const int* nn_count, const int* bin_index,
                                           const float* input, const float* filter, const float* grad_output,
                                           float* grad_input, float* grad_filter,
                                           void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    int rc = conv_dims_ok(B, N, M, F, C, r, K, "DepthwiseConv3dGrad");
    if (rc) return rc;
    const size_t tg = tgraph_ws_bytes(B, N, M, K, F);
    const size_t need = tg + sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, C, r);
    if (B > 0 && (workspace == nullptr || workspace_bytes < need)) {
        set_error("DepthwiseConv3dGrad: workspace %zu B < required %zu B", workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    if (B == 0) return sph3d_depthwise_conv3d_grad_t(B, N, M, F, C, r, nullptr, nullptr, nullptr, nullptr, nullptr, input, filter,
                                                     grad_output, grad_input, grad_filter, nullptr, 0, stream);
    TGraphWs t = tgraph_carve(workspace, B, N, M, K, F);
    rc = sph3d_graph_transpose(B, N, M, K, F, nn_index, nn_count, bin_index, nullptr, t.offsets, t.key, t.scale, t.active,
                               t.scratch, t.scratch_bytes, stream);
    if (rc) return rc;
    return sph3d_depthwise_conv3d_grad_t(B, N, M, F, C, r, t.offsets, t.key, t.scale, nullptr, t.active, input, filter, grad_output,
                                         grad_input, grad_filter, (char*)workspace + tg, workspace_bytes - tg, stream);
}


// ---- the same two ops on a channel concatenation [input_a (Ca) | input_b (Cb)] that is never materialised ---------------------
// (models/SPH3D_s3dis.py:100-104: tf.concat of the un-pooled features and the encoder's skip features feeds the next decoder
// level's separable convolution).  Channel slices of the kernels are 256 outputs = 256 / r inputs wide; Ca * r must be a
// multiple of 256 so that every slice reads one tensor.  Other shapes: SPH3D_EUNSUPPORTED (concatenate and call the plain ops).
static bool cat_ok(int F, int Ca, int Cb, int r)
{
    int V = 0;
    const int C = Ca + Cb;
    return Ca > 0 && Cb > 0 && (r == 1 || r == 2) && C % 4 == 0 && C > 128 && (Ca * r) % kSlice == 0 && vec_plan(F, C * r, r, V) && V == 4;
}

extern "C" int sph3d_depthwise_conv3d_cat_supported(int F, int Ca, int Cb, int r) { return cat_ok(F, Ca, Cb, r) ? 1 : 0; }

extern "C" int sph3d_depthwise_conv3d_cat(int B, int N, int M, int F, int Ca, int Cb, int r, int K, const int* nn_index,
                                          const int* nn_count, const int* bin_index, const float* input_a, const float* input_b,
                                          const float* filter, float* output, sph3d_stream_t stream)
{
    const int C = Ca + Cb;
    int rc = conv_dims_ok(B, N, M, F, C, r, K, "DepthwiseConv3d");
    if (rc) return rc;
    if (!cat_ok(F, Ca, Cb, r) || (unsigned long long)N * C + 256ull >= (1ull << 32)) {
        set_error("DepthwiseConv3d (two inputs): Ca=%d Cb=%d r=%d not covered (Ca*r must be a multiple of 256)", Ca, Cb, r);
        return SPH3D_EUNSUPPORTED;
    }
    if (B == 0 || M == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    const int CR = C * r;
    const int mblocks = (M + kFwdPointsPerWG - 1) / kFwdPointsPerWG;
    const int nslices = (CR + kSlice - 1) / kSlice;
    const size_t lds = (size_t)(F + 1) * kSlice * sizeof(float);
    const dim3 grid(xcd_grid(B, mblocks * nslices));
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            int e = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "conv3d: hipFuncSetAttribute");
            if (e) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, B, N, M, F, C, K, mblocks, nslices, nn_index, nn_count, bin_index, input_a,
                           filter, output, nullptr, input_b, Ca);
        return 0;
    };
    rc = r == 2 ? launch(dwconv_fwd_row<2>) : launch(dwconv_fwd_row<1>);
    if (rc) return rc;
    return check_launch("sph3d_depthwise_conv3d_cat");
}

extern "C" int sph3d_depthwise_conv3d_grad_t_cat(int B, int N, int M, int F, int Ca, int Cb, int r, const int* offsets, const int* ent_key,
                                                 const float* ent_scale, const int* source_order, const int* active_bins,
                                                 const float* input_a, const float* input_b, const float* filter,
                                                 const float* grad_output, float* grad_a, float* grad_b, float* grad_filter,
                                                 void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    const int C = Ca + Cb;
    int rc = conv_dims_ok(B, N, M, F, C, r, 1, "DepthwiseConv3dGrad");
    if (rc) return rc;
    if (!cat_ok(F, Ca, Cb, r) || (unsigned long long)M * C * r + 256ull >= (1ull << 32)) {
        set_error("DepthwiseConv3dGrad (two inputs): Ca=%d Cb=%d r=%d not covered (Ca*r must be a multiple of 256)", Ca, Cb, r);
        return SPH3D_EUNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    if (B == 0) return check_hip(hipMemsetAsync(grad_filter, 0, sizeof(float) * (size_t)F * C * r, st), "conv3d grad: memset");
    const size_t need = sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, C, r);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("DepthwiseConv3dGrad: workspace %zu B < required %zu B", workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    float* partial = (float*)workspace;
    if (r == 2)
        return launch_bwd_t_vec<2, 4, 33, 1>(B, N, M, F, C, offsets, ent_key, ent_scale, source_order, active_bins, input_a, filter,
                                                 grad_output, grad_a, grad_filter, partial, st, input_b, grad_b, Ca);
    return launch_bwd_t_vec<1, 4, 33, 1>(B, N, M, F, C, offsets, ent_key, ent_scale, source_order, active_bins, input_a, filter,
                                             grad_output, grad_a, grad_filter, partial, st, input_b, grad_b, Ca);
}
