// conv3d.hip — depthwise spherical convolution (forward + gradients) for gfx950.
//
// Replaces depthwise_conv3d_forward / depthwise_input_backward / depthwise_filter_backward
// (tf_ops/convolution/tf_conv3d_gpu.cu:7-101) and their launchers (:107-140).
//
// MI355X design:
//   * One WAVEFRONT per output point (the reference: one thread per output CHANNEL, every
//     channel-thread re-reading the neighbour/bin lists and doing K read-modify-writes on global
//     memory).  The point's neighbour ids and bin ids are read through the scalar cache (the row address is wave-uniform)
//     with s_load, so the inner loop's gather addresses
//     are scalar and every gathered feature row is one coalesced wave read (float2/float4 per lane).
//   * Lanes span output channels, 4 consecutive channels per lane: accumulate in registers, one
//     coalesced float4 store per point.  A "slice" is 256 output channels; wider layers loop slices.
//   * The filter table slice (F x 256 floats = 33 KB at F=33) lives in LDS, read as ds_read_b128.
//   * Backward fuses both gradients in one pass over the graph: grad_input by hardware fp32 atomics
//     (global_atomic_add_f32), grad_filter into an LDS table with ds_add_f32 (bank-conflict-free
//     permuted layout), flushed once per workgroup.  The reference re-ran the whole gather
//     ceil(F*C*r/12288) times (tf_conv3d_gpu.cu:126-139).
//   * Workgroups of one cloud are dealt to one XCD (xcd_decode) so the cloud's feature rows
//     (N*C*4 B, 4 MiB at N=8192,C=128) stay in that XCD's 4 MiB L2.
//   * Numerics: sum_k in*filt in fp32 FMA order k = 0..cnt-1, one division by cnt at the end (the
//     reference divides every term); agreement with the oracle is ~1e-7 relative, bound 1e-5.
#include "common.hpp"

namespace sph3d {

constexpr int kSlice = 256;       // output channels per wave pass (64 lanes x 4)
constexpr int kFwdPointsPerWG = 32;
constexpr int kBwdPointsPerWG = 64;

// ------------------------------------------------------------------------------------------
// forward, vectorised: R = depth multiplier (1 or 2), CR % 4 == 0
// ------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void dwconv_fwd_vec(
    int B, int N, int M, int F, int C, int K, int mblocks, int nslices,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount, const int* __restrict__ binIndex,
    const float* __restrict__ input, const float* __restrict__ filter, float* __restrict__ output)
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
    __syncthreads();

    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int cl0 = lane * 4;
    const bool act = cl0 < SL;
    const int cin0 = (slice0 + cl0) / R;      // first input channel of this lane
    const int m_begin = mb * kFwdPointsPerWG;
    const int m_end = (m_begin + kFwdPointsPerWG) < M ? (m_begin + kFwdPointsPerWG) : M;
    const float* inb = input + (size_t)b * N * C;

    for (int m = m_begin + wave; m < m_end; m += 4) {
        const size_t row = (size_t)b * M + m;
        const int cnt = uniform(nnCount[row]);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int* __restrict__ irow = nnIndex + row * K;   // wave-uniform address -> scalar loads
        const int* __restrict__ brow = binIndex + row * K;
        {
#pragma unroll 4
            for (int kk = 0; kk < cnt; kk++) {
                const int n = irow[kk];
                const int f = brow[kk];
                if (act) {
                    const float4 w = *reinterpret_cast<const float4*>(&lfilt[f * SL + cl0]);
                    if (R == 2) {
                        const float2 x = *reinterpret_cast<const float2*>(&inb[(size_t)n * C + cin0]);
                        acc.x = fmaf(x.x, w.x, acc.x);
                        acc.y = fmaf(x.x, w.y, acc.y);
                        acc.z = fmaf(x.y, w.z, acc.z);
                        acc.w = fmaf(x.y, w.w, acc.w);
                    } else {
                        const float4 x = *reinterpret_cast<const float4*>(&inb[(size_t)n * C + cin0]);
                        acc.x = fmaf(x.x, w.x, acc.x);
                        acc.y = fmaf(x.y, w.y, acc.y);
                        acc.z = fmaf(x.z, w.z, acc.z);
                        acc.w = fmaf(x.w, w.w, acc.w);
                    }
                }
            }
        }
        if (act) {
            const float fc = (float)cnt;   // cnt == 0 only for rows the caller marked empty: output 0
            float4 o;
            o.x = cnt > 0 ? acc.x / fc : 0.f;
            o.y = cnt > 0 ? acc.y / fc : 0.f;
            o.z = cnt > 0 ? acc.z / fc : 0.f;
            o.w = cnt > 0 ? acc.w / fc : 0.f;
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
                const int f = brow[kk];
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
// backward, vectorised, both gradients in one pass
// LDS table gtab holds grad_filter for this slice in a permuted layout:
//   local channel cl = 4*l + v  ->  gtab[f*SL + v*(SL/4) + l]   (consecutive lanes -> consecutive banks)
// ------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void dwconv_bwd_vec(
    int B, int N, int M, int F, int C, int K, int mblocks, int nslices,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount, const int* __restrict__ binIndex,
    const float* __restrict__ input, const float* __restrict__ filter, const float* __restrict__ gradOutput,
    float* __restrict__ gradInput, float* __restrict__ gradFilter)
{
    extern __shared__ __attribute__((aligned(16))) float gtab[];   // [F][SL] permuted
    const int CR = C * R;
    int b, part;
    xcd_decode((int)blockIdx.x, B, mblocks * nslices, b, part);
    if (b < 0) return;
    const int slice = part / mblocks;
    const int mb = part - slice * mblocks;
    const int slice0 = slice * kSlice;
    const int SL = (CR - slice0) < kSlice ? (CR - slice0) : kSlice;
    const int SLQ = SL >> 2;

    for (int e = threadIdx.x; e < F * SL; e += blockDim.x) gtab[e] = 0.f;
    __syncthreads();

    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int cl0 = lane * 4;
    const bool act = cl0 < SL;
    const int cin0 = (slice0 + cl0) / R;
    const int m_begin = mb * kBwdPointsPerWG;
    const int m_end = (m_begin + kBwdPointsPerWG) < M ? (m_begin + kBwdPointsPerWG) : M;
    const float* inb = input + (size_t)b * N * C;
    float* ginb = gradInput + (size_t)b * N * C;

    for (int m = m_begin + wave; m < m_end; m += 4) {
        const size_t row = (size_t)b * M + m;
        const int cnt = uniform(nnCount[row]);
        if (cnt <= 0) continue;
        float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) {
            go = *reinterpret_cast<const float4*>(&gradOutput[row * CR + slice0 + cl0]);
            const float fc = (float)cnt;
            go.x /= fc; go.y /= fc; go.z /= fc; go.w /= fc;     // the reference's /nnSize (:50, :87)
        }
        const int* __restrict__ irow = nnIndex + row * K;   // wave-uniform address -> scalar loads
        const int* __restrict__ brow = binIndex + row * K;
        {
#pragma unroll 4
            for (int kk = 0; kk < cnt; kk++) {
                const int n = irow[kk];
                const int f = brow[kk];
                if (act) {
                    const float4 w = *reinterpret_cast<const float4*>(&filter[(size_t)f * CR + slice0 + cl0]);
                    float* gt = &gtab[f * SL + lane];
                    if (R == 2) {
                        const float2 x = *reinterpret_cast<const float2*>(&inb[(size_t)n * C + cin0]);
                        unsafeAtomicAdd(&ginb[(size_t)n * C + cin0], fmaf(go.x, w.x, go.y * w.y));
                        unsafeAtomicAdd(&ginb[(size_t)n * C + cin0 + 1], fmaf(go.z, w.z, go.w * w.w));
                        unsafeAtomicAdd(gt, go.x * x.x);
                        unsafeAtomicAdd(gt + SLQ, go.y * x.x);
                        unsafeAtomicAdd(gt + 2 * SLQ, go.z * x.y);
                        unsafeAtomicAdd(gt + 3 * SLQ, go.w * x.y);
                    } else {
                        const float4 x = *reinterpret_cast<const float4*>(&inb[(size_t)n * C + cin0]);
                        unsafeAtomicAdd(&ginb[(size_t)n * C + cin0], go.x * w.x);
                        unsafeAtomicAdd(&ginb[(size_t)n * C + cin0 + 1], go.y * w.y);
                        unsafeAtomicAdd(&ginb[(size_t)n * C + cin0 + 2], go.z * w.z);
                        unsafeAtomicAdd(&ginb[(size_t)n * C + cin0 + 3], go.w * w.w);
                        unsafeAtomicAdd(gt, go.x * x.x);
                        unsafeAtomicAdd(gt + SLQ, go.y * x.y);
                        unsafeAtomicAdd(gt + 2 * SLQ, go.z * x.z);
                        unsafeAtomicAdd(gt + 3 * SLQ, go.w * x.w);
                    }
                }
            }
        }
    }
    __syncthreads();
    // flush: un-permute, skip untouched entries (many bins are never hit at small radii)
    for (int e = threadIdx.x; e < F * SL; e += blockDim.x) {
        const int f = e / SL;
        const int pe = e - f * SL;          // permuted position v*SLQ + l
        const int v = pe / SLQ;
        const int l = pe - v * SLQ;
        const float g = gtab[e];
        if (g != 0.f) unsafeAtomicAdd(&gradFilter[(size_t)f * CR + slice0 + 4 * l + v], g);
    }
}

// backward, generic (any C, r)
__global__ __launch_bounds__(256) void dwconv_bwd_generic(
    int B, int N, int M, int F, int C, int r, int K, int mblocks, int nslices,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount, const int* __restrict__ binIndex,
    const float* __restrict__ input, const float* __restrict__ filter, const float* __restrict__ gradOutput,
    float* __restrict__ gradInput, float* __restrict__ gradFilter)
{
    extern __shared__ __attribute__((aligned(16))) float gtab[];   // [F][SL]
    const int CR = C * r;
    int b, part;
    xcd_decode((int)blockIdx.x, B, mblocks * nslices, b, part);
    if (b < 0) return;
    const int slice = part / mblocks;
    const int mb = part - slice * mblocks;
    const int slice0 = slice * kSlice;
    const int SL = (CR - slice0) < kSlice ? (CR - slice0) : kSlice;

    for (int e = threadIdx.x; e < F * SL; e += blockDim.x) gtab[e] = 0.f;
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
    const int m_begin = mb * kBwdPointsPerWG;
    const int m_end = (m_begin + kBwdPointsPerWG) < M ? (m_begin + kBwdPointsPerWG) : M;
    const float* inb = input + (size_t)b * N * C;
    float* ginb = gradInput + (size_t)b * N * C;

    for (int m = m_begin + wave; m < m_end; m += 4) {
        const size_t row = (size_t)b * M + m;
        const int cnt = uniform(nnCount[row]);
        if (cnt <= 0) continue;
        float go[4];
#pragma unroll
        for (int t = 0; t < 4; t++) go[t] = act[t] ? gradOutput[row * CR + slice0 + cl[t]] / (float)cnt : 0.f;
        const int* __restrict__ irow = nnIndex + row * K;   // wave-uniform address -> scalar loads
        const int* __restrict__ brow = binIndex + row * K;
        {
#pragma unroll 4
            for (int kk = 0; kk < cnt; kk++) {
                const int n = irow[kk];
                const int f = brow[kk];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (act[t]) {
                        const float x = inb[(size_t)n * C + cin[t]];
                        const float w = filter[(size_t)f * CR + slice0 + cl[t]];
                        unsafeAtomicAdd(&ginb[(size_t)n * C + cin[t]], go[t] * w);
                        unsafeAtomicAdd(&gtab[f * SL + cl[t]], go[t] * x);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < F * SL; e += blockDim.x) {
        const int f = e / SL;
        const int c = e - f * SL;
        const float g = gtab[e];
        if (g != 0.f) unsafeAtomicAdd(&gradFilter[(size_t)f * CR + slice0 + c], g);
    }
}

static int conv_dims_ok(int B, int N, int M, int F, int C, int r, int K, const char* who)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0 && F > 0 && C > 0 && r > 0 && K > 0,
                  "%s: bad dims B=%d N=%d M=%d F=%d C=%d r=%d K=%d", who, B, N, M, F, C, r, K);
    SPH3D_REQUIRE((size_t)F * (C * r < kSlice ? C * r : kSlice) * sizeof(float) <= 160 * 1024,
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
    const int nslices = (CR + kSlice - 1) / kSlice;
    const int SLmax = CR < kSlice ? CR : kSlice;
    const size_t lds = (size_t)F * SLmax * sizeof(float);
    const dim3 grid(xcd_grid(B, mblocks * nslices));
    hipStream_t st = as_stream(stream);
    const bool vec = (CR % 4 == 0) && (r == 1 || r == 2);
#define SPH3D_BIG_LDS(kern)                                                                                       \
    if (lds > 64 * 1024) {                                                                                        \
        rc = check_hip(hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), \
                       "conv3d: hipFuncSetAttribute");                                                            \
        if (rc) return rc;                                                                                        \
    }
    if (vec && r == 2) {
        SPH3D_BIG_LDS(dwconv_fwd_vec<2>)
        hipLaunchKernelGGL(dwconv_fwd_vec<2>, grid, dim3(256), lds, st, B, N, M, F, C, K, mblocks, nslices,
                           nn_index, nn_count, bin_index, input, filter, output);
    } else if (vec) {
        SPH3D_BIG_LDS(dwconv_fwd_vec<1>)
        hipLaunchKernelGGL(dwconv_fwd_vec<1>, grid, dim3(256), lds, st, B, N, M, F, C, K, mblocks, nslices,
                           nn_index, nn_count, bin_index, input, filter, output);
    } else {
        SPH3D_BIG_LDS(dwconv_fwd_generic)
        hipLaunchKernelGGL(dwconv_fwd_generic, grid, dim3(256), lds, st, B, N, M, F, C, r, K, mblocks, nslices,
                           nn_index, nn_count, bin_index, input, filter, output);
    }
    return check_launch("sph3d_depthwise_conv3d");
}

extern "C" size_t sph3d_depthwise_conv3d_grad_workspace(int, int, int, int, int, int, int) { return 0; }

extern "C" int sph3d_depthwise_conv3d_grad(int B, int N, int M, int F, int C, int r, int K,
                                           const int* nn_index, const int* nn_count, const int* bin_index,
                                           const float* input, const float* filter, const float* grad_output,
                                           float* grad_input, float* grad_filter,
                                           void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    (void)workspace; (void)workspace_bytes;
    int rc = conv_dims_ok(B, N, M, F, C, r, K, "DepthwiseConv3dGrad");
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const int CR = C * r;
    // the op zero-fills both gradients first (tf_conv3d.cpp:152-153)
    rc = check_hip(hipMemsetAsync(grad_filter, 0, sizeof(float) * (size_t)F * CR, st), "conv3d grad: memset");
    if (rc) return rc;
    if (B == 0) return SPH3D_OK;
    rc = check_hip(hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)B * N * C, st), "conv3d grad: memset");
    if (rc) return rc;
    if (M == 0) return SPH3D_OK;
    const int mblocks = (M + kBwdPointsPerWG - 1) / kBwdPointsPerWG;
    const int nslices = (CR + kSlice - 1) / kSlice;
    const int SLmax = CR < kSlice ? CR : kSlice;
    const size_t lds = (size_t)F * SLmax * sizeof(float);
    const dim3 grid(xcd_grid(B, mblocks * nslices));
    const bool vec = (CR % 4 == 0) && (r == 1 || r == 2);
    if (vec && r == 2) {
        SPH3D_BIG_LDS(dwconv_bwd_vec<2>)
        hipLaunchKernelGGL(dwconv_bwd_vec<2>, grid, dim3(256), lds, st, B, N, M, F, C, K, mblocks, nslices,
                           nn_index, nn_count, bin_index, input, filter, grad_output, grad_input, grad_filter);
    } else if (vec) {
        SPH3D_BIG_LDS(dwconv_bwd_vec<1>)
        hipLaunchKernelGGL(dwconv_bwd_vec<1>, grid, dim3(256), lds, st, B, N, M, F, C, K, mblocks, nslices,
                           nn_index, nn_count, bin_index, input, filter, grad_output, grad_input, grad_filter);
    } else {
        SPH3D_BIG_LDS(dwconv_bwd_generic)
        hipLaunchKernelGGL(dwconv_bwd_generic, grid, dim3(256), lds, st, B, N, M, F, C, r, K, mblocks, nslices,
                           nn_index, nn_count, bin_index, input, filter, grad_output, grad_input, grad_filter);
    }
    return check_launch("sph3d_depthwise_conv3d_grad");
#undef SPH3D_BIG_LDS
}
