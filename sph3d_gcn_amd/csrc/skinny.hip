// skinny.hip — the 1x1 layer with a handful of outputs (the logits layer: 256 -> 13 classes over 131 072 points), gfx950.
//
// utils/sph3gcn_util.py:166-222 (pointwise_conv3d) applied to models/SPH3D_s3dis.py:104-108: tf.concat of the un-pooled features
// and the encoder's skip features, then tf.matmul with a [256, 13] weight matrix (+ biases).  For the big GEMM kernels this shape
// is all edge: 13 of 128 tile columns, guarded scalar loads / stores — 64 us forward, 78 us for the weight gradient where the
// operand streams in 27 us — and the concatenation in front of it is a 268-MB copy (23 us, plus two slice copies on the way back).
// Here the product reads its TWO operand halves where they are (the concatenation is never materialised) and is what it should
// be, a streaming kernel:
//   forward   Y [R, N]      = A1[R, K1] W[0:K1] + A2[R, K2] W[K1:K1+K2] (+ bias)        N <= 16, K1, K2 multiples of 16, K1 + K2 <= 256
//   weights   dW[K1+K2, N]  = [A1 | A2]^T dY                                            (slabs per workgroup + a fixed-order sum)
// v_mfma_f32_16x16x4_f32 does the arithmetic (exact fp32 FMAs).  Forward: a wave owns 16 rows; lane (row i, k-quarter q) loads
// A[i][16t + 4q .. +3] as ONE 16-byte load per k-group t and feeds its four values to four MFMAs — any k pairing is legal as long
// as both operands use it (sepconv.hip) —; the weights sit in registers for the whole launch (K/4 VGPRs).  Weight gradient:
// D[n][j] += sum over 4 rows of dY[row][n] * A[row][c(j)] with the 16 output blocks of a 64-channel group mapped so that lane j's
// four values of ONE 16-byte load (channels 64s + 4j + v) go to four different accumulators: the operand is read with full
// 16-byte lanes, 256 contiguous bytes per row.  (The input gradient keeps the general kernel: it is a K = 13 product writing
// 134 MB and runs near its store rate; called once per operand half it writes the two gradients where autograd wants them.)
#include "common.hpp"

namespace sph3d {

typedef float sk_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSkMaxK = 256;             // K1 + K2
constexpr int kSkKT = kSkMaxK / 16;      // k-groups of 16

__device__ __forceinline__ sk_f32x4 sk_mfma(float a, float b, sk_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- forward -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void skinny_nn_kernel(int R, int K1, int K2, int N, const float* __restrict__ A1,
                                                        const float* __restrict__ A2, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ Y)
{
    const int lane = lane_id();
    const int i16 = lane & 15, kq = lane >> 4;
    const int kt1 = K1 >> 4, kt = (K1 + K2) >> 4;
    // W[16t + 4kq + u][i16] for every k-group: resident for the launch (columns >= N are zero: padded outputs, never stored)
    float wreg[kSkKT][4];
#pragma unroll
    for (int t = 0; t < kSkKT; t++)
#pragma unroll
        for (int u = 0; u < 4; u++) wreg[t][u] = (t < kt && i16 < N) ? W[(size_t)(16 * t + 4 * kq + u) * N + i16] : 0.f;
    const float bv = (bias != nullptr && i16 < N) ? bias[i16] : 0.f;
    const int tiles = (R + 15) >> 4;
    const int wid = (int)blockIdx.x * 4 + uniform((int)threadIdx.x >> 6), nw = (int)gridDim.x * 4;
    for (int tile = wid; tile < tiles; tile += nw) {
        const int row = tile * 16 + i16;
        const int rowc = row < R ? row : R - 1;                 // ragged last tile: a valid row, its outputs are not stored
        const float* p1 = A1 + (size_t)rowc * K1 + 4 * kq;
        const float* p2 = A2 ? A2 + (size_t)rowc * K2 + 4 * kq : p1;
        sk_f32x4 a[kSkKT];
#pragma unroll
        for (int t = 0; t < kSkKT; t++) {
            if (t < kt) a[t] = *reinterpret_cast<const sk_f32x4*>(t < kt1 ? p1 + 16 * t : p2 + 16 * (t - kt1));   // wave-uniform selects
        }
        // four independent accumulation chains (one per k-slot of the 16-byte load): 64 dependent MFMAs in ONE chain left the
        // wave waiting for its own previous result
        sk_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
#pragma unroll
        for (int t = 0; t < kSkKT; t++) {
            if (t < kt) {
                d0 = sk_mfma(a[t].x, wreg[t][0], d0);
                d1 = sk_mfma(a[t].y, wreg[t][1], d1);
                d2 = sk_mfma(a[t].z, wreg[t][2], d2);
                d3 = sk_mfma(a[t].w, wreg[t][3], d3);
            }
        }
        const sk_f32x4 d = (d0 + d1) + (d2 + d3);
        // D: lane holds rows 4*(lane/16) + r, r < 4, of column lane % 16
        if (i16 < N) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const int ro = tile * 16 + 4 * kq + r4;
                if (ro < R) Y[(size_t)ro * N + i16] = d[r4] + bv;
            }
        }
    }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------------
// accumulator (s, v) of operand half h holds dW[koff_h + 64 s + 4 j + v][n] at lane (j = lane % 16, n-quad = lane / 16), element r
constexpr int kSkRowsPerTrip = 8;        // two 4-row MFMA steps per trip: their loads are issued together

__global__ __launch_bounds__(256) void skinny_tn_kernel(int R, int K1, int K2, int N, int rows_per_wg, const float* __restrict__ A1,
                                                        const float* __restrict__ A2, const float* __restrict__ dY,
                                                        float* __restrict__ slabs)
{
    __shared__ float tab[kSkMaxK * 16];
    const int lane = lane_id();
    const int wave = uniform((int)threadIdx.x >> 6);
    const int j16 = lane & 15, kq = lane >> 4;
    const int ns1 = (K1 + 63) >> 6, ns2 = (K2 + 63) >> 6;                 // 64-channel groups per half (<= 4 in total)
    sk_f32x4 acc[4][4];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int v = 0; v < 4; v++) acc[s][v] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
    const int r_begin = (int)blockIdx.x * rows_per_wg;
    const int r_end = (r_begin + rows_per_wg) < R ? (r_begin + rows_per_wg) : R;
    // the four waves take alternate trips of kSkRowsPerTrip rows
    for (int r0 = r_begin + wave * kSkRowsPerTrip; r0 < r_end; r0 += 4 * kSkRowsPerTrip) {
        float dv[kSkRowsPerTrip / 4];
        sk_f32x4 x[kSkRowsPerTrip / 4][4];
#pragma unroll
        for (int g = 0; g < kSkRowsPerTrip / 4; g++) {
            const int row = r0 + 4 * g + kq;
            const bool ok = row < r_end;
            const int rc = ok ? row : r_end - 1;
            dv[g] = (ok && j16 < N) ? dY[(size_t)rc * N + j16] : 0.f;     // A operand: lane (n = lane % 16, row kq); 0 masks the padding
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const bool h1 = s < ns1;                                  // wave-uniform: which half this 64-channel group belongs to
                const int ch = (h1 ? s : s - ns1) * 64 + 4 * j16;
                const int Kh = h1 ? K1 : K2;
                const float* base = h1 ? A1 : A2;
                x[g][s] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
                if (s < ns1 + ns2 && ch < Kh) x[g][s] = *reinterpret_cast<const sk_f32x4*>(base + (size_t)rc * Kh + ch);
            }
        }
#pragma unroll
        for (int g = 0; g < kSkRowsPerTrip / 4; g++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if (s < ns1 + ns2) {
                    acc[s][0] = sk_mfma(dv[g], x[g][s].x, acc[s][0]);
                    acc[s][1] = sk_mfma(dv[g], x[g][s].y, acc[s][1]);
                    acc[s][2] = sk_mfma(dv[g], x[g][s].z, acc[s][2]);
                    acc[s][3] = sk_mfma(dv[g], x[g][s].w, acc[s][3]);
                }
            }
    }
    // the workgroup's table [K1 + K2][16]: waves take turns (fixed order), then one slab per workgroup
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if (s < ns1 + ns2) {
                    const bool h1 = s < ns1;
                    const int ch0 = (h1 ? s : s - ns1) * 64 + 4 * j16;
                    const int Kh = h1 ? K1 : K2, koff = h1 ? 0 : K1;
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        if (ch0 + v < Kh) {
#pragma unroll
                            for (int r4 = 0; r4 < 4; r4++) {
                                float* p = &tab[(koff + ch0 + v) * 16 + 4 * kq + r4];        // D rows = n: 4*(lane/16) + r
                                *p = (w == 0) ? acc[s][v][r4] : (*p + acc[s][v][r4]);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    float* out = slabs + (size_t)blockIdx.x * (size_t)(K1 + K2) * 16;
    for (int e = (int)threadIdx.x; e < (K1 + K2) * 16; e += 256) out[e] = tab[e];
}

// dW[k][n] = sum over the slabs, fixed order; 16 padded columns -> N.  256 threads = 32 slab elements x 8 slab-lanes (a
// one-thread-per-element loop over 512 slabs is 64 dependent L2 round trips)
__global__ __launch_bounds__(256) void skinny_tn_reduce(int nslabs, int K, int N, const float* __restrict__ slabs, float* __restrict__ dW)
{
    __shared__ float red[8][32];
    const int cx = (int)threadIdx.x & 31, py = (int)threadIdx.x >> 5;
    const int e = (int)blockIdx.x * 32 + cx;                              // element of a [K][16] slab
    const int total = K * 16;
    float s = 0.f;
    if (e < total) {
        for (int p0 = py; p0 < nslabs; p0 += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int p = p0 + u * 8;
                v[u] = slabs[(size_t)(p < nslabs ? p : py) * total + e];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (p0 + u * 8 < nslabs) s += v[u];
        }
    }
    red[py][cx] = s;
    __syncthreads();
    if (py == 0 && e < total) {
        for (int k = 1; k < 8; k++) s += red[k][cx];
        const int kk = e >> 4, n = e & 15;
        if (n < N) dW[(size_t)kk * N + n] = s;
    }
}

static bool sk_ok(int R, int K1, int K2, int N)
{
    // (the weight gradient keeps one accumulator group per started 64 channels of a half: four groups)
    return R > 0 && N >= 1 && N <= 16 && K1 >= 16 && K1 % 16 == 0 && K2 >= 0 && K2 % 16 == 0 && K1 + K2 <= kSkMaxK &&
           (K1 + 63) / 64 + (K2 + 63) / 64 <= 4;
}

constexpr int kSkTnWGs = 512;

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_pointwise_gemm_skinny_supported(int R, int K1, int K2, int N) { return sk_ok(R, K1, K2, N) ? 1 : 0; }

extern "C" int sph3d_pointwise_gemm_skinny(int R, int K1, int K2, int N, const float* A1, const float* A2, const float* W,
                                           const float* bias, float* Y, sph3d_stream_t stream)
{
    if (!sk_ok(R, K1, K2, N) || (K2 > 0 && A2 == nullptr)) {
        set_error("pointwise_gemm_skinny: needs N <= 16, K1 and K2 multiples of 16, K1 + K2 <= 256 (got R=%d K1=%d K2=%d N=%d)", R, K1, K2, N);
        return SPH3D_EUNSUPPORTED;
    }
    SPH3D_REQUIRE((reinterpret_cast<size_t>(A1) & 15) == 0 && (reinterpret_cast<size_t>(A2) & 15) == 0,
                  "pointwise_gemm_skinny: operands must be 16-byte aligned");
    const int tiles = (R + 15) / 16;
    int wgs = (tiles + 7) / 8;                         // two tiles per wave
    if (wgs > 2048) wgs = 2048;
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL(skinny_nn_kernel, dim3(wgs), dim3(256), 0, as_stream(stream), R, K1, K2, N, A1, K2 > 0 ? A2 : nullptr, W, bias, Y);
    return check_launch("sph3d_pointwise_gemm_skinny");
}

extern "C" size_t sph3d_pointwise_gemm_skinny_tn_workspace(int R, int K1, int K2, int N)
{
    (void)R; (void)N;
    return sizeof(float) * (size_t)kSkTnWGs * (size_t)(K1 + K2) * 16;
}

extern "C" int sph3d_pointwise_gemm_skinny_tn(int R, int K1, int K2, int N, const float* A1, const float* A2, const float* dY,
                                              float* dW, void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    if (!sk_ok(R, K1, K2, N) || (K2 > 0 && A2 == nullptr)) {
        set_error("pointwise_gemm_skinny_tn: needs N <= 16, K1 and K2 multiples of 16, K1 + K2 <= 256 (got R=%d K1=%d K2=%d N=%d)", R, K1, K2, N);
        return SPH3D_EUNSUPPORTED;
    }
    const size_t need = sph3d_pointwise_gemm_skinny_tn_workspace(R, K1, K2, N);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("pointwise_gemm_skinny_tn: workspace %zu B < required %zu B", workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    SPH3D_REQUIRE((reinterpret_cast<size_t>(A1) & 15) == 0 && (reinterpret_cast<size_t>(A2) & 15) == 0,
                  "pointwise_gemm_skinny_tn: operands must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    // whole trips of 4 x kSkRowsPerTrip rows per workgroup
    const int unit = 4 * kSkRowsPerTrip;
    int rows_per_wg = ((R + kSkTnWGs - 1) / kSkTnWGs + unit - 1) / unit * unit;
    const int wgs = (R + rows_per_wg - 1) / rows_per_wg;
    hipLaunchKernelGGL(skinny_tn_kernel, dim3(wgs), dim3(256), 0, st, R, K1, K2, N, rows_per_wg, A1, K2 > 0 ? A2 : A1, dY,
                       (float*)workspace);
    const int K = K1 + K2;
    hipLaunchKernelGGL(skinny_tn_reduce, dim3((K * 16 + 31) / 32), dim3(256), 0, st, wgs, K, N, (const float*)workspace, dW);
    return check_launch("sph3d_pointwise_gemm_skinny_tn");
}
