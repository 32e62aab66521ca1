// loss.hip — the segmentation nets' training loss in one kernel, gfx950.
//
// models/SPH3D_s3dis.py:116-133: per block the mean over its INNER points (inner_label > 0) of the sparse softmax cross-entropy,
// summed over the batch (tf.nn.sparse_softmax_cross_entropy_with_logits + tf.boolean_mask + reduce_mean there; cross_entropy,
// mask, sum, clamp, divide, where, sum — ten launches forward and as many backward — as framework ops here before).  One
// launch computes the blocks' losses AND the gradient of the batch sum with respect to the logits,
//     dlogits[b,n,:] = inner[b,n] / cnt_b * (softmax(logits[b,n,:]) - onehot(label[b,n])),
// so the backward pass is one scaling by the upstream gradient.  Reductions run in a fixed order (deterministic).
#include "common.hpp"

namespace sph3d {

__device__ __forceinline__ float block_sum_1024(float v, float* red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = (int)threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < 16; k++) s += red[k];
    return s;
}

// grid (S, B): workgroup (s, b) counts the inner points of the WHOLE block b (32 KB of mask: cheaper than a second launch or a
// cross-workgroup exchange) and then handles the points [s * chunk, (s + 1) * chunk) of it: their share of the block's mean in
// lossPart[b * S + s] and their rows of the gradient.  A single workgroup per block left 240 of 256 CUs idle for 40 us.
__global__ __launch_bounds__(1024) void masked_xent_kernel(int N, int C, int chunk, const float* __restrict__ logits,
                                                           const long long* __restrict__ label, const float* __restrict__ inner,
                                                           float* __restrict__ lossPart, float* __restrict__ dlogits)
{
    __shared__ float red[16];
    const int b = (int)blockIdx.y, sl = (int)blockIdx.x, S = (int)gridDim.x;
    const float* lg = logits + (size_t)b * N * C;
    const long long* lb = label + (size_t)b * N;
    const float* in = inner + (size_t)b * N;
    float cnt = 0.f;
    for (int n = (int)threadIdx.x; n < N; n += 1024) cnt += in[n] > 0.f ? 1.f : 0.f;
    cnt = block_sum_1024(cnt, red);
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
    const int n0 = sl * chunk, n1 = (n0 + chunk) < N ? (n0 + chunk) : N;
    float* dg = dlogits + (size_t)b * N * C;
    float sum = 0.f;
    for (int n = n0 + (int)threadIdx.x; n < n1; n += 1024) {
        float* drow = dlogits != nullptr ? dg + (size_t)n * C : nullptr;      // dlogits == NULL: loss only (no gradient wanted)
        if (in[n] > 0.f) {
            const float* row = lg + (size_t)n * C;
            float m = row[0];
            for (int c = 1; c < C; c++) m = fmaxf(m, row[c]);
            float se = 0.f;
            for (int c = 0; c < C; c++) se += expf(row[c] - m);
            long long y = lb[n];
            // a label outside [0, C): the reference's op yields NaN for that row on the GPU (and the CPU path here raises):
            // the block's loss and the row's gradient are NaN, never a silently clamped class; reads stay in range
            const bool bad = y < 0 || y >= C;
            y = bad ? 0 : y;
            sum += bad ? __builtin_nanf("") : (m + logf(se)) - row[y];
            if (drow != nullptr) {
                const float r = inv / se;
                for (int c = 0; c < C; c++)
                    drow[c] = bad ? __builtin_nanf("") : expf(row[c] - m) * r - (c == (int)y ? inv : 0.f);
            }
        } else if (drow != nullptr) {
            for (int c = 0; c < C; c++) drow[c] = 0.f;
        }
    }
    sum = block_sum_1024(sum, red);
    if (threadIdx.x == 0) lossPart[(size_t)b * S + sl] = sum * inv;
}

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_masked_softmax_xent_parts(int N)
{
    const int s = (N + 1023) / 1024;
    return s < 1 ? 1 : (s > 16 ? 16 : s);
}

extern "C" int sph3d_masked_softmax_xent(int B, int N, int C, const float* logits, const long long* label, const float* inner_label,
                                         float* loss_part, float* dlogits, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && C > 0, "masked_softmax_xent: bad dims B=%d N=%d C=%d", B, N, C);
    if (B == 0) return SPH3D_OK;
    const int S = sph3d_masked_softmax_xent_parts(N);
    const int chunk = (N + S - 1) / S;
    hipLaunchKernelGGL(masked_xent_kernel, dim3(S, B), dim3(1024), 0, as_stream(stream), N, C, chunk, logits, label, inner_label,
                       loss_part, dlogits);
    return check_launch("sph3d_masked_softmax_xent");
}
