// optim.hip — the training loop's Adam update (train_s3dis.py:224: tf.train.AdamOptimizer(learning_rate, epsilon=1e-4)) over the
// harness's flat parameter / gradient buffers: ONE streaming pass (read p, g, m, v; write p, m, v — 28 B per parameter), 16-byte
// lanes.  The framework's fused multi-tensor kernel needs 50-86 us for the 3.9 M parameters of the S3DIS net where the bytes
// take 25 us.  Same arithmetic as torch.optim.Adam (no amsgrad, no weight decay, not maximising):
//     m = m + (1 - b1) (g - m);   v = b2 v + (1 - b2) g g;   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
#include "common.hpp"

namespace sph3d {

__global__ __launch_bounds__(256) void adam_kernel(long long n4, long long n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float lr_bc1, float inv_sqrt_bc2,
                                                   float b1, float b2, float eps)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        mm = mm + (1.f - b1) * (gg - mm);
        vv = b2 * vv + (1.f - b2) * gg * gg;
        pp = pp - lr_bc1 * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
    };
    if (i < n4) {
        float4 P = reinterpret_cast<float4*>(p)[i], M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
        reinterpret_cast<float4*>(p)[i] = P; reinterpret_cast<float4*>(m)[i] = M; reinterpret_cast<float4*>(v)[i] = V;
    } else {
        const long long e = n4 * 4 + (i - n4);          // the tail of a length that is not a multiple of 4
        if (e < n) upd(p[e], g[e], m[e], v[e]);
    }
}

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_adam_step(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                               float beta2, float eps, int step, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(n >= 0 && step >= 1, "adam_step: bad n=%lld step=%d", n, step);
    if (n == 0) return SPH3D_OK;
    const bool al = ((reinterpret_cast<size_t>(param) | reinterpret_cast<size_t>(grad) | reinterpret_cast<size_t>(exp_avg) |
                      reinterpret_cast<size_t>(exp_avg_sq)) & 15) == 0;
    const long long n4 = al ? n / 4 : 0;
    const long long threads = n4 + (n - n4 * 4);
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), n4, n, param, grad, exp_avg,
                       exp_avg_sq, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps);
    return check_launch("sph3d_adam_step");
}
