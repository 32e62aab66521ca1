// norm.hip — fused ELU + batch-normalisation tail of the separable convolution, gfx950.
//
// The reference's separable_conv3d / pointwise_conv3d end with  matmul -> (bias) -> ELU -> batch_norm
// (utils/sph3gcn_util.py:152-161, 208-220; tf.nn.elu + tf.layers.batch_normalization(momentum 0.99, eps 1e-3), stock TF
// ops there).  As separate passes that tail moves the [R,C] activation through HBM five times forward (ELU r+w, stats r,
// normalise r+w) and eight times backward; it was 3.7 ms of a 20 ms step.  Fused:
//   forward : stats pass  reads y            -> per-channel sum / sum of squares of z = elu(y)   (partials, no atomics)
//             apply pass  reads y, writes out   out = (elu(y) - mean) * rstd * gamma + beta
//   backward: reduce pass reads y, dout      -> dbeta = sum dout, dgamma = sum dout * zhat
//             apply pass  reads y, dout, writes dy = gamma*rstd*(dout - dbeta/R - zhat*dgamma/R) * elu'(y)
// Only y (the GEMM output) is kept for the backward pass; z is recomputed (one v_exp per element).
// All kernels are HBM-streaming: float4 per lane, rows strided over the workgroup, per-workgroup partial sums reduced in
// double by a single small kernel (deterministic).  Requires C % 4 == 0 and C <= 1024 (the plans' 64..512).
#include "common.hpp"

namespace sph3d {

constexpr int kNormMaxBlocks = 1024;

// exp(y) - 1 through v_exp_f32: absolute error ~1e-7 on (-inf, 0], far inside the 1e-5 activation bound; the software expm1f
// made these HBM-streaming kernels VALU-bound (3.7 ms/step, no better than the unfused torch ops)
__device__ __forceinline__ float elu1(float y) { return y > 0.f ? y : __expf(y) - 1.f; }

// partial[blk][0][C] = sum a, partial[blk][1][C] = sum b over the block's rows, where
//   BWD = false: a = z,    b = z*z            (z = elu(y))
//   BWD = true : a = dout, b = dout * zhat    (zhat = (z - mean) * rstd)
template <bool BWD>
__global__ __launch_bounds__(256) void norm_reduce_kernel(int R, int C, const float* __restrict__ y,
                                                          const float* __restrict__ dout, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ partial)
{
    __shared__ float4 red[2][256];
    const int cg = C >> 2;                          // channel groups of 4
    const int rows_per_iter = 256 / cg > 0 ? 256 / cg : 1;
    const int tx = (int)threadIdx.x % cg, ty = (int)threadIdx.x / cg;
    const bool act = ty < rows_per_iter;
    const int rows_per_block = (R + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r0 = (int)blockIdx.x * rows_per_block;
    const int r1 = (r0 + rows_per_block) < R ? (r0 + rows_per_block) : R;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    float4 mu = a, rs = a;
    if (BWD && act) {
        mu = *reinterpret_cast<const float4*>(&mean[tx * 4]);
        rs = *reinterpret_cast<const float4*>(&rstd[tx * 4]);
    }
    if (act) {
        // four rows per trip, their loads issued together: with one row per trip the kernel ran at a third of the HBM
        // rate (one or two 16-B loads in flight per lane)
        constexpr int UN = 4;
        for (int rb = r0 + ty; rb < r1; rb += UN * rows_per_iter) {
            float4 v[UN], g[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const int r = rb + u * rows_per_iter;
                const int rc = r < r1 ? r : rb;                      // clamped: re-reads the trip's first row
                v[u] = *reinterpret_cast<const float4*>(&y[(size_t)rc * C + tx * 4]);
                if (BWD) g[u] = *reinterpret_cast<const float4*>(&dout[(size_t)rc * C + tx * 4]);
            }
#pragma unroll
            for (int u = 0; u < UN; u++) {
                if (rb + u * rows_per_iter < r1) {
                    const float4 z = make_float4(elu1(v[u].x), elu1(v[u].y), elu1(v[u].z), elu1(v[u].w));
                    if (!BWD) {
                        a.x += z.x; a.y += z.y; a.z += z.z; a.w += z.w;
                        b.x = fmaf(z.x, z.x, b.x); b.y = fmaf(z.y, z.y, b.y); b.z = fmaf(z.z, z.z, b.z); b.w = fmaf(z.w, z.w, b.w);
                    } else {
                        a.x += g[u].x; a.y += g[u].y; a.z += g[u].z; a.w += g[u].w;
                        b.x = fmaf(g[u].x, (z.x - mu.x) * rs.x, b.x);
                        b.y = fmaf(g[u].y, (z.y - mu.y) * rs.y, b.y);
                        b.z = fmaf(g[u].z, (z.z - mu.z) * rs.z, b.z);
                        b.w = fmaf(g[u].w, (z.w - mu.w) * rs.w, b.w);
                    }
                }
            }
        }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (ty == 0 && tx < cg) {                        // fold the ty rows of the workgroup, fixed order
        for (int j = 1; j < rows_per_iter; j++) {
            const float4 p = red[0][j * cg + tx], q = red[1][j * cg + tx];
            a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
            b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
        }
        float* out = partial + (size_t)blockIdx.x * 2 * C;
        *reinterpret_cast<float4*>(&out[tx * 4]) = a;
        *reinterpret_cast<float4*>(&out[C + tx * 4]) = b;
    }
}

// Sum the per-workgroup partials of 32 channels with 1024 threads (32 partial-lanes per channel, double precision, fixed
// order); the result is valid for threads with py == 0.  (A one-thread-per-channel loop over 1024 partials took 67 us,
// 8 partial-lanes per channel 12 us per finalize launch = 0.4 ms per step over the 34 launches.)
constexpr int kFinLanes = 32;
__device__ __forceinline__ void sum_partials_32x8(int C, int nblk, const float* __restrict__ partial, int c, int py,
                                                  double& s, double& q)
{
    __shared__ double red[2][kFinLanes][32];
    s = 0.0; q = 0.0;
    if (c < C) {
        // eight rows per trip, their sixteen loads issued together: one row per trip is a chain of up to 32 dependent
        // round trips to L2 (the finalize kernels measured 8.5 us each, 34 of them per step, for a few microseconds of work)
        constexpr int UN = 8;
        for (int k0 = py; k0 < nblk; k0 += kFinLanes * UN) {
            float a[UN], b[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const int k = k0 + u * kFinLanes;
                const int kc = k < nblk ? k : py;                    // clamped (py < nblk here), masked below
                a[u] = partial[(size_t)kc * 2 * C + c];
                b[u] = partial[(size_t)kc * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < UN; u++) {
                if (k0 + u * kFinLanes < nblk) { s += (double)a[u]; q += (double)b[u]; }
            }
        }
    }
    const int cx = (int)threadIdx.x & 31;
    red[0][py][cx] = s;
    red[1][py][cx] = q;
    __syncthreads();
    if (py == 0) {
        for (int j = 1; j < kFinLanes; j++) { s += red[0][j][cx]; q += red[1][j][cx]; }
    }
}

// forward finalize: mean / rstd from the partials (double), running statistics update
// running = (1-m)*running + m*batch with the BIASED batch variance: tf.layers.batch_normalization on rank-2/3 inputs takes
// the non-fused path (tf.nn.moments), which stores the population variance — torch's F.batch_norm would store the
// Bessel-corrected one (3 % larger at the ModelNet fc layers, where R = batch size)
__global__ __launch_bounds__(1024) void norm_fwd_finalize(int R, int C, int nblk, const float* __restrict__ partial, float eps, float momentum,
                                  float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                  float* __restrict__ run_var)
{
    const int c = blockIdx.x * 32 + ((int)threadIdx.x & 31);
    const int py = (int)threadIdx.x >> 5;
    double s, q;
    sum_partials_32x8(C, nblk, partial, c, py, s, q);
    if (c >= C || py != 0) return;
    const double m = s / R;
    double var = q / R - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean != nullptr) {
        run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
        run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * var);
    }
}

// inference: statistics come from the running buffers
__global__ void norm_eval_stats(int C, const float* __restrict__ run_mean, const float* __restrict__ run_var, float eps,
                                float* __restrict__ mean, float* __restrict__ rstd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = run_mean[c];
    rstd[c] = (float)(1.0 / sqrt((double)run_var[c] + (double)eps));
}

// backward finalize: dgamma, dbeta, and the two per-channel means the apply pass needs
__global__ __launch_bounds__(1024) void norm_bwd_finalize(int R, int C, int nblk, int training, const float* __restrict__ partial,
                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                  float* __restrict__ coef /* [2][C]: dbeta/R, dgamma/R (0 in inference mode) */)
{
    const int c = blockIdx.x * 32 + ((int)threadIdx.x & 31);
    const int py = (int)threadIdx.x >> 5;
    double s, q;
    sum_partials_32x8(C, nblk, partial, c, py, s, q);
    if (c >= C || py != 0) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
    coef[c] = training ? (float)(s / R) : 0.f;          // with fixed (running) statistics the two mean terms vanish
    coef[C + c] = training ? (float)(q / R) : 0.f;
}

// BWD = false: out = (elu(y) - mean) * rstd * gamma + beta
// BWD = true : dy  = gamma * rstd * (dout - c0 - zhat * c1) * elu'(y)
template <bool BWD>
__global__ __launch_bounds__(256) void norm_apply_kernel(long long total4, int C, const float* __restrict__ y,
                                                         const float* __restrict__ dout, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ coef,
                                                         float* __restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int cg = C >> 2;
    // channel group of element i, carried from iteration to iteration: a 64-bit `i % cg` per float4 is ~60 VALU
    // instructions, as much time as the HBM traffic of this streaming kernel
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int cq = (int)(i0 % cg);
    const int cstep = (int)(stride % cg);
    for (long long i = i0; i < total4; i += stride) {
        const int c = cq * 4;
        cq += cstep;
        cq = cq >= cg ? cq - cg : cq;
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        const float4 mu = *reinterpret_cast<const float4*>(&mean[c]);
        const float4 rs = *reinterpret_cast<const float4*>(&rstd[c]);
        const float4 ga = *reinterpret_cast<const float4*>(&gamma[c]);
        float4 o;
        if (!BWD) {
            const float4 be = *reinterpret_cast<const float4*>(&beta[c]);
            o.x = fmaf((elu1(v.x) - mu.x) * rs.x, ga.x, be.x);
            o.y = fmaf((elu1(v.y) - mu.y) * rs.y, ga.y, be.y);
            o.z = fmaf((elu1(v.z) - mu.z) * rs.z, ga.z, be.z);
            o.w = fmaf((elu1(v.w) - mu.w) * rs.w, ga.w, be.w);
        } else {
            const float4 g = reinterpret_cast<const float4*>(dout)[i];
            const float4 c0 = *reinterpret_cast<const float4*>(&coef[c]);
            const float4 c1 = *reinterpret_cast<const float4*>(&coef[C + c]);
#define SPH3D_BN_BWD(e)                                                             \
    {                                                                               \
        const float z = elu1(v.e);                                                  \
        const float zh = (z - mu.e) * rs.e;                                         \
        const float dz = ga.e * rs.e * (g.e - c0.e - zh * c1.e);                    \
        o.e = v.e > 0.f ? dz : dz * (z + 1.f);                                      \
    }
            SPH3D_BN_BWD(x) SPH3D_BN_BWD(y) SPH3D_BN_BWD(z) SPH3D_BN_BWD(w)
#undef SPH3D_BN_BWD
        }
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

// one workgroup per ~32 rows (the small levels have only a few thousand rows: 128-row blocks left most CUs idle)
static int norm_blocks(int R) { int b = (R + 31) / 32; return b < 1 ? 1 : (b > kNormMaxBlocks ? kNormMaxBlocks : b); }

}  // namespace sph3d

using namespace sph3d;

extern "C" size_t sph3d_elu_bn_workspace(int R, int C)
{
    return sizeof(float) * ((size_t)norm_blocks(R) * 2 * C + 2 * (size_t)C);
}

// training != 0: batch statistics (saved to save_mean / save_rstd for the backward pass), running stats updated.
// training == 0: normalise with running_mean / running_var (save_* receive the statistics used).
extern "C" int sph3d_elu_bn_forward(int R, int C, const float* y, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, int training,
                                    float* out, float* save_mean, float* save_rstd,
                                    void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(R > 0 && C > 0 && C % 4 == 0 && C <= 1024, "elu_bn: needs R>0, C%%4==0, C<=1024 (got R=%d C=%d)", R, C);
    const size_t need = sph3d_elu_bn_workspace(R, C);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("elu_bn_forward: workspace %zu B < required %zu B", workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    float* partial = (float*)workspace;
    const int nblk = norm_blocks(R);
    if (training) {
        hipLaunchKernelGGL(norm_reduce_kernel<false>, dim3(nblk), dim3(256), 0, st, R, C, y, nullptr, nullptr, nullptr, partial);
        hipLaunchKernelGGL(norm_fwd_finalize, dim3((C + 31) / 32), dim3(32 * kFinLanes), 0, st, R, C, nblk, partial, eps, momentum,
                           save_mean, save_rstd, running_mean, running_var);
    } else {
        SPH3D_REQUIRE(running_mean != nullptr && running_var != nullptr, "elu_bn: inference needs running statistics");
        hipLaunchKernelGGL(norm_eval_stats, dim3((C + 255) / 256), dim3(256), 0, st, C, running_mean, running_var, eps,
                           save_mean, save_rstd);
    }
    const long long total4 = (long long)R * C / 4;
    long long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(norm_apply_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, total4, C, y, nullptr, save_mean,
                       save_rstd, gamma, beta, nullptr, out);
    return check_launch("sph3d_elu_bn_forward");
}

// The same op when the statistics' partial sums already exist (sph3d_pointwise_gemm_bnstats wrote them from the GEMM's
// epilogue: partial[nblk][2][C] = per row block sum elu(y), sum elu(y)^2): finalize + apply only, training mode.
extern "C" int sph3d_elu_bn_forward_partials(int R, int C, int nblk, const float* partial, const float* y, const float* gamma,
                                             const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                             float* out, float* save_mean, float* save_rstd, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(R > 0 && C > 0 && C % 4 == 0 && C <= 1024 && nblk > 0, "elu_bn_partials: needs R>0, C%%4==0, C<=1024, nblk>0 (got R=%d C=%d nblk=%d)",
                  R, C, nblk);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(norm_fwd_finalize, dim3((C + 31) / 32), dim3(32 * kFinLanes), 0, st, R, C, nblk, partial, eps, momentum,
                       save_mean, save_rstd, running_mean, running_var);
    const long long total4 = (long long)R * C / 4;
    long long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(norm_apply_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, total4, C, y, nullptr, save_mean,
                       save_rstd, gamma, beta, nullptr, out);
    return check_launch("sph3d_elu_bn_forward_partials");
}

extern "C" int sph3d_elu_bn_backward(int R, int C, const float* y, const float* dout, const float* gamma,
                                     const float* save_mean, const float* save_rstd, int training,
                                     float* dy, float* dgamma, float* dbeta,
                                     void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(R > 0 && C > 0 && C % 4 == 0 && C <= 1024, "elu_bn: needs R>0, C%%4==0, C<=1024 (got R=%d C=%d)", R, C);
    const size_t need = sph3d_elu_bn_workspace(R, C);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("elu_bn_backward: workspace %zu B < required %zu B", workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    float* partial = (float*)workspace;
    const int nblk = norm_blocks(R);
    float* coef = partial + (size_t)nblk * 2 * C;
    hipLaunchKernelGGL(norm_reduce_kernel<true>, dim3(nblk), dim3(256), 0, st, R, C, y, dout, save_mean, save_rstd, partial);
    hipLaunchKernelGGL(norm_bwd_finalize, dim3((C + 31) / 32), dim3(32 * kFinLanes), 0, st, R, C, nblk, training, partial, dgamma, dbeta, coef);
    const long long total4 = (long long)R * C / 4;
    long long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(norm_apply_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, total4, C, y, dout, save_mean,
                       save_rstd, gamma, nullptr, coef, dy);
    return check_launch("sph3d_elu_bn_backward");
}
