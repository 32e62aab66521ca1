/* sph3d_oracle.c — CPU restatement of the SPH3D-GCN tf_ops kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sph3d_gcn_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / timed CPU baseline.
 *
 * Each function restates one reference CUDA kernel (file:line cited) as plain
 * C99, arithmetic operation for arithmetic operation, so that integer outputs
 * (neighbour indices, counts, bin ids, FPS indices, arg-max ids) are
 * bit-identical to what that kernel computes under IEEE-754 float arithmetic
 * without fused multiply-add (build with -ffp-contract=off; see Makefile).
 * The reference has no CPU kernels (every REGISTER_KERNEL_BUILDER is
 * DEVICE_GPU), so this file IS the "reference rebuilt CPU-only".
 *
 * Pinning: the reference ships no tests or golden vectors.  The
 * oracle is pinned against (1) the known-answer cases of SURVEY §8c
 * (tests/test_oracle_kat.py), (2) golden vectors produced by the reference's
 * own .cu files compiled unmodified with hipcc for gfx950 and run on an MI355X
 * (oracle/_ref, tests/golden/, generator tests/golden/make_golden.py), and
 * (3) live comparison with oracle/_ref in the -m gpu tests.
 *
 * OpenMP is used only across independent work items (reference thread chains,
 * output points, channel slices); every output element is produced by exactly
 * one thread, in the reference's sequential order, so results do not depend on
 * the thread count.
 *
 * The one deliberate deviation: atan2f is include/sph3d_atan2f.h (shared with
 * the HIP kernels) instead of libm, see that header.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../include/sph3d_atan2f.h"

#define REF_GRID 32     /* every reference launch is <<<32,1024>>> */
#define REF_BLOCK 1024
#define ORACLE_MAX_GROWTH_PASSES 4096  /* == SPH3D_MAX_GROWTH_PASSES */

#define ORACLE_OK 0
#define ORACLE_EINVAL (-1)

static int imin(int a, int b) { return a < b ? a : b; }

int oracle_abi_version(void) { return 1; }

#ifdef _OPENMP
#include <omp.h>
void oracle_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int oracle_get_max_threads(void) { return omp_get_max_threads(); }
#else
void oracle_set_num_threads(int n) { (void)n; }
int oracle_get_max_threads(void) { return 1; }
#endif

/* ------------------------------------------------------------------------
 * cal_nn_binidx — tf_ops/nnquery/tf_nnquery_gpu.cu:15-65
 * Outputs zeroed first (tf_nnquery.cpp:100-102).
 * Parallel unit = one reference thread (block bb, thread t): it walks its
 * queries i = bb, bb+32, ... ; j = t, t+1024, ... in order and carries the
 * mutated `radius` parameter across them (:59 is never reset).
 * ---------------------------------------------------------------------- */
int oracle_build_sphere_neighbor(int B, int N, int M, int nnSample, float radius0,
                                 const float* database, const float* query,
                                 int* nnIndex, int* nnCount, float* nnDist)
{
    if (B < 0 || N <= 0 || M < 0 || nnSample <= 0 || !(radius0 > 0)) return ORACLE_EINVAL;
    memset(nnIndex, 0, sizeof(int) * (size_t)B * M * nnSample);
    memset(nnCount, 0, sizeof(int) * (size_t)B * M);
    memset(nnDist, 0, sizeof(float) * (size_t)B * M * nnSample);
    const int nb = imin(B, REF_GRID), nt = imin(M, REF_BLOCK);
#pragma omp parallel for collapse(2) schedule(dynamic, 8)
    for (int bb = 0; bb < nb; bb++) {
        for (int t = 0; t < nt; t++) {
            float radius = radius0;                       /* kernel parameter copy, per thread */
            for (int i = bb; i < B; i += REF_GRID) {      /* :21 */
                for (int j = t; j < M; j += REF_BLOCK) {  /* :23 */
                    const float qx = query[(size_t)i * M * 3 + j * 3];
                    const float qy = query[(size_t)i * M * 3 + j * 3 + 1];
                    const float qz = query[(size_t)i * M * 3 + j * 3 + 2];
                    int s = 0, passes = 0;
                    int* idx = nnIndex + ((size_t)i * M + j) * nnSample;
                    float* dst = nnDist + ((size_t)i * M + j) * nnSample;
                    while (s == 0) {                      /* :30 */
                        s = 0;
                        for (int k = 0; k < N; k++) {     /* :35 */
                            const float dx = database[(size_t)i * N * 3 + k * 3] - qx;
                            const float dy = database[(size_t)i * N * 3 + k * 3 + 1] - qy;
                            const float dz = database[(size_t)i * N * 3 + k * 3 + 2] - qz;
                            float dist2D = dx * dx + dy * dy;           /* :45 */
                            float dist3D = dist2D + dz * dz;            /* :46 */
                            dist3D = sqrtf(dist3D);                     /* :47 */
                            /* :49  float < float ; fabs(float) compared with the double 1e-6 */
                            if (dist3D < radius && (double)fabsf(dist3D - radius) > 1e-6) {
                                if (s < nnSample) {
                                    idx[s] = k;
                                    dst[s] = sqrtf(dist3D);             /* :54 sqrt of the distance */
                                }
                                s++;
                            }
                        }
                        radius = (float)((double)radius + 0.05);        /* :59 float += double literal */
                        if (++passes >= ORACLE_MAX_GROWTH_PASSES) break; /* reference would spin */
                    }
                    nnCount[(size_t)i * M + j] = s < nnSample ? s : nnSample; /* :62 */
                }
            }
        }
    }
    return ORACLE_OK;
}

/* Fixed-radius variant (sph3d_build_sphere_neighbor_fixed): NOT reference semantics — the radius is reset for every
 * query, so only the growth-until-one-neighbour loop of a single query remains. */
int oracle_build_sphere_neighbor_fixed(int B, int N, int M, int nnSample, float radius0,
                                 const float* database, const float* query,
                                 int* nnIndex, int* nnCount, float* nnDist)
{
    if (B < 0 || N <= 0 || M < 0 || nnSample <= 0 || !(radius0 > 0)) return ORACLE_EINVAL;
    memset(nnIndex, 0, sizeof(int) * (size_t)B * M * nnSample);
    memset(nnCount, 0, sizeof(int) * (size_t)B * M);
    memset(nnDist, 0, sizeof(float) * (size_t)B * M * nnSample);
    const int nb = imin(B, REF_GRID), nt = imin(M, REF_BLOCK);
#pragma omp parallel for collapse(2) schedule(dynamic, 8)
    for (int bb = 0; bb < nb; bb++) {
        for (int t = 0; t < nt; t++) {
            float radius = radius0;                       /* kernel parameter copy, per thread */
            for (int i = bb; i < B; i += REF_GRID) {      /* :21 */
                for (int j = t; j < M; j += REF_BLOCK) {  /* :23 */
                    radius = radius0;                     /* the deviation: no carry along the chain */
                    const float qx = query[(size_t)i * M * 3 + j * 3];
                    const float qy = query[(size_t)i * M * 3 + j * 3 + 1];
                    const float qz = query[(size_t)i * M * 3 + j * 3 + 2];
                    int s = 0, passes = 0;
                    int* idx = nnIndex + ((size_t)i * M + j) * nnSample;
                    float* dst = nnDist + ((size_t)i * M + j) * nnSample;
                    while (s == 0) {                      /* :30 */
                        s = 0;
                        for (int k = 0; k < N; k++) {     /* :35 */
                            const float dx = database[(size_t)i * N * 3 + k * 3] - qx;
                            const float dy = database[(size_t)i * N * 3 + k * 3 + 1] - qy;
                            const float dz = database[(size_t)i * N * 3 + k * 3 + 2] - qz;
                            float dist2D = dx * dx + dy * dy;           /* :45 */
                            float dist3D = dist2D + dz * dz;            /* :46 */
                            dist3D = sqrtf(dist3D);                     /* :47 */
                            /* :49  float < float ; fabs(float) compared with the double 1e-6 */
                            if (dist3D < radius && (double)fabsf(dist3D - radius) > 1e-6) {
                                if (s < nnSample) {
                                    idx[s] = k;
                                    dst[s] = sqrtf(dist3D);             /* :54 sqrt of the distance */
                                }
                                s++;
                            }
                        }
                        radius = (float)((double)radius + 0.05);        /* :59 float += double literal */
                        if (++passes >= ORACLE_MAX_GROWTH_PASSES) break; /* reference would spin */
                    }
                    nnCount[(size_t)i * M + j] = s < nnSample ? s : nnSample; /* :62 */
                }
            }
        }
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------
 * cal_nn_binidx_cube — tf_nnquery_gpu.cu:72-113 ; zero-init tf_nnquery.cpp:160-161
 * ---------------------------------------------------------------------- */
int oracle_build_cube_neighbor(int B, int N, int M, int gridSize, int nnSample, float length,
                               const float* database, const float* query,
                               int* nnIndex, int* nnCount)
{
    if (B < 0 || N <= 0 || M < 0 || nnSample <= 0 || gridSize <= 0 || !(length > 0)) return ORACLE_EINVAL;
    memset(nnIndex, 0, sizeof(int) * (size_t)B * M * nnSample * 2);
    memset(nnCount, 0, sizeof(int) * (size_t)B * M);
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < B; i++) {
        for (int j = 0; j < M; j++) {
            const float qx = query[(size_t)i * M * 3 + j * 3];
            const float qy = query[(size_t)i * M * 3 + j * 3 + 1];
            const float qz = query[(size_t)i * M * 3 + j * 3 + 2];
            int* idx = nnIndex + ((size_t)i * M + j) * nnSample * 2;
            const float half = length / 2;            /* length/2 : float / int */
            const float cell = length / gridSize;     /* length/gridSize : float / int */
            int s = 0;
            for (int k = 0; k < N; k++) {
                const float dx = database[(size_t)i * N * 3 + k * 3] - qx;
                const float dy = database[(size_t)i * N * 3 + k * 3 + 1] - qy;
                const float dz = database[(size_t)i * N * 3 + k * 3 + 2] - qz;
                if (fabsf(dx) < half && fabsf(dy) < half && fabsf(dz) < half && s < nnSample) { /* :96 */
                    int xId = (int)((dx + half) / cell);  /* :99-101 */
                    int yId = (int)((dy + half) / cell);
                    int zId = (int)((dz + half) / cell);
                    idx[s * 2] = k;
                    idx[s * 2 + 1] = xId * gridSize * gridSize + yId * gridSize + zId;
                    s++;
                }
            }
            nnCount[(size_t)i * M + j] = s;
        }
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------
 * build_spherical_kernel — tf_ops/buildkernel/tf_buildkernel_gpu.cu:20-79
 * M_PI is the glibc double macro: the #ifndef at :1-3 does not
 * fire, so every expression that mentions M_PI is evaluated in double and
 * rounded to float on assignment.  Attr checks tf_buildkernel.cpp:39-49.
 * ---------------------------------------------------------------------- */
#define ORACLE_PI 3.14159265358979323846 /* double, == glibc M_PI */

int oracle_sphere_bin(float dx, float dy, float dz, float dist, float radius, int n, int p, int q)
{
    const float M_EPSf = 1.01e-3F;                         /* :5-7 */
    float dist2D = dx * dx + dy * dy;                      /* :49 */
    dist2D = sqrtf(dist2D);                                /* :50 */
    if (!(dist > M_EPSf && (double)fabsf(dist - M_EPSf) > 1e-6)) return 0;   /* :52-53 */
    float theta = sph3d_atan2f(dy, dx);                    /* :55 */
    float phi = sph3d_atan2f(dz, dist2D);                  /* :56 */
    theta = (float)((double)theta < ORACLE_PI ? (double)theta : -ORACLE_PI);      /* :58 */
    theta = (float)((double)theta > -ORACLE_PI ? (double)theta : -ORACLE_PI);     /* :59 */
    theta = (float)((double)theta + ORACLE_PI);                                   /* :60 */
    phi = (float)((double)phi < (ORACLE_PI / 2) ? (double)phi : (ORACLE_PI / 2)); /* :62 */
    phi = (float)((double)phi > (-ORACLE_PI / 2) ? (double)phi : (-ORACLE_PI / 2)); /* :63 */
    phi = (float)((double)phi + ORACLE_PI / 2);                                   /* :64 */
    /* :66  theta*n -> float*int = float ; /2 -> float ; /M_PI -> double ; store float */
    float alpha = (float)((double)((theta * (float)n) / 2.0f) / ORACLE_PI);
    float beta = (float)((double)(phi * (float)p) / ORACLE_PI);                   /* :67 */
    float gamma = (dist * (float)q) / (radius + 1e-6F);                           /* :68 */
    int nID = imin(n - 1, (int)alpha);                                            /* :70-72 */
    int pID = imin(p - 1, (int)beta);
    int qID = imin(q - 1, (int)gamma);
    return qID * p * n + pID * n + nID + 1;                                       /* :74 */
}

int oracle_spherical_kernel(int B, int N, int M, int K, int n, int p, int q, float radius,
                            const float* database, const float* query,
                            const int* nnIndex, const int* nnCount, const float* nnDist,
                            int* filtIndex)
{
    if (!(radius > 0) || !(n > 2 && n % 2 == 0) || !(p > 0 && p % 2 == 0) || !(q > 0)) return ORACLE_EINVAL;
    memset(filtIndex, 0, sizeof(int) * (size_t)B * M * K);     /* tf_buildkernel.cpp:89 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < B; i++) {
        for (int j = 0; j < M; j++) {
            const float qx = query[(size_t)i * M * 3 + j * 3];
            const float qy = query[(size_t)i * M * 3 + j * 3 + 1];
            const float qz = query[(size_t)i * M * 3 + j * 3 + 2];
            const int nnSize = nnCount[(size_t)i * M + j];
            for (int k = 0; k < nnSize; k++) {
                const int ptID = nnIndex[((size_t)i * M + j) * K + k];
                const float dx = database[(size_t)i * N * 3 + ptID * 3] - qx;
                const float dy = database[(size_t)i * N * 3 + ptID * 3 + 1] - qy;
                const float dz = database[(size_t)i * N * 3 + ptID * 3 + 2] - qz;
                const float dist = nnDist[((size_t)i * M + j) * K + k];
                filtIndex[((size_t)i * M + j) * K + k] = oracle_sphere_bin(dx, dy, dz, dist, radius, n, p, q);
            }
        }
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------
 * depthwise_conv3d_forward — tf_ops/convolution/tf_conv3d_gpu.cu:7-29
 * output zeroed by the op (tf_conv3d.cpp:90).  Per output element the K terms
 * are added in k order, each as (in*filt)/nnSize (:25).
 * ---------------------------------------------------------------------- */
int oracle_depthwise_conv3d(int B, int N, int M, int F, int C, int r, int K,
                            const int* nnIndex, const int* nnCount, const int* binIndex,
                            const float* input, const float* filter, float* output)
{
    (void)F;
    const int CR = C * r;
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < B; i++) {
        for (int m = 0; m < M; m++) {
            const int nnSize = nnCount[(size_t)i * M + m];
            float* out = output + ((size_t)i * M + m) * CR;
            for (int cout = 0; cout < CR; cout++) out[cout] = 0.0f;
            for (int k = 0; k < nnSize; k++) {
                const int n = nnIndex[((size_t)i * M + m) * K + k];
                const int f = binIndex[((size_t)i * M + m) * K + k];
                const float* in = input + ((size_t)i * N + n) * C;
                const float* fl = filter + (size_t)f * CR;
                for (int cout = 0; cout < CR; cout++) {
                    out[cout] += in[cout / r] * fl[cout] / nnSize;   /* :25 */
                }
            }
        }
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------
 * depthwise_input_backward :32-55 and depthwise_filter_backward :58-101
 * (launcher :115-140; zero-init tf_conv3d.cpp:152-153).  The reference
 * accumulates with fp32 atomicAdd in a nondeterministic order; here each
 * thread owns a slice of input channels and walks (i, m, k) in order, which is
 * one of the orders the reference can produce for every element.
 * ---------------------------------------------------------------------- */
int oracle_depthwise_conv3d_grad(int B, int N, int M, int F, int C, int r, int K,
                                 const int* nnIndex, const int* nnCount, const int* binIndex,
                                 const float* input, const float* filter, const float* gradOutput,
                                 float* gradInput, float* gradFilter)
{
    const int CR = C * r;
    memset(gradInput, 0, sizeof(float) * (size_t)B * N * C);
    memset(gradFilter, 0, sizeof(float) * (size_t)F * CR);
    /* work item = (cloud i, slice of input channels): gradInput[i, :, slice] is private to it; the filter gradient
     * is accumulated per cloud (gfPart[i]) and the clouds are added afterwards in order i = 0..B-1, so the result
     * does not depend on the thread count */
    /* a work item's channels span whole 64-byte cache lines of gradInput rows (16 floats) and of the per-cloud filter
     * table (16 * r floats): with 4-channel slices the threads of one cloud wrote into each other's lines and the
     * loop ran 1.2x faster on 256 threads than on one (bench.py cpu_baseline, round 3).  The per-(cloud, channel)
     * summation order — edge order — does not depend on the slice width, so results are unchanged. */
    const int chunk = 16;
    const int nchunks = (C + chunk - 1) / chunk;
    float* gfPart = (float*)calloc((size_t)B * F * CR, sizeof(float));
    if (!gfPart) return ORACLE_EINVAL;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int i = 0; i < B; i++) {
        for (int ch = 0; ch < nchunks; ch++) {
            const int c0 = ch * chunk, c1 = imin(C, c0 + chunk);
            float* gfi = gfPart + (size_t)i * F * CR;
            for (int m = 0; m < M; m++) {
                const int nnSize = nnCount[(size_t)i * M + m];
                const float* go = gradOutput + ((size_t)i * M + m) * CR;
                for (int k = 0; k < nnSize; k++) {
                    const int n = nnIndex[((size_t)i * M + m) * K + k];
                    const int f = binIndex[((size_t)i * M + m) * K + k];
                    const float* in = input + ((size_t)i * N + n) * C;
                    float* gi = gradInput + ((size_t)i * N + n) * C;
                    for (int cin = c0; cin < c1; cin++) {
                        for (int rr = 0; rr < r; rr++) {
                            const int cout = cin * r + rr;
                            gi[cin] += go[cout] * filter[(size_t)f * CR + cout] / nnSize;   /* :50-51 */
                            gfi[(size_t)f * CR + cout] += go[cout] * in[cin] / nnSize;       /* :87 */
                        }
                    }
                }
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int j = 0; j < F * CR; j++) {
        float s = 0.0f;
        for (int i = 0; i < B; i++) s += gfPart[(size_t)i * F * CR + j];
        gradFilter[j] = s;
    }
    free(gfPart);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------
 * max_pool3d_forward — tf_ops/pooling/tf_pool3d_gpu.cu:5-34 (zero-init
 * tf_pool3d.cpp:101-102): first neighbour seeds, strict > replaces.
 * ---------------------------------------------------------------------- */
int oracle_max_pool3d(int B, int N, int M, int C, int K,
                      const int* nnIndex, const int* nnCount, const float* input,
                      float* output, int* maxIndex)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < B; i++) {
        for (int m = 0; m < M; m++) {
            const int nnSize = nnCount[(size_t)i * M + m];
            float* out = output + ((size_t)i * M + m) * C;
            int* mi = maxIndex + ((size_t)i * M + m) * C;
            for (int c = 0; c < C; c++) { out[c] = 0.0f; mi[c] = 0; }
            for (int k = 0; k < nnSize; k++) {
                const int n = nnIndex[((size_t)i * M + m) * K + k];
                const float* in = input + ((size_t)i * N + n) * C;
                for (int c = 0; c < C; c++) {
                    if (k == 0 || in[c] > out[c]) { out[c] = in[c]; mi[c] = n; }
                }
            }
        }
    }
    return ORACLE_OK;
}

/* max_pool3d_backward — tf_pool3d_gpu.cu:38-50 (zero-init tf_pool3d.cpp:142) */
int oracle_max_pool3d_grad(int B, int N, int M, int C,
                           const int* maxIndex, const float* gradOutput, float* gradInput)
{
    memset(gradInput, 0, sizeof(float) * (size_t)B * N * C);
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < B; i++) {
        for (int c = 0; c < C; c++) {          /* work item = (cloud, channel): private output column */
            for (int m = 0; m < M; m++) {
                const int n = maxIndex[((size_t)i * M + m) * C + c];
                gradInput[((size_t)i * N + n) * C + c] += gradOutput[((size_t)i * M + m) * C + c];
            }
        }
    }
    return ORACLE_OK;
}

/* avg_pool3d_forward — tf_pool3d_gpu.cu:53-70 : out += in/nnSize in k order */
int oracle_avg_pool3d(int B, int N, int M, int C, int K,
                      const int* nnIndex, const int* nnCount, const float* input, float* output)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < B; i++) {
        for (int m = 0; m < M; m++) {
            const int nnSize = nnCount[(size_t)i * M + m];
            float* out = output + ((size_t)i * M + m) * C;
            for (int c = 0; c < C; c++) out[c] = 0.0f;
            for (int k = 0; k < nnSize; k++) {
                const int n = nnIndex[((size_t)i * M + m) * K + k];
                const float* in = input + ((size_t)i * N + n) * C;
                for (int c = 0; c < C; c++) out[c] += in[c] / nnSize;
            }
        }
    }
    return ORACLE_OK;
}

/* avg_pool3d_backward — tf_pool3d_gpu.cu:73-90 (atomicAdd of go/nnSize) */
int oracle_avg_pool3d_grad(int B, int N, int M, int C, int K,
                           const int* nnIndex, const int* nnCount, const float* gradOutput,
                           float* gradInput)
{
    memset(gradInput, 0, sizeof(float) * (size_t)B * N * C);
    const int nch = (C + 15) / 16;    /* work item = (cloud, 16-channel slice): private output slice, whole cache lines */
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int i = 0; i < B; i++) {
        for (int ch = 0; ch < nch; ch++) {
            const int c0 = ch * 16, c1 = imin(C, c0 + 16);
            for (int m = 0; m < M; m++) {
                const int nnSize = nnCount[(size_t)i * M + m];
                const float* go = gradOutput + ((size_t)i * M + m) * C;
                for (int k = 0; k < nnSize; k++) {
                    const int n = nnIndex[((size_t)i * M + m) * K + k];
                    float* gi = gradInput + ((size_t)i * N + n) * C;
                    for (int c = c0; c < c1; c++) gi[c] += go[c] / nnSize;
                }
            }
        }
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------
 * mean_interpolate_forward/backward — tf_ops/unpooling/tf_unpool3d_gpu.cu:5-42
 * N = fine/output count, M = coarse/input count (reference naming).
 * ---------------------------------------------------------------------- */
int oracle_mean_interpolate(int B, int N, int M, int C, int K,
                            const int* nnIndex, const int* nnCount, const float* input, float* output)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < B; i++) {
        for (int n = 0; n < N; n++) {
            const int nnSize = nnCount[(size_t)i * N + n];
            float* out = output + ((size_t)i * N + n) * C;
            for (int c = 0; c < C; c++) out[c] = 0.0f;
            for (int k = 0; k < nnSize; k++) {
                const int m = nnIndex[((size_t)i * N + n) * K + k];
                const float* in = input + ((size_t)i * M + m) * C;
                for (int c = 0; c < C; c++) out[c] += in[c] / nnSize;
            }
        }
    }
    return ORACLE_OK;
}

int oracle_mean_interpolate_grad(int B, int N, int M, int C, int K,
                                 const int* nnIndex, const int* nnCount, const float* gradOutput,
                                 float* gradInput)
{
    memset(gradInput, 0, sizeof(float) * (size_t)B * M * C);
    const int nch = (C + 15) / 16;   /* whole cache lines per work item (see oracle_depthwise_conv3d_grad) */
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int i = 0; i < B; i++) {
        for (int ch = 0; ch < nch; ch++) {
            const int c0 = ch * 16, c1 = imin(C, c0 + 16);
            for (int n = 0; n < N; n++) {
                const int nnSize = nnCount[(size_t)i * N + n];
                const float* go = gradOutput + ((size_t)i * N + n) * C;
                for (int k = 0; k < nnSize; k++) {
                    const int m = nnIndex[((size_t)i * N + n) * K + k];
                    float* gi = gradInput + ((size_t)i * M + m) * C;
                    for (int c = c0; c < c1; c++) gi[c] += go[c] / nnSize;
                }
            }
        }
    }
    return ORACLE_OK;
}

/* weighted_interpolate_forward/backward — tf_unpool3d_gpu.cu:45-84 */
int oracle_weighted_interpolate(int B, int N, int M, int C, int K,
                                const int* nnIndex, const int* nnCount,
                                const float* input, const float* weight, float* output)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < B; i++) {
        for (int n = 0; n < N; n++) {
            const int nnSize = nnCount[(size_t)i * N + n];
            float* out = output + ((size_t)i * N + n) * C;
            for (int c = 0; c < C; c++) out[c] = 0.0f;
            for (int k = 0; k < nnSize; k++) {
                const int m = nnIndex[((size_t)i * N + n) * K + k];
                const float w = weight[((size_t)i * N + n) * K + k];
                const float* in = input + ((size_t)i * M + m) * C;
                for (int c = 0; c < C; c++) out[c] += in[c] * w;
            }
        }
    }
    return ORACLE_OK;
}

int oracle_weighted_interpolate_grad(int B, int N, int M, int C, int K,
                                     const int* nnIndex, const int* nnCount,
                                     const float* gradOutput, const float* weight,
                                     float* gradInput)
{
    memset(gradInput, 0, sizeof(float) * (size_t)B * M * C);
    const int nch = (C + 15) / 16;   /* whole cache lines per work item (see oracle_depthwise_conv3d_grad) */
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int i = 0; i < B; i++) {
        for (int ch = 0; ch < nch; ch++) {
            const int c0 = ch * 16, c1 = imin(C, c0 + 16);
            for (int n = 0; n < N; n++) {
                const int nnSize = nnCount[(size_t)i * N + n];
                const float* go = gradOutput + ((size_t)i * N + n) * C;
                for (int k = 0; k < nnSize; k++) {
                    const int m = nnIndex[((size_t)i * N + n) * K + k];
                    const float w = weight[((size_t)i * N + n) * K + k];
                    float* gi = gradInput + ((size_t)i * M + m) * C;
                    for (int c = c0; c < c1; c++) gi[c] += go[c] * w;
                }
            }
        }
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------
 * farthestpointsampleKernel — tf_ops/sampling/tf_sample_gpu.cu:7-73
 * Intended (race-free) semantics, see SURVEY §0.8: with one extra barrier
 * after :68 the kernel equals naive FPS with this tie-break:
 *   per thread t (k = t, t+1024, ...): strict > at :49 keeps the lowest k;
 *   tree :56-66 keeps the LEFT entry unless right is strictly greater, i.e.
 *   the lowest thread id among the maxima.  Idle threads (t >= n) hold
 *   best = -1, besti = 0 (:27-28).
 * temp[] is the running min distance, initialised to 1e38 (:19-21).
 * ---------------------------------------------------------------------- */
int oracle_farthest_point_sample(int b, int n, int m, const float* dataset, int* idxs)
{
    if (m <= 0) return ORACLE_OK;                         /* :8-9 */
    if (n <= 0) return ORACLE_EINVAL;
    int status = ORACLE_OK;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < b; i++) {
        float* temp = (float*)malloc(sizeof(float) * (size_t)n);
        float* tbest = (float*)malloc(sizeof(float) * REF_BLOCK);
        int* tbesti = (int*)malloc(sizeof(int) * REF_BLOCK);
        if (!temp || !tbest || !tbesti) { status = ORACLE_EINVAL; free(temp); free(tbest); free(tbesti); continue; }
        const float* pts = dataset + (size_t)i * n * 3;
        for (int k = 0; k < n; k++) temp[k] = 1e38f;
        int old = 0;
        idxs[(size_t)i * m] = old;
        for (int j = 1; j < m; j++) {
            const float x1 = pts[old * 3], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
            for (int t = 0; t < REF_BLOCK; t++) { tbest[t] = -1.0f; tbesti[t] = 0; }
            for (int k = 0; k < n; k++) {
                const int t = k % REF_BLOCK;
                const float x2 = pts[k * 3], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
                const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1); /* :45 */
                const float td = temp[k];
                const float d2 = d < td ? d : td;    /* min(d,td) :46 */
                if (d2 != td) temp[k] = d2;
                if (d2 > tbest[t]) { tbest[t] = d2; tbesti[t] = k; }   /* :49 (k ascending within a thread) */
            }
            /* tree reduction :56-66: left wins ties */
            for (int u = 0; (1 << u) < REF_BLOCK; u++) {
                for (int t = 0; t < (REF_BLOCK >> (u + 1)); t++) {
                    const int i1 = (t * 2) << u, i2 = (t * 2 + 1) << u;
                    if (tbest[i1] < tbest[i2]) { tbest[i1] = tbest[i2]; tbesti[i1] = tbesti[i2]; }
                }
            }
            old = tbesti[0];
            idxs[(size_t)i * m + j] = old;
        }
        free(temp); free(tbest); free(tbesti);
    }
    return status;
}
