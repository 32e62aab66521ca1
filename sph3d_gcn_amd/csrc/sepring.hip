// sepring.hip — the separable spherical convolution layer as ONE barrier-free kernel, training and inference, gfx950
// ("BN stats as a side reduction ... to remove the [B,M,C*r] intermediate" of utils/sph3gcn_util.py:134-161).
//
//     depthwise_conv3d (tf_conv3d_gpu.cu:7-29)  ->  tf.matmul with the pointwise weights  (+ biases)
//       training : raw product y, the depthwise tensor (the weight gradient's operand: written, never re-read in forward) and
//                  per-column partial sums of elu(y), elu(y)^2 in the layout sph3d_elu_bn_forward_partials consumes
//       inference: -> ELU -> per-channel affine (batch norm with the moving statistics), depthwise tensor never written
//
// Why a second formulation next to sepconv.hip: there the 16 waves of a CU meet at a barrier per 32-point tile, gather and
// product of a wave alternate inside the barrier interval, and the measured layer time is the SUM of the gather kernel and the
// product (298 us against 205 + 85 at level 0): a wave's gather takes 10 .. 64 neighbours, the barrier waits for the slowest.
// Here nothing waits for anything but its own data:
//   * the workgroup's points go through ONE claim counter in LDS; whichever wave is free claims the next point, gathers its
//     neighbour rows exactly like dwconv_fwd_multi (conv3d.hip) and writes the point's C*r depthwise outputs into a row of a
//     RING of 16-point row blocks in LDS (and, training, to HBM: 16 B per lane, one 1-KB row);
//   * a row block whose 16 rows are in (`filled` counter) is multiplied with the pointwise weights by the waves that own its
//     column blocks — W columns RESIDENT IN REGISTERS for the whole launch (v_mfma_f32_16x16x4_f32, four independent chains) —
//     whenever those waves come by: before claiming their next point, or while they wait for a ring slot;
//   * a ring slot is reused when all column-block owners have read it (`consumed` counter).
// So a wave in its 2048 cycles of MFMA shares a SIMD with three waves waiting for feature rows: the product hides in the
// gather's memory latency.  Forward progress: points are claimed in order, the oldest unfinished row block always has a free
// slot, and every waiting wave multiplies what is ready — no wave ever holds something another one needs.  LDS operations of
// one wave execute in order, so "row written, then counter raised" needs no fence, only a compiler barrier.
// Spins are bounded: a wave that polls ~2^22 times raises a device flag and the workgroup drains (results wrong, no hang);
// sph3d_separable_conv3d_ring_failures() reads the flag (tests).
#include <cstdlib>
#include "common.hpp"

namespace sph3d {

typedef float sr_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSrWaves = 16;
constexpr int kSrRows = 16;                 // points per row block (one MFMA row block)
constexpr int kSrSpinLimit = 1 << 22;

__device__ int g_sr_fail = 0;

__device__ __forceinline__ float sr_elu(float y) { return y > 0.f ? y : __expf(y) - 1.f; }      // == norm.hip: elu1

__device__ __forceinline__ int sr_peek(const int* p)
{
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return uniform(v);
}

__device__ __forceinline__ void sr_raise(int* p, int lane)
{
    // everything this wave sent to the LDS before is processed before the increment (in-order LDS queue per wave)
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// R = depth multiplier; LPE = lanes per edge in the gather (16: C <= 64, 32: C <= 128); KT = ceil(C*R / 16) k-groups;
// TRAIN: outputs as described above
template <int R, int LPE, int KT, bool TRAIN>
__global__ __launch_bounds__(1024) void sepconv_ring_kernel(
    int B, int N, int M, int F, int C, int K, int Cout, int act, int NB,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount, const int* __restrict__ binIndex,
    const float* __restrict__ input, const float* __restrict__ dwFilter, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ output, float* __restrict__ dwOut, float* __restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int EPL = 64 / LPE;
    constexpr int NO = 4 * R;
    constexpr int SLI = 4 * LPE;                 // input channels the gather layout covers
    constexpr int FSTB = SLI * R * 4;            // bytes per filter row in LDS
    constexpr int KP = KT * 16;                  // padded k extent (C*R rounded up to 16)
    constexpr int LDA = KP + 4;                  // ring row stride (floats): conflict-free ds_read_b128 across rows
    constexpr int RBF = kSrRows * LDA;           // floats per row block
    float* lfilt = lds;                                          // [F + 1][R][SLI]
    float* ring = lds + (size_t)(F + 1) * SLI * R;               // [NB][16][LDA]
    int* ctrl = reinterpret_cast<int*>(ring + (size_t)NB * RBF); // [0] claim counter, [1] abort, [2 .. 2+NB) filled, [2+NB .. 2+2NB) consumed
    int* filled = ctrl + 2;
    int* consumed = ctrl + 2 + NB;
    const int CR = C * R;
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();

    // ---- depthwise filter table -> LDS (dwconv_fwd_multi's layout), zero row F for padding slots ----
    for (int e = tid * 4; e < F * CR; e += kSrWaves * 64 * 4) {
        const int f = e / CR;
        const int cl = e - f * CR;
        const int l4 = cl / (4 * R), q = (cl >> 2) % R;
        *reinterpret_cast<float4*>(&lfilt[f * (SLI * R) + q * SLI + l4 * 4]) = *reinterpret_cast<const float4*>(&dwFilter[(size_t)f * CR + cl]);
    }
    for (int e = tid; e < SLI * R; e += kSrWaves * 64) lfilt[F * (SLI * R) + e] = 0.f;
    // ring columns beyond C*R (k padding) stay zero for the whole launch
    for (int e = tid; e < NB * RBF; e += kSrWaves * 64) ring[e] = 0.f;
    for (int e = tid; e < 2 + 2 * NB; e += kSrWaves * 64) ctrl[e] = 0;

    // ---- this wave's column block of the product; its W columns in registers ----
    const int ncb = Cout >> 4;                                   // column blocks (<= 16)
    const int G = kSrWaves / ncb;                                // wave groups: group g multiplies the row blocks j = g (mod G)
    const int grp = wave / ncb, cb = wave - grp * ncb;
    const bool gemm_wave = grp < G;
    const int i16 = lane & 15, kq = lane >> 4;
    float wreg[KT][4];
#pragma unroll
    for (int t = 0; t < KT; t++)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = 16 * t + 4 * kq + u;
            wreg[t][u] = (gemm_wave && k < CR) ? W[(size_t)k * Cout + cb * 16 + i16] : 0.f;
        }
    const int col = cb * 16 + i16;
    const float bv = (gemm_wave && bias != nullptr) ? bias[col] : 0.f;
    const float sc = (!TRAIN && gemm_wave && scale != nullptr) ? scale[col] : 1.f;
    const float sh = (!TRAIN && gemm_wave && shift != nullptr) ? shift[col] : 0.f;
    float sz = 0.f, sq = 0.f;                                    // TRAIN: this lane's share of sum elu(y), sum elu(y)^2 of column `col`
    __syncthreads();                                             // the only barrier of the kernel

    // ---- row blocks of this workgroup: a contiguous range of the row blocks of its XCD's clouds ----
    const int rpc = (M + kSrRows - 1) / kSrRows;                 // row blocks per cloud
    const int WPX = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7, wi = (int)blockIdx.x >> 3;
    const bool affine = (B & 7) == 0;
    long long total, part, parts;
    if (affine) { total = (long long)(B >> 3) * rpc; part = wi; parts = WPX; }
    else { total = (long long)B * rpc; part = (long long)xcd * WPX + wi; parts = 8LL * WPX; }
    const int j_begin = (int)(total * part / parts), j_end = (int)(total * (part + 1) / parts);
    const int nrb = j_end - j_begin;
    const int npts = nrb * kSrRows;

    const int g = lane / LPE, li = lane - g * LPE;
    const bool actl = li * 4 < C;
    const int cic = actl ? li * 4 : 0;
    const unsigned cicb = (unsigned)cic * 4u, rowb = (unsigned)C * 4u;
    const char* lfb = reinterpret_cast<const char*>(lfilt);

    // ---- gather of one claimed point p (local index) into its ring row ----
    auto gather_point = [&](int p, int s) {
        const int j = j_begin + (p >> 4);
        const int cl = j / rpc, rbi = j - cl * rpc;
        const int b = affine ? xcd + 8 * cl : cl;
        const int m = rbi * kSrRows + (p & 15);
        const char* inb = reinterpret_cast<const char*>(input + (size_t)b * N * C);
        float acc[NO];
#pragma unroll
        for (int v = 0; v < NO; v++) acc[v] = 0.f;
        int cnt = 0;
        const size_t row = (size_t)b * M + (m < M ? m : 0);
        if (m < M) {
            cnt = uniform(nnCount[row]);
            for (int kt = 0; kt < cnt; kt += 64) {
                const int myk = kt + lane;
                const int kn = (cnt - kt) < 64 ? (cnt - kt) : 64;
                const int mykc = myk < cnt ? myk : kt;
                const int idxv = nnIndex[row * K + mykc];
                int binv = binIndex[row * K + mykc];
                binv = binv < 0 ? 0 : (binv >= F ? F - 1 : binv);
                binv = myk < cnt ? binv : F;
                const unsigned pk = ((unsigned)idxv & 0xffffffu) | ((unsigned)binv << 24);
                for (int k0 = 0; k0 < kn; k0 += 4 * EPL) {
                    float4 x[4];
                    unsigned fo[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int kq2 = k0 + u * EPL + g;
                        const unsigned pp = (unsigned)__shfl((int)pk, kq2);
                        const unsigned off = __umul24(pp, rowb) + cicb;
                        fo[u] = __umul24(pp >> 24, (unsigned)FSTB) + cicb;
                        x[u] = *reinterpret_cast<const float4*>(inb + off);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                        for (int q = 0; q < R; q++) {
                            const float4 w4 = *reinterpret_cast<const float4*>(lfb + fo[u] + q * (SLI * 4));
                            acc[4 * q + 0] = fmaf(xs[(4 * q + 0) / R], w4.x, acc[4 * q + 0]);
                            acc[4 * q + 1] = fmaf(xs[(4 * q + 1) / R], w4.y, acc[4 * q + 1]);
                            acc[4 * q + 2] = fmaf(xs[(4 * q + 2) / R], w4.z, acc[4 * q + 2]);
                            acc[4 * q + 3] = fmaf(xs[(4 * q + 3) / R], w4.w, acc[4 * q + 3]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = LPE; o < 64; o <<= 1)
#pragma unroll
            for (int v = 0; v < NO; v++) acc[v] += __shfl_xor(acc[v], o);
        const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;            // rows past M / empty rows: zeros
#pragma unroll
        for (int v = 0; v < NO; v++) acc[v] *= inv;
        if (actl && g == 0) {
            float* ap = ring + (size_t)s * RBF + (size_t)(p & 15) * LDA + li * 4 * R;
#pragma unroll
            for (int q = 0; q < R; q++)
                *reinterpret_cast<float4*>(ap + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
        sr_raise(&filled[s], lane);
        if (TRAIN && m < M && actl && g == 0) {
            // the depthwise tensor, for the weight gradient (after the ring row: the product does not wait for HBM stores)
            float* op = dwOut + row * CR + (size_t)li * 4 * R;
#pragma unroll
            for (int q = 0; q < R; q++)
                *reinterpret_cast<float4*>(op + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    };

    // ---- product of local row block jl (ring slot s) with this wave's W columns ----
    auto product_block = [&](int jl, int s) {
        sr_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
        const float* arow = ring + (size_t)s * RBF + (size_t)i16 * LDA + 4 * kq;
#pragma unroll
        for (int t = 0; t < KT; t++) {
            const sr_f32x4 a = *reinterpret_cast<const sr_f32x4*>(arow + 16 * t);
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wreg[t][0], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wreg[t][1], d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wreg[t][2], d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wreg[t][3], d3, 0, 0, 0);
        }
        // the slot's rows are in registers / the matrix pipe: hand the slot back before the epilogue's HBM stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sr_raise(&consumed[s], lane);
        const sr_f32x4 d = (d0 + d1) + (d2 + d3);
        const int j = j_begin + jl;
        const int cl = j / rpc, rbi = j - cl * rpc;
        const int b = affine ? xcd + 8 * cl : cl;
        // D layout of 16x16x4: lane holds rows 4*(lane/16) + r, r < 4, of column lane % 16
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const int m = rbi * kSrRows + 4 * kq + r4;
            if (m < M) {
                float y = d[r4] + bv;
                if (TRAIN) {
                    output[((size_t)b * M + m) * Cout + col] = y;
                    const float z = sr_elu(y);
                    sz += z;
                    sq = fmaf(z, z, sq);
                } else {
                    if (act == 1) y = sr_elu(y);
                    output[((size_t)b * M + m) * Cout + col] = fmaf(y, sc, sh);
                }
            }
        }
    };

    // ---- the loop: multiply what is ready, else claim and gather a point, else wait ----
    int my_j = gemm_wave ? grp : 0x7fffffff;       // next local row block this wave multiplies
    int my_s = gemm_wave ? grp % NB : 0, my_e = gemm_wave ? grp / NB : 0;       // its ring slot and epoch
    bool exhausted = npts == 0;
    int spins = 0;
    auto try_product = [&]() -> bool {
        if (my_j >= nrb) return false;
        if (sr_peek(&filled[my_s]) < kSrRows * (my_e + 1)) return false;
        product_block(my_j, my_s);
        my_j += G;
        my_s += G;
        while (my_s >= NB) { my_s -= NB; my_e++; }
        return true;
    };
    auto spin = [&]() -> bool {                     // -> true: give up (the launch is broken; drain)
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSrSpinLimit) {
            if (lane == 0) {
                atomicExch(&g_sr_fail, 1);
                __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            return true;
        }
        return (spins & 1023) == 0 && sr_peek(&ctrl[1]) != 0;
    };
    for (;;) {
        if (try_product()) continue;
        if (!exhausted) {
            int p = 0;
            if (lane == 0) p = __hip_atomic_fetch_add(&ctrl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            p = uniform(p);
            if (p >= npts) {
                exhausted = true;
                continue;
            }
            const int jl = p >> 4;
            const int e = jl / NB, s = jl - e * NB;
            bool dead = false;
            while (sr_peek(&consumed[s]) < ncb * e) {          // the slot's previous row block has not been read by all its owners
                if (try_product()) continue;
                if (spin()) { dead = true; break; }
            }
            if (dead) break;
            gather_point(p, s);
            continue;
        }
        if (my_j >= nrb) break;
        if (spin()) break;
    }

    if (TRAIN && gemm_wave) {
        // partial statistics of this (workgroup, wave group): fold the four row quarters of a column, one store per column
        sz += __shfl_xor(sz, 16);
        sq += __shfl_xor(sq, 16);
        sz += __shfl_xor(sz, 32);
        sq += __shfl_xor(sq, 32);
        if (kq == 0) {
            float* sp = stats + ((size_t)blockIdx.x * G + grp) * 2 * Cout + col;
            sp[0] = sz;
            sp[Cout] = sq;
        }
    }
}

static size_t sr_filter_floats(int F, int C, int r)
{
    const size_t lpe = C <= 64 ? 16 : 32;
    return (size_t)(F + 1) * 4 * lpe * r;
}

static int sr_ktp(int C, int r)
{
    const int KT = (C * r + 15) / 16;
    return KT <= 4 ? 4 : (KT <= 8 ? 8 : 16);
}

// ring depth: as many 16-point row blocks as fit beside the filter table (at most 8)
static int sr_ring_depth(int F, int C, int r)
{
    const size_t rb = sizeof(float) * kSrRows * (size_t)(sr_ktp(C, r) * 16 + 4);
    const size_t fixed = sizeof(float) * sr_filter_floats(F, C, r) + 256;
    if (fixed + 3 * rb > 160 * 1024) return 0;
    size_t nb = (160 * 1024 - fixed) / rb;
    static const int forced = getenv("SPH3D_SR_NB") ? atoi(getenv("SPH3D_SR_NB")) : 0;      // (experiments)
    if (forced >= 3 && (size_t)forced <= nb) return forced;
    return nb > 8 ? 8 : (int)nb;
}

static bool sr_shape_ok(int N, int F, int C, int r, int K, int Cout)
{
    // Cout: 16 column-block owners must divide the 16 waves
    const bool cout_ok = Cout == 16 || Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256;
    return (r == 1 || r == 2) && C % 4 == 0 && C >= 4 && C <= 128 && C * r <= 256 && cout_ok && F <= 254 && N <= (1 << 24) && K > 0 &&
           (unsigned long long)N * C * 4ull + 1024ull < (1ull << 32) && sr_ring_depth(F, C, r) >= 3;
}

template <int R, int LPE, bool TRAIN>
static int sr_launch(int B, int N, int M, int F, int C, int K, int Cout, int act, const int* nn_index, const int* nn_count,
                     const int* bin_index, const float* input, const float* dw_filter, const float* W, const float* bias,
                     const float* scale, const float* shift, float* output, float* dw_out, float* stats, hipStream_t st)
{
    const int KTP = sr_ktp(C, R);
    const int NB = sr_ring_depth(F, C, R);
    const size_t lds = sizeof(float) * (sr_filter_floats(F, C, R) + (size_t)NB * kSrRows * (KTP * 16 + 4)) + sizeof(int) * (2 + 2 * NB);
    SPH3D_REQUIRE(NB >= 3 && lds <= 160 * 1024, "SeparableConv3dRing: %zu B of LDS needed", lds);
#define SPH3D_SR(KTT)                                                                                                        \
    {                                                                                                                        \
        auto kern = sepconv_ring_kernel<R, LPE, KTT, TRAIN>;                                                                 \
        if (lds > 48 * 1024) {                                                                                               \
            int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), \
                               "SeparableConv3dRing: hipFuncSetAttribute");                                                  \
            if (rc) return rc;                                                                                               \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, st, B, N, M, F, C, K, Cout, act, NB, nn_index, nn_count, bin_index, \
                           input, dw_filter, W, bias, scale, shift, output, dw_out, stats);                                  \
    }
    if (KTP == 4) SPH3D_SR(4)
    else if (KTP == 8) SPH3D_SR(8)
    else SPH3D_SR(16)
#undef SPH3D_SR
    return check_launch("sph3d_separable_conv3d_ring");
}

template <bool TRAIN>
static int sr_dispatch(int B, int N, int M, int F, int C, int r, int K, int Cout, int act, const int* nn_index, const int* nn_count,
                       const int* bin_index, const float* input, const float* dw_filter, const float* W, const float* bias,
                       const float* scale, const float* shift, float* output, float* dw_out, float* stats, hipStream_t st)
{
    if (C <= 64)
        return r == 2 ? sr_launch<2, 16, TRAIN>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, dw_filter, W, bias, scale,
                                                shift, output, dw_out, stats, st)
                      : sr_launch<1, 16, TRAIN>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, dw_filter, W, bias, scale,
                                                shift, output, dw_out, stats, st);
    return r == 2 ? sr_launch<2, 32, TRAIN>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, dw_filter, W, bias, scale,
                                            shift, output, dw_out, stats, st)
                  : sr_launch<1, 32, TRAIN>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, dw_filter, W, bias, scale,
                                            shift, output, dw_out, stats, st);
}

// used by sepconv.hip: the inference layer through the ring kernel where it covers the shape
bool sepring_infer_ok(int N, int F, int C, int r, int K, int Cout) { return sr_shape_ok(N, F, C, r, K, Cout); }
int sepring_infer(int B, int N, int M, int F, int C, int r, int K, int Cout, int act, const int* nn_index, const int* nn_count,
                  const int* bin_index, const float* input, const float* dw_filter, const float* W, const float* bias,
                  const float* scale, const float* shift, float* output, hipStream_t st)
{
    return sr_dispatch<false>(B, N, M, F, C, r, K, Cout, act, nn_index, nn_count, bin_index, input, dw_filter, W, bias, scale, shift,
                              output, nullptr, nullptr, st);
}

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_separable_conv3d_train_supported(int N, int F, int C, int r, int K, int Cout)
{
    return sr_shape_ok(N, F, C, r, K, Cout) ? 1 : 0;
}

extern "C" int sph3d_separable_conv3d_train_blocks(int Cout)
{
    if (Cout < 16 || Cout > 256 || (Cout & (Cout - 1)) != 0) return 0;
    return 256 * (kSrWaves / (Cout >> 4));
}

extern "C" int sph3d_separable_conv3d_train(int B, int N, int M, int F, int C, int r, int K, int Cout,
                                            const int* nn_index, const int* nn_count, const int* bin_index,
                                            const float* input, const float* depthwise_filter, const float* pointwise_weights,
                                            const float* bias, float* depthwise_output, float* y, float* partial,
                                            sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0 && F > 0 && C > 0 && K > 0 && Cout > 0,
                  "SeparableConv3dTrain: bad dims B=%d N=%d M=%d F=%d C=%d K=%d Cout=%d", B, N, M, F, C, K, Cout);
    if (!sr_shape_ok(N, F, C, r, K, Cout)) {
        set_error("SeparableConv3dTrain: shape C=%d r=%d Cout=%d F=%d not covered (C %% 4 == 0, C <= 128, C*r <= 256, "
                  "Cout in {16, 32, 64, 128, 256}, r in {1, 2})", C, r, Cout, F);
        return SPH3D_EUNSUPPORTED;
    }
    SPH3D_REQUIRE(depthwise_output != nullptr && y != nullptr && partial != nullptr, "SeparableConv3dTrain: NULL output");
    // (B == 0 or M == 0: the launch still writes the all-zero partial rows the statistics finalize reads)
    return sr_dispatch<true>(B, N, M, F, C, r, K, Cout, 0, nn_index, nn_count, bin_index, input, depthwise_filter, pointwise_weights,
                             bias, nullptr, nullptr, y, depthwise_output, partial, as_stream(stream));
}

extern "C" int sph3d_separable_conv3d_ring_failures(void)
{
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sr_fail), sizeof(int), 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}
