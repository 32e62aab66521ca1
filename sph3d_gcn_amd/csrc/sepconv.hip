// sepconv.hip — fused separable spherical convolution for inference, gfx950 (second half).
//
// One kernel for the whole layer of utils/sph3gcn_util.py:134-161 in inference mode:
//     depthwise_conv3d (tf_conv3d_gpu.cu:7-29)  ->  tf.matmul with the pointwise weights  ->  + biases  ->  ELU  ->
//     batch norm with the moving statistics (an affine map per output channel)
// The [B, M, C*r] depthwise tensor never leaves the CU: a persistent 16-wave workgroup per CU takes tiles of 32 output
// points; every wave gathers two of them exactly like dwconv_fwd_multi (conv3d.hip: 16 / 32 lanes per edge, packed id | bin
// hand-over, LDS filter table) and writes the 32 x C*r tile to LDS; then the 16 waves multiply the tile with the pointwise
// weights on the matrix cores — v_mfma_f32_16x16x4_f32, one 16 x 16 output block per wave, its 16 columns of W
// (C*r x 16 floats = 64 VGPRs at C*r = 256) RESIDENT IN REGISTERS for the whole launch — and apply the epilogue.
// In training the depthwise tensor is the weight gradient's operand and has to be written anyway, so this
// path is taken when no gradient is recorded.  Shapes outside C <= 128, C*r <= 256, Cout <= 128 — every deeper layer of the
// S3DIS / ShapeNet plans — run sepconv_general_kernel below (accumulators resident, W streamed per k slice).
//
// k-order of the product: lane (i = lane % 16, kq = lane / 16) supplies A[i][16t + 4kq + u] and B[16t + 4kq + u][j] at step
// (t, u): any pairing is legal as long as A and B use the same one, and this one makes a lane's four A values of a t ONE
// ds_read_b128.  fp32 MFMA is exact fp32 FMA arithmetic; results differ from the separate kernels by summation order only.
#include <cstdlib>
#include "common.hpp"

namespace sph3d {

typedef float sc_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kScWaves = 16;
constexpr int kScTile = 32;                 // output points per tile (two per wave in the gather phase)

__device__ __forceinline__ float sc_elu(float y) { return y > 0.f ? y : __expf(y) - 1.f; }      // == norm.hip: elu1

// R = depth multiplier; LPE = lanes per edge in the gather phase (16: C <= 64, 32: C <= 128); KT = ceil(C*R / 16) k-groups
template <int R, int LPE, int KT>
__global__ __launch_bounds__(1024) void sepconv_fused_kernel(
    int B, int N, int M, int F, int C, int K, int Cout, int act,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount, const int* __restrict__ binIndex,
    const float* __restrict__ input, const float* __restrict__ dwFilter, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ output)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int EPL = 64 / LPE;
    constexpr int NO = 4 * R;
    constexpr int SLI = 4 * LPE;                 // input channels the gather layout covers
    constexpr int FSTB = SLI * R * 4;            // bytes per filter row in LDS
    constexpr int KP = KT * 16;                  // padded k extent (C*R rounded up to 16)
    constexpr int LDA = KP + 4;                  // A-tile row stride (floats): conflict-free ds_read_b128 across rows
    float* lfilt = lds;                                          // [F + 1][R][SLI]
    float* atile = lds + (size_t)(F + 1) * SLI * R;              // [2][kScTile][LDA]: double buffered
    const int CR = C * R;
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();

    // ---- depthwise filter table -> LDS (dwconv_fwd_multi's layout), zero row F for padding slots ----
    for (int e = tid * 4; e < F * CR; e += kScWaves * 64 * 4) {
        const int f = e / CR;
        const int cl = e - f * CR;
        const int l4 = cl / (4 * R), q = (cl >> 2) % R;
        *reinterpret_cast<float4*>(&lfilt[f * (SLI * R) + q * SLI + l4 * 4]) = *reinterpret_cast<const float4*>(&dwFilter[(size_t)f * CR + cl]);
    }
    for (int e = tid; e < SLI * R; e += kScWaves * 64) lfilt[F * (SLI * R) + e] = 0.f;
    // columns of the A tile beyond C*R (k padding) stay zero for the whole launch
    for (int e = tid; e < 2 * kScTile * LDA; e += kScWaves * 64) atile[e] = 0.f;

    // ---- this wave's block of the product: rows rb*16.., columns cb*16..; its W columns in registers ----
    const int ncb = Cout >> 4;
    const int rb = wave / ncb, cb = wave - rb * ncb;
    const bool gemm_wave = rb < (kScTile / 16);
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
    const float sc = (gemm_wave && scale != nullptr) ? scale[col] : 1.f;
    const float sh = (gemm_wave && shift != nullptr) ? shift[col] : 0.f;
    __syncthreads();

    // ---- tiles of this workgroup: contiguous range of the tiles of its XCD's clouds ----
    const int tpc = (M + kScTile - 1) / kScTile;                  // tiles per cloud
    const int WPX = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7, wi = (int)blockIdx.x >> 3;
    const bool affine = (B & 7) == 0;
    long long total, part, parts;
    if (affine) { total = (long long)(B >> 3) * tpc; part = wi; parts = WPX; }
    else { total = (long long)B * tpc; part = (long long)xcd * WPX + wi; parts = 8LL * WPX; }
    const int f_begin = (int)(total * part / parts), f_end = (int)(total * (part + 1) / parts);

    const int g = lane / LPE, li = lane - g * LPE;
    const bool actl = li * 4 < C;
    const int cic = actl ? li * 4 : 0;
    const unsigned cicb = (unsigned)cic * 4u, rowb = (unsigned)C * 4u;
    const char* lfb = reinterpret_cast<const char*>(lfilt);

    // ---- software pipeline over the tiles: in iteration `it` the workgroup gathers tile `it` into A buffer it & 1 and
    // multiplies tile it - 1 out of the other buffer; half of the waves of every SIMD (wave / 4 odd) run the product first
    // and the gather second, so that the matrix pipe (64 x 32-cycle MFMAs per wave and tile: the product alone is
    // MFMA-rate bound at the fp32 peak) and the gather's VALU / LDS / VMEM work overlap inside a SIMD.
    auto gather_tile = [&](int b, int m0, float* abuf) {
        const char* inb = reinterpret_cast<const char*>(input + (size_t)b * N * C);
#pragma unroll 1
        for (int h = 0; h < 2; h++) {                 // wave w takes points m0 + w and m0 + w + 16
            const int p = wave + kScWaves * h;
            const int m = m0 + p;
            float acc[NO];
#pragma unroll
            for (int v = 0; v < NO; v++) acc[v] = 0.f;
            int cnt = 0;
            if (m < M) {
                const size_t row = (size_t)b * M + m;
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
            if (actl && g == 0) {
                const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;        // rows past M / empty rows: zeros
                float* ap = abuf + (size_t)p * LDA + li * 4 * R;
#pragma unroll
                for (int q = 0; q < R; q++)
                    *reinterpret_cast<float4*>(ap + 4 * q) =
                        make_float4(acc[4 * q] * inv, acc[4 * q + 1] * inv, acc[4 * q + 2] * inv, acc[4 * q + 3] * inv);
            }
        }
    };
    auto product_tile = [&](int b, int m0, const float* abuf) {
        if (!gemm_wave) return;
        // four independent accumulation chains (one per k-slot of the 16-byte read), added at the end: one chain of 4 KT
        // dependent MFMAs leaves the wave waiting for its own previous result (skinny.hip: 46 -> 33 us from the same change)
        sc_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
        const float* arow = abuf + (size_t)(rb * 16 + i16) * LDA + 4 * kq;
#pragma unroll
        for (int t = 0; t < KT; t++) {
            const sc_f32x4 a = *reinterpret_cast<const sc_f32x4*>(arow + 16 * t);
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wreg[t][0], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wreg[t][1], d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wreg[t][2], d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wreg[t][3], d3, 0, 0, 0);
        }
        const sc_f32x4 d = (d0 + d1) + (d2 + d3);
        // D layout of 16x16x4: lane holds rows 4*(lane/16) + r, r < 4, of column lane % 16
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const int m = m0 + rb * 16 + 4 * kq + r4;
            if (m < M) {
                float y = d[r4] + bv;
                if (act == 1) y = sc_elu(y);
                output[((size_t)b * M + m) * Cout + col] = fmaf(y, sc, sh);
            }
        }
    };

    const bool product_first = ((wave >> 2) & 1) != 0;
    int cl = f_begin / tpc, tl = f_begin - cl * tpc;
    int pb = 0, pm0 = 0;
    for (int it = f_begin; it <= f_end; it++) {
        const int b = affine ? xcd + 8 * cl : cl;
        const int m0 = tl * kScTile;
        float* gbuf = atile + (size_t)(it & 1) * (kScTile * LDA);
        const float* mbuf = atile + (size_t)((it & 1) ^ 1) * (kScTile * LDA);
        const bool do_g = it < f_end, do_p = it > f_begin;
        if (product_first) {
            if (do_p) product_tile(pb, pm0, mbuf);
            if (do_g) gather_tile(b, m0, gbuf);
        } else {
            if (do_g) gather_tile(b, m0, gbuf);
            if (do_p) product_tile(pb, pm0, mbuf);
        }
        __syncthreads();
        pb = b;
        pm0 = m0;
        if (++tl == tpc) { tl = 0; cl++; }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// General shapes (round 4): any layer of the S3DIS / ShapeNet plans (C up to 1024 in slices of 128 input channels, Cout up to
// 512).  The pointwise weights no longer fit a wave's registers (C*r x Cout = up to 2048 x 512 floats), so the product is
// organised the other way round: the ACCUMULATORS stay resident — wave w owns the output column blocks cb = w, w + 16 of a
// tile of TP = 16 * RBK points (RBK x 2 blocks of 16 x 16: <= 32 VGPRs) across the whole k range — and the layer is walked
// slice by slice: filter slice -> LDS, the 16 waves gather the slice's depthwise outputs of the tile's points into the
// LDS A tile [TP][SLI*R] (same gather as above), then every wave multiplies the tile with ITS columns of the slice's
// k-rows of W, fetched from global memory / L2 as they are needed (a lane's four B values of a t are four 4-byte loads of
// 16 consecutive columns; each fragment feeds RBK row blocks).  Per tile W is read once (C*r*Cout*4 B from L2): the kernel
// trades the [B, M, C*r] depthwise tensor's HBM round trip (write + read) for L2 reads of W per tile.
// ------------------------------------------------------------------------------------------------------------------
template <int R, int LPE, int RBK>
__global__ __launch_bounds__(1024) void sepconv_general_kernel(
    int B, int N, int M, int F, int C, int K, int Cout, int act,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount, const int* __restrict__ binIndex,
    const float* __restrict__ input, const float* __restrict__ dwFilter, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ output)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int EPL = 64 / LPE;
    constexpr int NO = 4 * R;
    constexpr int SLI = 4 * LPE;                 // input channels per slice
    constexpr int FSTB = SLI * R * 4;            // bytes per filter row in LDS
    constexpr int KP = SLI * R;                  // k extent of a slice
    constexpr int KT = KP / 16;
    constexpr int LDA = KP + 4;
    constexpr int TP = 16 * RBK;                 // points per tile
    constexpr int PPW = (TP + kScWaves - 1) / kScWaves;      // points a wave gathers per tile and slice
    float* lfilt = lds;                                          // [F + 1][R][SLI]
    float* atile = lds + (size_t)(F + 1) * SLI * R;              // [TP][LDA]
    const int CR = C * R;
    const int tid = (int)threadIdx.x;
    const int wave = uniform(tid >> 6);
    const int lane = lane_id();
    const int nslices = (C + SLI - 1) / SLI;
    const int ncb = Cout >> 4;                                   // <= 32
    const int i16 = lane & 15, kq = lane >> 4;

    const int tpc = (M + TP - 1) / TP;                           // tiles per cloud
    const int WPX = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7, wi = (int)blockIdx.x >> 3;
    const bool affine = (B & 7) == 0;
    long long total, part, parts;
    if (affine) { total = (long long)(B >> 3) * tpc; part = wi; parts = WPX; }
    else { total = (long long)B * tpc; part = (long long)xcd * WPX + wi; parts = 8LL * WPX; }
    const int f_begin = (int)(total * part / parts), f_end = (int)(total * (part + 1) / parts);

    const int g = lane / LPE, li = lane - g * LPE;
    const unsigned rowb = (unsigned)C * 4u;
    const char* lfb = reinterpret_cast<const char*>(lfilt);
    for (int e = tid; e < SLI * R; e += kScWaves * 64) lfilt[F * (SLI * R) + e] = 0.f;      // zero filter row of padding slots

    int cl = f_begin / tpc, tl = f_begin - cl * tpc;
    for (int it = f_begin; it < f_end; it++) {
        const int b = affine ? xcd + 8 * cl : cl;
        const int m0 = tl * TP;
        sc_f32x4 d[2][RBK];
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
            for (int rb = 0; rb < RBK; rb++) d[c2][rb] = sc_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nslices; s++) {
            const int c0 = s * SLI;
            const int SLc = (C - c0) < SLI ? (C - c0) : SLI;             // channels of this slice (multiple of 4)
            __syncthreads();                                              // the previous slice's product is done with LDS
            // ---- filter slice -> LDS ----
            for (int e = tid * 4; e < F * SLc * R; e += kScWaves * 64 * 4) {
                const int f = e / (SLc * R);
                const int cl4 = e - f * (SLc * R);
                const int l4 = cl4 / (4 * R), q = (cl4 >> 2) % R;
                *reinterpret_cast<float4*>(&lfilt[f * (SLI * R) + q * SLI + l4 * 4]) =
                    *reinterpret_cast<const float4*>(&dwFilter[(size_t)f * CR + (size_t)c0 * R + cl4]);
            }
            __syncthreads();
            // ---- gather: wave w takes points m0 + w + 16 h of the tile ----
            const bool actl = li * 4 < SLc;
            const unsigned cicb = (unsigned)(c0 + (actl ? li * 4 : 0)) * 4u;
            const unsigned ficb = (unsigned)(actl ? li * 4 : 0) * 4u;
            const char* inb = reinterpret_cast<const char*>(input + (size_t)b * N * C);
#pragma unroll 1
            for (int h = 0; h < PPW; h++) {
                const int p = wave + kScWaves * h;
                if (p >= TP) break;
                const int m = m0 + p;
                float acc[NO];
#pragma unroll
                for (int v = 0; v < NO; v++) acc[v] = 0.f;
                int cnt = 0;
                if (m < M) {
                    const size_t row = (size_t)b * M + m;
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
                                fo[u] = __umul24(pp >> 24, (unsigned)FSTB) + ficb;
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
                if (g == 0) {
                    // lanes beyond the slice's channels write zeros: k columns the product multiplies with W rows that do not exist
                    const float inv = (cnt > 0 && actl) ? 1.0f / (float)cnt : 0.f;
                    float* ap = atile + (size_t)p * LDA + li * 4 * R;
#pragma unroll
                    for (int q = 0; q < R; q++)
                        *reinterpret_cast<float4*>(ap + 4 * q) =
                            make_float4(acc[4 * q] * inv, acc[4 * q + 1] * inv, acc[4 * q + 2] * inv, acc[4 * q + 3] * inv);
                }
            }
            __syncthreads();
            // ---- product: this wave's column blocks x every row block of the tile, k rows of this slice ----
            const int kbase = c0 * R;                                    // first k row of the slice in W
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++) {
                const int cb = wave + kScWaves * c2;
                if (cb >= ncb) break;
                const float* wp = W + (size_t)kbase * Cout + cb * 16 + i16;
#pragma unroll 1
                for (int t = 0; t < KT; t++) {
                    float bw[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int k = 16 * t + 4 * kq + u;
                        bw[u] = k < SLc * R ? wp[(size_t)k * Cout] : 0.f;
                    }
                    sc_f32x4 a[RBK];
#pragma unroll
                    for (int rb = 0; rb < RBK; rb++)
                        a[rb] = *reinterpret_cast<const sc_f32x4*>(atile + (size_t)(rb * 16 + i16) * LDA + 16 * t + 4 * kq);
                    // (u outer, row blocks inner: consecutive MFMAs write different accumulators)
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int rb = 0; rb < RBK; rb++)
                            d[c2][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][u], bw[u], d[c2][rb], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: bias -> ELU -> per-channel affine; D layout: lane holds rows 4*(lane/16) + r of column lane % 16 ----
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
            const int cb = wave + kScWaves * c2;
            if (cb >= ncb) break;
            const int col = cb * 16 + i16;
            const float bv = bias != nullptr ? bias[col] : 0.f;
            const float sc = scale != nullptr ? scale[col] : 1.f;
            const float sh = shift != nullptr ? shift[col] : 0.f;
#pragma unroll
            for (int rb = 0; rb < RBK; rb++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const int m = m0 + rb * 16 + 4 * kq + r4;
                    if (m < M) {
                        float y = d[c2][rb][r4] + bv;
                        if (act == 1) y = sc_elu(y);
                        output[((size_t)b * M + m) * Cout + col] = fmaf(y, sc, sh);
                    }
                }
        }
        if (++tl == tpc) { tl = 0; cl++; }
    }
}

// shapes of the general kernel: C <= 128 (one, possibly partial, slice) or a multiple of 128; Cout a multiple of 16, <= 512
// LDS of the general kernel: the depthwise filter slice (F + 1 rows) + the A tile of rbk row blocks
static size_t sc_general_lds(int F, int C, int r, int rbk)
{
    const size_t lpe = C <= 64 ? 16 : 32;
    return sizeof(float) * ((size_t)(F + 1) * 4 * lpe * r + (size_t)(16 * rbk) * (4 * lpe * r + 4));
}

static bool sc_general_ok(int N, int F, int C, int r, int K, int Cout)
{
    // "supported" is exactly what the launcher can run: the smallest tile (one row block) must fit the CU's LDS next to the
    // filter slice (F <= 254 alone let kernels with ~95+ bins through, which then failed at launch)
    return (r == 1 || r == 2) && sc_general_lds(F, C, r, 1) <= 160 * 1024 && C % 4 == 0 && C >= 4 && (C <= 128 || (C % 128 == 0 && C <= 4096)) && Cout % 16 == 0 && Cout >= 16 &&
           Cout <= 512 && F <= 254 && N <= (1 << 24) && K > 0 && (unsigned long long)N * C * 4ull + 1024ull < (1ull << 32);
}

template <int R, int LPE>
static int sc_launch_general(int B, int N, int M, int F, int C, int K, int Cout, int act, const int* nn_index, const int* nn_count,
                             const int* bin_index, const float* input, const float* dw_filter, const float* W, const float* bias,
                             const float* scale, const float* shift, float* output, hipStream_t st)
{
    // tile height: as many row blocks as still leave every CU a tile
    const long long pts = (long long)B * M;
    int rbk = pts >= 64LL * 512 ? 4 : (pts >= 32LL * 256 ? 2 : 1);
    while (rbk > 1 && sc_general_lds(F, C, R, rbk) > 160 * 1024) rbk >>= 1;      // many bins: a lower tile beside the larger filter slice
    const size_t lds = sizeof(float) * ((size_t)(F + 1) * 4 * LPE * R + (size_t)(16 * rbk) * (4 * LPE * R + 4));
    SPH3D_REQUIRE(lds <= 160 * 1024, "SeparableConv3dFused: %zu B of LDS needed", lds);
#define SPH3D_SCG(RB)                                                                                                        \
    {                                                                                                                        \
        auto kern = sepconv_general_kernel<R, LPE, RB>;                                                                      \
        if (lds > 48 * 1024) {                                                                                               \
            int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), \
                               "SeparableConv3dFused: hipFuncSetAttribute");                                                 \
            if (rc) return rc;                                                                                               \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, st, B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, \
                           dw_filter, W, bias, scale, shift, output);                                                        \
    }
    if (rbk == 4) SPH3D_SCG(4)
    else if (rbk == 2) SPH3D_SCG(2)
    else SPH3D_SCG(1)
#undef SPH3D_SCG
    return check_launch("sph3d_separable_conv3d_fused (general)");
}

static size_t sc_small_lds(int F, int C, int r)
{
    const size_t lpe = C <= 64 ? 16 : 32;
    const int KT = (C * r + 15) / 16;
    const int KTP = KT <= 4 ? 4 : (KT <= 8 ? 8 : 16);
    return sizeof(float) * ((size_t)(F + 1) * 4 * lpe * r + 2 * (size_t)kScTile * (KTP * 16 + 4));
}

static bool sc_shape_ok(int N, int F, int C, int r, int K, int Cout)
{
    return (r == 1 || r == 2) && sc_small_lds(F, C, r) <= 160 * 1024 && C % 4 == 0 && C >= 4 && C <= 128 && C * r <= 256 && Cout % 16 == 0 && Cout >= 16 && Cout <= 128 &&
           F <= 254 && N <= (1 << 24) && K > 0 && (unsigned long long)N * C * 4ull + 1024ull < (1ull << 32);
}

template <int R, int LPE>
static int sc_launch(int B, int N, int M, int F, int C, int K, int Cout, int act, const int* nn_index, const int* nn_count,
                     const int* bin_index, const float* input, const float* dw_filter, const float* W, const float* bias,
                     const float* scale, const float* shift, float* output, hipStream_t st)
{
    const int KT = (C * R + 15) / 16;
    const int KTP = KT <= 4 ? 4 : (KT <= 8 ? 8 : 16);           // the k extent the kernel is instantiated for
    const size_t lds = sizeof(float) * ((size_t)(F + 1) * 4 * LPE * R + 2 * (size_t)kScTile * (KTP * 16 + 4));
    SPH3D_REQUIRE(lds <= 160 * 1024, "SeparableConv3dFused: %zu B of LDS needed", lds);
#define SPH3D_SC(KTT)                                                                                                        \
    {                                                                                                                        \
        auto kern = sepconv_fused_kernel<R, LPE, KTT>;                                                                       \
        if (lds > 48 * 1024) {                                                                                               \
            int rc = check_hip(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), \
                               "SeparableConv3dFused: hipFuncSetAttribute");                                                 \
            if (rc) return rc;                                                                                               \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, st, B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, \
                           dw_filter, W, bias, scale, shift, output);                                                        \
    }
    if (KTP == 4) SPH3D_SC(4)
    else if (KTP == 8) SPH3D_SC(8)
    else SPH3D_SC(16)
#undef SPH3D_SC
    return check_launch("sph3d_separable_conv3d_fused");
}

// sepring.hip: the same layer without barriers (claim counter + ring of row blocks); covers C <= 128, C*r <= 256, Cout a power of two
bool sepring_infer_ok(int N, int F, int C, int r, int K, int Cout);
int sepring_infer(int B, int N, int M, int F, int C, int r, int K, int Cout, int act, const int* nn_index, const int* nn_count,
                  const int* bin_index, const float* input, const float* dw_filter, const float* W, const float* bias,
                  const float* scale, const float* shift, float* output, hipStream_t st);

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_separable_conv3d_fused_supported(int N, int F, int C, int r, int K, int Cout)
{
    return (sc_shape_ok(N, F, C, r, K, Cout) || sc_general_ok(N, F, C, r, K, Cout)) ? 1 : 0;
}

extern "C" int sph3d_separable_conv3d_fused(int B, int N, int M, int F, int C, int r, int K, int Cout, int act,
                                            const int* nn_index, const int* nn_count, const int* bin_index,
                                            const float* input, const float* depthwise_filter, const float* pointwise_weights,
                                            const float* bias, const float* scale, const float* shift, float* output,
                                            sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0 && F > 0 && C > 0 && K > 0 && Cout > 0,
                  "SeparableConv3dFused: bad dims B=%d N=%d M=%d F=%d C=%d K=%d Cout=%d", B, N, M, F, C, K, Cout);
    SPH3D_REQUIRE(act == 0 || act == 1, "SeparableConv3dFused: act must be 0 (none) or 1 (ELU), got %d", act);
    const bool small = sc_shape_ok(N, F, C, r, K, Cout);
    if (!small && !sc_general_ok(N, F, C, r, K, Cout)) {
        set_error("SeparableConv3dFused: shape C=%d r=%d Cout=%d F=%d not covered (C %% 4 == 0 and C <= 128 or a multiple of 128; "
                  "Cout <= 512 in multiples of 16; r in {1, 2})", C, r, Cout, F);
        return SPH3D_EUNSUPPORTED;
    }
    if (B == 0 || M == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    // SPH3D_SC_RING: 1 (default) the barrier-free kernel wherever it covers the shape, 0 never (A/B measurements)
    static const bool use_ring = !(getenv("SPH3D_SC_RING") && atoi(getenv("SPH3D_SC_RING")) == 0);
    if (use_ring && sepring_infer_ok(N, F, C, r, K, Cout))
        return sepring_infer(B, N, M, F, C, r, K, Cout, act, nn_index, nn_count, bin_index, input, depthwise_filter,
                             pointwise_weights, bias, scale, shift, output, st);
    if (!small) {
        // the layers whose pointwise weights do not fit a wave's registers: accumulators resident, W streamed per k slice
        if (C <= 64)
            return r == 2 ? sc_launch_general<2, 16>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, depthwise_filter,
                                                     pointwise_weights, bias, scale, shift, output, st)
                          : sc_launch_general<1, 16>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, depthwise_filter,
                                                     pointwise_weights, bias, scale, shift, output, st);
        return r == 2 ? sc_launch_general<2, 32>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, depthwise_filter,
                                                 pointwise_weights, bias, scale, shift, output, st)
                      : sc_launch_general<1, 32>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, depthwise_filter,
                                                 pointwise_weights, bias, scale, shift, output, st);
    }
    if (C <= 64)
        return r == 2 ? sc_launch<2, 16>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, depthwise_filter,
                                         pointwise_weights, bias, scale, shift, output, st)
                      : sc_launch<1, 16>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, depthwise_filter,
                                         pointwise_weights, bias, scale, shift, output, st);
    return r == 2 ? sc_launch<2, 32>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, depthwise_filter,
                                     pointwise_weights, bias, scale, shift, output, st)
                  : sc_launch<1, 32>(B, N, M, F, C, K, Cout, act, nn_index, nn_count, bin_index, input, depthwise_filter,
                                     pointwise_weights, bias, scale, shift, output, st);
}
