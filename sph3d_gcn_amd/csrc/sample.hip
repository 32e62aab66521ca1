// sample.hip — farthest point sampling for gfx950.
//
// Replaces farthestpointsampleKernel (tf_ops/sampling/tf_sample_gpu.cu:7-73).
//
// FPS is m strictly sequential rounds, so the design minimises the latency of ONE round:
//   * one workgroup per cloud, thread t owns points k = t, t+1024, ... (the reference's own mapping, which
//     also defines its tie-break); their xyz AND running min-distance live in registers for the whole
//     kernel (the reference re-read a global `temp` array and 3072-point shared cache every round);
//   * per round each thread updates its P points and keeps its best (value, candidate xyz);
//   * the wave arg-max is a 6-instruction DPP max (row shifts + row broadcasts, no LDS) of the
//     order-preserving float bits, then ballot(value == max) + count-trailing-zeros picks the LOWEST lane:
//     "larger value, then lower thread id" — exactly the reference's tree (:56-66, left entry wins ties)
//     composed with its strict > per-thread scan (:49);
//   * one 16-byte LDS slot per wave {value bits, x, y, z}, double-buffered -> ONE barrier per round
//     (reference: 11); after it each wave loads the 16 slots with ONE ds_read_b128 (lane l reads slot l&15),
//     repeats the DPP max + ballot inside the row and readlanes the winner's coordinates.
//   * race-free by construction (the reference reads dists_i[0] unfenced at :68).
// Clouds larger than 1024 * 24 points fall back to a kernel that keeps the running distance in a
// caller-provided workspace (the reference's `temp`) and re-reads xyz from L2.
// -ffp-contract=off: d = (dx*dx + dy*dy) + dz*dz must round like the oracle.
#include "common.hpp"

namespace sph3d {

__device__ __forceinline__ unsigned order_bits(float v)
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// wave64 max of an unsigned value with DPP row shifts + row broadcasts (gfx9 scan idiom: 6 VALU ops, no LDS);
// the result is valid in lane 63 and returned as a wave-uniform scalar.  Identity 0.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = v > t ? v : t;   // row_shr:1
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = v > t ? v : t;   // row_shr:2
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = v > t ? v : t;   // row_shr:4
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = v > t ? v : t;   // row_shr:8
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;   // row_bcast:15
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;   // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, m, 64);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

struct __attribute__((aligned(16))) FpsSlot {
    unsigned long long key;
    float x, y, z;
    int pad[3];
};

// one 16-byte LDS slot per wave and round: ordered value bits + the candidate's coordinates
struct __attribute__((aligned(16))) FpsSlot16 {
    unsigned vbits;
    float x, y, z;
};

// (Round 2: capping this kernel and the fused neighbour search at 64 VGPRs, so that a CU hosting one of their 16-wave
// workgroups keeps room for two 128-VGPR waves per SIMD of the feature path, cut the slow-down of a GEMM running beside
// the FPS chain from 1.2x to 1.07x (tools/exp_gemm_fps.py) but the training step got 0.5 % slower: A/B in one call,
// tools/gpu_ab.sh.  Not kept.)
template <int P>
__global__ __launch_bounds__(1024) void fps_reg_kernel(int b, int n, int m,
                                                       const float* __restrict__ dataset, int* __restrict__ idxs)
{
    __shared__ FpsSlot16 slots[2][16];
    __shared__ int slot_k[2][16];           // the candidate's point index, read by thread 0 only
    const int t = (int)threadIdx.x;
    const int lane = t & 63;
    const int wave = uniform(t >> 6);
    const int nwaves = (int)(blockDim.x >> 6);

    for (int i = (int)blockIdx.x; i < b; i += (int)gridDim.x) {
        const float* pts = dataset + (size_t)i * n * 3;
        float px[P], py[P], pz[P], td[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int k = t + p * kRefBlock;
            const bool ok = k < n;
            px[p] = ok ? pts[k * 3] : 0.f;
            py[p] = ok ? pts[k * 3 + 1] : 0.f;
            pz[p] = ok ? pts[k * 3 + 2] : 0.f;
            td[p] = ok ? 1e38f : -1.f;      // tf_sample_gpu.cu:19-21; absent points can never win (best starts at -1)
        }
        float x1 = pts[0], y1 = pts[1], z1 = pts[2];         // old = 0 (:16)
        if (t == 0) idxs[(size_t)i * m] = 0;
        if (t < 32) {                                        // unused slots never win
            slots[t >> 4][t & 15].vbits = 0u;
        }
        __syncthreads();

        for (int j = 1; j < m; j++) {
            float best = -1.f;                               // :27-28
            int bestp = 0;
            float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
            for (int p = 0; p < P; p++) {
                const float dx = px[p] - x1, dy = py[p] - y1, dz = pz[p] - z1;
                const float d = (dx * dx + dy * dy) + dz * dz;           // :45
                const float d2 = d < td[p] ? d : td[p];                  // :46 min(d, td); absent points stay -1
                td[p] = d2;
                if (d2 > best) { best = d2; bestp = p; bx = px[p]; by = py[p]; bz = pz[p]; }   // :49 strict >
            }
            // wave arg-max: max value, then lowest lane (= lowest reference thread id in this wave)
            const unsigned vb = order_bits(best);
            const unsigned wmax = wave_max_u32(vb);
            const unsigned long long tie = __ballot(vb == wmax);
            const int wl = (int)__builtin_ctzll(tie);
            const int buf = j & 1;
            if (lane == wl) {
                FpsSlot16 sl;
                sl.vbits = vb; sl.x = bx; sl.y = by; sl.z = bz;
                slots[buf][wave] = sl;
                slot_k[buf][wave] = (best >= 0.f) ? (t + bestp * kRefBlock) : 0;   // idle thread: besti = 0
            }
            __syncthreads();
            // every wave: lanes 0..15 read the 16 slots, reduce inside the row, lowest wave wins ties
            const FpsSlot16 s = slots[buf][lane & 15];
            const unsigned gmax = wave_max_u32(s.vbits);
            const unsigned long long gt = __ballot(s.vbits == gmax);
            const int gw = (int)__builtin_ctzll(gt);          // < 16: lanes 0..15 hold slots 0..15
            x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.x), gw));
            y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.y), gw));
            z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.z), gw));
            if (t == 0) idxs[(size_t)i * m + j] = slot_k[buf][gw];
            (void)nwaves;
        }
    }
}

// Fallback for very large clouds: running distance in global workspace (the reference's temp[32][n]),
// xyz re-read from L2 each round.  Same arithmetic and tie-break.
__global__ __launch_bounds__(1024) void fps_big_kernel(int b, int n, int m, const float* __restrict__ dataset,
                                                       float* __restrict__ temp, int* __restrict__ idxs)
{
    __shared__ FpsSlot slots[2][16];
    const int t = (int)threadIdx.x;
    const int wave = uniform(t >> 6);
    const int nwaves = (int)(blockDim.x >> 6);
    float* td = temp + (size_t)blockIdx.x * n;
    for (int i = (int)blockIdx.x; i < b; i += (int)gridDim.x) {
        const float* pts = dataset + (size_t)i * n * 3;
        for (int k = t; k < n; k += kRefBlock) td[k] = 1e38f;
        float x1 = pts[0], y1 = pts[1], z1 = pts[2];
        if (t == 0) idxs[(size_t)i * m] = 0;
        __syncthreads();
        for (int j = 1; j < m; j++) {
            float best = -1.f;
            int bestk = 0;
            float bx = 0.f, by = 0.f, bz = 0.f;
            for (int k = t; k < n; k += kRefBlock) {
                const float x2 = pts[k * 3], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
                const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const float o = td[k];
                const float d2 = d < o ? d : o;
                if (d2 != o) td[k] = d2;
                if (d2 > best) { best = d2; bestk = k; bx = x2; by = y2; bz = z2; }
            }
            unsigned long long key = ((unsigned long long)order_bits(best) << 32) |
                                     ((unsigned)(kRefBlock - 1 - t) << 8) | (unsigned)((bestk >> 10) & 0xff);
            unsigned long long wk = key;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const unsigned long long o = shfl_xor_u64(wk, s);
                wk = o > wk ? o : wk;
            }
            const int buf = j & 1;
            if (key == wk) {
                slots[buf][wave].key = key;
                slots[buf][wave].x = bx;
                slots[buf][wave].y = by;
                slots[buf][wave].z = bz;
            }
            __syncthreads();
            unsigned long long gk = slots[buf][0].key;
            int gw = 0;
            for (int w = 1; w < nwaves; w++) {
                const unsigned long long o = slots[buf][w].key;
                if (o > gk) { gk = o; gw = w; }
            }
            x1 = slots[buf][gw].x;
            y1 = slots[buf][gw].y;
            z1 = slots[buf][gw].z;
            if (t == 0) {
                const int wt = kRefBlock - 1 - (int)((gk >> 8) & 0x3ffu);
                const int wp = (int)(gk & 0xffu);
                idxs[(size_t)i * m + j] = (wt < n) ? wt + wp * kRefBlock : 0;
            }
        }
    }
}

constexpr int kFpsMaxRegPoints = 24;   // points per thread held in registers
constexpr int kFpsBigGrid = 64;

}  // namespace sph3d

using namespace sph3d;

extern "C" size_t sph3d_farthest_point_sample_workspace(int b, int n, int m)
{
    (void)m;
    if (n <= kRefBlock * kFpsMaxRegPoints) return 0;
    const int g = b < kFpsBigGrid ? b : kFpsBigGrid;
    return sizeof(float) * (size_t)g * n;
}

extern "C" int sph3d_farthest_point_sample(int b, int n, int m, const float* inp, int* out,
                                           void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(m > 0, "FarthestPointSample expects positive npoint");                     // tf_sample.cpp:35
    SPH3D_REQUIRE(b >= 0 && n > 0, "FarthestPointSample expects (batch_size,num_points,3) inp shape");
    SPH3D_REQUIRE(n <= kRefBlock * 256, "FarthestPointSample: n=%d exceeds the supported 262144 points", n);
    if (b == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    const int P = (n + kRefBlock - 1) / kRefBlock;
    int bs = n < kRefBlock ? ((n + 63) / 64) * 64 : kRefBlock;
    const dim3 grid(b), block(bs);
#define SPH3D_FPS(PP) hipLaunchKernelGGL(fps_reg_kernel<PP>, grid, block, 0, st, b, n, m, inp, out)
    if (P <= 1) SPH3D_FPS(1);
    else if (P <= 2) SPH3D_FPS(2);
    else if (P <= 4) SPH3D_FPS(4);
    else if (P <= 8) SPH3D_FPS(8);
    else if (P <= 12) SPH3D_FPS(12);
    else if (P <= 16) SPH3D_FPS(16);
    else if (P <= kFpsMaxRegPoints) SPH3D_FPS(24);
    else {
        const size_t need = sph3d_farthest_point_sample_workspace(b, n, m);
        if (workspace == nullptr || workspace_bytes < need) {
            set_error("FarthestPointSample: workspace %zu B < required %zu B", workspace_bytes, need);
            return SPH3D_EWORKSPACE;
        }
        const int g = b < kFpsBigGrid ? b : kFpsBigGrid;
        hipLaunchKernelGGL(fps_big_kernel, dim3(g), dim3(kRefBlock), 0, st, b, n, m, inp, (float*)workspace, out);
    }
#undef SPH3D_FPS
    return check_launch("sph3d_farthest_point_sample");
}
