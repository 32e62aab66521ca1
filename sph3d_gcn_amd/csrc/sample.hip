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
#include <cstdlib>
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
// `gate` != nullptr: run only if *gate != 0 (the co-operative kernel's error word: its repair pass, see the launcher).
__global__ __launch_bounds__(1024) void fps_big_kernel(int b, int n, int m, const float* __restrict__ dataset,
                                                       float* __restrict__ temp, int* __restrict__ idxs,
                                                       const int* __restrict__ gate = nullptr)
{
    __shared__ FpsSlot slots[2][16];
    if (gate != nullptr && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
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

// ---------------------------------------------------------------------------------------------------------------
// Large clouds (n > 24 576 points: BASELINE config 5, one 65 536-point block per GPU), round 3.  fps_big_kernel above keeps
// one workgroup per cloud and re-reads the cloud's coordinates and running distances from L2 every round: 15 us per round,
// 245 ms for 65 536 -> 16 384.  Here G = ceil(n / 4096) workgroups SHARE a cloud: workgroup g owns points [4096 g, 4096 g +
// 4096) in registers (thread t: k = 4096 g + t + 1024 p, p < 4), finds its candidate like fps_reg_kernel, and the G
// candidates meet through 8-byte data-tagged granules in global memory:
//     [63:32] order-preserving bits of the candidate's min-distance   [31:22] 1023 - t   [21:14] 255 - (k >> 10)   [13:0] round
// one relaxed agent-scope store per workgroup and round, polled by lane g' < G of wave 0 with relaxed agent-scope loads until
// every tag equals the round (MI355X_MICROARCH.md: a naturally aligned 8-byte granule written by ONE store needs no further
// ordering; slots alternate with the round's parity, so a workgroup one round ahead never overwrites what a slower one still
// reads).  The maximum of the upper 50 bits is the reference's winner: larger distance, then lower thread id t = k mod 1024,
// then lower k (tf_sample_gpu.cu:49,56-66: strict > inside a thread, left entry wins in the tree).  The winner's coordinates
// are read from the (read-only) cloud.  Every spin is bounded: a workgroup that waits 2^22 polls sets the error word and the
// kernel ends.  The launch is an ordinary one: its B * G <= 128 workgroups of 1024 threads are co-resident on an idle GPU, but
// nothing guarantees that beside other streams' kernels or another process — so a time-out must not cost
// correctness: the launcher queues fps_big_kernel behind it, gated on the error word, which recomputes every cloud's samples
// the slow way when (and only when) the co-operative pass gave up.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCoopP = 4;                  // points per thread
constexpr int kCoopPts = kRefBlock * kCoopP;

__global__ __launch_bounds__(1024) void fps_coop_kernel(int n, int m, int G, const float* __restrict__ dataset,
                                                        unsigned long long* __restrict__ slots, int* __restrict__ err,
                                                        int* __restrict__ idxs)
{
    __shared__ FpsSlot16 lslots[2][16];
    __shared__ int lslot_tp[2][16];          // winner thread's t | p << 16
    __shared__ int win_k[2];
    const int t = (int)threadIdx.x;
    const int lane = t & 63;
    const int wave = uniform(t >> 6);
    const int i = (int)blockIdx.x / G, g = (int)blockIdx.x % G;
    const float* pts = dataset + (size_t)i * n * 3;
    unsigned long long* myslots = slots + (size_t)i * 2 * G;
    float px[kCoopP], py[kCoopP], pz[kCoopP], td[kCoopP];
#pragma unroll
    for (int p = 0; p < kCoopP; p++) {
        const int k = g * kCoopPts + t + p * kRefBlock;
        const bool ok = k < n;
        px[p] = ok ? pts[k * 3] : 0.f;
        py[p] = ok ? pts[k * 3 + 1] : 0.f;
        pz[p] = ok ? pts[k * 3 + 2] : 0.f;
        td[p] = ok ? 1e38f : -1.f;
    }
    float x1 = pts[0], y1 = pts[1], z1 = pts[2];
    if (g == 0 && t == 0) idxs[(size_t)i * m] = 0;
    if (t < 32) lslots[t >> 4][t & 15].vbits = 0u;
    __syncthreads();

    for (int j = 1; j < m; j++) {
        float best = -1.f;
        int bestp = 0;
#pragma unroll
        for (int p = 0; p < kCoopP; p++) {
            const float dx = px[p] - x1, dy = py[p] - y1, dz = pz[p] - z1;
            const float d = (dx * dx + dy * dy) + dz * dz;               // tf_sample_gpu.cu:45
            const float d2 = d < td[p] ? d : td[p];
            td[p] = d2;
            if (d2 > best) { best = d2; bestp = p; }                     // :49 strict >
        }
        const unsigned vb = order_bits(best);
        const unsigned wmax = wave_max_u32(vb);
        const int wl = (int)__builtin_ctzll(__ballot(vb == wmax));
        const int buf = j & 1;
        if (lane == wl) {
            lslots[buf][wave].vbits = vb;
            lslot_tp[buf][wave] = t | (bestp << 16);
        }
        __syncthreads();
        if (wave == 0) {
            // this workgroup's candidate (lowest wave wins ties: lower t) ...
            const unsigned sv = lslots[buf][lane & 15].vbits;
            const unsigned gmax = wave_max_u32(sv);
            const int gw = (int)__builtin_ctzll(__ballot(sv == gmax));           // < 16
            const int tp = lslot_tp[buf][gw];
            const int wt = tp & 0xffff, wp = tp >> 16;
            const int q = g * kCoopP + wp;                                       // k >> 10 of the candidate (absent points: vb of -1)
            const unsigned long long gran = ((unsigned long long)gmax << 32) | ((unsigned long long)(1023 - wt) << 22) |
                                            ((unsigned long long)(255 - q) << 14) | (unsigned long long)(j & 0x3fff);
            if (lane == 0) __hip_atomic_store(&myslots[buf * G + g], gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ... meets the other workgroups' candidates
            const int gl = lane < G ? lane : 0;
            unsigned long long v = 0ull;
            int spins = 0;
            for (;;) {
                v = __hip_atomic_load(&myslots[buf * G + gl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (int)(v & 0x3fffull) == (j & 0x3fff);
                if (__ballot(ok) == ~0ull) break;
                if (++spins > (1 << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = ~0ull;                                                   // poison: the round loop ends below
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            unsigned long long key = v >> 14;
            if (v == ~0ull) key = ~0ull;
#pragma unroll
            for (int s2 = 1; s2 < 64; s2 <<= 1) {
                const unsigned long long o = shfl_xor_u64(key, s2);
                key = o > key ? o : key;
            }
            if (lane == 0) {
                int k = -1;
                if (key != ~0ull) {
                    const int kt = 1023 - (int)((key >> 8) & 0x3ffull);
                    const int kq = 255 - (int)(key & 0xffull);
                    const float bv = __uint_as_float(0);
                    (void)bv;
                    k = kq * kRefBlock + kt;
                    // best < 0 everywhere (no point left: cannot happen for m <= n) -> index 0 like the reference's idle threads
                    if ((unsigned)(key >> 18) == order_bits(-1.f)) k = 0;
                    if (k >= n) k = 0;
                }
                win_k[buf] = k;
                if (g == 0 && k >= 0) idxs[(size_t)i * m + j] = k;
            }
        }
        __syncthreads();
        const int k = win_k[buf];
        if (k < 0) break;                                                        // time-out: give up (err is set)
        x1 = pts[(size_t)k * 3];
        y1 = pts[(size_t)k * 3 + 1];
        z1 = pts[(size_t)k * 3 + 2];
    }
}

constexpr int kFpsMaxRegPoints = 24;   // points per thread held in registers
constexpr int kFpsBigGrid = 64;

}  // namespace sph3d

using namespace sph3d;

// co-operative kernel: G workgroups per cloud, all B * G resident at once (<= 128 of the 256 CUs)
static int coop_groups(int b, int n)
{
    const int G = (n + kCoopPts - 1) / kCoopPts;
    return (G <= 64 && (long long)b * G <= 128) ? G : 0;
}

static size_t coop_slot_bytes(int b, int G) { return (256 + sizeof(unsigned long long) * (size_t)b * 2 * G + 255) & ~(size_t)255; }

// test hook: SPH3D_FPS_FORCE_TIMEOUT=1 starts the co-operative kernel with its error word already set, so the repair pass runs
static int fps_force_timeout()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SPH3D_FPS_FORCE_TIMEOUT");
        v = (e && atoi(e) != 0) ? 1 : 0;
    }
    return v;
}

extern "C" size_t sph3d_farthest_point_sample_workspace(int b, int n, int m)
{
    (void)m;
    if (n <= kRefBlock * kFpsMaxRegPoints) return 0;
    const int G = coop_groups(b, n);
    const int g = b < kFpsBigGrid ? b : kFpsBigGrid;
    const size_t big = sizeof(float) * (size_t)g * n;                            // fps_big_kernel's running distances
    if (G) return coop_slot_bytes(b, G) + big;                                   // error word + granule slots, then the repair pass's
    return big;
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
        const int G = coop_groups(b, n);
        if (G) {
            const size_t head = coop_slot_bytes(b, G);
            int rc = check_hip(hipMemsetAsync(workspace, 0, head, st), "FarthestPointSample: memset");
            if (rc) return rc;
            int* err = (int*)workspace;
            if (fps_force_timeout()) {
                rc = check_hip(hipMemsetAsync(err, 1, 1, st), "FarthestPointSample: memset");
                if (rc) return rc;
            }
            unsigned long long* slots = (unsigned long long*)((char*)workspace + 256);
            hipLaunchKernelGGL(fps_coop_kernel, dim3(b * G), dim3(kRefBlock), 0, st, n, m, G, inp, slots, err, out);
            rc = check_launch("sph3d_farthest_point_sample (co-operative pass)");
            if (rc) return rc;
            // repair pass: returns at once unless the co-operative pass timed out (then every cloud is resampled, bit-exactly)
            const int g = b < kFpsBigGrid ? b : kFpsBigGrid;
            hipLaunchKernelGGL(fps_big_kernel, dim3(g), dim3(kRefBlock), 0, st, b, n, m, inp, (float*)((char*)workspace + head), out,
                               (const int*)err);
        } else {
            const int g = b < kFpsBigGrid ? b : kFpsBigGrid;
            hipLaunchKernelGGL(fps_big_kernel, dim3(g), dim3(kRefBlock), 0, st, b, n, m, inp, (float*)workspace, out, (const int*)nullptr);
        }
    }
#undef SPH3D_FPS
    return check_launch("sph3d_farthest_point_sample");
}
