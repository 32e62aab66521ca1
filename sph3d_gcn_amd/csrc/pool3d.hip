// pool3d.hip — graph max/avg pooling and mean/weighted un-pooling (forward + gradients), gfx950.
//
// Replaces max_pool3d_forward/backward, avg_pool3d_forward/backward (tf_ops/pooling/tf_pool3d_gpu.cu:5-90)
// and mean/weighted_interpolate_forward/backward (tf_ops/unpooling/tf_unpool3d_gpu.cu:5-84).
//
// All eight kernels are the same shape of work — "for every output point, walk its neighbour list and
// combine gathered feature rows" — so they share one skeleton: one wavefront per output point, the
// neighbour row read through the scalar cache (its address is wave-uniform), lanes spanning channels
// with float4 (or scalar when C % 4 != 0) coalesced row reads, accumulation in registers, one store.
// The reference used one thread per (point, channel) with a global read-modify-write per neighbour
// and a cudaDeviceSynchronize after every launch (tf_pool3d_gpu.cu:97,104,111,118).
// Un-pooling "mean" is avg-pooling with the roles of the point sets swapped, so it reuses that kernel.
// avg / mean / weighted gradients gather over the transposed graph (graph.hip); only the max-pool gradient
// (one element per output, data-dependent target) still uses hardware fp32 atomics.
#include "common.hpp"

namespace sph3d {

constexpr int kPtsPerWG = 16;   // 4 waves x 4 points: the launches with many points; small ones take 4 (pts_per_wg)

enum class Mode { Max, Avg, Weighted };

// V = 4: channels c0 + 4*lane + {0..3} per pass of 256 channels; V = 1: c0 + lane (64 per pass)
template <Mode MODE, int V>
__global__ __launch_bounds__(256) void gather_fwd(
    int B, int Nin, int Mout, int C, int K, int mblocks,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount,
    const float* __restrict__ input, const float* __restrict__ weight,
    float* __restrict__ output, int* __restrict__ maxIndex, int ppwg)
{
    int b, mb;
    xcd_decode((int)blockIdx.x, B, mblocks, b, mb);
    if (b < 0) return;
    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int m_begin = mb * ppwg;
    const int m_end = (m_begin + ppwg) < Mout ? (m_begin + ppwg) : Mout;
    const float* inb = input + (size_t)b * Nin * C;

    for (int m = m_begin + wave; m < m_end; m += 4) {
        const size_t row = (size_t)b * Mout + m;
        const int cnt = uniform(nnCount[row]);
        for (int c0 = 0; c0 < C; c0 += 64 * V) {
            const int c = c0 + lane * V;
            const bool act = c < C;
            const int cc = act ? c : 0;
            float acc[V];
            int arg[V];
#pragma unroll
            for (int v = 0; v < V; v++) { acc[v] = 0.f; arg[v] = 0; }
            // wave-uniform address -> scalar loads, eight consecutive ids per s_load_dwordx8.  (Round 2: the vector-load +
            // v_readlane scheme that helped the convolution gradient made the step 3.7 % SLOWER here: these loops have no
            // clamps or interleaved scale loads for it to remove, and a readlane per edge costs more than an eighth of a wide s_load.)
            const int* __restrict__ irow = nnIndex + row * K;
            const float* __restrict__ wrow = weight + row * K;
            {
#pragma unroll 8
                for (int kk = 0; kk < cnt; kk++) {
                    const int n = irow[kk];
                    float w = 1.f;
                    if (MODE == Mode::Weighted) w = wrow[kk];
                    {
                        float x[V];          // branch-free: inactive lanes read channel 0 (keeps the unrolled gathers in flight)
                        if (V == 4) {
                            const float4 t = *reinterpret_cast<const float4*>(&inb[(size_t)n * C + cc]);
                            x[0] = t.x; x[1 % V] = t.y; x[2 % V] = t.z; x[3 % V] = t.w;
                        } else {
                            x[0] = inb[(size_t)n * C + cc];
                        }
#pragma unroll
                        for (int v = 0; v < V; v++) {
                            if (MODE == Mode::Max) {
                                // first neighbour seeds, strict > replaces (tf_pool3d_gpu.cu:17-29)
                                if (kk == 0 || x[v] > acc[v]) { acc[v] = x[v]; arg[v] = n; }
                            } else if (MODE == Mode::Avg) {
                                acc[v] += x[v];
                            } else {
                                acc[v] = fmaf(x[v], w, acc[v]);     // tf_unpool3d_gpu.cu:59
                            }
                        }
                    }
                }
            }
            if (act) {
                const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
                for (int v = 0; v < V; v++) {
                    float o = acc[v];
                    if (MODE == Mode::Avg) o = o * inv;
                    output[row * C + c + v] = o;
                    if (MODE == Mode::Max) maxIndex[row * C + c + v] = arg[v];
                }
            }
        }
    }
}

// C <= 128 (C % 4 == 0): a row is at most 32 lanes of 16 bytes, so the two halves of the wave take ALTERNATE neighbours (one
// wave load = two rows): these launches are bound by the L1's 16 cycles per wave load (3.1 M edges at the decoder's last
// level = 0.08 ms of L1 issue with one row per load, measured 0.128 ms), and with one row per load half the lanes repeat
// lane 0's address.  Max: each half keeps (value, id, slot) of its own neighbours — half 0 is seeded by neighbour 0 as in
// the reference (tf_pool3d_gpu.cu:17-29), half 1 starts from -inf — and the halves are merged with "larger value, then
// earlier slot", which is what the reference's sequential strict-> scan computes.
template <Mode MODE>
__global__ __launch_bounds__(256) void gather_fwd_half(
    int B, int Nin, int Mout, int C, int K, int mblocks,
    const int* __restrict__ nnIndex, const int* __restrict__ nnCount,
    const float* __restrict__ input, const float* __restrict__ weight,
    float* __restrict__ output, int* __restrict__ maxIndex, int ppwg)
{
    int b, mb;
    xcd_decode((int)blockIdx.x, B, mblocks, b, mb);
    if (b < 0) return;
    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int half = lane >> 5;
    const int c = (lane & 31) * 4;
    const bool act = c < C;
    const int m_begin = mb * ppwg;
    const int m_end = (m_begin + ppwg) < Mout ? (m_begin + ppwg) : Mout;
    const float* inb = input + (size_t)b * Nin * C + (act ? c : 0);

    for (int m = m_begin + wave; m < m_end; m += 4) {
        const size_t row = (size_t)b * Mout + m;
        const int cnt = uniform(nnCount[row]);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int arg[4] = {0, 0, 0, 0}, pos[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
        if (MODE == Mode::Max && half == 1) {
#pragma unroll
            for (int v = 0; v < 4; v++) acc[v] = -__builtin_inff();
        }
        const int* __restrict__ irow = nnIndex + row * K;
        const float* __restrict__ wrow = weight + row * K;
#pragma unroll 4
        for (int kk = 0; kk < cnt; kk += 2) {
            const bool two = (kk + 1) < cnt;                         // wave-uniform
            const int n0 = irow[kk], n1 = irow[two ? kk + 1 : kk];
            float w0 = 1.f, w1 = two ? 1.f : 0.f;                    // the odd tail: the same row again with weight 0 ...
            if (MODE == Mode::Weighted) { w0 = wrow[kk]; w1 = two ? wrow[kk + 1] : 0.f; }
            const int n = half ? n1 : n0;
            const float w = half ? w1 : w0;
            const int k = kk + half;
            const bool real = half == 0 || two;                      // ... or, for max, skipped
            const float4 t = *reinterpret_cast<const float4*>(&inb[(size_t)n * C]);
            const float x[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int v = 0; v < 4; v++) {
                if (MODE == Mode::Max) {
                    if (real && (k == 0 || x[v] > acc[v])) { acc[v] = x[v]; arg[v] = n; pos[v] = k; }
                } else {
                    acc[v] = fmaf(x[v], w, acc[v]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const float ov = __shfl_xor(acc[v], 32);
            if (MODE == Mode::Max) {
                const int oa = __shfl_xor(arg[v], 32), op = __shfl_xor(pos[v], 32);
                // larger value wins, equal values: the earlier slot.  Half 0's seed (slot 0) is kept even when it is NaN,
                // like the reference's unconditional first assignment: nothing compares greater than NaN
                if (ov > acc[v] || (ov == acc[v] && op < pos[v])) { acc[v] = ov; arg[v] = oa; pos[v] = op; }
            } else {
                acc[v] += ov;
            }
        }
        if (act && half == 0) {
            float4 o;
            if (MODE == Mode::Avg) {
                const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;      // one reciprocal per point (see conv3d.hip)
                o = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
            } else {
                o = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
            *reinterpret_cast<float4*>(&output[row * C + c]) = o;
            if (MODE == Mode::Max) *reinterpret_cast<int4*>(&maxIndex[row * C + c]) = make_int4(arg[0], arg[1], arg[2], arg[3]);
        }
    }
}

// gradient of avg-pool / mean- and weighted-interpolate as a GATHER over the transposed graph (graph.hip):
//   gradInput[b, n, c] = sum over in-edges (m, scale) of gradOutput[b, m, c] * scale
// scale = 1/nn_count[m] (avg / mean) or weight[b,m,k] (weighted).  One wave per source point n, each
// gradInput element written exactly once: no atomics, no memset (the reference: atomicAdd per
// (point, neighbour, channel), tf_pool3d_gpu.cu:86, tf_unpool3d_gpu.cu:38,80).
template <int V>
__global__ __launch_bounds__(256) void gather_bwd_t(
    int B, int Nin, int Mout, int C, int nblocks,
    const int* __restrict__ offsets, const int* __restrict__ entKey, const float* __restrict__ entScale,
    const float* __restrict__ gradOutput, float* __restrict__ gradInput)
{
    int b, nb;
    xcd_decode((int)blockIdx.x, B, nblocks, b, nb);
    if (b < 0) return;
    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int n_begin = nb * kPtsPerWG;
    const int n_end = (n_begin + kPtsPerWG) < Nin ? (n_begin + kPtsPerWG) : Nin;
    const float* gob = gradOutput + (size_t)b * Mout * C;
    const int* __restrict__ offb = offsets + (size_t)b * (Nin + 1);
    for (int n = n_begin + wave; n < n_end; n += 4) {
        const int e0 = offb[n], e1 = offb[n + 1];
        for (int c0 = 0; c0 < C; c0 += 64 * V) {
            const int c = c0 + lane * V;
            const bool act = c < C;
            const int cc = act ? c : 0;
            float acc[V];
#pragma unroll
            for (int v = 0; v < V; v++) acc[v] = 0.f;
#pragma unroll 8
            for (int e = e0; e < e1; e++) {
                const int wk = entKey[e];
                const int m = tg_key(wk, entScale == nullptr);          // (packed entries: common.hpp)
                const float sc = entScale == nullptr ? tg_packed_scale(wk) : entScale[e];
                {   // branch-free: inactive lanes read channel 0
                    if (V == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(&gob[(size_t)m * C + cc]);
                        acc[0] = fmaf(t.x, sc, acc[0]);
                        acc[1 % V] = fmaf(t.y, sc, acc[1 % V]);
                        acc[2 % V] = fmaf(t.z, sc, acc[2 % V]);
                        acc[3 % V] = fmaf(t.w, sc, acc[3 % V]);
                    } else {
                        acc[0] = fmaf(gob[(size_t)m * C + cc], sc, acc[0]);
                    }
                }
            }
            if (act) {
#pragma unroll
                for (int v = 0; v < V; v++) gradInput[((size_t)b * Nin + n) * C + c + v] = acc[v];
            }
        }
    }
}

// max-pool gradient: gradInput[b, maxIndex[b,m,c], c] += gradOutput[b,m,c]   (tf_pool3d_gpu.cu:38-50)
__global__ __launch_bounds__(256) void maxpool_bwd(
    int B, int N, int M, int C, const int* __restrict__ maxIndex,
    const float* __restrict__ gradOutput, float* __restrict__ gradInput)
{
    // grid.y = cloud: no 64-bit division per element; the channel is carried from iteration to iteration
    const int b = (int)blockIdx.y;
    const long long per_b = (long long)M * C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int c = (int)(i0 % C);
    const int cstep = (int)(stride % C);
    const int* __restrict__ mi = maxIndex + (size_t)b * per_b;
    const float* __restrict__ go = gradOutput + (size_t)b * per_b;
    float* __restrict__ gi = gradInput + (size_t)b * N * C;
    for (long long e = i0; e < per_b; e += stride) {
        unsafeAtomicAdd(&gi[(size_t)mi[e] * C + c], go[e]);
        c += cstep;
        c = c >= C ? c - C : c;
    }
}

// max-pool gradient as a GATHER over the transposed pooling graph (round 3): source point n walks its in-edges m and takes
// gradOutput[b,m,c] where maxIndex[b,m,c] == n.  Every gradInput element is written exactly once: no memset, no float
// atomics (the scatter above: 4.2 M single-lane atomics at S3DIS level 0, 151 us; 47-83 us on the small levels), and the sum
// of the (few) terms of an element has a fixed order.  Rows m WITHOUT neighbours keep maxIndex 0 in the forward pass and the
// scatter adds their gradient to point 0; they have no edge here, so the wave of point 0 looks for them in nn_count.
template <int V>
__global__ __launch_bounds__(256) void maxpool_bwd_t(
    int B, int Nin, int Mout, int C, int nblocks, int ppwg,
    const int* __restrict__ offsets, const int* __restrict__ entKey, const int* __restrict__ nnCount,
    const int* __restrict__ maxIndex, const float* __restrict__ gradOutput, const float* __restrict__ addend,
    float* __restrict__ gradInput)
{
    int b, nb;
    xcd_decode((int)blockIdx.x, B, nblocks, b, nb);
    if (b < 0) return;
    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const int n_begin = nb * ppwg;
    const int n_end = (n_begin + ppwg) < Nin ? (n_begin + ppwg) : Nin;
    const float* gob = gradOutput + (size_t)b * Mout * C;
    const int* mib = maxIndex + (size_t)b * Mout * C;
    const int* __restrict__ offb = offsets + (size_t)b * (Nin + 1);
    for (int n = n_begin + wave; n < n_end; n += 4) {
        const int e0 = offb[n], e1 = offb[n + 1];
        for (int c0 = 0; c0 < C; c0 += 64 * V) {
            const int c = c0 + lane * V;
            const bool act = c < C;
            const int cc = act ? c : 0;
            // addend (optional, [B, Nin, C]): a second gradient of the same tensor — the pooled tensor's other consumer, the
            // encoder's skip connection — summed here instead of by a separate elementwise kernel over three tensors
            float acc[V];
#pragma unroll
            for (int v = 0; v < V; v++) acc[v] = addend != nullptr ? addend[((size_t)b * Nin + n) * C + cc + v] : 0.f;
            auto take = [&](int m) {       // branch-free: inactive lanes read channel 0
                if (V == 4) {
                    const int4 a = *reinterpret_cast<const int4*>(&mib[(size_t)m * C + cc]);
                    const float4 g = *reinterpret_cast<const float4*>(&gob[(size_t)m * C + cc]);
                    acc[0] += a.x == n ? g.x : 0.f;
                    acc[1 % V] += a.y == n ? g.y : 0.f;
                    acc[2 % V] += a.z == n ? g.z : 0.f;
                    acc[3 % V] += a.w == n ? g.w : 0.f;
                } else {
                    acc[0] += mib[(size_t)m * C + cc] == n ? gob[(size_t)m * C + cc] : 0.f;
                }
            };
#pragma unroll 4
            // (an entry word may be packed with the row's count above bit 24: common.hpp; rows are < 2^24 whenever it is)
            for (int e = e0; e < e1; e++) take(Mout <= (1 << 24) ? (int)((unsigned)entKey[e] & kTgKeyMask) : entKey[e]);
            if (n == 0) {                  // rows without neighbours: maxIndex 0 (see above); 64 rows per trip
                for (int m0 = 0; m0 < Mout; m0 += 64) {
                    const int mm = m0 + lane;
                    const int cnt = mm < Mout ? nnCount[(size_t)b * Mout + mm] : 1;
                    unsigned long long empty = __ballot(cnt == 0);
                    while (empty != 0ull) {
                        const int bit = (int)__builtin_ctzll(empty);
                        empty &= empty - 1ull;
                        take(m0 + bit);
                    }
                }
            }
            if (act) {
#pragma unroll
                for (int v = 0; v < V; v++) gradInput[((size_t)b * Nin + n) * C + c + v] = acc[v];
            }
        }
    }
}

// Few sources with many in-edges each (un-pooling: 2048 coarse points collect ~100 fine points each): one WORKGROUP per
// source, its four waves take every fourth in-edge and the partial sums meet in LDS.  One wave per source (gather_bwd_t)
// leaves such a launch with eight waves per SIMD in total and a 100-edge dependent chain per wave: 0.13 ms at the decoder's
// last level whatever the channel count.  (Two edges per wave load for C <= 128 was measured first: no gain — the launch is
// latency-, not load-bound.)
template <int V>
__global__ __launch_bounds__(256) void gather_bwd_t_split(
    int B, int Nin, int Mout, int C,
    const int* __restrict__ offsets, const int* __restrict__ entKey, const float* __restrict__ entScale,
    const float* __restrict__ gradOutput, float* __restrict__ gradInput)
{
    __shared__ float part[4][64 * V];
    const long long src = (long long)blockIdx.x;
    const int b = (int)(src / Nin);
    const int n = (int)(src - (long long)b * Nin);
    const int wave = uniform((int)threadIdx.x >> 6);
    const int lane = lane_id();
    const float* gob = gradOutput + (size_t)b * Mout * C;
    const int* __restrict__ offb = offsets + (size_t)b * (Nin + 1);
    const int e0 = offb[n], e1 = offb[n + 1];
    for (int c0 = 0; c0 < C; c0 += 64 * V) {
        const int c = c0 + lane * V;
        const bool act = c < C;
        const int cc = act ? c : 0;
        float acc[V];
#pragma unroll
        for (int v = 0; v < V; v++) acc[v] = 0.f;
        if (V == 4 && C <= 128) {
            // a row is at most 32 lanes wide: the two halves of the wave take alternate edges (one wave load = two rows; the
            // launch is bound by the L1's 16 cycles per wave load once there are enough waves)
            const int half = lane >> 5;
            const int ch = (lane & 31) * 4;
            const int cch = ch < C ? ch : 0;
#pragma unroll 4
            for (int e = e0 + 2 * wave; e < e1; e += 8) {
                const bool two = (e + 1) < e1;                           // wave-uniform
                const int w0 = entKey[e], w1 = entKey[two ? e + 1 : e];
                const bool pk = entScale == nullptr;
                const int m0 = tg_key(w0, pk), m1 = tg_key(w1, pk);
                const float s0 = pk ? tg_packed_scale(w0) : entScale[e], s1 = two ? (pk ? tg_packed_scale(w1) : entScale[e + 1]) : 0.f;
                const int m = half ? m1 : m0;
                const float sc = half ? s1 : s0;
                const float4 t = *reinterpret_cast<const float4*>(&gob[(size_t)m * C + cch]);
                acc[0] = fmaf(t.x, sc, acc[0]);
                acc[1 % V] = fmaf(t.y, sc, acc[1 % V]);
                acc[2 % V] = fmaf(t.z, sc, acc[2 % V]);
                acc[3 % V] = fmaf(t.w, sc, acc[3 % V]);
            }
#pragma unroll
            for (int v = 0; v < V; v++) acc[v] += __shfl_xor(acc[v], 32);       // lanes 0..31 now hold channels 4*lane..
        } else
#pragma unroll 4
        for (int e = e0 + wave; e < e1; e += 4) {
            const int wk = entKey[e];
            const int m = tg_key(wk, entScale == nullptr);
            const float sc = entScale == nullptr ? tg_packed_scale(wk) : entScale[e];
            if (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(&gob[(size_t)m * C + cc]);
                acc[0] = fmaf(t.x, sc, acc[0]);
                acc[1 % V] = fmaf(t.y, sc, acc[1 % V]);
                acc[2 % V] = fmaf(t.z, sc, acc[2 % V]);
                acc[3 % V] = fmaf(t.w, sc, acc[3 % V]);
            } else {
                acc[0] = fmaf(gob[(size_t)m * C + cc], sc, acc[0]);
            }
        }
        if (c0 > 0) __syncthreads();              // the previous pass's sums have been read
#pragma unroll
        for (int v = 0; v < V; v++) part[wave][lane * V + v] = acc[v];
        __syncthreads();
        if (wave == 0 && act) {
#pragma unroll
            for (int v = 0; v < V; v++)           // fixed order: deterministic
                gradInput[((size_t)b * Nin + n) * C + c + v] =
                    ((part[0][lane * V + v] + part[1][lane * V + v]) + part[2][lane * V + v]) + part[3][lane * V + v];
        }
    }
}

template <Mode MODE>
static int launch_fwd(const char* who, int B, int Nin, int Mout, int C, int K,
                      const int* nn_index, const int* nn_count, const float* input, const float* weight,
                      float* output, int* max_index, hipStream_t st)
{
    SPH3D_REQUIRE(B >= 0 && Nin > 0 && Mout >= 0 && C > 0 && K > 0, "%s: bad dims B=%d N=%d M=%d C=%d K=%d",
                  who, B, Nin, Mout, C, K);
    if (B == 0 || Mout == 0) return SPH3D_OK;
    // Output points per workgroup: these kernels have no per-workgroup prologue, so the small levels (a few thousand output
    // points) take ONE point per wave — four per wave left them with 2-8 waves per SIMD walking 4 x ~50 dependent gathers each
    // (round 3, tools/exp_calls.py: max pooling 36 -> 17 us at 16 x 128 points of 512 channels, 80 -> 64 us at level 0);
    // launches of >= 65536 points keep four per wave (mean interpolation at level 0: 80 vs 82 us).
    const int ppwg = (long long)B * Mout >= 65536 ? kPtsPerWG : 4;
    const int mblocks = (Mout + ppwg - 1) / ppwg;
    const dim3 grid(xcd_grid(B, mblocks));
    if (C % 4 == 0 && C <= 128)
        hipLaunchKernelGGL((gather_fwd_half<MODE>), grid, dim3(256), 0, st, B, Nin, Mout, C, K, mblocks,
                           nn_index, nn_count, input, weight, output, max_index, ppwg);
    else if (C % 4 == 0)
        hipLaunchKernelGGL((gather_fwd<MODE, 4>), grid, dim3(256), 0, st, B, Nin, Mout, C, K, mblocks,
                           nn_index, nn_count, input, weight, output, max_index, ppwg);
    else
        hipLaunchKernelGGL((gather_fwd<MODE, 1>), grid, dim3(256), 0, st, B, Nin, Mout, C, K, mblocks,
                           nn_index, nn_count, input, weight, output, max_index, ppwg);
    return check_launch(who);
}

static int launch_bwd_t(const char* who, int B, int Nin, int Mout, int C,
                        const int* offsets, const int* ent_key, const float* ent_scale,
                        const float* grad_output, float* grad_input, hipStream_t st)
{
    SPH3D_REQUIRE(B >= 0 && Nin > 0 && Mout >= 0 && C > 0, "%s: bad dims B=%d N=%d M=%d C=%d", who, B, Nin, Mout, C);
    if (B == 0) return SPH3D_OK;
    const int nblocks = (Nin + kPtsPerWG - 1) / kPtsPerWG;
    const dim3 grid(xcd_grid(B, nblocks));
    // few sources, each with (Mout / Nin) x more in-edges than a target has neighbours: one workgroup per source
    if ((long long)B * Nin <= 65536 && Mout >= 2 * Nin) {
        const dim3 sgrid((unsigned)((long long)B * Nin));
        if (C % 4 == 0)
            hipLaunchKernelGGL(gather_bwd_t_split<4>, sgrid, dim3(256), 0, st, B, Nin, Mout, C, offsets, ent_key, ent_scale,
                               grad_output, grad_input);
        else
            hipLaunchKernelGGL(gather_bwd_t_split<1>, sgrid, dim3(256), 0, st, B, Nin, Mout, C, offsets, ent_key, ent_scale,
                               grad_output, grad_input);
        return check_launch(who);
    }
    if (C % 4 == 0)
        hipLaunchKernelGGL(gather_bwd_t<4>, grid, dim3(256), 0, st, B, Nin, Mout, C, nblocks, offsets, ent_key,
                           ent_scale, grad_output, grad_input);
    else
        hipLaunchKernelGGL(gather_bwd_t<1>, grid, dim3(256), 0, st, B, Nin, Mout, C, nblocks, offsets, ent_key,
                           ent_scale, grad_output, grad_input);
    return check_launch(who);
}

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_max_pool3d(int B, int N, int M, int C, int K, const int* nn_index, const int* nn_count,
                                const float* input, float* output, int* max_index, sph3d_stream_t stream)
{
    return launch_fwd<Mode::Max>("sph3d_max_pool3d", B, N, M, C, K, nn_index, nn_count, input, nullptr, output,
                                 max_index, as_stream(stream));
}

extern "C" int sph3d_max_pool3d_grad(int B, int N, int M, int C, const int* max_index, const float* grad_output,
                                     float* grad_input, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0 && C > 0, "MaxPool3dGrad: bad dims B=%d N=%d M=%d C=%d", B, N, M, C);
    if (B == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    int rc = zero_async(grad_input, sizeof(float) * (size_t)B * N * C, st, "sph3d_max_pool3d_grad");
    if (rc) return rc;
    const long long per_b = (long long)M * C;
    if (per_b == 0) return SPH3D_OK;
    SPH3D_REQUIRE(B <= 65535, "MaxPool3dGrad: batch %d exceeds the grid's second dimension", B);
    long long blocks = (per_b + 255) / 256;
    const long long cap = (8192 + B - 1) / B > 1 ? (8192 + B - 1) / B : 1;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(maxpool_bwd, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, st, B, N, M, C, max_index, grad_output, grad_input);
    return check_launch("sph3d_max_pool3d_grad");
}

extern "C" int sph3d_max_pool3d_grad_t(int B, int N, int M, int C, const int* offsets, const int* ent_key, const int* nn_count,
                                       const int* max_index, const float* grad_output, const float* addend, float* grad_input,
                                       sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && N > 0 && M >= 0 && C > 0, "MaxPool3dGrad: bad dims B=%d N=%d M=%d C=%d", B, N, M, C);
    if (B == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    const int ppwg = (long long)B * N >= 65536 ? kPtsPerWG : 4;
    const int nblocks = (N + ppwg - 1) / ppwg;
    const dim3 grid(xcd_grid(B, nblocks));
    if (C % 4 == 0)
        hipLaunchKernelGGL(maxpool_bwd_t<4>, grid, dim3(256), 0, st, B, N, M, C, nblocks, ppwg, offsets, ent_key, nn_count, max_index,
                           grad_output, addend, grad_input);
    else
        hipLaunchKernelGGL(maxpool_bwd_t<1>, grid, dim3(256), 0, st, B, N, M, C, nblocks, ppwg, offsets, ent_key, nn_count, max_index,
                           grad_output, addend, grad_input);
    return check_launch("sph3d_max_pool3d_grad_t");
}

extern "C" int sph3d_avg_pool3d(int B, int N, int M, int C, int K, const int* nn_index, const int* nn_count,
                                const float* input, float* output, sph3d_stream_t stream)
{
    return launch_fwd<Mode::Avg>("sph3d_avg_pool3d", B, N, M, C, K, nn_index, nn_count, input, nullptr, output,
                                 nullptr, as_stream(stream));
}

// ---- gradients: transposed-graph entry point + the reference-surface wrappers that build the transpose ----


// grad_input[B,Nin,C] = gather over in-edges of grad_output[B,Mout,C] * scale
extern "C" int sph3d_scatter_grad_t(int B, int Nin, int Mout, int C, const int* offsets, const int* ent_key,
                                    const float* ent_scale, const float* grad_output, float* grad_input,
                                    sph3d_stream_t stream)
{
    return launch_bwd_t("sph3d_scatter_grad_t", B, Nin, Mout, C, offsets, ent_key, ent_scale, grad_output, grad_input,
                        as_stream(stream));
}

static int grad_via_transpose(const char* who, int B, int Nin, int Mout, int C, int K, const int* nn_index,
                              const int* nn_count, const float* weight, const float* grad_output, float* grad_input,
                              void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(B >= 0 && Nin > 0 && Mout >= 0 && C > 0 && K > 0, "%s: bad dims B=%d N=%d M=%d C=%d K=%d", who, B,
                  Nin, Mout, C, K);
    if (B == 0) return SPH3D_OK;
    const size_t need = sph3d_scatter_grad_workspace(B, Nin, Mout, K);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("%s: workspace %zu B < required %zu B", who, workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char* p = (char*)workspace;
    int* offsets = (int*)p; p += al(sizeof(int) * (size_t)B * (Nin + 1));
    int* key = (int*)p; p += al(sizeof(int) * (size_t)B * Mout * K);
    float* scale = (float*)p; p += al(sizeof(int) * (size_t)B * Mout * K);
    int rc = sph3d_graph_transpose(B, Nin, Mout, K, 1, nn_index, nn_count, nullptr, weight, offsets, key, scale, nullptr, p,
                                   al(sph3d_graph_transpose_workspace(B, Nin, Mout, K, 1)), stream);
    if (rc) return rc;
    return launch_bwd_t(who, B, Nin, Mout, C, offsets, key, scale, grad_output, grad_input, as_stream(stream));
}

extern "C" int sph3d_avg_pool3d_grad(int B, int N, int M, int C, int K, const int* nn_index, const int* nn_count,
                                     const float* grad_output, float* grad_input,
                                     void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    return grad_via_transpose("sph3d_avg_pool3d_grad", B, N, M, C, K, nn_index, nn_count, nullptr, grad_output,
                              grad_input, workspace, workspace_bytes, stream);
}

// un-pooling: the reference's N is the fine (output) count and M the coarse (input) count
extern "C" int sph3d_mean_interpolate(int B, int N, int M, int C, int K, const int* nn_index, const int* nn_count,
                                      const float* input, float* output, sph3d_stream_t stream)
{
    return launch_fwd<Mode::Avg>("sph3d_mean_interpolate", B, /*Nin=*/M, /*Mout=*/N, C, K, nn_index, nn_count, input,
                                 nullptr, output, nullptr, as_stream(stream));
}

extern "C" int sph3d_mean_interpolate_grad(int B, int N, int M, int C, int K, const int* nn_index,
                                           const int* nn_count, const float* grad_output, float* grad_input,
                                           void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    return grad_via_transpose("sph3d_mean_interpolate_grad", B, /*Nin=*/M, /*Mout=*/N, C, K, nn_index, nn_count, nullptr,
                              grad_output, grad_input, workspace, workspace_bytes, stream);
}

extern "C" int sph3d_weighted_interpolate(int B, int N, int M, int C, int K, const int* nn_index,
                                          const int* nn_count, const float* input, const float* weight,
                                          float* output, sph3d_stream_t stream)
{
    return launch_fwd<Mode::Weighted>("sph3d_weighted_interpolate", B, M, N, C, K, nn_index, nn_count, input, weight,
                                      output, nullptr, as_stream(stream));
}

extern "C" int sph3d_weighted_interpolate_grad(int B, int N, int M, int C, int K, const int* nn_index,
                                               const int* nn_count, const float* grad_output, const float* weight,
                                               float* grad_input, void* workspace, size_t workspace_bytes,
                                               sph3d_stream_t stream)
{
    return grad_via_transpose("sph3d_weighted_interpolate_grad", B, M, N, C, K, nn_index, nn_count, weight, grad_output,
                              grad_input, workspace, workspace_bytes, stream);
}
