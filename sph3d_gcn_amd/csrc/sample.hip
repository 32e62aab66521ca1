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
#include <atomic>
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

// ---------------------------------------------------------------------------------------------------------------
// fps_prune_kernel (round 5): the same sampling — same arithmetic per point, same winner per round, bit for bit — with the
// per-round distance update PRUNED.  A round only changes min(d, td) of a point when the new sample is closer to it than its
// running distance td; late in the chain that is a handful of points around the sample, yet fps_reg_kernel recomputes all n
// distances every round — and it is bound by exactly that: ~150 VALU instructions per wave and round at 8 points per
// thread, four waves per SIMD, one wave instruction per four cycles = the measured 1.09 us per round (8192 -> 2048).
//   * prologue: the cloud is counting-sorted by the Morton code of a 16^3 grid over its bounding box (LDS histogram); wave w
//     takes sorted positions [w * 64 P, (w + 1) * 64 P), register slot p of the wave the 64 consecutive positions
//     (w P + p) * 64 + lane: a SLOT is a spatial blob of 64 points with a bounding box;
//   * the sample of a round is the arg-max of the running distances: its own running distance g bounds EVERY point's (and
//     running distances only shrink).  Per round a wave tests all its P slots at once (lane p < P holds slot p's box): with
//     q = the sample clamped into the box, dbox = (dx*dx + dy*dy) + dz*dz of q — the SAME expression as a point's distance,
//     every operation monotone in |dx|, |dy|, |dz|, and |q - s| <= |pt - s| per axis for every point of the box, so dbox <= the
//     computed distance of every point of the slot in floating point, not just in exact arithmetic.  dbox >= g  =>  d >= td
//     for every point: min(d, td) = td, the slot is untouched.  Only the slots that fail the test are updated;
//   * a lane keeps the arg-max of its own P points cached (key, slot).  Running distances only shrink, so that cache goes
//     stale only when the lane's arg-max point itself was lowered: the wave rescans its points (and re-elects its candidate)
//     only in the rounds in which that happened to some lane — otherwise its published candidate stands;
//   * 16 waves as before, and consecutive blobs dealt to the waves in turn.  What bounds a round is the LONGEST dependent
//     instruction path of any wave between two barriers (cycle counters per phase, tools/exp_fps_prof.py ->
//     profiles/r05_exp_fps_prune_cycles*.log: an idle wave needs 140 cycles for its box test, a wave with work 650-1300, the
//     candidate reduction after the barrier 450-800, ~10 cycles per dependent instruction whatever the instruction count),
//     not the instructions a SIMD issues in total: measured and dropped — exact per-slot maxima with a candidate election per
//     active wave (1.06 us per round at 8192 points against 1.09 for the plain kernel); 4 or 8 waves with 32 / 16 points per
//     lane (1.53 / 1.15: fewer waves hide less of each other's latency); updating ALL slots of an active wave in straight-line
//     code instead of branching per slot (0.95, 1.10 with the blobs dealt in turn); a wave owning P CONSECUTIVE blobs (0.97).
//     This form (with the atomic-max exchange and the winning lane publishing its own coordinates, below): 0.82 us per round at
//     8192 points (plain 1.09), 0.87 at 10 000 (1.39), 0.70 at 4096 (0.77), 0.69 at 2500 (0.77); at 2048 the plain kernel stays
//     ahead (0.61 against 0.69).  Two diagnostic builds (-DSPH3D_FPS_EXP) bound what is left: a round WITHOUT any update (box
//     test, publish, barrier, exchange) takes 0.24 us; the rest is the dependent path of the waves with work — above all of the
//     wave that owned the winner: its candidate just became the sample, so it updates, rescans and re-elects every round;
//   * the reference's tie-break (tf_sample_gpu.cu:49,56-66: larger distance, then lower thread id k mod 1024, then lower k)
//     no longer follows from the thread mapping — the points are permuted — so it is carried explicitly: a point's key is
//     (bits of td) << 32 | sec(k), sec(k) = (1023 - k mod 1024) << 8 | (255 - k div 1024); the largest key wins and the
//     winner's index is decoded from sec.  (td >= +0, so its bits order as unsigned integers; points past the cloud's end
//     carry td = +0 and sec = 0: below every real point.)
// Which points share a slot (the order inside a Morton cell is the arrival order of an LDS atomic) never changes a result.
// ---------------------------------------------------------------------------------------------------------------
template <bool MAX>
__device__ __forceinline__ float wave_red_f32(float v)
{
    // gfx9 DPP reduction like wave_max_u32 (prologue only); lanes without a source keep `old` = the identity
    const int ident = MAX ? (int)0xff800000u : (int)0x7f800000u;
    float t;
#define SPH3D_RED_STEP(ctrl, rmask)                                                                             \
    t = __int_as_float(__builtin_amdgcn_update_dpp(ident, __float_as_int(v), ctrl, rmask, 0xf, false));         \
    v = MAX ? fmaxf(v, t) : fminf(v, t);
    SPH3D_RED_STEP(0x111, 0xf)
    SPH3D_RED_STEP(0x112, 0xf)
    SPH3D_RED_STEP(0x114, 0xf)
    SPH3D_RED_STEP(0x118, 0xf)
    SPH3D_RED_STEP(0x142, 0xa)
    SPH3D_RED_STEP(0x143, 0xc)
#undef SPH3D_RED_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ unsigned morton4(unsigned v)      // 4 bits -> every third bit
{
    v &= 0xfu;
    v = (v | (v << 4)) & 0xc3u;
    v = (v | (v << 2)) & 0x249u;
    return v;
}

typedef unsigned long long u64;

struct __attribute__((aligned(16))) FpsSlotP {
    unsigned vbits;        // bits of the candidate's running distance (>= +0: ordered as unsigned)
    float x, y, z;
};

constexpr int kPruneCells = 4096;

// -DSPH3D_FPS_PROF (diagnostic builds only, tools/gpu_fps_prof.sh): cycle counts of the round loop's phases, per wave of cloud 0
#ifdef SPH3D_FPS_PROF
__device__ unsigned long long g_fps_prof[16][8];
#define FPS_CLK() __builtin_readcyclecounter()
#else
#define FPS_CLK() 0ull
#endif

// P points per lane (lane p < P holds slot p's box), NW waves per workgroup
template <int P, int NW>
__global__ __launch_bounds__(64 * NW) void fps_prune_kernel(int b, int n, int m, const float* __restrict__ dataset, int* __restrict__ idxs)
{
    constexpr int NT = 64 * NW;
    static_assert((P == 2 || P == 4 || P % 8 == 0) && P <= 32 && (NW == 4 || NW == 8 || NW == 16), "fps_prune_kernel: shape");
    constexpr int G8 = P < 8 ? P : 8;
    __shared__ int hist[kPruneCells];
    __shared__ unsigned short sidx[P * NT];
    __shared__ float red[6][NW];
    __shared__ int wsum[NW];
    __shared__ FpsSlotP slots[2][NW];
    __shared__ u64 cell[3];
    const int t = (int)threadIdx.x;
    const int lane = t & 63;
    const int wave = uniform(t >> 6);

    for (int i = (int)blockIdx.x; i < b; i += (int)gridDim.x) {
        const float* pts = dataset + (size_t)i * n * 3;
        // ---- bounding box of the cloud (non-finite coordinates are ignored by fminf / fmaxf; a cloud of them alone sorts into cell 0) ----
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int k = t; k < n; k += NT)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float v = pts[k * 3 + a];
                lo[a] = fminf(lo[a], v);
                hi[a] = fmaxf(hi[a], v);
            }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = wave_red_f32<false>(lo[a]);
            hi[a] = wave_red_f32<true>(hi[a]);
        }
        __syncthreads();                                     // (the previous cloud's rounds are done with red / hist / sidx)
        if (lane == 0)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                red[a][wave] = lo[a];
                red[3 + a][wave] = hi[a];
            }
        for (int c = t; c < kPruneCells; c += NT) hist[c] = 0;
        __syncthreads();
        float inv[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float l = red[a][0], h = red[3 + a][0];
            for (int w = 1; w < NW; w++) {
                l = fminf(l, red[a][w]);
                h = fmaxf(h, red[3 + a][w]);
            }
            lo[a] = l;
            const float ext = h - l;
            inv[a] = (ext > 0.f && ext < 3.0e38f) ? 16.0f / ext : 0.f;
        }
        auto cell_of = [&](int k) -> int {
            unsigned c = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float f = (pts[k * 3 + a] - lo[a]) * inv[a];
                int q = (f >= 0.f) ? (f < 15.f ? (int)f : 15) : 0;            // NaN -> 0
                c |= morton4((unsigned)q) << a;
            }
            return (int)c;
        };
        for (int k = t; k < n; k += NT) atomicAdd(&hist[cell_of(k)], 1);
        __syncthreads();
        {   // exclusive scan of the 4096 counters: PER consecutive ones per thread
            constexpr int PER = kPruneCells / NT;
            int s = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) s += hist[t * PER + j];
            int incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; w++) base += wsum[w];
            int run = base + incl - s;
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const int c = hist[t * PER + j];
                hist[t * PER + j] = run;
                run += c;
            }
        }
        __syncthreads();
        for (int k = t; k < n; k += NT) sidx[atomicAdd(&hist[cell_of(k)], 1)] = (unsigned short)k;
        __syncthreads();

        // ---- this lane's P points: slot p of wave w = sorted positions (w P + p) * 64 + lane ----
        float px[P], py[P], pz[P];
        u64 key[P];                                            // (bits of the running distance) << 32 | sec
        float blx = 0.f, bly = 0.f, blz = 0.f, bhx = 0.f, bhy = 0.f, bhz = 0.f;      // lane p < P: slot p's bounding box
#pragma unroll
        for (int p = 0; p < P; p++) {
            // slot p of wave w = blob p * NW + w of 64 consecutive sorted positions: consecutive blobs are dealt to the waves in
            // turn, so the few blobs around a sample — neighbours in the sorted order — belong to different waves and are
            // updated side by side (a wave owning P consecutive blobs: 0.97 instead of 0.89 us per round at 8192 points)
            const int blob = p * NW + wave;
            const int q = blob * 64 + lane;
            const bool ok = q < n;
            const int k = ok ? (int)sidx[q] : 0;
            px[p] = pts[k * 3];
            py[p] = pts[k * 3 + 1];
            pz[p] = pts[k * 3 + 2];
            // tf_sample_gpu.cu:19-21 (td = 1e38); a position past the end: td = +0 (never lowered) and sec 0 (below every point)
            const unsigned sec = ok ? (((unsigned)(1023 - (k & 1023)) << 8) | (unsigned)(255 - (k >> 10))) : 0u;
            key[p] = ((u64)(ok ? __float_as_uint(1e38f) : 0u) << 32) | sec;
            const float lx = wave_red_f32<false>(ok ? px[p] : 3.0e38f), hx = wave_red_f32<true>(ok ? px[p] : -3.0e38f);
            const float ly = wave_red_f32<false>(ok ? py[p] : 3.0e38f), hy = wave_red_f32<true>(ok ? py[p] : -3.0e38f);
            const float lz = wave_red_f32<false>(ok ? pz[p] : 3.0e38f), hz = wave_red_f32<true>(ok ? pz[p] : -3.0e38f);
            // (an empty slot gets a box far away: its dbox is +inf, it is never updated)
            const bool any = blob * 64 < n;
            if (lane == p) { blx = lx; bly = ly; blz = lz; bhx = any ? hx : 3.0e38f; bhy = any ? hy : 3.0e38f; bhz = any ? hz : 3.0e38f; }
            // (one slot after the other: interleaving the P iterations' loads and reductions costs ~100 VGPRs at P = 32, and the
            // round loop below then runs out of the accumulation registers)
            __builtin_amdgcn_sched_barrier(0);
        }
        float x1 = pts[0], y1 = pts[1], z1 = pts[2];           // old = 0 (:16)
        if (t == 0) idxs[(size_t)i * m] = 0;
        // g = the running distance of the current sample = the maximum over ALL points; the wave's published candidate;
        // the lane's cached arg-max (key, slot)
        float g = __builtin_inff();
        unsigned c_vb = 0u, c_sec = 0u;
        int c_wl = 0, dirty = 0;
        u64 kb = 0ull;
        int bp = 0;
        float bx = 0.f, by = 0.f, bz = 0.f;
        int c3 = 1;                                            // j mod 3
        if (t < 3) cell[t] = 0ull;
        __syncthreads();

#ifdef SPH3D_FPS_PROF
        unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // own work (active / idle rounds), their counts, publish + barrier, post, rescans, their own cycles
#endif
        for (int j = 1; j < m; j++) {
            const unsigned long long tk0 = FPS_CLK(); (void)tk0;
            // ---- which slots can the new sample change? (lane p tests slot p; v_med3_f32 = the clamp into [lo, hi]) ----
            const float qx = __builtin_amdgcn_fmed3f(x1, blx, bhx), qy = __builtin_amdgcn_fmed3f(y1, bly, bhy),
                        qz = __builtin_amdgcn_fmed3f(z1, blz, bhz);
            const float ex = qx - x1, ey = qy - y1, ez = qz - z1;
            const float dbox = (ex * ex + ey * ey) + ez * ez;
            const bool need = lane < P && !(dbox >= g);        // NaN on either side: update
#ifndef SPH3D_FPS_EXP
#define SPH3D_FPS_EXP 0        // diagnostic builds only (WRONG samples): 1 = never rescan after round 1, 2 = never update after round 1
#endif
            const unsigned mask = (SPH3D_FPS_EXP == 2 && j > 1) ? 0u : (unsigned)__ballot(need);
            u64 stale = j == 1 ? ~0ull : 0ull;                 // lanes whose cached arg-max was lowered this round
            if (mask != 0u) {
#pragma unroll
                for (int g8 = 0; g8 < P; g8 += G8) {
                    if ((mask >> g8) & ((1u << G8) - 1u)) {    // wave-uniform, like the bit tests below
#pragma unroll
                        for (int p = g8; p < g8 + G8; p++) {
                            if ((mask >> p) & 1u) {
                                const float dx = px[p] - x1, dy = py[p] - y1, dz = pz[p] - z1;
                                const float d = (dx * dx + dy * dy) + dz * dz;                    // :45
                                const bool lower = d < __uint_as_float((unsigned)(key[p] >> 32));    // :46 min(d, td)
                                key[p] = lower ? (((u64)__float_as_uint(d) << 32) | (unsigned)key[p]) : key[p];
                                stale |= __ballot(lower && bp == p);
                            }
                        }
                    }
                }
            }
            if (SPH3D_FPS_EXP == 1 && j > 1) stale = 0ull;
            if (stale != 0ull) {
                // ---- some lane's arg-max moved: every lane rescans its points (a tournament: log2 P dependent steps), the
                // wave re-elects its candidate ----
                u64 tk[P];
                int tb[P];
                float tx[P], ty[P], tz[P];
#pragma unroll
                for (int p = 0; p < P; p++) { tk[p] = key[p]; tb[p] = p; tx[p] = px[p]; ty[p] = py[p]; tz[p] = pz[p]; }
#pragma unroll
                for (int st = 1; st < P; st <<= 1)
#pragma unroll
                    for (int q = 0; q + st < P; q += 2 * st) {
                        const bool gt = tk[q + st] > tk[q];
                        tk[q] = gt ? tk[q + st] : tk[q];
                        tb[q] = gt ? tb[q + st] : tb[q];
                        tx[q] = gt ? tx[q + st] : tx[q];
                        ty[q] = gt ? ty[q + st] : ty[q];
                        tz[q] = gt ? tz[q + st] : tz[q];
                    }
                kb = tk[0];
                bp = tb[0];
                bx = tx[0];                                    // the lane's best point travels with its key: the winning lane
                by = ty[0];                                    // publishes its coordinates from its own registers (below)
                bz = tz[0];
                const unsigned kh = (unsigned)(kb >> 32);
                const unsigned wmb = wave_max_u32(kh);
                const u64 tie = __ballot(kh == wmb);
                c_wl = (int)__builtin_ctzll(tie);
                if (tie & (tie - 1ull)) {                      // several lanes at the maximum: the largest sec among them
                    const unsigned ls = kh == wmb ? (unsigned)kb : 0u;
                    const unsigned ws = wave_max_u32(ls);
                    c_wl = (int)__builtin_ctzll(__ballot(kh == wmb && ls == ws));
                    c_sec = ws;
                } else {
                    c_sec = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)kb, c_wl);
                }
                c_vb = wmb;
                dirty = 2;                                     // both parities of the wave's slot take the new coordinates
            }
            const unsigned long long tk1 = FPS_CLK(); (void)tk1;
            // ---- exchange: every wave publishes its candidate's coordinates and raises the round's cell to its key with ONE LDS
            // atomic max: (distance bits) << 32 | sec << 4 | wave — the largest key is the reference's winner (sec is unique per
            // point, so the wave bits never decide) and names the slot that holds its coordinates.  After the barrier a wave reads
            // the cell and that slot: ~10 instructions, where reducing the 16 candidates inside every wave (LDS read, six DPP steps,
            // ballots, read-lanes) took 35 — and the SIMD issues the post-barrier code of its four waves one after the other
            // (cycle counters: 450 cycles for the oldest wave of a SIMD, 790 for the youngest; 320-430 with the cell).  The plain
            // kernel above keeps its reduction: there all 16 waves reach the exchange together and their atomics on one address
            // queue up on the round's critical path (0.70 instead of 0.62 us per round at 2048 points); here the idle waves'
            // atomics are long done when the waves with work arrive.  Cells rotate over three rounds:
            // the one of round j + 2 is cleared after barrier j, when round j - 1's readers are done and before barrier j + 1 lets
            // round j + 2's writers through.
            // The coordinates never pass through scalar registers: after a re-election the winning LANE writes them from its own
            // registers into the wave's slot of this round's parity and of the next round's (a first version read the lane's slot index
            // back, branched to the matching registers and read-laned four values: ~300 of a rescan round's ~1250 cycles).
            const int buf = j & 1;
            if (dirty > 0) {                                   // wave-uniform
                if (lane == c_wl) {
                    FpsSlotP sl;
                    sl.vbits = c_vb; sl.x = bx; sl.y = by; sl.z = bz;
                    slots[buf][wave] = sl;
                }
                dirty--;
            }
            if (lane == 0)
                __hip_atomic_fetch_max(&cell[c3], ((u64)c_vb << 32) | ((u64)c_sec << 4) | (u64)wave, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            const unsigned long long tk2 = FPS_CLK(); (void)tk2;
            const u64 win = cell[c3];                              // the same address in every lane: a broadcast read
            const int c3n = c3 == 2 ? 0 : c3 + 1;
            if (t == 0) cell[c3n == 2 ? 0 : c3n + 1] = 0ull;       // = (c3 + 2) % 3
            c3 = c3n;
            const FpsSlotP s = slots[buf][(int)(win & 15ull)];
            const unsigned gsec = (unsigned)(win >> 4) & 0x3ffffu;
            x1 = uniformf(s.x);
            y1 = uniformf(s.y);
            z1 = uniformf(s.z);
            g = uniformf(__uint_as_float((unsigned)(win >> 32)));
            if (t == 0) {
                // sec -> k; sec == 0: no point at all (cannot happen for n >= 1) -> the reference's idle-thread index 0
                const int k = gsec != 0u ? (((255 - (int)(gsec & 255u)) << 10) | (1023 - (int)(gsec >> 8))) : 0;
                idxs[(size_t)i * m + j] = k;
            }
#ifdef SPH3D_FPS_PROF
            {
                const unsigned long long tk3 = FPS_CLK();
                const bool active = mask != 0u;
                pf[active ? 0 : 1] += tk1 - tk0;
                pf[active ? 2 : 3] += 1;
                pf[4] += tk2 - tk1;
                pf[5] += tk3 - tk2;
                pf[6] += stale != 0ull ? 1 : 0;
                pf[7] += (stale != 0ull) ? (tk1 - tk0) : 0ull;       // own cycles of the rounds with a rescan
            }
#endif
        }
#ifdef SPH3D_FPS_PROF
        if (i == 0 && lane == 0)
            for (int q = 0; q < 8; q++) g_fps_prof[wave][q] = pf[q];
#endif
    }
}

#ifdef SPH3D_FPS_PROF
}  // namespace sph3d
extern "C" int sph3d_debug_fps_prof(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sph3d::g_fps_prof), sizeof(unsigned long long) * 16 * 8) == hipSuccess ? 0 : -3;
}
namespace sph3d {
#endif

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
#ifndef SPH3D_FPS_COOP_P
#define SPH3D_FPS_COOP_P 4
#endif
constexpr int kCoopP = SPH3D_FPS_COOP_P;   // points per thread
constexpr int kCoopPts = kRefBlock * kCoopP;
#ifndef SPH3D_FPS_COOP_ERRPOLL
#define SPH3D_FPS_COOP_ERRPOLL 0
#endif
#ifndef SPH3D_FPS_COOP_SLOAD
#define SPH3D_FPS_COOP_SLOAD 1
#endif

__global__ __launch_bounds__(1024) void fps_coop_kernel(int b, int xcd_local, int n, int m, int G, const float* __restrict__ dataset,
                                                        unsigned long long* __restrict__ slots, int* __restrict__ err,
                                                        int* __restrict__ idxs)
{
    __shared__ FpsSlot16 lslots[2][16];
    __shared__ int lslot_tp[2][16];          // winner thread's t | p << 16
    __shared__ int win_k[2];
    const int t = (int)threadIdx.x;
    const int lane = t & 63;
    const int wave = uniform(t >> 6);
    // xcd_local: the G workgroups of a cloud on ONE XCD (block b runs on XCD b % 8 — observed, used for speed only: an exchange
    // inside one L2 is 0.1-0.3 us shorter than across the fabric); the grid is 8 x as large and the surplus workgroups leave
    int i, g;
    if (xcd_local) {
        const int slot = (int)blockIdx.x >> 3;
        i = (slot / G) * 8 + (((int)blockIdx.x + 9 - xcd_local) & 7);             // xcd_local - 1 = the XCD of cloud 0 (rotates per launch)
        g = slot % G;
        if (i >= b) return;
    } else {
        i = (int)blockIdx.x / G;
        g = (int)blockIdx.x % G;
    }
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
            // (round 5, measured and dropped, 65 536 points, us per round: the candidates' coordinates as three more tagged granules
            //  per workgroup, one 64-lane sweep — 2.18 against 1.96, and 2.55 with every wave polling for itself instead of wave 0 +
            //  barrier: the poll's price is the number of 8-byte requests in the polling CU's memory queue; a lane fetching its
            //  candidate's point as soon as its granule is in, under the poll for the others: 1.97 against 1.72)
            const int gl = lane < G ? lane : 0;
            unsigned long long v = 0ull;
            int spins = 0;
            for (;;) {
                v = __hip_atomic_load(&myslots[buf * G + gl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (int)(v & 0x3fffull) == (j & 0x3fff);
                if (__ballot(ok) == ~0ull) break;
                // (the error word is looked at every 16th poll only: a second fabric round trip in EVERY failed poll doubled the
                //  period of the poll, i.e. the mean delay between a granule's arrival and its detection)
                if (++spins > (1 << 22) || ((spins & 15) == SPH3D_FPS_COOP_ERRPOLL && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = ~0ull;                                                   // poison: the round loop ends below
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            // the winner: max of the granules' upper 50 bits (value, then 1023 - t, then 255 - (k >> 10)) in two 32-bit DPP steps
            // (a 64-bit butterfly is twelve dependent ds_bpermute)
            const bool poisoned = __ballot(v == ~0ull) != 0ull;
            const unsigned hi = lane < G ? (unsigned)(v >> 32) : 0u;
            const unsigned hmax = wave_max_u32(hi);
            const unsigned lo = (lane < G && hi == hmax) ? (unsigned)((v >> 14) & 0x3ffffull) + 1u : 0u;
            const unsigned lmax = wave_max_u32(lo);
            int k = -1;
            if (!poisoned) {
                const unsigned low18 = lmax - 1u;
                const int kt = 1023 - (int)((low18 >> 8) & 0x3ffu);
                const int kq = 255 - (int)(low18 & 0xffu);
                k = kq * kRefBlock + kt;
                // best < 0 everywhere (no point left: cannot happen for m <= n) -> index 0 like the reference's idle threads
                if (hmax == order_bits(-1.f)) k = 0;
                if (k >= n) k = 0;
            }
            if (lane == 0) {
                win_k[buf] = k;
                if (g == 0 && k >= 0) idxs[(size_t)i * m + j] = k;
            }
        }
        __syncthreads();
#if SPH3D_FPS_COOP_SLOAD
        const int k = uniform(win_k[buf]);                                       // scalar loads of the winner's point (read-only cloud)
#else
        const int k = win_k[buf];
#endif
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

// cloud size from which the pruned kernel is used (SPH3D_FPS_PRUNE=<points>, 0 = never; read once)
#ifndef SPH3D_FPS_PRUNE_MIN
#define SPH3D_FPS_PRUNE_MIN 2049
#endif
static int fps_prune_min_points()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SPH3D_FPS_PRUNE");
        v = e ? atoi(e) : SPH3D_FPS_PRUNE_MIN;
        if (v <= 0) v = 1 << 30;
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
    // clouds of 2049 .. 16384 points: the pruned kernel (same samples bit for bit; at 2048 points it only ties with the plain
    // kernel: 0.620 vs 0.614 us per round).  SPH3D_FPS_PRUNE=0 switches it off, =<n>: for clouds of at least n points
    if (n >= fps_prune_min_points() && n > kRefBlock && n <= 16384 && m > 1) {
        const int need = (n + kRefBlock - 1) / kRefBlock;
        const dim3 pblock(kRefBlock);
#define SPH3D_FPSP(PP) hipLaunchKernelGGL((fps_prune_kernel<PP, 16>), grid, pblock, 0, st, b, n, m, inp, out)
        if (need <= 4) SPH3D_FPSP(4);
        else if (need <= 8) SPH3D_FPSP(8);
        else SPH3D_FPSP(16);
#undef SPH3D_FPSPN
#undef SPH3D_FPSP
    }
    else if (P <= 1) SPH3D_FPS(1);
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
            // all workgroups of a cloud on one XCD while they take at most half of its 32 CUs (one 1024-thread workgroup per CU)
            static const int xcd_env = getenv("SPH3D_FPS_COOP_XCD") ? atoi(getenv("SPH3D_FPS_COOP_XCD")) : 1;      // (experiments)
            const int rounds8 = (b + 7) / 8;
            // (launches rotate over the XCDs: two sampling streams' kernels of one-cloud batches would share XCD 0 otherwise)
            static std::atomic<unsigned> rotate{0};
            const int xcd_local = (xcd_env != 0 && rounds8 * G <= 16) ? 1 + (int)(rotate.fetch_add(1, std::memory_order_relaxed) & 7u) : 0;      // (32 per XCD measured slower than spread: 2.88 vs 2.46 us per round at 131 072 points)
            const unsigned nblk = xcd_local ? (unsigned)(8 * rounds8 * G) : (unsigned)(b * G);
            hipLaunchKernelGGL(fps_coop_kernel, dim3(nblk), dim3(kRefBlock), 0, st, b, xcd_local, n, m, G, inp, slots, err, out);
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
