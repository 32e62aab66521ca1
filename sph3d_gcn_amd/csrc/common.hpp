// common.hpp — shared host/device helpers for libsph3d (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sph3d.h"

namespace sph3d {

constexpr int kWave = 64;           // CDNA wavefront width
constexpr int kRefGrid = 32;        // the reference launches every kernel <<<32,1024>>>;
constexpr int kRefBlock = 1024;     // its thread->work mapping defines the radius-growth chains and FPS tie-break

// ---- host-side status plumbing ------------------------------------------------
void set_error(const char* fmt, ...);   // stores a thread-local message (api.cpp)

#define SPH3D_REQUIRE(cond, ...)                    \
    do {                                            \
        if (!(cond)) {                              \
            ::sph3d::set_error(__VA_ARGS__);        \
            return SPH3D_EINVAL;                    \
        }                                           \
    } while (0)

static inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return SPH3D_ELAUNCH;
    }
    return SPH3D_OK;
}

// api.cpp: a library-owned device buffer of at least `bytes` for work queued on `stream` of the current device (one per
// (device, stream, kind), grown on demand and kept; nullptr if the allocation fails or the stream is being captured).
// kind 0: only for state that is dead when the call's last kernel has run.  kind 1: zero-initialised; its users leave the
// first 16 KB zero (arrival counters, gemm.hip)
void* stream_scratch(hipStream_t stream, size_t bytes, int kind = 0);

static inline int check_hip(hipError_t e, const char* what)
{
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SPH3D_ELAUNCH;
    }
    return SPH3D_OK;
}

static inline hipStream_t as_stream(sph3d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// graph.hip: stream-ordered zero fill of `bytes` bytes at `p` (4-byte aligned, bytes % 4 == 0).  hipMemsetAsync's fill kernel
// runs 256 workgroups whatever the size: 245 us for the 17 MB of level-0 segment counters (70 GB/s; rocprofv3 trace of round 5);
// large fills go through a kernel that covers the chip (8 us for the same buffer), small ones stay with the runtime.
int zero_async(void* p, size_t bytes, hipStream_t stream, const char* what);

// ---- workspace layout of the transposed-graph build (graph.hip; the fused neighbour search of nnquery.hip / nngrid.hip writes
// the counting phase's three arrays itself) ----
//   deg[B*L]        in-degree of every (source, bin) segment, L = N*F; the fill cursors' zero state after the scan
//   bin_used[F]     1 for every bin that occurs
//   status[B*chunks]  64-bit (flag << 32 | value) words of the single-pass scan (decoupled look-back), chunks = ceil(L / 2048)
//   slot_pos[B*M*K] position of every edge inside its segment
// deg .. status are ONE zero fill (tg_zero_words words from deg).
constexpr int kTgChunk = 2048;
struct TgWs {
    int* deg; int* bin_used; unsigned long long* status; int* slot_pos; int L; int chunks; size_t zero_words; size_t total_words;
};
static inline TgWs tg_ws(void* workspace, int B, int N, int M, int K, int F)
{
    TgWs w;
    w.L = N * F;
    w.chunks = (w.L + kTgChunk - 1) / kTgChunk;
    const size_t head = ((size_t)B * w.L + (size_t)F + 1) & ~(size_t)1;         // status words are 8-byte aligned
    w.deg = (int*)workspace;
    w.bin_used = w.deg + (size_t)B * w.L;
    w.status = reinterpret_cast<unsigned long long*>(w.deg + head);
    w.zero_words = head + 2 * (size_t)B * w.chunks;
    w.slot_pos = w.deg + w.zero_words;
    w.total_words = w.zero_words + (size_t)B * M * K;
    return w;
}

// ---- packed entries of the transposed graph (round 6) ----------------------------------------------------------------------------
// ent_scale == NULL: an entry word of ent_key is  m | nn_count[m] << 24  (un-weighted graphs with at most 2^24 rows and K <= 255):
// the fill pass then makes ONE scattered 4-byte store per edge instead of two, and a consumer takes the row and its 1 / count
// (the same correctly rounded division the fill pass would have done) from one word.
constexpr unsigned kTgKeyMask = 0x00ffffffu;
static inline bool tg_packable(int M, int K, const void* weight) { return weight == nullptr && M <= (1 << 24) && K <= 255; }

// ---- device helpers -----------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ int tg_key(int word, bool packed) { return packed ? (int)((unsigned)word & kTgKeyMask) : word; }
__device__ __forceinline__ float tg_packed_scale(int word) { return 1.0f / (float)((unsigned)word >> 24); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int prefix_popc(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// make a wave-uniform int visibly scalar to the compiler
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniformf(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// XCD-affine work decode.  Workgroup `bid` runs on XCD (bid % 8) (observed placement, used for
// L2 locality only, never for correctness).  Clouds are dealt to XCDs round-robin so that all
// workgroups of one cloud share one L2: returns (cloud, part) for this block or cloud = -1.
// Batches of fewer than five clouds would leave XCDs without work that way (ONE cloud: the whole launch on 32 of the 256 CUs —
// the ScanNet-shape line until round 5): such a batch gives every cloud 2, 4 or 8 XCDs and deals the cloud's parts over them.
__host__ __device__ __forceinline__ int xcd_spread_log2(int B) { return B >= 5 ? 0 : (B >= 3 ? 1 : (B == 2 ? 2 : 3)); }
__device__ __forceinline__ void xcd_decode(int bid, int B, int parts, int& cloud, int& part)
{
    const int ls = xcd_spread_log2(B), s = 1 << ls;       // XCDs per cloud
    const int x = bid & 7;
    const int y = bid >> 3;
    const int pps = (parts + s - 1) >> ls;                // parts per XCD of a cloud
    const int row = y / pps;
    cloud = (x >> ls) + (8 >> ls) * row;
    part = (x & (s - 1)) + s * (y - row * pps);
    if (cloud >= B || part >= parts) cloud = -1;
}
static inline int xcd_grid(int B, int parts)
{
    const int ls = xcd_spread_log2(B), s = 1 << ls, cpr = 8 >> ls;
    return 8 * ((B + cpr - 1) / cpr) * ((parts + s - 1) >> ls);
}

}  // namespace sph3d
