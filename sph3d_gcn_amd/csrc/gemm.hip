// gemm.hip — pointwise 1x1 feature GEMM on the fp32 matrix cores of gfx950.
//
// Replaces the tf.matmul inside separable_conv3d / pointwise_conv3d / fully_connected
// (utils/sph3gcn_util.py:146-150, 204-206, 260; cuBLAS SGEMM in the reference) and its two backward products.
//
// All three products are tall-skinny: R = B*points rows (up to 131072), channel dims 3..2048.
//   NN : Y [R,Cout]   = X [R,Cin]  * W[Cin,Cout]   (+bias, optional ELU)          forward
//   NT : dX[R,Cin]    = dY[R,Cout] * W^T                                           input gradient
//   TN : dW[Cin,Cout] = X^T        * dY[R,Cout]     (reduction over R, split-K)    weight gradient
//
// MI355X design: v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain, no TF32-style truncation, so the
// 1e-5 activation bound holds).  Workgroup = 4 waves, block tile 128x128 / 128x64 / 64x64 (the largest that still
// gives all 256 CUs work: the row count shrinks 64-fold from level 0 to level 3), BK = 16 so that the 128x128 kernel
// fits 128 VGPRs and 40 KB of LDS = FOUR workgroups per CU (BK = 32 at two per CU measured slower for all three
// products); each wave owns a quarter of the tile as independent 32x32 accumulators.  Operand tiles go global -> registers (prefetch of tile t+1 during
// the MFMAs of tile t) -> a double-buffered LDS image lds[row][k] (one barrier per k-tile) from which a lane's
// operands for four MFMA steps are a single conflict-free ds_read_b128 (Stage / kmap comments below).
// Ragged edges (Cin = 3, Cout = 13, R not a multiple of 128) take guarded scalar loads / stores.
// TN splits R over workgroups; every split writes its partial tile to a workspace slab and a second kernel adds the
// slabs in a fixed order (deterministic, no float atomics).
#include <cstdlib>
#include "common.hpp"

namespace sph3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// prefetch registers are NATIVE vectors: HIP's float4 is a struct, whole-struct copies become memcpy's through a private
// array that the compiler then keeps in scratch memory (round 2: scratch_store behind every A-tile load of the main loop)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 as_v4(const float4 v) { f32x4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }

#ifndef SPH3D_GEMM_EXP
#define SPH3D_GEMM_EXP 0      // diagnostic builds only (wrong results): 1 no global loads, 2 no LDS stores, 4 no barrier in the k loop
#endif
#ifndef SPH3D_GEMM_SETPRIO
#define SPH3D_GEMM_SETPRIO 1
#endif
constexpr bool kSetPrio = SPH3D_GEMM_SETPRIO != 0;
constexpr int BM = 128;
constexpr int BKS = 16;      // k-tile for small grids (more workgroups per CU)
constexpr int BKL = 32;      // k-tile for large grids (half the barriers)

// load a 4-wide chunk of a [rows x cols] row-major matrix at (r, c..c+3), zero outside
__device__ __forceinline__ float4 load4_guard(const float* __restrict__ p, int ld, int r, int c, int rows, int cols, bool vec_ok)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
        const float* q = p + (size_t)r * ld + c;
        if (vec_ok && c + 3 < cols) {
            v = *reinterpret_cast<const float4*>(q);
        } else {
            if (c < cols) v.x = q[0];
            if (c + 1 < cols) v.y = q[1];
            if (c + 2 < cols) v.z = q[2];
            if (c + 3 < cols) v.w = q[3];
        }
    }
    return v;
}

// Operand tile staging.  Whatever the memory order, the LDS image is lds[row][k] with row stride BK+4 floats:
//   * a lane's MFMA fragment for FOUR consecutive k-steps is one aligned ds_read_b128 (see kmap below);
//   * 16 rows x stride 20 dwords hit 16 distinct 4-dword bank slots -> the b128 reads are conflict-free;
//   * every source is stored with ds_write_b128 (row-contiguous sources are transposed 4x4 in registers first).
// KMAJ = true : memory is [row][k] (k contiguous);  KMAJ = false: memory is [k][row] (row contiguous).
// BT = tile extent along the row dimension.  256 threads.
template <bool KMAJ, int BT, int BK>
struct Stage {
    static constexpr int LDK = BK + 4;
    // KMAJ : one float4 (4 consecutive k of one row) per chunk, BT*BK/4 chunks, stored with ds_write_b128.
    // !KMAJ: one KU(k) x 4(row) block per unit: KU float4 loads along the row dimension (lanes with the same k
    //        cover contiguous bytes), transposed in registers, stored as four ds_write_b128 (KU = 4) or ds_write_b64
    //        (KU = 2) along k.  Lane -> (k-unit = lane % (BK/KU), row-quad = lane / (BK/KU)) puts the lanes of a store
    //        group on distinct bank slots.  (Round-1 PMC: the earlier scalar transposing store was 16-way
    //        bank-conflicted, 77 % of LDS cycles.)  KU = 2 for BK = 16: every one of the 256 threads then holds 8 prefetch
    //        registers for a 128-row tile instead of half of them holding 16 (the 128x128 kernel is at the 128-VGPR
    //        limit of four workgroups per CU; round 2 found its A prefetch spilled to scratch with a vmcnt wait
    //        right behind the load).
    static constexpr int KU = KMAJ ? 4 : (BK == 16 ? 2 : 4);
    static constexpr int KUN = BK / KU;                       // k-units per tile (!KMAJ), k-quads (KMAJ: BK/4)
    static constexpr int UNITS = KMAJ ? BT * BK / 4 : (BT / 4) * KUN;
    static constexpr int NCH = (UNITS + 255) / 256;
    static constexpr int NREG = KMAJ ? NCH : NCH * KU;
    static constexpr int LDS_FLOATS = BT * LDK;
    f32x4 r[NREG];

    template <bool GUARD>
    __device__ __forceinline__ void load(const float* __restrict__ p, int ld, int row0, int k0, int rows, int kdim, bool vec_ok)
    {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int u = (int)threadIdx.x + i * 256;
            if (KMAJ) {
                if (UNITS < 256 && u >= UNITS) { r[i] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
                const int row = u / (BK / 4), kq = u % (BK / 4);
                if (!GUARD) r[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(row0 + row) * ld + k0 + kq * 4);
                else r[i] = as_v4(load4_guard(p, ld, row0 + row, k0 + kq * 4, rows, kdim, vec_ok));
            } else {
                if (UNITS % 256 != 0 && u >= UNITS) {
#pragma unroll
                    for (int t = 0; t < KU; t++) r[i * KU + t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    continue;
                }
                const int kq = u % KUN, rq = u / KUN;
#pragma unroll
                for (int t = 0; t < KU; t++) {
                    if (!GUARD) r[i * KU + t] = *reinterpret_cast<const f32x4*>(p + (size_t)(k0 + kq * KU + t) * ld + row0 + rq * 4);
                    else r[i * KU + t] = as_v4(load4_guard(p, ld, k0 + kq * KU + t, row0 + rq * 4, kdim, rows, vec_ok));
                }
            }
        }
    }

    // Unguarded whole tiles: `base` is the WORKGROUP-UNIFORM address of the tile's (row0, k0) element (scalar registers,
    // advanced by the caller with scalar adds), the lane's part is ONE loop-invariant 32-bit offset: the loads are
    // global_load_dwordx4 v, v_off, s[base] instead of six 64-bit pointer VGPR pairs with a v_lshl_add_u64 each per tile.
    __device__ __forceinline__ unsigned lane_offset(int ld) const
    {
        const int u = (int)threadIdx.x;
        if (KMAJ) return (unsigned)((u / (BK / 4)) * ld + (u % (BK / 4)) * 4);
        return (unsigned)((u % KUN) * KU * ld + (u / KUN) * 4);
    }
    __device__ __forceinline__ void load_fast(const float* __restrict__ base, int ld, unsigned voff)
    {
        if (UNITS < 256 && (int)threadIdx.x >= UNITS) return;      // (registers of idle threads are never stored)
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            if (KMAJ) {
                const float* b = base + (size_t)(i * (256 / (BK / 4))) * ld;                // uniform
                r[i] = *reinterpret_cast<const f32x4*>(b + voff);
            } else {
#pragma unroll
                for (int t = 0; t < KU; t++) {
                    const float* b = base + (size_t)t * ld + i * (256 / KUN) * 4;            // uniform
                    r[i * KU + t] = *reinterpret_cast<const f32x4*>(b + voff);
                }
            }
        }
    }
    __device__ __forceinline__ void store(float* lds) const
    {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int u = (int)threadIdx.x + i * 256;
            if (KMAJ) {
                if (UNITS < 256 && u >= UNITS) continue;
                const int row = u / (BK / 4), kq = u % (BK / 4);
                *reinterpret_cast<f32x4*>(lds + row * LDK + kq * 4) = r[i];
            } else {
                if (UNITS % 256 != 0 && u >= UNITS) continue;
                const int kq = u % KUN, rq = u / KUN;
                float* base = lds + (rq * 4) * LDK + kq * KU;
                if (KU == 4) {
                    const f32x4 a = r[i * 4], b = r[i * 4 + 1], c = r[i * 4 + 2], d = r[i * 4 + 3];   // k = 4kq + 0..3
                    *reinterpret_cast<f32x4*>(base) = f32x4{a.x, b.x, c.x, d.x};
                    *reinterpret_cast<f32x4*>(base + LDK) = f32x4{a.y, b.y, c.y, d.y};
                    *reinterpret_cast<f32x4*>(base + 2 * LDK) = f32x4{a.z, b.z, c.z, d.z};
                    *reinterpret_cast<f32x4*>(base + 3 * LDK) = f32x4{a.w, b.w, c.w, d.w};
                } else {
                    const f32x4 a = r[i * KU], b = r[i * KU + (KU > 1 ? 1 : 0)];                         // k = 2kq, 2kq + 1
                    *reinterpret_cast<f32x2*>(base) = f32x2{a.x, b.x};
                    *reinterpret_cast<f32x2*>(base + LDK) = f32x2{a.y, b.y};
                    *reinterpret_cast<f32x2*>(base + 2 * LDK) = f32x2{a.z, b.z};
                    *reinterpret_cast<f32x2*>(base + 3 * LDK) = f32x2{a.w, b.w};
                }
            }
        }
    }
};

// ---- LDS-DMA staging (unguarded whole tiles, BK = 16) -------------------------------------------------------------------
// global_load_lds_dwordx4 writes a wave's 64 x 16 B straight into LDS at (uniform base + lane * 16): no prefetch registers,
// no ds_write, no vmcnt wait in front of a store — phase-skip timing of the register-staged loop (tools/gpu_gemm_exp.sh) put
// 16 of the 93 us of the level-0 forward product into exactly that (global-load issue 6, LDS stores 7, barrier skew 3).
// The LDS image is therefore the MEMORY order of the tile, 16-byte granule by granule, and the only freedom left is WHICH
// global granule a lane fetches into its slot.  That freedom carries the bank swizzles:
//   k-contiguous operand ([row][16 k], 4 granules per row): slot = kq ^ ((row >> 2) & 3).  A lane's fragment for four MFMA
//       steps is still one ds_read_b128 at granule row*4 + ((2g + lk) ^ swizzle); the 16 rows of a b128 phase hit 16
//       distinct bank quads (rows r, r+4, r+8, r+12 would collide at stride 64 B without it).
//   row-contiguous operand ([16 k][BT rows]): element (k, row) sits at k*BT + (row ^ (bit 2 of k) << 5).  A lane reads its
//       four steps as four ds_read_b32 (k = 8g + 4lk + s, so bit 2 of k is the lane half lk): the two lane halves then use
//       opposite halves of the 64 banks instead of the same 32.
template <bool KM, int BT>
struct Dma {
    static constexpr int GR = BT * 4;                 // 16-byte granules of a BT x 16 tile
    static constexpr int NI = GR / 256;               // wave instructions per wave (4 waves), 64 granules each
    static_assert(GR % 256 == 0, "tile must split into whole wave instructions");
    unsigned voff[NI];
    __device__ __forceinline__ void init(int ld)
    {
        const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int p = (wave * NI + i) * 64 + lane;
            if (KM) {
                const int row = p >> 2, kq = (p & 3) ^ ((row >> 2) & 3);
                voff[i] = (unsigned)(row * ld + kq * 4);
            } else {
                const int k = p / (BT / 4), mq = (p % (BT / 4)) ^ (((k >> 2) & 1) << 3);
                voff[i] = (unsigned)(k * ld + mq * 4);
            }
        }
    }
    // base: workgroup-uniform address of the tile's first element; image: LDS address of the tile image
    __device__ __forceinline__ void issue(const float* __restrict__ base, float* image) const
    {
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#pragma unroll
        for (int i = 0; i < NI; i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff[i]),
                                             (__attribute__((address_space(3))) void*)(image + (wave * NI + i) * 256), 16, 0, 0);
        }
    }
    // fragment of the lane for the four MFMA steps of k-group g: row = tile row of the lane, lk = lane half
    static __device__ __forceinline__ f32x4 frag(const float* image, int row, int g, int lk)
    {
        if (KM) return *reinterpret_cast<const f32x4*>(image + ((row << 2) + ((2 * g + lk) ^ ((row >> 2) & 3))) * 4);
        const float* q = image + (8 * g + 4 * lk) * BT + (row ^ (lk << 5));
        return f32x4{q[0], q[BT], q[2 * BT], q[3 * BT]};
    }
};

#ifndef SPH3D_GEMM_DMA
#define SPH3D_GEMM_DMA 1
#endif
#ifndef SPH3D_GEMM_NBUF
#define SPH3D_GEMM_NBUF 2     // tile images of the LDS-DMA pipeline; 3 (tiles below 128 x 128): measured round 6, per shape and in the step: no change (profiles/r06_exp_gemm_nbuf.log)
#endif
#ifndef SPH3D_GEMM_SPLIT
#define SPH3D_GEMM_SPLIT 1    // whole-tile products on the bf16 matrix pipe with three-way split operands (gemm_split_mfma); 0: fp32 MFMA
#endif
#ifndef SPH3D_GEMM_WGS
#define SPH3D_GEMM_WGS 5      // workgroups per CU the unguarded BK = 16 kernels are built for
#endif
constexpr int cmax_i(int a, int b) { return a > b ? a : b; }

// epilogue shared by the fp32-MFMA kernel and the split-bf16 kernel (same C/D register layout: it is dtype-independent)
template <int BMT, int BN, int TM, int TN, bool SPLITK, bool GUARD, bool STATS, bool EPX, int LDSF>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TM][TN], float* lds, int M, int N, float* __restrict__ Cmat, int ldc,
                                              const float* __restrict__ bias, int act, float* __restrict__ stats, int tm, int ksplit,
                                              int m0, int n0, int wave, int lane, int wm, int wn, int li, int lk)
{
    constexpr int WN = BN / 2;
    // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    float* Cout = SPLITK ? Cmat + (size_t)ksplit * ((size_t)M * ldc) : Cmat;
    if (!GUARD) {
        // whole tiles: stage each wave's 32 x WN slab through its own LDS region (the operand buffers are free after the
        // loop's last barrier) and write it back as float4 rows: WN/4 lanes cover one row, 16 B per lane, instead of
        // sixteen 4-byte stores per MFMA tile (the scalar epilogue was store-issue-bound: ~25 % of the kernel)
        // row stride of the staging slab (floats): padded by 4, or — LDS-DMA kernels, whose operand buffers are exactly 32 KB at
        // 128 x 128: FIVE workgroups per CU — unpadded with the column XOR-ed by 32 * (bit 2 of the row) = the lane half lk, so
        // that the two lane halves of a store (rows 4 apart) and the four rows of a float4 read still use distinct banks
        constexpr int EP = EPX ? WN : WN + 4;
        static_assert(4 * 32 * EP <= LDSF, "epilogue staging must fit the operand LDS");
        float* stage = lds + wave * (32 * EP);
        constexpr int LPR = WN / 4;                          // lanes per row
        constexpr int RPI = 64 / LPR;                        // rows per store instruction
        if constexpr (STATS) {
            // a lane's 16 x TM values of column block j all sit in ONE column (col = lane & 31): sum, fold the two lane halves,
            // one store per (row half of the tile, column); fixed order, no atomics
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const float bv = bias != nullptr ? bias[n0 + wn + j * 32 + li] : 0.f;
                float sz = 0.f, sq = 0.f;
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const float y = acc[i][j][e] + bv;
                        const float z = y > 0.f ? y : __expf(y) - 1.f;         // == norm.hip: elu1
                        sz += z;
                        sq = fmaf(z, z, sq);
                    }
                sz += __shfl_xor(sz, 32);
                sq += __shfl_xor(sq, 32);
                if (lk == 0) {
                    float* sp = stats + (size_t)(tm * 2 + (wave >> 1)) * 2 * N + n0 + wn + j * 32 + li;
                    sp[0] = sz;
                    sp[N] = sq;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int col = wn + j * 32 + li;
                const float bv = (!SPLITK && bias != nullptr) ? bias[n0 + col] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    float v = acc[i][j][e] + bv;
                    if (!SPLITK && act == 1) v = v > 0.f ? v : expm1f(v);
                    stage[((e & 3) + 8 * (e >> 2) + 4 * lk) * EP + ((j * 32 + li) ^ (EPX ? (lk << 5) : 0))] = v;
                }
            }
            // same wave wrote and reads: LDS ops of one wave complete in order, no barrier needed
#pragma unroll
            for (int it = 0; it < 32 / RPI; it++) {
                const int r = it * RPI + lane / LPR;
                const int c4 = (lane % LPR) * 4;
                const float4 v = *reinterpret_cast<const float4*>(&stage[r * EP + (c4 ^ (EPX ? (((r >> 2) & 1) << 5) : 0))]);
                *reinterpret_cast<float4*>(&Cout[(size_t)(m0 + wm + i * 32 + r) * ldc + n0 + wn + c4]) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = n0 + wn + j * 32 + li;
            const float bv = (!SPLITK && bias != nullptr && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (row < M && col < N) {
                    float v = acc[i][j][e] + bv;
                    if (!SPLITK && act == 1) v = v > 0.f ? v : expm1f(v);     // ELU (alpha = 1)
                    Cout[(size_t)row * ldc + col] = v;
                }
            }
        }
    }
}

// k assignment inside a group of 8 consecutive k: MFMA step s (0..3) multiplies k = 8g + s (lanes 0..31) with
// k = 8g + 4 + s (lanes 32..63).  Any pairing is legal (the sum over k is what matters, A and B use the same one);
// this one makes a lane's operands for the 4 steps of a group ONE float4 at lds[row][8g + 4*(lane>>5)].
//
// C[M,N] = A * B for one BMT x BN tile and the k range [k_begin, k_end).
// A(m,k): AK ? A[m*lda + k] : A[k*lda + m].   B(k,n): BKM ? B[n*ldb + k] : B[k*ldb + n].
// 128x128 / BK = 16 unguarded tiles: 124 VGPRs and 40 KB of LDS -> FOUR workgroups per CU, so the common grids of
// 1024 / 2048 tiles have no partial last wave of workgroups (3 per CU left a quarter of the run at 1/3 occupancy).
// STATS (forward product of a layer whose tail is ELU -> batch norm, utils/sph3gcn_util.py:152-161): the epilogue also emits,
// per half tile of rows and per column, sum z and sum z*z with z = elu(y) — the partial sums the fused ELU+BN op's statistics
// pass would otherwise produce by reading Y once more (norm.hip: norm_reduce_kernel, same [block][2][C] layout).
template <bool AK, bool BKM, int BMT, int BN, int BK, bool SPLITK, bool GUARD, bool STATS = false>
__global__ __launch_bounds__(256, (BK == 16 && !GUARD) ? (SPH3D_GEMM_DMA != 0 && SPH3D_GEMM_WGS > 4 ? SPH3D_GEMM_WGS : 4) : 2) void gemm_f32_mfma(int M, int N, int Kd, const float* __restrict__ A, int lda,
                                                     const float* __restrict__ B, int ldb, float* __restrict__ Cmat,
                                                     int ldc, const float* __restrict__ bias, int act, int kchunk,
                                                     float* __restrict__ stats = nullptr, int nsplit = 1)
{
    using SA = Stage<AK, BMT, BK>;
    using SB = Stage<BKM, BN, BK>;
    constexpr int LDK = BK + 4;
    constexpr bool DMA = SPH3D_GEMM_DMA != 0 && !GUARD && BK == 16;       // LDS-DMA staging (see Dma)
    constexpr int WM = BMT / 2, WN = BN / 2;     // wave sub-tile
    // LDS-DMA staging depth: tile t+2 is in flight while tile t is multiplied wherever three tile images still leave the
    // occupancy alone (every tile but 128 x 128); with two images a k-tile iteration cannot be shorter than one memory round
    // trip (the loads issued at its top are waited for at its bottom): SPH3D_GEMM_NBUF
    constexpr int NBUF = DMA ? ((SPH3D_GEMM_NBUF >= 3 && (BMT + BN) <= 192) ? 3 : 2) : 2;
    constexpr int LDSF = DMA ? cmax_i(NBUF * (BMT + BN) * BK, 4 * 32 * (WN == 64 ? WN : WN + 4)) : 2 * (SA::LDS_FLOATS + SB::LDS_FLOATS);
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    constexpr int TM = WM / 32, TN = WN / 32;    // MFMA tiles per wave

    // XCD-aware workgroup -> tile map (workgroup b runs on XCD b % 8, each XCD has its own L2): the workgroups that read the SAME
    // operand rows go to the same XCD, back to back.  Products: the tiles_n column tiles of a tile row (they share the A rows;
    // row-major tile ids put them on tiles_n different XCDs and every XCD fetched the rows from HBM itself).  Weight gradient:
    // the tiles of one k-split (they share the split's X and dY rows: round-3 counters, 268 MB fetched for 151 MB of operands
    // at (32768, 1024 -> 128), the kernel within 15 % of the HBM rate).  Ids are dealt in groups of 8 rows (splits) x all their
    // tiles; the grid is padded to whole groups and the surplus workgroups leave at once.
    const int tiles_n = (N + BN - 1) / BN;
    const int tiles_m = (M + BMT - 1) / BMT;
    const int wg = (int)blockIdx.x;
    int tm, tn, ksplit = 0;
    if (SPLITK) {
        const int T = tiles_m * tiles_n;
        const int g = wg / (8 * T), r = wg - g * 8 * T;
        ksplit = g * 8 + (r & 7);
        if (ksplit >= nsplit) return;
        const int tile = r >> 3;
        tm = tile / tiles_n;
        tn = tile - tm * tiles_n;
    } else {
        const int g = wg / (8 * tiles_n), r = wg - g * 8 * tiles_n;
        tm = g * 8 + (r & 7);
        tn = r >> 3;
        if (tm >= tiles_m) return;
    }
    const int m0 = tm * BMT, n0 = tn * BN;
    const int k_begin = SPLITK ? ksplit * kchunk : 0;
    const int k_end = SPLITK ? ((k_begin + kchunk) < Kd ? (k_begin + kchunk) : Kd) : Kd;

    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int li = lane & 31, lk = lane >> 5;

    const bool a_vec = (lda % 4 == 0) && ((reinterpret_cast<size_t>(A) & 15) == 0);
    const bool b_vec = (ldb % 4 == 0) && ((reinterpret_cast<size_t>(B) & 15) == 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    if constexpr (DMA) {
        constexpr int ABUF = BMT * BK, BUFD = (BMT + BN) * BK;          // buffer b: A image at lds + b*BUFD, B image after it
        Dma<AK, BMT> da;
        Dma<BKM, BN> db;
        da.init(lda);
        db.init(ldb);
        const float* abase = AK ? A + (size_t)m0 * lda + k_begin : A + (size_t)k_begin * lda + m0;
        const float* bbase = BKM ? B + (size_t)n0 * ldb + k_begin : B + (size_t)k_begin * ldb + n0;
        const size_t astep = AK ? (size_t)BK : (size_t)BK * lda;
        const size_t bstep = BKM ? (size_t)BK : (size_t)BK * ldb;
        da.issue(abase, lds);
        db.issue(bbase, lds + ABUF);
        if constexpr (NBUF == 3) {
            constexpr int NL = Dma<AK, BMT>::NI + Dma<BKM, BN>::NI;          // loads a wave has in flight per tile
            const bool two = (k_begin + BK) < k_end;
            if (two) {
                abase += astep;
                bbase += bstep;
                da.issue(abase, lds + BUFD);
                db.issue(bbase, lds + BUFD + ABUF);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");      // tile 0 has landed, tile 1 may still fly
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            int buf = 0;
            for (int k0 = k_begin; k0 < k_end; k0 += BK) {
                const bool more = (k0 + BK) < k_end, more2 = (k0 + 2 * BK) < k_end;
                if (more2) {                  // tile t+2 -> the image tile t-1 was read from (every wave is past that iteration's barrier)
                    const int nb = buf >= 1 ? buf - 1 : 2;
                    abase += astep;
                    bbase += bstep;
                    da.issue(abase, lds + nb * BUFD);
                    db.issue(bbase, lds + nb * BUFD + ABUF);
                }
                const float* ca = lds + buf * BUFD;
                const float* cb = ca + ABUF;
                if (kSetPrio) __builtin_amdgcn_s_setprio(2);
#pragma unroll
                for (int g = 0; g < BK / 8; g++) {
                    f32x4 af[TM], bf[TN];
#pragma unroll
                    for (int i = 0; i < TM; i++) af[i] = Dma<AK, BMT>::frag(ca, wm + i * 32 + li, g, lk);
#pragma unroll
                    for (int j = 0; j < TN; j++) bf[j] = Dma<BKM, BN>::frag(cb, wn + j * 32 + li, g, lk);
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) {
#pragma unroll
                        for (int i = 0; i < TM; i++)
#pragma unroll
                            for (int j = 0; j < TN; j++) {
                                const float av = s4 == 0 ? af[i].x : (s4 == 1 ? af[i].y : (s4 == 2 ? af[i].z : af[i].w));
                                const float bv = s4 == 0 ? bf[j].x : (s4 == 1 ? bf[j].y : (s4 == 2 ? bf[j].z : bf[j].w));
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                            }
                    }
                }
                if (kSetPrio) __builtin_amdgcn_s_setprio(0);
                if (more) {                   // this wave's share of tile t+1 has landed (tile t+2 may still fly)
                    if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();              // ONE barrier per k-tile
                buf = buf == 2 ? 0 : buf + 1;
            }
        } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            const bool more = (k0 + BK) < k_end;
            if (more) {                   // tile t+1 -> the other buffer (nobody reads it during this iteration), no registers
                abase += astep;
                bbase += bstep;
                da.issue(abase, lds + (buf ^ 1) * BUFD);
                db.issue(bbase, lds + (buf ^ 1) * BUFD + ABUF);
            }
            const float* ca = lds + buf * BUFD;
            const float* cb = ca + ABUF;
            if (kSetPrio) __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int g = 0; g < BK / 8; g++) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) af[i] = Dma<AK, BMT>::frag(ca, wm + i * 32 + li, g, lk);
#pragma unroll
                for (int j = 0; j < TN; j++) bf[j] = Dma<BKM, BN>::frag(cb, wn + j * 32 + li, g, lk);
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++) {
                            const float av = s4 == 0 ? af[i].x : (s4 == 1 ? af[i].y : (s4 == 2 ? af[i].z : af[i].w));
                            const float bv = s4 == 0 ? bf[j].x : (s4 == 1 ? bf[j].y : (s4 == 2 ? bf[j].z : bf[j].w));
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        }
                }
            }
            if (kSetPrio) __builtin_amdgcn_s_setprio(0);
            if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of tile t+1 has landed
            __syncthreads();              // ONE barrier per k-tile
            buf ^= 1;
        }
        }
    } else {
    SA sa;
    SB sb;
    constexpr int BUF = SA::LDS_FLOATS + SB::LDS_FLOATS;     // buffer b: A image at lds + b*BUF, B image after it
    // unguarded tiles: uniform tile addresses + one loop-invariant lane offset per operand (Stage::load_fast)
    const float* abase = AK ? A + (size_t)m0 * lda + k_begin : A + (size_t)k_begin * lda + m0;
    const float* bbase = BKM ? B + (size_t)n0 * ldb + k_begin : B + (size_t)k_begin * ldb + n0;
    const size_t astep = AK ? (size_t)BK : (size_t)BK * lda;
    const size_t bstep = BKM ? (size_t)BK : (size_t)BK * ldb;
    const unsigned aoff = sa.lane_offset(lda), boff = sb.lane_offset(ldb);
    if (GUARD) {
        sa.template load<GUARD>(A, lda, m0, k_begin, M, k_end, a_vec);
        sb.template load<GUARD>(B, ldb, n0, k_begin, N, k_end, b_vec);
    } else {
        sa.load_fast(abase, lda, aoff);
        sb.load_fast(bbase, ldb, boff);
    }
    sa.store(lds);
    sb.store(lds + SA::LDS_FLOATS);
    __syncthreads();

    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = (k0 + BK) < k_end;
        if (more && !(SPH3D_GEMM_EXP & 1)) {                       // global -> registers for tile t+1 while tile t is multiplied
            if (GUARD) {
                sa.template load<GUARD>(A, lda, m0, k0 + BK, M, k_end, a_vec);
                sb.template load<GUARD>(B, ldb, n0, k0 + BK, N, k_end, b_vec);
            } else {
                abase += astep;
                bbase += bstep;
                sa.load_fast(abase, lda, aoff);
                sb.load_fast(bbase, ldb, boff);
            }
        }
        const float* ca = lds + buf * BUF;
        const float* cb = ca + SA::LDS_FLOATS;
        if (kSetPrio) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int g = 0; g < BK / 8; g++) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) af[i] = *reinterpret_cast<const f32x4*>(ca + (wm + i * 32 + li) * LDK + g * 8 + lk * 4);
#pragma unroll
            for (int j = 0; j < TN; j++) bf[j] = *reinterpret_cast<const f32x4*>(cb + (wn + j * 32 + li) * LDK + g * 8 + lk * 4);
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        const float av = s4 == 0 ? af[i].x : (s4 == 1 ? af[i].y : (s4 == 2 ? af[i].z : af[i].w));
                        const float bv = s4 == 0 ? bf[j].x : (s4 == 1 ? bf[j].y : (s4 == 2 ? bf[j].z : bf[j].w));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
            }
        }
        if (kSetPrio) __builtin_amdgcn_s_setprio(0);
        if (more && !(SPH3D_GEMM_EXP & 2)) {                       // the other buffer: nobody reads it during this iteration
            sa.store(lds + (buf ^ 1) * BUF);
            sb.store(lds + (buf ^ 1) * BUF + SA::LDS_FLOATS);
        }
        if (!(SPH3D_GEMM_EXP & 4)) __syncthreads();                  // ONE barrier per k-tile
        buf ^= 1;
    }

    }

    gemm_epilogue<BMT, BN, TM, TN, SPLITK, GUARD, STATS, DMA && WN == 64, LDSF>(acc, lds, M, N, Cmat, ldc, bias, act, stats, tm, ksplit,
                                                                                  m0, n0, wave, lane, wm, wn, li, lk);
}

// ------------------------------------------------------------------------------------------------------------------------
// Split-bf16 products (round 6): the same three products on the BF16 matrix pipe — 16x the fp32 MFMA rate — at fp32 accuracy.
// Every fp32 operand x is cut into three bf16 pieces by TRUNCATION, x = h + m + l EXACTLY (8 + 8 + 8 significant bits: h = the
// top 16 bits of x, m = the top 16 bits of x - h, l = x - h - m, each subtraction exact), and
//     x * y  ~=  h h' + (h m' + m h') + (m m' + h l' + l h')
// six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block (192 matrix-pipe cycles against 512 for eight v_mfma_f32_32x32x2_f32),
// products of bf16 pairs exact in the fp32 accumulator.  Dropped: m l' + l m' + l l' <= 2^-23 |x y| per product — the size of ONE
// fp32 rounding, where the fp32 kernel's k-ordered fmaf chain commits one rounding per term; measured against float64 the two
// kernels' errors are the same size (tests/test_gpu_gemm_split.py).  Non-finite operands: Inf - Inf makes the lower pieces NaN, so
// an output the fp32 kernel makes +-Inf may be NaN (never finite).
// Pipeline: global -> registers (Stage's loads: tile t+1 in flight during the MFMAs of tile t) -> split in registers -> LDS piece
// planes [piece][k-group of 8][row][8 bf16]: a lane's MFMA operand (row = lane % 32, k-group = lane / 32) is ONE ds_read_b128, the
// 32 lanes of a half wave read 512 contiguous bytes.  One barrier per k-tile, two plane images.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

struct Split3 { unsigned h, m, l; };          // each: the bf16 piece of a in the low half, of b in the high half
__device__ __forceinline__ Split3 split_pair(float a, float b)
{
    Split3 r;
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    r.h = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                       // (ua >> 16) | (ub & 0xffff0000)
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    r.m = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
    r.l = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
    return r;
}

// Operand tile of the split kernel: global -> registers -> three bf16 piece planes in LDS.
// KMAJ (memory [row][k]): a unit = one float4 (4 consecutive k of a row) -> one 8-byte store per piece.
// !KMAJ (memory [k][row]): a unit = 2 (k) x 4 (rows): two float4 loads along the rows (the lanes of a k cover contiguous bytes) ->
// per row the k pair is ONE packed dword per piece (no transposing shuffles: split_pair packs its two arguments).  Always pairs —
// with 4 x 4 blocks half of the threads have no unit at 64-row tiles and the busy half carries twice the conversion work.
template <bool KMAJ, int BT, int BK>
struct SplitStage {
    static constexpr int KG = BK / 8;                       // k-groups of 8
    static constexpr int PLANE = KG * BT * 16;              // bytes of one piece's plane
    static constexpr int BYTES = 3 * PLANE;
    // 16-byte slot of (k-group g, row): the row index XOR-ed with a few bits of (g, row bit 4), constant over the 16 consecutive rows
    // a fragment read covers (it stays a permutation of one 256-byte bank group: conflict-free), different for the writers that
    // would otherwise meet in a bank — the two k-groups of a k-contiguous unit's store (same row) and, for the row-contiguous
    // operands, the rows 16 apart and the two k-groups of one 4-byte store (4-way conflicts before:
    // profiles/r06_pmc_gemm_split_nn_vs_nt_6144x2048x256.log, 14 M against 2.4 M conflict cycles)
    __device__ __forceinline__ static int slot(int g, int row)
    {
        return (g * BT + row) ^ (((row >> 4) & 1) | ((g & 1) << 1) | ((g & 1) << 3) | (((g >> 1) & 1) << 2));
    }
    static constexpr int KUN = KMAJ ? BK / 4 : BK / 2;      // units along k
    static constexpr int UNITS = KMAJ ? BT * (BK / 4) : (BT / 4) * (BK / 2);
    static constexpr int NCH = (UNITS + 255) / 256;
    static constexpr int NREG = KMAJ ? NCH : 2 * NCH;
    f32x4 r[NREG];
    __device__ __forceinline__ unsigned lane_offset(int ld) const
    {
        const int u = (int)threadIdx.x;
        if (KMAJ) return (unsigned)((u / KUN) * ld + (u % KUN) * 4);
        return (unsigned)((u % KUN) * 2 * ld + (u / KUN) * 4);
    }
    // `base`: workgroup-uniform address of the tile's first element (see Stage::load_fast)
    __device__ __forceinline__ void load(const float* __restrict__ base, int ld, unsigned voff)
    {
        if (UNITS < 256 && (int)threadIdx.x >= UNITS) return;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            if (KMAJ) {
                const float* b = base + (size_t)(i * (256 / KUN)) * ld;                     // uniform
                r[i] = *reinterpret_cast<const f32x4*>(b + voff);
            } else {
                const float* b = base + i * (256 / KUN) * 4;                                 // uniform
                r[2 * i] = *reinterpret_cast<const f32x4*>(b + voff);
                r[2 * i + 1] = *reinterpret_cast<const f32x4*>(b + ld + voff);
            }
        }
    }
    // ragged tiles: element-wise bounds, zeros outside (rows = extent of the tile's row dimension, kdim = end of the k range)
    __device__ __forceinline__ void load_guard(const float* __restrict__ p, int ld, int row0, int k0, int rows, int kdim, bool vec_ok)
    {
        if (UNITS < 256 && (int)threadIdx.x >= UNITS) return;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int u = (int)threadIdx.x + i * 256;
            if (KMAJ) {
                r[i] = as_v4(load4_guard(p, ld, row0 + u / KUN, k0 + (u % KUN) * 4, rows, kdim, vec_ok));
            } else {
                const int kq = u % KUN, rq = u / KUN;
                r[2 * i] = as_v4(load4_guard(p, ld, k0 + kq * 2, row0 + rq * 4, kdim, rows, vec_ok));
                r[2 * i + 1] = as_v4(load4_guard(p, ld, k0 + kq * 2 + 1, row0 + rq * 4, kdim, rows, vec_ok));
            }
        }
    }
    __device__ __forceinline__ void store(char* img) const
    {
        if (UNITS < 256 && (int)threadIdx.x >= UNITS) return;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int u = (int)threadIdx.x + i * 256;
            if (KMAJ) {
                const int row = u / KUN, kq = u % KUN;                  // k = 4 kq .. 4 kq + 3 of `row`
                const f32x4 v = r[i];
                const Split3 p0 = split_pair(v.x, v.y), p1 = split_pair(v.z, v.w);
                char* q = img + (slot(kq >> 1, row) << 4) + ((kq & 1) << 3);
                *reinterpret_cast<u32x2v*>(q) = u32x2v{p0.h, p1.h};
                *reinterpret_cast<u32x2v*>(q + PLANE) = u32x2v{p0.m, p1.m};
                *reinterpret_cast<u32x2v*>(q + 2 * PLANE) = u32x2v{p0.l, p1.l};
            } else {
                const int kq = u % KUN, rq = u / KUN;                   // k = 2 kq, 2 kq + 1 of rows 4 rq .. 4 rq + 3
                const int k0 = kq * 2;
                const f32x4 a = r[2 * i], b = r[2 * i + 1];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const Split3 sp = split_pair(a[j], b[j]);
                    char* q = img + (slot(k0 >> 3, rq * 4 + j) << 4) + ((k0 & 7) << 1);
                    *reinterpret_cast<unsigned*>(q) = sp.h;
                    *reinterpret_cast<unsigned*>(q + PLANE) = sp.m;
                    *reinterpret_cast<unsigned*>(q + 2 * PLANE) = sp.l;
                }
            }
        }
    }
    // piece `pc` of the lane's operand: tile row `row`, k-group g
    __device__ __forceinline__ static bf16x8 frag(const char* img, int pc, int row, int g)
    {
        return *reinterpret_cast<const bf16x8*>(img + pc * PLANE + (slot(g, row) << 4));
    }
};

#ifndef SPH3D_SPLIT_DB
#define SPH3D_SPLIT_DB 0      // 1: two plane images (one barrier per k-tile, 3 workgroups per CU at 128 x 128); 0: one image, two barriers, 4-5 per CU
#endif
__device__ int g_xs_fail = 0;      // exchange launches that gave up waiting (never expected)

// XS (in-kernel split-K exchange, for products whose tile grid is too small to fill the chip and whose k loop is long): the grid holds
// `nsplit` workgroups per tile, each multiplying `kchunk` of k.  Splits 1 .. nsplit-1 (the LOWER workgroup ids: dispatched first,
// they never wait) store their accumulators — in register layout, 16 contiguous bytes per lane — to their slab of `xslab`, fence,
// and count themselves in at xflag[tile]; split 0 (dispatched last) waits for the count, adds the slabs in split order
// (deterministic), clears the counter and runs the ordinary epilogue (bias / ELU / BN statistics / float4 rows).  No slab-sum
// launch, no float atomics.
template <bool AK, bool BKM, int BMT, int BN, int BK, bool SPLITK, bool STATS, bool GUARD = false, bool XS = false>
__global__ __launch_bounds__(256, SPH3D_SPLIT_DB ? 2 : 4) void gemm_split_mfma(int M, int N, int Kd, const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb, float* __restrict__ Cmat, int ldc,
                                                          const float* __restrict__ bias, int act, int kchunk,
                                                          float* __restrict__ stats = nullptr, int nsplit = 1,
                                                          float* __restrict__ xslab = nullptr, int* __restrict__ xflag = nullptr)
{
    static_assert(!(XS && SPLITK) && !(XS && GUARD), "the exchange variant is an unguarded, non-slab kernel");
    using PA = SplitStage<AK, BMT, BK>;
    using PB = SplitStage<BKM, BN, BK>;
    constexpr int WM = BMT / 2, WN = BN / 2;     // wave sub-tile
    constexpr int TM = WM / 32, TN = WN / 32;    // MFMA tiles per wave
    constexpr int BUFB = PA::BYTES + PB::BYTES;  // bytes of one plane image (A pieces, then B pieces)
    constexpr bool DB = SPH3D_SPLIT_DB != 0;
    constexpr bool EPX = WN == 64;               // un-padded, XOR-swizzled epilogue slab (32 KB at 128 x 128)
    constexpr int LDSF = cmax_i((DB ? 2 : 1) * BUFB / 4, 4 * 32 * (EPX ? WN : WN + 4));
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    char* img = reinterpret_cast<char*>(lds);

    // XCD-aware workgroup -> tile map: as gemm_f32_mfma
    const int tiles_n = (N + BN - 1) / BN;
    const int tiles_m = (M + BMT - 1) / BMT;
    const int wg = (int)blockIdx.x;
    int tm, tn, ksplit = 0;
    if (XS) {
        const int Tg = ((tiles_m + 7) / 8) * 8 * tiles_n;      // == gemm_grid(tiles_m, tiles_n)
        const int rev = wg / Tg, wl = wg - rev * Tg;
        ksplit = nsplit - 1 - rev;
        const int g = wl / (8 * tiles_n), r = wl - g * 8 * tiles_n;
        tm = g * 8 + (r & 7);
        tn = r >> 3;
        if (tm >= tiles_m) return;
    } else if (SPLITK) {
        const int T = tiles_m * tiles_n;
        const int g = wg / (8 * T), r = wg - g * 8 * T;
        ksplit = g * 8 + (r & 7);
        if (ksplit >= nsplit) return;
        const int tile = r >> 3;
        tm = tile / tiles_n;
        tn = tile - tm * tiles_n;
    } else {
        const int g = wg / (8 * tiles_n), r = wg - g * 8 * tiles_n;
        tm = g * 8 + (r & 7);
        tn = r >> 3;
        if (tm >= tiles_m) return;
    }
    const int m0 = tm * BMT, n0 = tn * BN;
    const int k_begin = (SPLITK || XS) ? ksplit * kchunk : 0;
    const int k_end = (SPLITK || XS) ? ((k_begin + kchunk) < Kd ? (k_begin + kchunk) : Kd) : Kd;
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int li = lane & 31, lk = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    PA sa;
    PB sb;
    const float* abase = AK ? A + (size_t)m0 * lda + k_begin : A + (size_t)k_begin * lda + m0;
    const float* bbase = BKM ? B + (size_t)n0 * ldb + k_begin : B + (size_t)k_begin * ldb + n0;
    const size_t astep = AK ? (size_t)BK : (size_t)BK * lda;
    const size_t bstep = BKM ? (size_t)BK : (size_t)BK * ldb;
    const unsigned aoff = sa.lane_offset(lda), boff = sb.lane_offset(ldb);
    const bool a_vec = (lda % 4 == 0) && ((reinterpret_cast<size_t>(A) & 15) == 0);
    const bool b_vec = (ldb % 4 == 0) && ((reinterpret_cast<size_t>(B) & 15) == 0);
    if (GUARD) {
        sa.load_guard(A, lda, m0, k_begin, M, k_end, a_vec);
        sb.load_guard(B, ldb, n0, k_begin, N, k_end, b_vec);
    } else {
        sa.load(abase, lda, aoff);
        sb.load(bbase, ldb, boff);
    }
    sa.store(img);
    sb.store(img + PA::BYTES);
    __syncthreads();

    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = (k0 + BK) < k_end;
        if (more) {                       // global -> registers for tile t+1 while tile t is multiplied
            if (GUARD) {
                sa.load_guard(A, lda, m0, k0 + BK, M, k_end, a_vec);
                sb.load_guard(B, ldb, n0, k0 + BK, N, k_end, b_vec);
            } else {
                abase += astep;
                bbase += bstep;
                sa.load(abase, lda, aoff);
                sb.load(bbase, ldb, boff);
            }
        }
        const char* ca = img + (DB ? buf * BUFB : 0);
        const char* cb = ca + PA::BYTES;
        if (kSetPrio) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int s16 = 0; s16 < BK / 16; s16++) {
            const int g = 2 * s16 + lk;
            bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int pc = 0; pc < 3; pc++) af[i][pc] = PA::frag(ca, pc, wm + i * 32 + li, g);
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int pc = 0; pc < 3; pc++) bf[j][pc] = PB::frag(cb, pc, wn + j * 32 + li, g);
            // small terms first; consecutive MFMAs write different accumulators where there is more than one
#pragma unroll
            for (int t = 0; t < 6; t++) {
                const int pa = t == 0 ? 0 : (t == 1 ? 2 : (t == 2 ? 1 : (t == 3 ? 0 : (t == 4 ? 1 : 0))));      // A piece: h l m h m h
                const int pb = t == 0 ? 2 : (t == 1 ? 0 : (t == 2 ? 1 : (t == 3 ? 1 : (t == 4 ? 0 : 0))));      // B piece: l h m m h h
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0);
            }
        }
        if (kSetPrio) __builtin_amdgcn_s_setprio(0);
        if (DB) {
            if (more) {                   // the other image: nobody reads it during this iteration
                sa.store(img + (buf ^ 1) * BUFB);
                sb.store(img + (buf ^ 1) * BUFB + PA::BYTES);
            }
            __syncthreads();              // ONE barrier per k-tile
            buf ^= 1;
        } else {
            __syncthreads();              // every wave has its fragments of tile t in registers / the matrix pipe
            if (more) {
                sa.store(img);
                sb.store(img + PA::BYTES);
            }
            __syncthreads();
        }
    }
    if (XS) {
        // The slabs and the counters are exchanged between workgroups on different XCDs (private L2s).  Every access to them is an
        // agent-scope relaxed atomic (sc1: written through / read past the non-coherent cache lines), ordered by hand: the
        // producer waits for its stores to be acknowledged (vmcnt) before it counts itself in; the consumer's slab loads are
        // issued after it has seen the full count.  A release / acquire FENCE would be correct too and is 4x slower than not
        // splitting at all: it writes back / invalidates the XCD's whole L2 (`buffer_wbl2` / `buffer_inv sc1`), once per
        // workgroup, while the L2 is full of the dirty lines of the layer's operands (profiles/r06_exp_gemm_xs.log).
        constexpr int NW = TM * TN * 16;                         // dwords per lane
        const int tile = tm * tiles_n + tn, ntile = tiles_m * tiles_n;
        if (ksplit > 0) {
            float* sl = xslab + ((size_t)(ksplit - 1) * ntile + tile) * (NW * 256) + threadIdx.x;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        __hip_atomic_store(&sl[((i * TN + j) * 16 + e) * 256], acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&xflag[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (threadIdx.x == 0) {
            // bounded: the producers are dispatched first and never wait, so the count always arrives; a launch that would still
            // spin after ~1 s (a fault elsewhere) gives up, counts the failure (sph3d_pointwise_gemm_exchange_failures) and ends
            int polls = 0;
            while (__hip_atomic_load(&xflag[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nsplit - 1) {
                __builtin_amdgcn_s_sleep(4);
                if (++polls > (1 << 20)) {
                    atomicAdd(&g_xs_fail, 1);
                    break;
                }
            }
            __hip_atomic_store(&xflag[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the buffer's invariant: counters are zero between launches
        }
        __syncthreads();
        for (int ks = 1; ks < nsplit; ks++) {
            const float* sl = xslab + ((size_t)(ks - 1) * ntile + tile) * (NW * 256) + threadIdx.x;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        acc[i][j][e] += __hip_atomic_load(&sl[((i * TN + j) * 16 + e) * 256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    gemm_epilogue<BMT, BN, TM, TN, SPLITK, GUARD, STATS, EPX, LDSF>(acc, lds, M, N, Cmat, ldc, bias, act, stats, tm, ksplit, m0, n0, wave,
                                                                      lane, wm, wn, li, lk);
}

__global__ __launch_bounds__(256) void gemm_reduce_splits(int nsplit, int total, const float* __restrict__ partial,
                                                          float* __restrict__ out)
{
    // 256 threads = 32 outputs x 8 split-lanes, fixed summation order (deterministic); the split count is small (<= 32):
    // 32 lanes per output measured slower (15.8 vs 10.7 us per call)
    __shared__ float red[8][32];
    const int cx = (int)threadIdx.x & 31, py = (int)threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + cx;
    float s = 0.f;
    if (j < total) {
        // eight slabs per trip, loads issued together (one per trip = up to 32 dependent L2 round trips: 8.5 us per launch)
        for (int p0 = py; p0 < nsplit; p0 += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int p = p0 + u * 8;
                v[u] = partial[(size_t)(p < nsplit ? p : py) * total + j];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (p0 + u * 8 < nsplit) s += v[u];
        }
    }
    red[py][cx] = s;
    __syncthreads();
    if (py == 0 && j < total) {
        for (int k = 1; k < 8; k++) s += red[k][cx];
        out[j] = s;
    }
}

// padded 1-D grid of the XCD-aware tile map (see the kernel): groups of 8 tile rows (k-splits) x all their tiles
static unsigned gemm_grid(long long tiles_m, long long tiles_n, long long nsplit = 0)
{
    if (nsplit > 0) return (unsigned)(((nsplit + 7) / 8) * 8 * tiles_m * tiles_n);
    return (unsigned)(((tiles_m + 7) / 8) * 8 * tiles_n);
}

static bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// whole-tile products on the bf16 pipe with split operands (gemm_split_mfma) unless SPH3D_GEMM_SPLIT=0 (A/B runs; the exact
// fp32-MFMA kernels then); ragged shapes always take the fp32 kernels
static int g_split_mode = -1;          // -1: not decided yet (environment / build default)
static bool split_on()
{
    if (g_split_mode < 0) g_split_mode = getenv("SPH3D_GEMM_SPLIT") ? (atoi(getenv("SPH3D_GEMM_SPLIT")) != 0) : (SPH3D_GEMM_SPLIT != 0);
    return g_split_mode != 0;
}

// In-kernel split-K exchange (gemm_split_mfma<..., XS>): splits per tile for a whole-tile product that would run as T 64 x 64 tiles
// (grid Tg after XCD padding), 1 = do not split.  Up to ~1024 workgroups (4 per CU: all resident), at least 256 of k per split.
// SPH3D_GEMM_XS: 0 off, n = at most n splits (experiments)
static int xs_cap()
{
    static const int cap = getenv("SPH3D_GEMM_XS") ? atoi(getenv("SPH3D_GEMM_XS")) : 4;
    return cap;
}
static int xs_splits(long long Tg, int Kd, int bk, int cap)
{
    static const int max_wgs = getenv("SPH3D_GEMM_XS_WGS") ? atoi(getenv("SPH3D_GEMM_XS_WGS")) : 1024;      // (experiments)
    static const int max_tg = getenv("SPH3D_GEMM_XS_TG") ? atoi(getenv("SPH3D_GEMM_XS_TG")) : 384;
    int ns = (int)(max_wgs / (Tg > 0 ? Tg : 1));
    if (Tg > max_tg) ns = 1;      // (512 tiles split in two: 22 vs 20 us at (2048, 512 -> 1024))
    if (ns > cap) ns = cap;
    while (ns > 1 && (Kd % (bk * ns) != 0 || Kd / ns < 256)) ns--;
    return ns < 1 ? 1 : ns;
}
constexpr size_t kXsFlagBytes = 16384;      // 4096 arrival counters in front of the slabs
// -> the exchange buffer of the stream (counters zero), or nullptr (capture in progress and nothing allocated yet / out of memory /
// too many tiles): the caller then launches the ordinary kernel
static char* xs_buffer(hipStream_t st, long long ntile, int nsplit, size_t tile_floats)
{
    if (ntile > (long long)(kXsFlagBytes / sizeof(int))) return nullptr;
    return (char*)stream_scratch(st, kXsFlagBytes + sizeof(float) * (size_t)(nsplit - 1) * (size_t)ntile * tile_floats, 1);
}

template <bool AK, bool BKM, bool GUARD>
static void launch_gemm_tiles(int M, int N, int Kd, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                              const float* bias, int act, hipStream_t st)
{
    // tile choice: the largest tile that still gives every CU TWO workgroups (measured: with >= 256 tiles as the rule the
    // mid-size levels ran 50 TF, with >= 512 they run 80-90 TF: one partial wave of workgroups leaves half the CUs idle)
    auto ntiles = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    static const long long kMinTiles = getenv("SPH3D_GEMM_MINTILES") ? atoi(getenv("SPH3D_GEMM_MINTILES")) : 512;      // (experiments)
    // (BK = 32 at two workgroups per CU for the big grids: measured slower than BK = 16 at four, before and after the
    //  round-2 register fix: 0.101 vs 0.094 ms at (131072, 256 -> 128))
    if constexpr (GUARD) {
        // ragged shapes (the ModelNet plan's 35 / 67 / 131-channel layers, Cin = 3): the same kernels with element-wise bounds on the
        // loads and stores; not for the very narrow ones (K < 16 or N < 32: mostly zero padding on the matrix pipe)
        if (split_on() && Kd >= 16 && N >= 32) {
            if (N > 64 && ntiles(128, 128) >= kMinTiles)
                hipLaunchKernelGGL((gemm_split_mfma<AK, BKM, 128, 128, 16, false, false, true>), dim3(gemm_grid((M + 127) / 128, (N + 127) / 128)),
                                   dim3(256), 0, st, M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
            else if (!BKM && N <= 64 && ntiles(128, 64) >= kMinTiles)
                hipLaunchKernelGGL((gemm_split_mfma<AK, BKM, 128, 64, 16, false, false, true>), dim3(gemm_grid((M + 127) / 128, (N + 63) / 64)),
                                   dim3(256), 0, st, M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
            else
                hipLaunchKernelGGL((gemm_split_mfma<AK, BKM, 64, 64, 16, false, false, true>), dim3(gemm_grid((M + 63) / 64, (N + 63) / 64)), dim3(256),
                                   0, st, M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
            return;
        }
    }
    if constexpr (!GUARD) {
        if (split_on()) {
            // 64 x 64 tiles take two MFMA k-steps per barrier where K allows.  SPH3D_SPLIT_MINTILES: experiments
            static const long long kMinTiles = getenv("SPH3D_SPLIT_MINTILES") ? atoi(getenv("SPH3D_SPLIT_MINTILES")) : 512;
            if (N > 64 && ntiles(128, 128) >= kMinTiles)
                hipLaunchKernelGGL((gemm_split_mfma<AK, BKM, 128, 128, 16, false, false>), dim3(gemm_grid((M + 127) / 128, (N + 127) / 128)), dim3(256),
                                   0, st, M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
            else if (!BKM && N <= 64 && ntiles(128, 64) >= kMinTiles)
                hipLaunchKernelGGL((gemm_split_mfma<AK, BKM, 128, 64, 16, false, false>), dim3(gemm_grid((M + 127) / 128, (N + 63) / 64)), dim3(256), 0,
                                   st, M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
            else if (Kd % 32 == 0) {
                const long long Tg = gemm_grid(M / 64, N / 64);
                const int ns = xs_splits(Tg, Kd, 32, xs_cap());
                char* xb = ns > 1 ? xs_buffer(st, (long long)(M / 64) * (N / 64), ns, 64 * 64) : nullptr;
                if (xb != nullptr)
                    hipLaunchKernelGGL((gemm_split_mfma<AK, BKM, 64, 64, 32, false, false, false, true>), dim3((unsigned)(Tg * ns)), dim3(256), 0, st,
                                       M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, Kd / ns, nullptr, ns, (float*)(xb + kXsFlagBytes), (int*)xb);
                else
                    hipLaunchKernelGGL((gemm_split_mfma<AK, BKM, 64, 64, 32, false, false>), dim3(gemm_grid((M + 63) / 64, (N + 63) / 64)), dim3(256), 0,
                                       st, M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
            } else
                hipLaunchKernelGGL((gemm_split_mfma<AK, BKM, 64, 64, 16, false, false>), dim3(gemm_grid((M + 63) / 64, (N + 63) / 64)), dim3(256), 0,
                                   st, M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
            return;
        }
    }
    if (BKM && !(N > 64 && ntiles(128, 128) >= kMinTiles)) {
        // input-gradient product (W stored [n][k]) below 2 big tiles per CU: 64x64 tiles (0.071 vs 0.088 ms at
        // (6144, 2048 -> 256) ... ) -- with >= kMinTiles big tiles the 128x128 kernel wins since its prefetch registers stopped
        // going through scratch: 0.100 vs 0.106 ms at (131072, 256 -> 128), 0.089 vs 0.098 at (32768, 512 -> 256),
        // 0.062 vs 0.071 at (6144, 256 -> 2048)
        hipLaunchKernelGGL((gemm_f32_mfma<AK, BKM, 64, 64, BKS, false, GUARD>), dim3(gemm_grid((M + 63) / 64, (N + 63) / 64)), dim3(256), 0, st, M,
                           N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
    } else if (N > 64 && ntiles(128, 128) >= kMinTiles) {
        constexpr int BKX = BKS;
        hipLaunchKernelGGL((gemm_f32_mfma<AK, BKM, 128, 128, BKX, false, GUARD>), dim3(gemm_grid((M + 127) / 128, (N + 127) / 128)), dim3(256), 0, st,
                           M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
    } else if (N <= 64 && ntiles(128, 64) >= kMinTiles) {      // (wider outputs with fewer rows: 64x64, e.g. 0.092 vs 0.111 ms at (32768, 1024 -> 128))
        hipLaunchKernelGGL((gemm_f32_mfma<AK, BKM, 128, 64, BKS, false, GUARD>), dim3(gemm_grid((M + 127) / 128, (N + 63) / 64)), dim3(256), 0, st,
                           M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
    } else {
        hipLaunchKernelGGL((gemm_f32_mfma<AK, BKM, 64, 64, BKS, false, GUARD>), dim3(gemm_grid((M + 63) / 64, (N + 63) / 64)), dim3(256), 0, st, M,
                           N, Kd, A, lda, B, ldb, C, ldc, bias, act, 0);
    }
}

template <bool AK, bool BKM>
static int launch_gemm(int M, int N, int Kd, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                       const float* bias, int act, hipStream_t st)
{
    // unguarded kernels need whole tiles in every variant the tile chooser may pick (128 | M, 128 | N or 64 | N, 16 | K)
    const bool whole = (M % 128 == 0) && (N % 64 == 0) && (N <= 64 || N % 128 == 0) && (Kd % BKS == 0) &&
                       (lda % 4 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) && aligned16(A) && aligned16(B) && aligned16(C);
    if (whole) launch_gemm_tiles<AK, BKM, false>(M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, st);
    else launch_gemm_tiles<AK, BKM, true>(M, N, Kd, A, lda, B, ldb, C, ldc, bias, act, st);
    return check_launch("sph3d_pointwise_gemm");
}

// forward product + BN partial statistics: unguarded NN tiles only.  -> row blocks of the statistics (2 per tile row), 0 if
// the shape does not qualify (the caller then runs the plain product and the op's own statistics pass)
static int nn_stats_tile(int M, int N, int Kd, int& bm, int& bn)
{
    const bool whole = (M % 128 == 0) && (N % 64 == 0) && (N <= 64 || N % 128 == 0) && (Kd % BKS == 0) && (Kd % 4 == 0) && (N % 4 == 0);
    if (!whole) return 0;
    auto ntiles = [&](int a, int b) { return (long long)((M + a - 1) / a) * ((N + b - 1) / b); };
    static const long long kMinTiles = getenv("SPH3D_GEMM_MINTILES") ? atoi(getenv("SPH3D_GEMM_MINTILES")) : 512;      // (experiments)
    if (N > 64 && ntiles(128, 128) >= kMinTiles) { bm = 128; bn = 128; }
    else if (ntiles(128, 64) >= kMinTiles) { bm = 128; bn = 64; }      // (N > 64 too: 32768 x 1024 -> 128 measured 81 vs 90 us with 64 x 64)
    else { bm = 64; bn = 64; }
    return 2 * (M / bm);
}

// split-K plan for the weight gradient: enough (tile, split) workgroups to fill 256 CUs, k chunks multiple of BK
static void tn_plan(int R, int Cin, int Cout, int& bn, int& tiles, int& nsplit, int& kchunk)
{
    // the weight gradient always has thousands of (tile, split) workgroups: wide k-tile
    bn = Cout > 64 ? 128 : 64;
    tiles = ((Cin + BM - 1) / BM) * ((Cout + bn - 1) / bn);
    // ~2 workgroups per CU in total: measured round 2 over the 13 S3DIS shapes, 512 workgroups 0.97 ms against 1.10 ms
    // with 1024 (twice the partial tiles to write and re-read, half the k-loop to amortise prologue and epilogue) and
    // 1.11 ms with 256
    // SPH3D_TN_WGS: experiments.  Round 5 sweep over 256 / 384 / 512 / 768 / 1024 (tools/exp_gemm_knobs.py,
    // profiles/r05_exp_gemm_knobs.log): 512 is the best single value (0.79 ms over the 13 shapes; per-shape optimum 0.77);
    // only the short products with few tiles ((12288, 512 -> 256), (6144, 512 -> 512): <= 16 tiles, <= 12288 rows) want fewer,
    // fatter splits: 38 / 40 us at 256 workgroups against 46 / 46
    static const int forced = getenv("SPH3D_TN_WGS") ? atoi(getenv("SPH3D_TN_WGS")) : 0;
    const int target = forced > 0 ? forced : ((R <= 12288 && tiles <= 16) ? 256 : 512);
    int want = (target + tiles - 1) / tiles;   // (re-measured with the LDS-DMA kernel: 512 -> 2.25 ms over the 13 shapes, 768 2.46, 1024 2.31, 2048 2.28)
    int maxsplit = (R + 255) / 256;                   // at least 256 rows of k per split
    nsplit = want < maxsplit ? want : maxsplit;
    if (nsplit < 1) nsplit = 1;
    kchunk = (R + nsplit - 1) / nsplit;
    kchunk = ((kchunk + BKL - 1) / BKL) * BKL;
    nsplit = (R + kchunk - 1) / kchunk;
}

}  // namespace sph3d

using namespace sph3d;

extern "C" int sph3d_pointwise_gemm_exchange_failures(void)
{
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_xs_fail), sizeof(int), 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

extern "C" int sph3d_pointwise_gemm_mode(int mode)
{
    const int prev = split_on() ? 1 : 0;
    if (mode == 0 || mode == 1) g_split_mode = mode;
    return prev;
}

extern "C" int sph3d_pointwise_gemm(int R, int Cin, int Cout, const float* X, const float* W, const float* bias, int act,
                                    int trans_w, float* Y, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(R >= 0 && Cin > 0 && Cout > 0, "pointwise_gemm: bad dims R=%d Cin=%d Cout=%d", R, Cin, Cout);
    SPH3D_REQUIRE(act == 0 || act == 1, "pointwise_gemm: act must be 0 (none) or 1 (ELU), got %d", act);
    if (R == 0) return SPH3D_OK;
    hipStream_t st = as_stream(stream);
    // Y[R,Cout] = X[R,Cin] * op(W);  trans_w: W is stored [Cout,Cin] (k contiguous), else [Cin,Cout] (n contiguous)
    if (trans_w) return launch_gemm<true, true>(R, Cout, Cin, X, Cin, W, Cin, Y, Cout, bias, act, st);
    return launch_gemm<true, false>(R, Cout, Cin, X, Cin, W, Cout, Y, Cout, bias, act, st);
}

extern "C" int sph3d_pointwise_gemm_bnstats_blocks(int R, int Cin, int Cout)
{
    int bm = 0, bn = 0;
    return (R > 0 && Cin > 0 && Cout > 0) ? nn_stats_tile(R, Cout, Cin, bm, bn) : 0;
}

extern "C" int sph3d_pointwise_gemm_bnstats(int R, int Cin, int Cout, const float* X, const float* W, const float* bias, float* Y,
                                            float* partial, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(R > 0 && Cin > 0 && Cout > 0, "pointwise_gemm_bnstats: bad dims R=%d Cin=%d Cout=%d", R, Cin, Cout);
    int bm = 0, bn = 0;
    const int nblk = nn_stats_tile(R, Cout, Cin, bm, bn);
    if (nblk == 0 || !aligned16(X) || !aligned16(W) || !aligned16(Y)) {
        set_error("pointwise_gemm_bnstats: shape (%d, %d -> %d) needs whole tiles and 16-byte aligned operands", R, Cin, Cout);
        return SPH3D_EUNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    const unsigned tiles = gemm_grid(R / bm, Cout / bn);
    if (split_on()) {
        if (bm == 128 && bn == 128)
            hipLaunchKernelGGL((gemm_split_mfma<true, false, 128, 128, 16, false, true>), dim3(tiles), dim3(256), 0, st, R, Cout, Cin, X, Cin, W,
                               Cout, Y, Cout, bias, 0, 0, partial);
        else if (bm == 128)
            hipLaunchKernelGGL((gemm_split_mfma<true, false, 128, 64, 16, false, true>), dim3(tiles), dim3(256), 0, st, R, Cout, Cin, X, Cin, W,
                               Cout, Y, Cout, bias, 0, 0, partial);
        else if (Cin % 32 == 0) {
            const int ns = xs_splits(tiles, Cin, 32, xs_cap());
            char* xb = ns > 1 ? xs_buffer(st, (long long)(R / 64) * (Cout / 64), ns, 64 * 64) : nullptr;
            if (xb != nullptr)
                hipLaunchKernelGGL((gemm_split_mfma<true, false, 64, 64, 32, false, true, false, true>), dim3(tiles * ns), dim3(256), 0, st, R, Cout,
                                   Cin, X, Cin, W, Cout, Y, Cout, bias, 0, Cin / ns, partial, ns, (float*)(xb + kXsFlagBytes), (int*)xb);
            else
                hipLaunchKernelGGL((gemm_split_mfma<true, false, 64, 64, 32, false, true>), dim3(tiles), dim3(256), 0, st, R, Cout, Cin, X, Cin, W,
                                   Cout, Y, Cout, bias, 0, 0, partial);
        } else
            hipLaunchKernelGGL((gemm_split_mfma<true, false, 64, 64, 16, false, true>), dim3(tiles), dim3(256), 0, st, R, Cout, Cin, X, Cin, W,
                               Cout, Y, Cout, bias, 0, 0, partial);
        return check_launch("sph3d_pointwise_gemm_bnstats");
    }
    if (bm == 128 && bn == 128)
        hipLaunchKernelGGL((gemm_f32_mfma<true, false, 128, 128, BKS, false, false, true>), dim3(tiles), dim3(256), 0, st, R, Cout, Cin, X,
                           Cin, W, Cout, Y, Cout, bias, 0, 0, partial);
    else if (bm == 128)
        hipLaunchKernelGGL((gemm_f32_mfma<true, false, 128, 64, BKS, false, false, true>), dim3(tiles), dim3(256), 0, st, R, Cout, Cin, X,
                           Cin, W, Cout, Y, Cout, bias, 0, 0, partial);
    else
        hipLaunchKernelGGL((gemm_f32_mfma<true, false, 64, 64, BKS, false, false, true>), dim3(tiles), dim3(256), 0, st, R, Cout, Cin, X,
                           Cin, W, Cout, Y, Cout, bias, 0, 0, partial);
    return check_launch("sph3d_pointwise_gemm_bnstats");
}

extern "C" size_t sph3d_pointwise_gemm_tn_workspace(int R, int Cin, int Cout)
{
    int bn, tiles, nsplit, kchunk;
    tn_plan(R, Cin, Cout, bn, tiles, nsplit, kchunk);
    return nsplit > 1 ? sizeof(float) * (size_t)nsplit * Cin * Cout : 0;
}

extern "C" int sph3d_pointwise_gemm_tn(int R, int Cin, int Cout, const float* X, const float* dY, float* dW,
                                       void* workspace, size_t workspace_bytes, sph3d_stream_t stream)
{
    SPH3D_REQUIRE(R > 0 && Cin > 0 && Cout > 0, "pointwise_gemm_tn: bad dims R=%d Cin=%d Cout=%d", R, Cin, Cout);
    hipStream_t st = as_stream(stream);
    int bn, tiles, nsplit, kchunk;
    tn_plan(R, Cin, Cout, bn, tiles, nsplit, kchunk);
    const size_t need = sph3d_pointwise_gemm_tn_workspace(R, Cin, Cout);
    if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
        set_error("pointwise_gemm_tn: workspace %zu B < required %zu B", workspace_bytes, need);
        return SPH3D_EWORKSPACE;
    }
    float* out = nsplit > 1 ? (float*)workspace : dW;
    // dW[Cin,Cout] = X^T * dY : A(m=cin, k=r) = X[r*Cin + cin] (row contiguous), B(k=r, n=cout) = dY[r*Cout + cout]
    const bool whole = (Cin % 128 == 0) && (Cout % bn == 0) && (R % kchunk == 0) && aligned16(X) && aligned16(dY) && aligned16(out);
#define SPH3D_TN(BNN, G)                                                                                                   \
    hipLaunchKernelGGL((gemm_f32_mfma<false, false, 128, BNN, BKL, true, G>), dim3(gemm_grid(tiles, 1, nsplit)), dim3(256), 0, st, Cin, Cout, R, X, \
                       Cin, dY, Cout, out, Cout, nullptr, 0, kchunk, nullptr, nsplit)
#define SPH3D_TN16(BNN)                                                                                                    \
    hipLaunchKernelGGL((gemm_f32_mfma<false, false, 128, BNN, BKS, true, false>), dim3(gemm_grid(tiles, 1, nsplit)), dim3(256), 0, st, Cin, Cout, R, X, \
                       Cin, dY, Cout, out, Cout, nullptr, 0, kchunk, nullptr, nsplit)
    // whole tiles: BK = 16 at four workgroups per CU (8 + 8 prefetch registers with the 2x4 transposing units): 0.786 vs
    // 0.855 ms over the step's shapes against BK = 32 at two per CU
    // short products (few rows of k, many output tiles): 64 x 64 tiles with the in-kernel exchange — up to 8 splits whose slabs are
    // 16 KB — instead of 128 x 128 tiles + slab-sum launch.  (The exchange with 128 x 128 tiles and 8-16 splits was measured and is
    // SLOWER than the slab-sum kernel: 72 vs 38, 82 vs 50, 84 vs 62 us — split 0 reads 7-15 64-KB slabs past the cache, one dword per
    // lane and instruction; profiles/r06_exp_gemm_xs.log.)  SPH3D_GEMM_TN_XS: largest R that takes this path (experiments)
    static const int tn_xs_rows = getenv("SPH3D_GEMM_TN_XS") ? atoi(getenv("SPH3D_GEMM_TN_XS")) : 2048;
    if (split_on() && R <= tn_xs_rows && Cin % 64 == 0 && Cout % 64 == 0 && R % 32 == 0 && aligned16(X) && aligned16(dY) && aligned16(dW)) {
        const long long Tg = gemm_grid(Cin / 64, Cout / 64);
        const int ns = xs_splits(Tg, R, 32, 8);
        char* xb = ns > 1 ? xs_buffer(st, (long long)(Cin / 64) * (Cout / 64), ns, 64 * 64) : nullptr;
        if (xb != nullptr) {
            hipLaunchKernelGGL((gemm_split_mfma<false, false, 64, 64, 32, false, false, false, true>), dim3((unsigned)(Tg * ns)), dim3(256), 0, st, Cin,
                               Cout, R, X, Cin, dY, Cout, dW, Cout, nullptr, 0, R / ns, nullptr, ns, (float*)(xb + kXsFlagBytes), (int*)xb);
            return check_launch("sph3d_pointwise_gemm_tn");
        }
    }
    if (whole && split_on()) {
        if (bn == 128)
            hipLaunchKernelGGL((gemm_split_mfma<false, false, 128, 128, 16, true, false>), dim3(gemm_grid(tiles, 1, nsplit)), dim3(256), 0, st, Cin,
                               Cout, R, X, Cin, dY, Cout, out, Cout, nullptr, 0, kchunk, nullptr, nsplit);
        else
            hipLaunchKernelGGL((gemm_split_mfma<false, false, 128, 64, 16, true, false>), dim3(gemm_grid(tiles, 1, nsplit)), dim3(256), 0, st, Cin,
                               Cout, R, X, Cin, dY, Cout, out, Cout, nullptr, 0, kchunk, nullptr, nsplit);
    } else if (!whole && split_on() && Cin >= 32 && Cout >= 32) {
        if (bn == 128)
            hipLaunchKernelGGL((gemm_split_mfma<false, false, 128, 128, 16, true, false, true>), dim3(gemm_grid(tiles, 1, nsplit)), dim3(256), 0, st,
                               Cin, Cout, R, X, Cin, dY, Cout, out, Cout, nullptr, 0, kchunk, nullptr, nsplit);
        else
            hipLaunchKernelGGL((gemm_split_mfma<false, false, 128, 64, 16, true, false, true>), dim3(gemm_grid(tiles, 1, nsplit)), dim3(256), 0, st,
                               Cin, Cout, R, X, Cin, dY, Cout, out, Cout, nullptr, 0, kchunk, nullptr, nsplit);
    } else
    if (whole) { if (bn == 128) SPH3D_TN16(128); else SPH3D_TN16(64); }
    else if (bn == 128) SPH3D_TN(128, true);
    else SPH3D_TN(64, true);
#undef SPH3D_TN16
#undef SPH3D_TN
    if (nsplit > 1) {
        const int total = Cin * Cout;
        hipLaunchKernelGGL(gemm_reduce_splits, dim3((total + 31) / 32), dim3(256), 0, st, nsplit, total, out, dW);
    }
    return check_launch("sph3d_pointwise_gemm_tn");
}
