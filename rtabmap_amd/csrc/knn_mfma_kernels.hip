// knn_mfma_kernels.hip -- squared-L2 2-NN as a dense contraction on the matrix cores, with an exactness certificate.
//
// The reference's arithmetic, sum_k (v_k - q_k)^2 in rtflann's order (dist.h:150-177), costs 3 VALU ops per element and
// cannot use FMA.  Here the scan is split in two:
//
//   1. FILTER (MFMA): s(i, j) = |v_i|^2 + |q_j|^2 - 2 v_i . q_j for every (vocabulary row i, query j).  |v|^2 and |q|^2 ride
//      along as one extra f32 k-step (A = (|v_i|^2, 1), B = (1, |q_j|^2)) and the queries are pre-scaled by -2, so the
//      accumulator IS the approximate squared distance.  Every lane keeps a running top-3 of 32-bit keys (score bits | in-strip
//      row index) for the queries it sees -- no cross-lane traffic in the loop.  Two variants:
//        knn_bf16_filter_kernel  (default)  three bf16 MFMA chains per product on a hi/lo split of the operands
//                                           (v_mfma_f32_32x32x16_bf16, 16x the f32 rate), one vocabulary tile shared by the
//                                           four waves of a workgroup; the same launch also computes the same-frame distance matrix
//        knn_mfma_filter_kernel             v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate: an exact fp32 FMA chain
//   2. RE-RANK (knn_mfma_rerank_kernel): per query the few kept keys that can still be a neighbour (filter score within
//      2 eps of the second best) are re-evaluated with the reference's own arithmetic (bit-exact distances, lower row wins
//      ties) and the two best are returned.  The result is PROVEN equal to the exact scan when every row the filter dropped
//      is certainly farther than the exact second neighbour:
//            bound - eps > d2_exact,
//      bound = the smallest filter score any dropped row can have (tracked through every merge level), eps = a bound on
//      |filter score - reference distance| (eps_for() / eps_bf16()).  Queries that fail the certificate (near-duplicate
//      clusters) are re-done exactly by rowpar_body.cuh (one lane per vocabulary row, the whole chip on each rejected
//      query), so the output is always the reference's bit-exact answer.
//
// MFMA operand layout (both variants): lane (row or query l&31, half l>>5) holds the contiguous elements [32h, 32h+32) of its
// row, i.e. a k-step multiplies the same element subset on both sides -- A and B use the same k permutation, which a dot
// product does not see.  D layout: lane holds query (l&31), 16 rows (reg&3) + 8*(reg>>2) + 4*(l>>5).
#include <hip/hip_ext.h>
#include "lcd_kernels.h"
#include "rowpar_body.cuh"
#include "shard_body.cuh"
#include "frame_tail_body.cuh"
#include "score_body.cuh"

#include <cstdlib>

namespace lcd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MF_BLOCK = 256;
constexpr int MF_WAVES = 4;
constexpr int MF_KEEP = 4;     // keys kept per (row block, query)

// Augmentation table of the vocabulary: aug[2r] = |row r|^2 (any summation order: the filter only needs it to ~dim ulps;
// +inf for tombstones), aug[2r + 1] = 1, plus a sentinel entry {+inf, 1} at r = n_rows for the padding rows of the last
// tile.  The MFMA filter reads aug[2 * min(row, n_rows) + half] with ONE unconditional load per lane.
__global__ void row_norm_kernel(const float* __restrict__ vocab, const int32_t* __restrict__ row_id, int first, int n, int dim,
                                float* __restrict__ aug, uint32_t* __restrict__ norm_max_bits) {
    const int r = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= first + n) return;
    float s = __int_as_float(0x7f800000);
    if (row_id[r] != 0) {
        s = 0.0f;
        const float* v = vocab + (size_t)r * dim;
        for (int k = 0; k < dim; ++k) s = fmaf(v[k], v[k], s);
        atomicMax(norm_max_bits, __float_as_uint(s));
    }
    aug[2 * (size_t)r] = s;
    aug[2 * (size_t)r + 1] = 1.0f;
    if (r == first + n - 1) { aug[2 * (size_t)(r + 1)] = __int_as_float(0x7f800000); aug[2 * (size_t)(r + 1) + 1] = 1.0f; }
}
__global__ void norm_tombstone_kernel(float* __restrict__ aug, const int32_t* __restrict__ rows, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) aug[2 * (size_t)rows[i]] = __int_as_float(0x7f800000);
}

// bf16 split of the vocabulary for the bf16x3 filter: row r -> 256 bytes = 64 bf16 "hi" (the float rounded to bf16, RNE) then
// 64 bf16 "lo" (the exact remainder float - hi, rounded to bf16).  hi + lo carries ~17 significant bits of the float.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bf16_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));      // v_cvt_pk_bf16_f32
    const float ra = __fsub_rn(a, __uint_as_float(hi << 16)), rb = __fsub_rn(b, __uint_as_float(hi & 0xFFFF0000u));   // exact
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){ra, rb}, bf16x2_t));
}
// The fp16 filter (LCD_KNN_F16, one product per fp32 product: the operand format north_star names for the SURF distance GEMM): the SAME
// table layout with IEEE half "hi" (RNE, 11 significant bits) and half "lo" (the remainder, unused by the one-product filter) -- the
// matrix pipe runs v_mfma_f32_32x32x16_f16 at the bf16 rate, a third of the products, an eps of ~2^-10 (|q|^2 + |v|^2) instead of ~2^-14.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void f16_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f16x2_t h = __builtin_convertvector((f32x2_t){a, b}, f16x2_t);                          // v_cvt_pkrtz would truncate: this rounds to nearest even
    hi = __builtin_bit_cast(uint32_t, h);
    const f32x2_t back = __builtin_convertvector(h, f32x2_t);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){__fsub_rn(a, back.x), __fsub_rn(b, back.y)}, f16x2_t));
}
template <int M> __device__ __forceinline__ void op_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    if (M == 1) f16_split2(a, b, hi, lo); else bf16_split2(a, b, hi, lo);
}
__device__ __forceinline__ void op_split2_rt(int f16, float a, float b, uint32_t& hi, uint32_t& lo) {
    if (f16) f16_split2(a, b, hi, lo); else bf16_split2(a, b, hi, lo);
}
__global__ void vocab_bf16_kernel(const float* __restrict__ vocab, int first, int n, uint32_t* __restrict__ bf, int f16) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per 4 floats of a 64-float row
    if (i >= n * 16) return;
    const int r = first + (i >> 4), c = i & 15;
    const float4 x = reinterpret_cast<const float4*>(vocab + (size_t)r * 64)[c];
    uint2 hi, lo;
    op_split2_rt(f16, x.x, x.y, hi.x, lo.x);
    op_split2_rt(f16, x.z, x.w, hi.y, lo.y);
    reinterpret_cast<uint2*>(bf + (size_t)r * 64)[c] = hi;
    reinterpret_cast<uint2*>(bf + (size_t)r * 64 + 32)[c] = lo;
}

// rows [first, first + n) that hold no word yet (capacity behind the vocabulary, filled by the device-side append): |row|^2 = +inf so
// that no filter ever ranks them, a zero bf16 split so that the product with them is finite
__global__ void vocab_tail_kernel(float* __restrict__ aug, uint32_t* __restrict__ bf, long long first, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per 16 bytes of the 256-byte split row
    if (i >= n * 16) return;
    const long long r = first + (i >> 4);
    const int c = (int)(i & 15);
    reinterpret_cast<uint4*>(bf + r * 64)[c] = make_uint4(0u, 0u, 0u, 0u);
    if (c == 0) { aug[2 * r] = __int_as_float(0x7f800000); aug[2 * r + 1] = 1.0f; }
}

// ------------------------------------------------------------------------------------------------ filter
// In-loop candidate key: 32 bits = the score's float bits with the low MF_IDX_BITS mantissa bits replaced by the
// candidate's position inside the wave's strip (tile-in-strip << 4 | accumulator register).  Scores are >= 0, so the keys
// strip (tile-in-strip << 4 | accumulator register).  Keys are compared as SIGNED integers: non-negative floats order like
// their bit patterns, and a score that rounding pushed slightly below zero (an exact duplicate of the query) sorts first,
// which is where it belongs; a top-3 update is one v_min_i32 and two v_med3_i32.  Truncation only LOWERS a key
// (by < 2^-16 relative): a dropped row's true score is >= its key >= the bound derived from kept keys, so the
// certificate stays valid; at most MF_STRIP_TILES tiles per wave strip keep the index in 7 bits.
constexpr int MF_IDX_BITS = 7;
constexpr int MF_STRIP_TILES = 1 << (MF_IDX_BITS - 4);
constexpr uint32_t MF_IDX_MASK = (1u << MF_IDX_BITS) - 1;
constexpr int32_t MF_KEY_NONE = 0x7FFFFFFF;

// sorted insertion into k0 <= k1 <= k2 from the OLD values only (three independent VALU, no dependent chain):
//   k0' = min(k0, k), k1' = med3(k0, k1, k), k2' = med3(k1, k2, k)
__device__ __forceinline__ int32_t med3_i32(int32_t a, int32_t b, int32_t c) {
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void top3_push32(int32_t& k0, int32_t& k1, int32_t& k2, int32_t k) {
    const int32_t n2 = med3_i32(k1, k2, k);
    const int32_t n1 = med3_i32(k0, k1, k);
    k0 = min(k0, k);
    k1 = n1;
    k2 = n2;
}
// strip key of accumulator register r of strip tile tl: the score bits with the index in the low mantissa bits -- one v_and_or_b32
// (the index is wave-uniform)
// (the index is wave-uniform; the mask is kept in a VGPR the compiler cannot see through, or it would pick v_and + v_or with
// a literal)
__device__ __forceinline__ uint32_t strip_mask() {
    uint32_t m;
    asm("v_mov_b32 %0, 0xffffff80" : "=v"(m));
    static_assert(MF_IDX_BITS == 7, "literal above");
    return m;
}
__device__ __forceinline__ int32_t strip_key(float score, uint32_t mask, uint32_t idx) {
    return (int32_t)((__float_as_uint(score) & mask) | idx);
}
// strip key -> merge key (score bits << 32 | vocabulary row); a slightly negative score (rounding of a distance ~ 0) becomes +0
__device__ __forceinline__ uint64_t widen_key(int32_t k, int t_begin, int half) {
    if (k == MF_KEY_NONE) return KEY_NONE;
    const uint32_t idx = (uint32_t)k & MF_IDX_MASK, r = idx & 15u;
    const uint32_t row = (uint32_t)(t_begin + (int)(idx >> 4)) * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (uint32_t)half;
    const uint32_t bits = k < 0 ? 0u : ((uint32_t)k & ~MF_IDX_MASK);
    return ((uint64_t)bits << 32) | row;
}

// A tile = 32 vocabulary rows x DIM floats (8 KB for DIM = 64).  It goes global -> LDS with the asynchronous LDS-DMA
// (global_load_lds, 16 B per lane, no staging VGPRs, fully coalesced: every instruction moves 1 KiB = 8 whole 128-B lines),
// then LDS -> VGPRs in MFMA operand order (lane (row l&31, half l>>5) gets the contiguous floats [KH*half, KH*half + KH) of
// its row).  The DMA writes LDS linearly (wave-uniform base + 16 * lane), so the bank-conflict fix is an XOR swizzle applied
// to the SOURCE address and to the read address alike (cdna_hip_programming.md rule 21): 16-B chunk c of row r lives at
// chunk position c ^ (r & 15).  A per-lane gather straight from global memory (row stride 256 B across lanes) touches 64
// lines per load and thrashes the 32 KiB L1: it made the kernel load-bound.
template <int KH>
__device__ __forceinline__ void dma_a_tile(const float* __restrict__ vocab, int n_rows, int t, int lane, float* __restrict__ lds_slot) {
    constexpr int DIM = 2 * KH;
    constexpr int CPR = DIM / 4;                          // 16-B chunks per row
    static_assert(CPR == 16 || CPR == 32, "swizzle written for 64- or 128-float rows");
#pragma unroll
    for (int i = 0; i < (32 * CPR) / 64; ++i) {           // 8 instructions for DIM = 64
        const int p = i * 64 + lane;                      // linear chunk position in the LDS slot
        const int r = p / CPR, cpos = p % CPR;
        const int c = cpos ^ (r & 15);                    // the global chunk that belongs at this position
        const int row = min(t * 32 + r, n_rows - 1);      // padding rows repeat the last row (their score is forced to +inf)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vocab + (size_t)row * DIM + c * 4),
                                         (__attribute__((address_space(3))) void*)(lds_slot + i * 256), 16, 0, 0);
    }
}
template <int KH>
__device__ __forceinline__ void read_a_tile(const float* __restrict__ lds_slot, int col, int half, float (&a)[KH]) {
    constexpr int DIM = 2 * KH;
    const float* rowp = lds_slot + col * DIM;
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) {
        const int c = half * (KH / 4) + v;                // chunk of the row this lane needs
        const float4 x = *reinterpret_cast<const float4*>(rowp + ((c ^ (col & 15)) << 2));
        a[4 * v + 0] = x.x; a[4 * v + 1] = x.y; a[4 * v + 2] = x.z; a[4 * v + 3] = x.w;
    }
}

// one 32-row tile against one 32-query group: 33 MFMAs
template <int KH>
__device__ __forceinline__ f32x16 mfma_group(const float (&a)[KH], float a_aug, const float (&b)[KH], float b_aug) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#if LCD_MFMA_ABLATE == 2
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = a[r] * b[r] + a_aug * b_aug;
    return acc;
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b_aug, acc, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < KH; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
    return acc;
}
// one 32-row tile against TWO 32-query groups with the two accumulator chains interleaved k-step by k-step: consecutive
// MFMAs are independent, so the matrix pipe never waits for a dependent accumulator
template <int KH>
__device__ __forceinline__ void mfma_pair(const float (&a)[KH], float a_aug, const float (&b0)[KH], float b0_aug, const float (&b1)[KH],
                                          float b1_aug, f32x16& acc0, f32x16& acc1) {
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b0_aug, z, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b1_aug, z, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b0[k], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b1[k], acc1, 0, 0, 0);
    }
}

// acc[r] = approximate squared distance between the lane's query and row t*32 + (r&3) + 8*(r>>2) + 4*half
#ifndef LCD_MFMA_ABLATE
#define LCD_MFMA_ABLATE 0      // 1: skip the top-3 update (timing experiment only), 2: skip the MFMAs
#endif
__device__ __forceinline__ void push_group(const f32x16& acc, uint32_t tl, int32_t& k0, int32_t& k1, int32_t& k2) {
#if LCD_MFMA_ABLATE == 1
    asm volatile("" :: "v"(acc[0]), "v"(acc[5]), "v"(acc[10]), "v"(acc[15]));
    k0 = min(k0, __float_as_int(acc[3])); (void)tl; (void)k1; (void)k2;
    return;
#endif
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tl << 4));
    const uint32_t mask = strip_mask();
#pragma unroll
    for (int r = 0; r < 16; ++r) top3_push32(k0, k1, k2, strip_key(acc[r], mask, base | (uint32_t)r));
}

// The bf16 / fp16 filters select GROUPS: the four accumulator registers 4g .. 4g + 3 of a lane are four CONSECUTIVE vocabulary rows
// (t * 32 + 8 g + 4 half + {0, 1, 2, 3}); the lane keeps the three groups with the smallest minimum -- one key per group (the minimum's bits,
// the index of the group's FIRST register), 6 VALU per four scores (v_min3_f32, v_min_f32, v_and_or_b32, v_med3_i32 x 2, v_min_i32)
// instead of 16.  Complete for the re-rank, which evaluates all four rows of a kept group exactly: every row of the true top-3 lies in a
// group whose minimum is <= that row's score, and a group WITHOUT such a row has a minimum >= the third-best row's score -- so the
// three groups with the smallest minima contain the three best rows, and the bound on what was dropped (the third kept key) holds as
// before.  (The filter loop was issue-bound 2:1 on exactly this selection: 354 instructions per 32-row tile against 768 matrix-pipe
// cycles with one fp16 product per fp32 product)
__device__ __forceinline__ float min4(float a, float b, float c, float d) { return fminf(fminf(fminf(a, b), c), d); }
__device__ __forceinline__ void push_group4(const f32x16& acc, uint32_t tl, int32_t& k0, int32_t& k1, int32_t& k2) {
#if LCD_MFMA_ABLATE == 1
    asm volatile("" :: "v"(acc[0]), "v"(acc[5]), "v"(acc[10]), "v"(acc[15]));
    k0 = min(k0, __float_as_int(acc[3])); (void)tl; (void)k1; (void)k2;
    return;
#endif
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tl << 4));
    const uint32_t mask = strip_mask();
#pragma unroll
    for (int g = 0; g < 4; ++g) top3_push32(k0, k1, k2, strip_key(min4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]), mask, base | (uint32_t)(4 * g)));
}

// One software-pipeline step in explicit program order: the 66 MFMAs of a group pair (two interleaved accumulator chains)
// with the top-3 update of the PREVIOUS pair's 32 scores spread between them -- 4 MFMAs, then the update of one score of
// each pending accumulator (~14 VALU), sixteen times.  A wave issues in order and the compiler otherwise emits the MFMAs
// back to back and the VALU afterwards (measured: MFMA-busy 57 % of the wave cycles, VALU time additive), so the order is
// pinned with sched_barrier(0): the VALU then issues in the shadow of the 64-cycle MFMAs.
template <int KH>
__device__ __forceinline__ void mfma_pair_push(const float (&a)[KH], float a_aug, const float (&b0)[KH], float b0_aug, const float (&b1)[KH],
                                               float b1_aug, f32x16& c0, f32x16& c1, const f32x16& p0, const f32x16& p1, uint32_t tl,
                                               int32_t& k00, int32_t& k01, int32_t& k02, int32_t& k10, int32_t& k11, int32_t& k12) {
    static_assert(KH == 32, "interleave pattern written for 64-float rows");
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tl << 4));
    const uint32_t mask = strip_mask();
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b0_aug, z, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b1_aug, z, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * r], b0[2 * r], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * r], b1[2 * r], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * r + 1], b0[2 * r + 1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * r + 1], b1[2 * r + 1], c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        top3_push32(k00, k01, k02, strip_key(p0[r], mask, base | (uint32_t)r));
        top3_push32(k10, k11, k12, strip_key(p1[r], mask, base | (uint32_t)r));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// partial_keys [qpad][n_blocks][MF_KEEP] u64, partial_lmin [qpad][n_blocks] f32 bits (query-major: the re-rank wave of a query
// reads one contiguous run).
// NG = 32-query column groups per wave (wave tile = NG*32 queries x 32 rows).  NG = 4 runs ONE wave per SIMD with four
// independent accumulator chains (A tiles reused 4x, the VALU top-3 update of one accumulator issues under the MFMAs of
// the next); NG = 2 runs two waves per SIMD.
#ifdef LCD_MFMA_TIMING   // timing experiment only: per-wave timestamps (100 MHz) at kernel entry, loop entry, loop exit, kernel exit
__device__ unsigned long long g_mf_timing[4 * 4096];
#define MF_STAMP(i) do { if (lane == 0) g_mf_timing[4 * (((blockIdx.y * gridDim.x + blockIdx.x) & 1023) * MF_WAVES + wave) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned long long g_mf_timing2[8 * 4096];   // finer stamps inside one loop trip of the bf16 filter
#define MF_STAMP2(i) do { if (lane == 0 && (i) < 8) { g_mf_timing2[8 * (((blockIdx.y * gridDim.x + blockIdx.x) & 1023) * MF_WAVES + wave) + (i)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define MF_STAMP(i) do { } while (0)
#define MF_STAMP2(i) do { } while (0)
#endif

template <int DIM, int NG>
__global__ __launch_bounds__(MF_BLOCK, (NG == 2 ? 2 : 1)) void knn_mfma_filter_kernel(const float* __restrict__ vocab,
                                                                                     const float* __restrict__ row_norm, int n_rows,
                                                                                     const float* __restrict__ queries, int nq, int qpad,
                                                                                     int tiles_per_block, uint64_t* __restrict__ partial_keys,
                                                                                     uint32_t* __restrict__ partial_lmin) {
    constexpr int KH = DIM / 2;                    // floats of a row held by one lane
    constexpr int QW = NG * 32;                    // queries per wave / workgroup
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int q0 = blockIdx.y * QW;
    MF_STAMP(0);

    // B operand: the NG 32-query groups, pre-scaled by -2 (exact), + |q|^2 for the extra k-step
    float b[NG][KH];
    float b_aug[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int qi = min(q0 + g * 32 + col, nq - 1);
        const float4* src = reinterpret_cast<const float4*>(queries + (size_t)qi * DIM + half * KH);
        float part = 0.0f;
#pragma unroll
        for (int v = 0; v < KH / 4; ++v) {
            const float4 x = src[v];
            b[g][4 * v + 0] = -2.0f * x.x; b[g][4 * v + 1] = -2.0f * x.y; b[g][4 * v + 2] = -2.0f * x.z; b[g][4 * v + 3] = -2.0f * x.w;
            part = fmaf(x.x, x.x, part); part = fmaf(x.y, x.y, part); part = fmaf(x.z, x.z, part); part = fmaf(x.w, x.w, part);
        }
        const float qn = part + __shfl_xor(part, 32, 64);          // both halves of the row
        b_aug[g] = half == 0 ? 1.0f : qn;                          // B[k0][j] = 1, B[k1][j] = |q_j|^2
    }

    const int tile0 = blockIdx.x * tiles_per_block;
    const int n_tiles = (n_rows + 31) / 32;
    const int tile1 = min(tile0 + tiles_per_block, n_tiles);
    const int per_wave = (tile1 - tile0 + MF_WAVES - 1) / MF_WAVES;
    const int t_begin = min(tile0 + wave * per_wave, tile1);
    const int t_end = min(t_begin + per_wave, tile1);

    // every lane keeps its three best keys per query group: a row the lane drops is no better than its third key
    int32_t k0[NG], k1[NG], k2[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { k0[g] = MF_KEY_NONE; k1[g] = MF_KEY_NONE; k2[g] = MF_KEY_NONE; }
    // Software pipeline per wave: while tile t occupies the matrix pipe, the LDS-DMA of tile t+1 is in flight into the wave's
    // other LDS slot; it is waited for (vmcnt) and read back in operand order right before it is needed.  The two accumulator
    // chains of a group pair are interleaved (independent consecutive MFMAs); the VALU top-3 update of a pair is issued under
    // the MFMAs of the next pair.  Each wave owns its two slots: no workgroup barrier in the loop.
    __shared__ __attribute__((aligned(16))) float s_tile[MF_WAVES][2][32 * DIM];
    float a[KH];
    float aug = 0.0f;
    if (t_begin < t_end) {
        dma_a_tile<KH>(vocab, n_rows, t_begin, lane, s_tile[wave][0]);
        aug = row_norm[2 * (size_t)min(t_begin * 32 + col, n_rows) + half];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        read_a_tile<KH>(s_tile[wave][0], col, half, a);
        dma_a_tile<KH>(vocab, n_rows, min(t_begin + 1, t_end - 1), lane, s_tile[wave][1]);        // prefetch tile t_begin + 1
        float aug_next = row_norm[2 * (size_t)min(min(t_begin + 1, t_end - 1) * 32 + col, n_rows) + half];
        f32x16 p0, p1;                                               // pending accumulators (previous pair)
        int pend_t = t_begin;
        MF_STAMP(1);
        mfma_pair<KH>(a, aug, b[0], b_aug[0], b[1], b_aug[1], p0, p1);
        for (int t = t_begin; t < t_end; ++t) {
#pragma unroll
            for (int gp = 1; gp < NG / 2; ++gp) {                    // remaining pairs of tile t
                f32x16 c0, c1;
                mfma_pair_push<KH>(a, aug, b[2 * gp], b_aug[2 * gp], b[2 * gp + 1], b_aug[2 * gp + 1], c0, c1, p0, p1,
                                   (uint32_t)(pend_t - t_begin), k0[2 * gp - 2], k1[2 * gp - 2], k2[2 * gp - 2], k0[2 * gp - 1],
                                   k1[2 * gp - 1], k2[2 * gp - 1]);
                p0 = c0; p1 = c1;
            }
            if (t + 1 >= t_end) break;
            // tile t+1 has landed in the other slot: operand order -> registers (the MFMAs that read `a` are already issued),
            // then start the DMA of tile t+2 into the slot just vacated
            const int cur = (t + 1 - t_begin) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_a_tile<KH>(s_tile[wave][cur], col, half, a);
            aug = aug_next;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot reads above precede the DMA that overwrites the other slot's twin
            dma_a_tile<KH>(vocab, n_rows, min(t + 2, t_end - 1), lane, s_tile[wave][cur ^ 1]);
            aug_next = row_norm[2 * (size_t)min(min(t + 2, t_end - 1) * 32 + col, n_rows) + half];
            {
                f32x16 c0, c1;                                       // (t+1, pair 0) with the update of (t, last pair)
                mfma_pair_push<KH>(a, aug, b[0], b_aug[0], b[1], b_aug[1], c0, c1, p0, p1, (uint32_t)(t - t_begin), k0[NG - 2], k1[NG - 2],
                                   k2[NG - 2], k0[NG - 1], k1[NG - 1], k2[NG - 1]);
                p0 = c0; p1 = c1; pend_t = t + 1;
            }
        }
        // the last pending pair: the last pair of groups of the last tile
        push_group(p0, (uint32_t)(pend_t - t_begin), k0[NG - 2], k1[NG - 2], k2[NG - 2]);
        push_group(p1, (uint32_t)(pend_t - t_begin), k0[NG - 1], k1[NG - 1], k2[NG - 1]);
    }

    MF_STAMP(2);
    // workgroup merge: 8 partitions (4 waves x 2 halves) x top-3 per query -> top-MF_KEEP + the smallest partition third
    __shared__ uint64_t s_key[QW][MF_WAVES * 2][3];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        s_key[g * 32 + col][wave * 2 + half][0] = widen_key(k0[g], t_begin, half);
        s_key[g * 32 + col][wave * 2 + half][1] = widen_key(k1[g], t_begin, half);
        s_key[g * 32 + col][wave * 2 + half][2] = widen_key(k2[g], t_begin, half);
    }
    __syncthreads();
    if (threadIdx.x < QW) {
        const int ql = threadIdx.x;
        uint64_t keep[MF_KEEP];
#pragma unroll
        for (int i = 0; i < MF_KEEP; ++i) keep[i] = KEY_NONE;
        uint32_t lmin = 0x7f800000u;                                 // +inf
        for (int p = 0; p < MF_WAVES * 2; ++p) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                uint64_t k = s_key[ql][p][e];
                if (e == 2) lmin = min(lmin, (uint32_t)min(k >> 32, (uint64_t)0x7f800000u));   // rows hidden behind a partition's top-3
#pragma unroll
                for (int i = 0; i < MF_KEEP; ++i) {                  // sorted insertion
                    const uint64_t lo = keep[i] < k ? keep[i] : k;
                    k = keep[i] < k ? k : keep[i];
                    keep[i] = lo;
                }
            }
        }
        const int qi = q0 + ql;
        if (qi < qpad) {
#pragma unroll
            for (int i = 0; i < MF_KEEP; ++i) partial_keys[((size_t)qi * gridDim.x + blockIdx.x) * MF_KEEP + i] = keep[i];
            partial_lmin[(size_t)qi * gridDim.x + blockIdx.x] = lmin;
        }
    }
    MF_STAMP(3);
}


// ------------------------------------------------------------------------------------------------ bf16x3 filter
// The same filter with the contraction on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate): every float
// is split hi + lo (two bf16), and q . v ~ qh.vh + qh.vl + ql.vh -- three bf16 MFMA chains accumulated in fp32 into the SAME
// accumulator that the f32 augmentation step (|v|^2 + |q|^2, exact) initialised.  The neglected ql.vl and the bf16 rounding of
// the lo parts cost < 2^-16 relative to |q||v|, which eps_bf16() adds to the certificate -- the result stays the exact scan's.
//
// Workgroup = 4 waves x 128 queries (512 queries) against ONE shared strip of vocabulary tiles: a tile (32 rows x {hi, lo} =
// 8 KiB) is brought in once by LDS-DMA (each wave issues a quarter), double-buffered, one barrier per tile, and read by all four
// waves -- the vocabulary crosses L2 -> LDS once per 512 queries.  The queries are staged the same way (coalesced DMA, then
// operand order), split on the fly, and stay in registers for the whole kernel.
constexpr int BF_KEEP = 2;                       // keys kept per (row block, query); the third best is the block's bound
constexpr int BF_QW = 128;                       // queries per wave
constexpr int BF_QB = BF_QW * MF_WAVES;          // queries per workgroup
constexpr int BF_TILE_F = 32 * 64;               // floats (dwords) per staged tile
constexpr size_t BF_LDS_BYTES = (size_t)(MF_WAVES * 4 + 2) * BF_TILE_F * 4 + (size_t)MF_STRIP_TILES * 64 * 4;   // + the strip's augmentation entries
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// DMA instructions [i0, i1) of the 8 that move one 32-row x 256-byte tile (same swizzle as dma_a_tile)
// HI (round 6, the fp16 filter): only the row's first 128-byte line -- the "hi" operands, the only ones the one-product filter multiplies -- is
// requested: the lanes whose chunk lies in the "lo" line sit the instruction out.  Same instruction count (the waits that count instructions
// stay valid), same LDS layout (the lo positions are simply never written or read), HALF the bytes: the opening burst of launch A -- every
// strip asks for its whole strip in the first microsecond -- is 6.3 MB instead of 12.5 MB at 49 000 words.
template <bool HI = false>
__device__ __forceinline__ void dma_tile_part(const float* __restrict__ base, int n_rows, int t, int lane, float* __restrict__ lds_slot,
                                              int i0, int i1) {
    for (int i = i0; i < i1; ++i) {
        const int p = i * 64 + lane;
        const int r = p >> 4, cpos = p & 15;
        const int c = cpos ^ (r & 15);
        if (HI && c >= 8) continue;
        const int row = min(t * 32 + r, n_rows - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)row * 64 + c * 4),
                                         (__attribute__((address_space(3))) void*)(lds_slot + i * 256), 16, 0, 0);
    }
}
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <int M>
__device__ __forceinline__ f32x16 bf_mfma(const uint4& a, const uint4& b, const f32x16& c) {
    if (M == 1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// One 32-row tile against two 32-query groups: 2 x (1 f32 augmentation step + 12 bf16 steps), the two accumulator chains
// interleaved; with PUSH the top-3 update of the previous pair's 32 scores is spread between the steps.
// M = 0: bf16, three products per fp32 product (hi.hi + hi.lo + lo.hi); M = 1: fp16, the hi.hi product alone.
template <bool PUSH, int M>
__device__ __forceinline__ void bf_pair(const uint4 (&ah)[4], const uint4 (&al)[4], float a_aug, const uint4 (&bh0)[4], const uint4 (&bl0)[4],
                                        float b0_aug, const uint4 (&bh1)[4], const uint4 (&bl1)[4], float b1_aug, f32x16& c0, f32x16& c1,
                                        const f32x16& p0, const f32x16& p1, uint32_t tl, int32_t& k00, int32_t& k01, int32_t& k02,
                                        int32_t& k10, int32_t& k11, int32_t& k12) {
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tl << 4));
    const uint32_t mask = strip_mask();
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b0_aug, z, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b1_aug, z, 0, 0, 0);
#pragma unroll
    for (int st = 0; st < 12; ++st) {
        const int s = st / 3, term = st % 3;                        // (hi, hi), (hi, lo), (lo, hi)
        const uint4& a = term == 2 ? al[s] : ah[s];
#if LCD_MFMA_ABLATE != 3      // 4 / 5: only the (hi, hi) / the (hi, hi) + (hi, lo) products -- the MFMA count of a one- / two-product filter (timing only)
        if ((M == 0 && LCD_MFMA_ABLATE < 4) || term == 0 || (LCD_MFMA_ABLATE == 5 && term == 1)) c0 = bf_mfma<M>(a, term == 1 ? bl0[s] : bh0[s], c0);
#endif
        if (PUSH && LCD_MFMA_ABLATE == 1) asm volatile("" :: "v"(p0[st]), "v"(p1[st]));   // keep the ablated chains alive
        if (PUSH && LCD_MFMA_ABLATE != 1) {                         // one MFMA, then the VALU that fits in its 32-cycle shadow
            __builtin_amdgcn_sched_barrier(0);
            if (st < 4) top3_push32(k00, k01, k02, strip_key(min4(p0[4 * st], p0[4 * st + 1], p0[4 * st + 2], p0[4 * st + 3]), mask, base | (uint32_t)(4 * st)));   // group st of the previous pair
            __builtin_amdgcn_sched_barrier(0);
        }
#if LCD_MFMA_ABLATE != 3
        if ((M == 0 && LCD_MFMA_ABLATE < 4) || term == 0 || (LCD_MFMA_ABLATE == 5 && term == 1)) c1 = bf_mfma<M>(a, term == 1 ? bl1[s] : bh1[s], c1);
#endif
        if (PUSH && LCD_MFMA_ABLATE != 1) {
            __builtin_amdgcn_sched_barrier(0);
            if (st < 4) top3_push32(k10, k11, k12, strip_key(min4(p1[4 * st], p1[4 * st + 1], p1[4 * st + 2], p1[4 * st + 3]), mask, base | (uint32_t)(4 * st)));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Eight 16-byte LDS reads + the wait for them, as ONE inline-assembly statement the compiler does not see as LDS traffic (see the
// filter loop).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_read8_b128(const uint32_t (&addr)[8], uint4 (&out)[8], uint32_t addr32, float& out32) {
    u32x4_t v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile(
        "ds_read_b128 %0, %9\n\tds_read_b128 %1, %10\n\tds_read_b128 %2, %11\n\tds_read_b128 %3, %12\n\t"
        "ds_read_b128 %4, %13\n\tds_read_b128 %5, %14\n\tds_read_b128 %6, %15\n\tds_read_b128 %7, %16\n\t"
        "ds_read_b32 %8, %17\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7), "=&v"(out32)
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]), "v"(addr32)
        : "memory");
    out[0] = __builtin_bit_cast(uint4, v0); out[1] = __builtin_bit_cast(uint4, v1); out[2] = __builtin_bit_cast(uint4, v2);
    out[3] = __builtin_bit_cast(uint4, v3); out[4] = __builtin_bit_cast(uint4, v4); out[5] = __builtin_bit_cast(uint4, v5);
    out[6] = __builtin_bit_cast(uint4, v6); out[7] = __builtin_bit_cast(uint4, v7);
}
// the same for the "hi" operands alone (the fp16 filter): four 16-byte reads
__device__ __forceinline__ void lds_read4_b128(const uint32_t (&addr)[8], uint4 (&out)[8], uint32_t addr32, float& out32) {
    u32x4_t v0, v1, v2, v3;
    asm volatile(
        "ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %8\n\t"
        "ds_read_b32 %4, %9\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(out32)
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr32)
        : "memory");
    out[0] = __builtin_bit_cast(uint4, v0); out[1] = __builtin_bit_cast(uint4, v1); out[2] = __builtin_bit_cast(uint4, v2);
    out[3] = __builtin_bit_cast(uint4, v3);
    out[4] = out[5] = out[6] = out[7] = make_uint4(0u, 0u, 0u, 0u);     // ("lo": never multiplied by the one-product filter)
}
template <int M>
__device__ __forceinline__ void lds_read_ops(const uint32_t (&addr)[8], uint4 (&out)[8], uint32_t addr32, float& out32) {
    if (M == 1) lds_read4_b128(addr, out, addr32, out32); else lds_read8_b128(addr, out, addr32, out32);
}
// third smallest of two sorted triples
__device__ __forceinline__ uint64_t third_of_two_triples(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t b0, uint64_t b1, uint64_t b2) {
    const uint64_t x = a1 > b0 ? a1 : b0, y = a0 > b1 ? a0 : b1;
    uint64_t m = a2 < b2 ? a2 : b2;
    m = m < x ? m : x;
    return m < y ? m : y;
}

// ---- same-frame distance matrix, computed by extra workgroups of the filter launch (independent of the 2-NN; a launch of its own
// costs more than the work).  One workgroup = one 64 x 64 tile of the upper triangle of D[r][c] = |q_r - q_c|^2 in the reference's
// arithmetic (dist.h:150-177; (a - b)^2 == (b - a)^2 bit for bit, so the mirrored tile is a copy).  Both 64-query tiles are staged
// in LDS (16-byte chunks XOR-swizzled by the row so that lanes with different queries read conflict-free); a thread owns a 4 x 4 block
// of the tile (rows ty + 16 m, columns tx + 16 n), so a 16-byte LDS read feeds four outputs.  64 x 64 and not 32 x 32: every workgroup
// of the launch holds a whole compute unit's LDS, and 500 descriptors are 36 tiles -- which fit on the compute units the 192 filter
// workgroups leave free -- instead of 136, which did not (the launch then ran in two rounds: 23 us instead of 15).
struct SelfdistJob {
    const float* queries = nullptr;    // [nq x 64]
    int nq = 0;
    float* out = nullptr;              // [nq x ld]
    int ld = 0;
    int n_tiles = 0;                   // workgroups: n_self + the cross-frame tiles; 0 = no job
    int n_self = 0;                    // T (T + 1) / 2, T = ceil(nq / 64): the tiles of the upper triangle of the same-frame matrix (0: not asked for)
    // cross-frame tiles (round 5): X[r][c] = |q_r - o_c|^2 against the descriptors of the frame BEFORE this one, ceil(nq / 64) x ceil(n_other / 64)
    // full tiles.  The rows that frame appends to the vocabulary ARE descriptors of it, and this frame's re-rank (launch B of the same pair) has to
    // scan them exactly -- they are not in the filter's snapshot.  With this matrix that scan is one gathered read per pending row instead of
    // 38 KB of rows staged through LDS by each of 250 workgroups (+9.6 MB per launch) and ~150 distances per query.
    const float* other = nullptr; int n_other = 0;
    float* xout = nullptr; int xld = 0;
};
inline int selfdist_tiles(int q) { const int T = (q + 63) / 64; return T * (T + 1) / 2; }
__device__ __forceinline__ void selfdist_tile(const SelfdistJob& sd, int k, float* __restrict__ lds) {
    const int T = (sd.nq + 63) / 64;
    const bool xj = k >= sd.n_self;                    // uniform: a cross-frame tile
    const float* __restrict__ colsrc = xj ? sd.other : sd.queries;
    const int ncol = xj ? sd.n_other : sd.nq;
    float* __restrict__ out = xj ? sd.xout : sd.out;
    const int ld = xj ? sd.xld : sd.ld;
    int ti = 0, tj;
    if (xj) {
        const int Tc = (ncol + 63) / 64;
        ti = (k - sd.n_self) / Tc; tj = (k - sd.n_self) % Tc;
    } else {
        int rem = k;
        while (rem >= T - ti) { rem -= T - ti; ++ti; }
        tj = ti + rem;
    }
    const int tid = threadIdx.x;
    const bool act = tid < 256;                        // the tile is the work of 256 threads; a larger workgroup's other threads idle
    float* sA = lds;                   // rows of tile ti   [64][64] swizzled
    float* sB = lds + 4096;            // rows of tile tj
    float* sT = lds + 8192;            // [64][65] transposed result
    if (act) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * 256;                   // float4 index within a tile: row e / 16, chunk e % 16
            const int r = e >> 4, c = e & 15;
            const int ra = min(ti * 64 + r, sd.nq - 1), rb = min(tj * 64 + r, ncol - 1);
            const float4 a = reinterpret_cast<const float4*>(sd.queries + (size_t)ra * 64)[c];
            const float4 b = reinterpret_cast<const float4*>(colsrc + (size_t)rb * 64)[c];
            *reinterpret_cast<float4*>(sA + r * 64 + ((c ^ (r & 15)) << 2)) = a;
            *reinterpret_cast<float4*>(sB + r * 64 + ((c ^ (r & 15)) << 2)) = b;
        }
    }
    __syncthreads();
    const int tx = tid & 15, ty = (tid >> 4) & 15;     // columns tx + 16 n of tile tj; rows ty + 16 m of tile ti
    if (act) {
        float res[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) res[m][n] = 0.0f;
#pragma unroll 2
        for (int g = 0; g < 16; ++g) {
            float4 a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) { const int r = ty + 16 * m; a[m] = *reinterpret_cast<const float4*>(sA + r * 64 + ((g ^ (r & 15)) << 2)); }
#pragma unroll
            for (int n = 0; n < 4; ++n) { const int c = tx + 16 * n; b[n] = *reinterpret_cast<const float4*>(sB + c * 64 + ((g ^ (c & 15)) << 2)); }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const float d0 = __fsub_rn(a[m].x, b[n].x), d1 = __fsub_rn(a[m].y, b[n].y), d2 = __fsub_rn(a[m].z, b[n].z), d3 = __fsub_rn(a[m].w, b[n].w);
                    float t = __fmul_rn(d0, d0);
                    t = __fadd_rn(t, __fmul_rn(d1, d1));
                    t = __fadd_rn(t, __fmul_rn(d2, d2));
                    t = __fadd_rn(t, __fmul_rn(d3, d3));
                    res[m][n] = __fadd_rn(res[m][n], t);
                }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int r = ti * 64 + ty + 16 * m, c = tj * 64 + tx + 16 * n;
                if (r < sd.nq && c < ncol) out[(size_t)r * ld + c] = res[m][n];
                sT[(ty + 16 * m) * 65 + tx + 16 * n] = res[m][n];
            }
    }
    if (xj || ti == tj) return;                        // uniform
    __syncthreads();
    if (act) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {              // mirrored tile: D[tj*64 + x][ti*64 + y] = D[ti*64 + y][tj*64 + x]
                const int x = ty + 16 * m, y = tx + 16 * n;
                const int r = tj * 64 + x, cc = ti * 64 + y;
                if (r < sd.nq && cc < sd.nq) sd.out[(size_t)r * sd.ld + cc] = sT[y * 65 + x];
            }
    }
}

// partial_keys [n_blocks][qpad][BF_KEEP] u64, partial_bound [n_blocks][qpad] f32 bits (block-major: the filter's lanes own consecutive
// queries of ONE block, so its records leave as full lines; query-major they were 16-byte pieces of 112 000 different lines per frame, and
// the write-back of those partial lines at the end of the launch was a third of launch A's wall time).  Grid (1-D): n_blocks x ceil(nq / 512) filter
// workgroups first, then sd.n_tiles distance-matrix workgroups.
template <int NG, int M>
__device__ __forceinline__ void knn_bf16_filter_body(float* s_dyn, int bid, const float* __restrict__ vocab_bf, const float* __restrict__ row_norm,
                                                     int n_rows, const float* __restrict__ queries, int nq, int qpad,
                                                     int tiles_per_block, int n_blocks, uint64_t* __restrict__ partial_keys,
                                                     uint32_t* __restrict__ partial_bound, const SelfdistJob& sd,
                                                     const int32_t* __restrict__ n_lo = nullptr) {
    // n_lo: the number of rows this search may see, on the device (a pipelined handle appends the previous frames' new words while
    // this launch runs: rows at or beyond n_lo[0] -- up to n_rows, the host's upper bound -- are masked with an infinite |row|^2)
    // (n_rows itself may be an ESTIMATE below the device's count -- the launch plan of a growing vocabulary: nothing at or beyond it is
    // ranked either, the re-rank scans from min(n_rows, n_lo[0]) on)
    const int lo_rows = min(n_lo ? n_lo[0] : 0x7fffffff, n_rows);
    // NG = 32-query groups per wave: 4 -> four waves, one per SIMD; 2 -> eight waves, two per SIMD (one wave's tile
    // synchronisation, LDS reads and top-3 update hide behind the other's MFMAs).  The workgroup covers BF_QB queries either way.
    static_assert(NG == 4 || NG == 2, "wave tile");
    constexpr int KH = 32;
    constexpr int NW = BF_QB / (NG * 32);          // waves per workgroup
    constexpr int QW = NG * 32;                    // queries per wave
    constexpr int DPW = 8 / NW;                    // DMA instructions of a tile issued by one wave
    // the filter workgroups come FIRST in the grid: a workgroup holds a whole compute unit's LDS, workgroups are dispatched in index order,
    // and the (short) distance-matrix tiles in front used to take 136 of the 256 compute units at the start of the launch -- a third of the
    // filter workgroups then began only when a tile, or another filter workgroup, had finished (entry spread 0 .. 12 us for a 12 us workgroup)
    const int n_fwg = n_blocks * ((nq + BF_QB - 1) / BF_QB);
    if (bid >= n_fwg) { selfdist_tile(sd, bid - n_fwg, s_dyn); return; }
    const int fb = bid;
    const int bx = fb % n_blocks, by = fb / n_blocks;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int q0 = by * BF_QB + wave * QW;
    float* s_q = s_dyn + (size_t)wave * NG * BF_TILE_F;             // this wave's query staging (prologue only)
    float* s_tile = s_dyn + (size_t)NW * NG * BF_TILE_F;      // [2] vocabulary tiles shared by the workgroup
    float* s_aug = s_tile + 2 * BF_TILE_F;                    // [MF_STRIP_TILES][64] augmentation entries of the strip, per lane
    MF_STAMP(0);

    const int tile0 = bx * tiles_per_block;
    const int n_tiles = (n_rows + 31) / 32;
    const int tile1 = min(tile0 + tiles_per_block, n_tiles);

    // Everything the prologue needs is put in flight at once: the wave's query groups, its share of the first TWO vocabulary tiles
    // and the augmentation entries of every tile of the strip.  The strip's other tiles follow as soon as the query staging area
    // is free (below): the whole strip is requested long before it is needed -- with one tile of look-ahead the loop ran at the
    // memory LATENCY (a tile trip is shorter than a round trip to HBM), not at the matrix rate.
#pragma unroll
    for (int g = 0; g < NG; ++g) dma_a_tile<KH>(queries, nq, q0 / 32 + g, lane, s_q + g * BF_TILE_F);
    float augs[MF_STRIP_TILES];
#pragma unroll
    for (int i = 0; i < MF_STRIP_TILES; ++i) {
        const int t = min(tile0 + i, max(tile1 - 1, tile0));
        augs[i] = row_norm[2 * (size_t)min(t * 32 + col, n_rows) + half];
    }
    if (tile0 < tile1) {
        dma_tile_part<M == 1>(vocab_bf, n_rows, tile0, lane, s_tile, DPW * wave, DPW * wave + DPW);
        dma_tile_part<M == 1>(vocab_bf, n_rows, min(tile0 + 1, tile1 - 1), lane, s_tile + BF_TILE_F, DPW * wave, DPW * wave + DPW);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 0) {                                                 // the loop takes them from LDS: a VMEM load there would wait behind the whole prefetch
#pragma unroll
        for (int i = 0; i < MF_STRIP_TILES; ++i) {
            const int t = min(tile0 + i, max(tile1 - 1, tile0));
            s_aug[i * 64 + lane] = (half == 0 && t * 32 + col >= lo_rows) ? __int_as_float(0x7f800000) : augs[i];
        }
    }

    // B operands: -2 q split hi/lo in operand order (lane (query l&31, half l>>5) holds floats [32h, 32h + 32) of its query: k-step s
    // multiplies elements 32h + 8s .. + 8 -- A uses the same k permutation), + |q|^2 for the augmentation step
    uint4 bh[NG][4], bl[NG][4];
    float b_aug[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float x[KH];
        read_a_tile<KH>(s_q + g * BF_TILE_F, col, half, x);
        float part = 0.0f;
#pragma unroll
        for (int k = 0; k < KH; ++k) part = fmaf(x[k], x[k], part);
        const float qn = part + __shfl_xor(part, 32, 64);
        b_aug[g] = half == 0 ? 1.0f : qn;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            op_split2<M>(-2.0f * x[8 * s + 0], -2.0f * x[8 * s + 1], bh[g][s].x, bl[g][s].x);
            op_split2<M>(-2.0f * x[8 * s + 2], -2.0f * x[8 * s + 3], bh[g][s].y, bl[g][s].y);
            op_split2<M>(-2.0f * x[8 * s + 4], -2.0f * x[8 * s + 5], bh[g][s].z, bl[g][s].z);
            op_split2<M>(-2.0f * x[8 * s + 6], -2.0f * x[8 * s + 7], bh[g][s].w, bl[g][s].w);
        }
    }

    int32_t k0[NG], k1[NG], k2[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { k0[g] = MF_KEY_NONE; k1[g] = MF_KEY_NONE; k2[g] = MF_KEY_NONE; }
    f32x16 p0, p1;                                                   // pending accumulators (groups NG-2, NG-1 of the previous tile)
    f32x16 r0, r1;                                                   // NG == 2: the other pair (odd tiles)
    // every wave has its queries in registers (and the first two tiles have landed for everybody): the staging area now takes
    // tiles 2.. of the strip, all requested at once
    __syncthreads();
    for (int t = tile0 + 2; t < tile1; ++t)
        dma_tile_part<M == 1>(vocab_bf, n_rows, t, lane, s_dyn + (size_t)(t - tile0 - 2) * BF_TILE_F, DPW * wave, DPW * wave + DPW);
    MF_STAMP(1);
    for (int t = tile0; t < tile1; ++t) {
        const int ti = t - tile0;
        MF_STAMP2(ti);
        if (ti == 2) {                                               // tiles 2.. : one wait and one barrier for all of them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // A operands: hi chunks 4h .. 4h+3 and lo chunks 8+4h .. 8+4h+3 of row `col` (16-byte chunks, XOR-swizzled).  The reads
        // are issued as inline assembly: the compiler orders every LDS read it knows of behind ALL outstanding LDS-DMA
        // (s_waitcnt vmcnt(0)), which would serialise the loop behind the strip's prefetch.
        uint4 ah[4], al[4];
        float aug;
        {
            const float* slot = ti < 2 ? s_tile + ti * BF_TILE_F : s_dyn + (size_t)(ti - 2) * BF_TILE_F;
            const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(slot + col * 64);
            uint32_t addr[8];
            uint4 av[8];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                addr[v] = base + ((((uint32_t)(4 * half + v)) ^ (uint32_t)(col & 15)) << 4);
                addr[4 + v] = base + ((((uint32_t)(8 + 4 * half + v)) ^ (uint32_t)(col & 15)) << 4);
            }
            lds_read_ops<M>(addr, av, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(s_aug + ti * 64 + lane), aug);
#pragma unroll
            for (int v = 0; v < 4; ++v) { ah[v] = av[v]; al[v] = av[4 + v]; }
        }
        const uint32_t tl = (uint32_t)ti;
        if constexpr (NG == 4) {
            // two accumulator pairs take turns (no copies): groups 0,1 are computed into (x0, x1) while the pending scores of
            // groups 2,3 of the previous tile (p0, p1) are pushed, then groups 2,3 into (p0, p1) while (x0, x1) are pushed
            f32x16 x0, x1;
            if (t == tile0) bf_pair<false, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], x0, x1, p0, p1, 0u, k0[2], k1[2], k2[2],
                                           k0[3], k1[3], k2[3]);
            else bf_pair<true, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], x0, x1, p0, p1, tl - 1u, k0[2], k1[2], k2[2], k0[3],
                               k1[3], k2[3]);
            bf_pair<true, M>(ah, al, aug, bh[2], bl[2], b_aug[2], bh[3], bl[3], b_aug[3], p0, p1, x0, x1, tl, k0[0], k1[0], k2[0], k0[1], k1[1], k2[1]);
        } else {
            // one pair per tile: the two accumulator pairs take turns from tile to tile (even tiles -> (p0, p1), odd -> (r0, r1))
            if (t == tile0) bf_pair<false, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], p0, p1, r0, r1, 0u, k0[0], k1[0], k2[0],
                                           k0[1], k1[1], k2[1]);
            else if (tl & 1u) bf_pair<true, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], r0, r1, p0, p1, tl - 1u, k0[0], k1[0],
                                            k2[0], k0[1], k1[1], k2[1]);
            else bf_pair<true, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], p0, p1, r0, r1, tl - 1u, k0[0], k1[0], k2[0], k0[1],
                               k1[1], k2[1]);
        }
    }
    if (tile0 < tile1) {
        const uint32_t tlast = (uint32_t)(tile1 - 1 - tile0);
        if (NG == 2 && (tlast & 1u)) {
            push_group4(r0, tlast, k0[0], k1[0], k2[0]);
            push_group4(r1, tlast, k0[1], k1[1], k2[1]);
        } else {
            push_group4(p0, tlast, k0[NG - 2], k1[NG - 2], k2[NG - 2]);
            push_group4(p1, tlast, k0[NG - 1], k1[NG - 1], k2[NG - 1]);
        }
    }
    MF_STAMP(2);

    // the two halves of a query's rows meet in registers: best two of the six keys, and the third as the bound on everything
    // this workgroup dropped for the query
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const uint64_t a0 = widen_key(k0[g], tile0, half), a1 = widen_key(k1[g], tile0, half), a2 = widen_key(k2[g], tile0, half);
        const uint64_t b0 = shfl_xor_u64(a0, 32), b1 = shfl_xor_u64(a1, 32), b2 = shfl_xor_u64(a2, 32);
        const uint64_t m0 = a0 < b0 ? a0 : b0;
        const uint64_t hx = a0 < b0 ? b0 : a0, lx = a1 < b1 ? a1 : b1;
        const uint64_t m1 = hx < lx ? hx : lx;
        const uint64_t third = third_of_two_triples(a0, a1, a2, b0, b1, b2);
        const int qi = q0 + g * 32 + col;
        if (half == 0 && qi < qpad) {
            uint64_t* dst = partial_keys + ((size_t)bx * qpad + qi) * BF_KEEP;        // block-major: a wave's 32 queries make one 512-byte write
            dst[0] = m0;
            dst[1] = m1;
            partial_bound[(size_t)bx * qpad + qi] = (uint32_t)min(third >> 32, (uint64_t)0x7f800000u);
        }
    }
    MF_STAMP(3);
}

// ------------------------------------------------------------------------------------------------ pre-split queries (pipelined frames)
// Every strip's workgroup of the body above stages the SAME 512 queries through 128 KB of LDS and splits them into bf16 operands --
// 192 times per frame, and the staging area is what makes a filter workgroup own a whole compute unit's LDS.  A pipelined handle
// knows a frame one launch before its filter runs: the launch that carries the previous frame's filter also converts the new
// frame's queries ONCE (qsplit_body, a handful of small workgroups) into MFMA operand order in global memory:
//     qsplit[(((query / 32) * 4 + s) * 2 + kind) * 64 + lane]   uint4 = eight bf16 of  -2 q[32 * (lane >> 5) + 8 s .. + 8)  (kind 0: hi, 1: lo)
//     qnorm[query] = |q|^2
// so that the filter's prologue is 32 coalesced 16-byte loads per lane straight into the registers the operands live in, its LDS holds
// only the strip (8 tiles + augmentation entries = 66 KB) and TWO workgroups share a compute unit.
constexpr size_t BF_LDS_BYTES_Q = (size_t)MF_STRIP_TILES * BF_TILE_F * 4 + (size_t)MF_STRIP_TILES * 64 * 4;

__device__ __forceinline__ void qsplit_item(const QSplitArgs& qs, int t, const float4& a, const float4& b) {
    const int qi = t >> 3, h = (t >> 2) & 1, sx = t & 3;
    uint4 hi, lo;
    op_split2_rt(qs.f16, -2.0f * a.x, -2.0f * a.y, hi.x, lo.x);
    op_split2_rt(qs.f16, -2.0f * a.z, -2.0f * a.w, hi.y, lo.y);
    op_split2_rt(qs.f16, -2.0f * b.x, -2.0f * b.y, hi.z, lo.z);
    op_split2_rt(qs.f16, -2.0f * b.z, -2.0f * b.w, hi.w, lo.w);
    const size_t base = ((size_t)(qi >> 5) * 4 + sx) * 2;
    qs.qsplit[(base + 0) * 64 + h * 32 + (qi & 31)] = hi;
    qs.qsplit[(base + 1) * 64 + h * 32 + (qi & 31)] = lo;
    float part = fmaf(b.w, b.w, fmaf(b.z, b.z, fmaf(b.y, b.y, fmaf(b.x, b.x, fmaf(a.w, a.w, fmaf(a.z, a.z, fmaf(a.y, a.y, a.x * a.x)))))));
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    if ((t & 7) == 0) qs.qnorm[qi] = part;
    if (qs.shadow_bf) {                                                  // the descriptor as a ROW of an operand table (vocab_bf16_kernel's layout): floats [32 h + 8 sx, + 8)
        uint4 rh, rl;
        op_split2_rt(qs.f16, a.x, a.y, rh.x, rl.x);
        op_split2_rt(qs.f16, a.z, a.w, rh.y, rl.y);
        op_split2_rt(qs.f16, b.x, b.y, rh.z, rl.z);
        op_split2_rt(qs.f16, b.z, b.w, rh.w, rl.w);
        uint4* row = reinterpret_cast<uint4*>(qs.shadow_bf + (size_t)qi * 64);
        row[4 * h + sx] = rh;
        row[8 + 4 * h + sx] = rl;
        if ((t & 7) == 0) {                                              // padding rows (they repeat the last descriptor) never rank: |row|^2 = +inf
            qs.shadow_norm[2 * (size_t)qi] = qi < qs.nq ? part : __int_as_float(0x7f800000);
            qs.shadow_norm[2 * (size_t)qi + 1] = 1.0f;
            if (t == 0) { qs.shadow_norm[2 * (size_t)qs.qpad] = __int_as_float(0x7f800000); qs.shadow_norm[2 * (size_t)qs.qpad + 1] = 1.0f; }   // the sentinel
        }
        // the filter's error bound is made from the largest |row|^2 the filter may have multiplied: these rows are among them from the next launch on
        // (a running maximum: raising it early only widens the bound); one atomic per wave
        float nm = qi < qs.nq ? part : 0.0f;
#pragma unroll
        for (int m = 32; m >= 8; m >>= 1) nm = fmaxf(nm, __shfl_xor(nm, m, 64));
        if (qs.norm_max_bits && (threadIdx.x & 63) == 0 && nm > 0.0f) atomicMax(qs.norm_max_bits, __float_as_uint(nm));
    }
}
// One item = eight floats of one query.  A thread's items are READ first and written afterwards: loads and stores share one in-order
// counter, so a second trip's loads behind a first trip's stores wait for a store round trip that carries no data (round 4's stamps:
// the eight two-trip workgroups took 17 us, the longest chain of launch A).  Frames of up to 1 024 descriptors get one item per thread.
__device__ __forceinline__ void qsplit_body(const QSplitArgs& qs, int wg) {
    const int n_items = qs.qpad * 8, stride = qs.n_wgs * (int)blockDim.x;   // (qpad * 8 is a multiple of 64: a wave's lanes take part together)
    for (int t0 = wg * (int)blockDim.x + (int)threadIdx.x; t0 < n_items; t0 += 2 * stride) {
        const int t1 = t0 + stride;
        const bool two = t1 < n_items;
        const int tt = two ? t1 : t0;
        const float4* s0 = reinterpret_cast<const float4*>(qs.queries + (size_t)min(t0 >> 3, qs.nq - 1) * 64 + 32 * ((t0 >> 2) & 1) + 8 * (t0 & 3));   // padding repeats the last query
        const float4* s1 = reinterpret_cast<const float4*>(qs.queries + (size_t)min(tt >> 3, qs.nq - 1) * 64 + 32 * ((tt >> 2) & 1) + 8 * (tt & 3));
        const float4 a0 = s0[0], b0 = s0[1], a1 = s1[0], b1 = s1[1];
        qsplit_item(qs, t0, a0, b0);
        if (two) qsplit_item(qs, t1, a1, b1);
    }
}

// the filter body over pre-split queries: as knn_bf16_filter_body<4> (one strip per workgroup), without the query staging
template <int M>
__device__ __forceinline__ void knn_bf16_filter_body_q(float* s_dyn, int bid, const float* __restrict__ vocab_bf, const float* __restrict__ row_norm,
                                                       int n_rows, const uint4* qsplit, const float* qnorm, int nq, int qpad,
                                                       int tiles_per_block, int n_blocks, uint64_t* __restrict__ partial_keys,
                                                       uint32_t* __restrict__ partial_bound, const SelfdistJob& sd, const int32_t* __restrict__ n_lo) {
    constexpr int NG = 4;
    constexpr int NW = MF_WAVES;
    constexpr int QW = NG * 32;
    constexpr int DPW = 8 / NW;
    // the rows this search may see (see knn_bf16_filter_body), through the scalar cache: a vector load would sit in the same in-order
    // queue as the strip's requests below, and its first use would wait for all of them
    int lo_rows = 0x7fffffff;
    if (n_lo) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lo_rows) : "s"(n_lo) : "memory");
    lo_rows = min(lo_rows, n_rows);                                       // (the plan's n_rows may be an estimate below the device's count)
    const int n_fwg = n_blocks * ((nq + BF_QB - 1) / BF_QB);
    if (bid >= n_fwg) { selfdist_tile(sd, bid - n_fwg, s_dyn); return; }
    const int bx = bid % n_blocks, by = bid / n_blocks;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int q0 = by * BF_QB + wave * QW;
    float* s_aug = s_dyn + (size_t)MF_STRIP_TILES * BF_TILE_F;          // [MF_STRIP_TILES][64]
    MF_STAMP(0);
    const int tile0 = bx * tiles_per_block;
    const int n_tiles = (n_rows + 31) / 32;
    const int tile1 = min(tile0 + tiles_per_block, n_tiles);
    // First what the loop needs to start -- two tiles, the augmentation entries, the query operands -- and, once that has arrived, the
    // rest of the strip, which lands while the first two tiles are multiplied.  (Requested all at once and awaited in front of the loop,
    // the strip cost 3.6 us per workgroup: every workgroup of the launch asks at the same moment, so the last byte of anybody's strip
    // arrives when the whole vocabulary has crossed the fabric.  Requested all at once and awaited tile by tile is not expressible:
    // the compiler waits for EVERY outstanding request at the first use of an operand register while LDS-DMA is in flight.)
    const int tile_last = max(tile1 - 1, tile0);
    dma_tile_part<M == 1>(vocab_bf, n_rows, tile0, lane, s_dyn, DPW * wave, DPW * wave + DPW);
    dma_tile_part<M == 1>(vocab_bf, n_rows, min(tile0 + 1, tile_last), lane, s_dyn + BF_TILE_F, DPW * wave, DPW * wave + DPW);
    // the augmentation entries of the strip go straight to LDS as well, two tiles per wave (the mask of rows that do not exist yet
    // is patched into them below)
    static_assert(MF_STRIP_TILES == 2 * NW, "two augmentation rows per wave");
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = 2 * wave + j;
        const int t = min(tile0 + i, tile_last);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row_norm + 2 * (size_t)min(t * 32 + col, n_rows) + half),
                                         (__attribute__((address_space(3))) void*)(s_aug + i * 64), 4, 0, 0);
    }
    uint4 bh[NG][4], bl[NG][4];
    float b_aug[NG];
    const int grp0 = q0 >> 5;
    const int n_grp = qpad >> 5;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int grp = min(grp0 + g, n_grp - 1);                       // (a wave beyond the padded queries repeats the last group: its keys are not written)
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            const size_t base = ((size_t)grp * 4 + sx) * 2;
            bh[g][sx] = qsplit[(base + 0) * 64 + lane];
            bl[g][sx] = qsplit[(base + 1) * 64 + lane];
        }
        const float qn = qnorm[min(grp * 32 + col, qpad - 1)];
        b_aug[g] = half == 0 ? 1.0f : qn;
    }
    // (a use of the youngest request here: the compiler waits for it -- and with it for everything older -- at this point, and knows
    // from then on that the operand registers are complete; left alone it would wait at their first use, behind the requests below)
    asm volatile("" : "+v"(b_aug[NG - 1]) : : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int t = tile0 + 2; t < tile1; ++t)
        dma_tile_part<M == 1>(vocab_bf, n_rows, t, lane, s_dyn + (size_t)(t - tile0) * BF_TILE_F, DPW * wave, DPW * wave + DPW);
    if ((tile0 + MF_STRIP_TILES) * 32 > lo_rows) {                       // (rare: the strip reaches rows that are being appended while this launch runs)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * wave + j;
            if (half == 0 && (tile0 + i) * 32 + col >= lo_rows)
                asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(s_aug + i * 64 + lane)),
                             "v"(__int_as_float(0x7f800000)) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                        // every wave's share of tiles 0, 1 and the augmentation entries are in LDS
    MF_STAMP(1);
    int32_t k0[NG], k1[NG], k2[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { k0[g] = MF_KEY_NONE; k1[g] = MF_KEY_NONE; k2[g] = MF_KEY_NONE; }
    f32x16 p0, p1;
    for (int t = tile0; t < tile1; ++t) {
        const int ti = t - tile0;
        if (ti == 2) {                                                   // tiles 2.. : one wait and one barrier for all of them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        uint4 ah[4], al[4];
        float aug;
        {
            const float* slot = s_dyn + (size_t)ti * BF_TILE_F;
            const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(slot + col * 64);
            uint32_t addr[8];
            uint4 av[8];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                addr[v] = base + ((((uint32_t)(4 * half + v)) ^ (uint32_t)(col & 15)) << 4);
                addr[4 + v] = base + ((((uint32_t)(8 + 4 * half + v)) ^ (uint32_t)(col & 15)) << 4);
            }
            lds_read_ops<M>(addr, av, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(s_aug + ti * 64 + lane), aug);
#pragma unroll
            for (int v = 0; v < 4; ++v) { ah[v] = av[v]; al[v] = av[4 + v]; }
        }
        const uint32_t tl = (uint32_t)ti;
        f32x16 x0, x1;
        if (t == tile0) bf_pair<false, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], x0, x1, p0, p1, 0u, k0[2], k1[2], k2[2],
                                       k0[3], k1[3], k2[3]);
        else bf_pair<true, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], x0, x1, p0, p1, tl - 1u, k0[2], k1[2], k2[2], k0[3],
                           k1[3], k2[3]);
        bf_pair<true, M>(ah, al, aug, bh[2], bl[2], b_aug[2], bh[3], bl[3], b_aug[3], p0, p1, x0, x1, tl, k0[0], k1[0], k2[0], k0[1], k1[1], k2[1]);
    }
    if (tile0 < tile1) {
        const uint32_t tlast = (uint32_t)(tile1 - 1 - tile0);
        push_group4(p0, tlast, k0[NG - 2], k1[NG - 2], k2[NG - 2]);
        push_group4(p1, tlast, k0[NG - 1], k1[NG - 1], k2[NG - 1]);
    }
    MF_STAMP(2);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const uint64_t a0 = widen_key(k0[g], tile0, half), a1 = widen_key(k1[g], tile0, half), a2 = widen_key(k2[g], tile0, half);
        const uint64_t b0 = shfl_xor_u64(a0, 32), b1 = shfl_xor_u64(a1, 32), b2 = shfl_xor_u64(a2, 32);
        const uint64_t m0 = a0 < b0 ? a0 : b0;
        const uint64_t hx = a0 < b0 ? b0 : a0, lx = a1 < b1 ? a1 : b1;
        const uint64_t m1 = hx < lx ? hx : lx;
        const uint64_t third = third_of_two_triples(a0, a1, a2, b0, b1, b2);
        const int qi = q0 + g * 32 + col;
        if (half == 0 && qi < qpad) {
            uint64_t* dst = partial_keys + ((size_t)bx * qpad + qi) * BF_KEEP;        // block-major: a wave's 32 queries make one 512-byte write
            dst[0] = m0;
            dst[1] = m1;
            partial_bound[(size_t)bx * qpad + qi] = (uint32_t)min(third >> 32, (uint64_t)0x7f800000u);
        }
    }
    MF_STAMP(3);
}

// ------------------------------------------------------------------------------------------------ shadow scores (round 6)
// The words frame t-2 is about to create (its decision loop rides in THIS launch) are not rows of the vocabulary when the filter of frame t-1 runs
// beside it -- but they are descriptors of frame t-2, and that frame's query pre-split left ALL its descriptors as rows of an operand table (256 B
// each, the layout of vocab_bf; QSplitArgs::shadow_bf).  One workgroup per 32-row tile of that table multiplies it with the frame's 512 pre-split
// queries exactly as a filter strip does (the same MFMA chains on top of |v|^2 + |q|^2: the same error bound) and -- instead of selecting -- writes
// every score: x[query][descriptor], ld floats per query.  Launch B's re-rank, which knows by then which of those descriptors became words, reads
// its query's row, keeps the words' scores at or below its threshold and evaluates them exactly with its other candidates (knn_mfma_rerank_body):
// nobody stages or scans the ~150 new rows any more (250 workgroups x 38 KB and ~4.5 us of the re-rank's chain in round 5).
template <int M>
__device__ __forceinline__ void shadow_scores_body(float* s_dyn, int wg, const float* __restrict__ sh_bf, const float* __restrict__ sh_norm, int sh_rows,
                                                   const uint4* qsplit, const float* qnorm, int nq, int qpad, float* __restrict__ x, int ld) {
    constexpr int NG = 4, NW = MF_WAVES, QW = NG * 32, DPW = 8 / NW;
    const int n_tiles = (sh_rows + 31) / 32;
    const int t = wg % n_tiles, by = wg / n_tiles;                       // tile of the table, block of 512 queries
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int q0 = by * BF_QB + wave * QW;
    float* s_aug = s_dyn + (size_t)BF_TILE_F;
    dma_tile_part<M == 1>(sh_bf, sh_rows, t, lane, s_dyn, DPW * wave, DPW * wave + DPW);
    if (wave == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sh_norm + 2 * (size_t)min(t * 32 + col, sh_rows) + half),
                                         (__attribute__((address_space(3))) void*)s_aug, 4, 0, 0);
    uint4 bh[NG][4], bl[NG][4];
    float b_aug[NG];
    const int grp0 = q0 >> 5, n_grp = qpad >> 5;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int grp = min(grp0 + g, n_grp - 1);
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            const size_t base = ((size_t)grp * 4 + sx) * 2;
            bh[g][sx] = qsplit[(base + 0) * 64 + lane];
            bl[g][sx] = qsplit[(base + 1) * 64 + lane];
        }
        const float qn = qnorm[min(grp * 32 + col, qpad - 1)];
        b_aug[g] = half == 0 ? 1.0f : qn;
    }
    asm volatile("" : "+v"(b_aug[NG - 1]) : : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint4 ah[4], al[4];
    float aug;
    {
        const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(s_dyn + col * 64);
        uint32_t addr[8];
        uint4 av[8];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            addr[v] = base + ((((uint32_t)(4 * half + v)) ^ (uint32_t)(col & 15)) << 4);
            addr[4 + v] = base + ((((uint32_t)(8 + 4 * half + v)) ^ (uint32_t)(col & 15)) << 4);
        }
        lds_read_ops<M>(addr, av, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(s_aug + lane), aug);
#pragma unroll
        for (int v = 0; v < 4; ++v) { ah[v] = av[v]; al[v] = av[4 + v]; }
    }
    __builtin_amdgcn_s_barrier();                                        // every wave has its operands: the tile's LDS is free (the scores cross it below)
    f32x16 c[NG];
    int32_t kd0 = MF_KEY_NONE, kd1 = MF_KEY_NONE, kd2 = MF_KEY_NONE, kd3 = MF_KEY_NONE, kd4 = MF_KEY_NONE, kd5 = MF_KEY_NONE;   // (bf_pair<false> touches no keys)
    const f32x16 none = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bf_pair<false, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], c[0], c[1], none, none, 0u, kd0, kd1, kd2, kd3, kd4, kd5);
    bf_pair<false, M>(ah, al, aug, bh[2], bl[2], b_aug[2], bh[3], bl[3], b_aug[3], c[2], c[3], none, none, 0u, kd0, kd1, kd2, kd3, kd4, kd5);
    // Accumulator register r of lane (col, half) is row (r & 3) + 8 (r >> 2) + 4 half of the tile for query col of the group: stored straight from
    // the registers a query's 128 bytes would leave as eight 16-byte pieces of four different instructions (partial lines: the workgroups ended at
    // 11.4-12 us, 5 us of it waiting for those stores to be acknowledged).  So the scores cross LDS once -- each wave its own 128 queries x 32 rows
    // = 16 KB, 16-byte chunk k of a query at position k ^ (query & 7): no bank conflicts either way -- and leave as whole 128-byte lines, eight
    // queries per instruction.  (The barrier in front of the MFMAs has seen every wave finish reading the tile: its LDS is free.)
    float* xs = s_dyn + (size_t)wave * (QW * 32);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int ql = g * 32 + col;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            *reinterpret_cast<float4*>(xs + (size_t)ql * 32 + (((2 * m + half) ^ (ql & 7)) << 2)) = make_float4(c[g][4 * m], c[g][4 * m + 1], c[g][4 * m + 2], c[g][4 * m + 3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // (the wave reads back what the wave wrote: no barrier)
#pragma unroll 4
    for (int it = 0; it < QW / 8; ++it) {
        const int ql = it * 8 + (lane >> 3), ck = lane & 7;
        const int qi = q0 + ql;
        const float4 v = *reinterpret_cast<const float4*>(xs + (size_t)ql * 32 + ((ck ^ (ql & 7)) << 2));
        if (qi < nq) *reinterpret_cast<float4*>(x + (size_t)qi * ld + t * 32 + 4 * ck) = v;
    }
}

// ------------------------------------------------------------------------------------------------ persistent variant
// Vocabularies of more than BF_PX strips (> ~60k words): a workgroup keeps its queries in registers and walks several strips
// (strip bx0, bx0 + px, ...) instead of staging and splitting the same 512 queries once per strip -- with one workgroup per compute
// unit (146 KB of LDS each) the strips of the non-persistent launch ran in ceil(strips / 256) rounds of prologue + loop + epilogue.
// LDS holds two strips: the one being multiplied and the next one, requested (LDS-DMA, augmentation entries included) as soon as
// the strip before it has been read by every wave.  Strip 0 uses the slots of the one-strip kernel (tiles 0,1 behind the query
// staging area, tiles 2.. in it), odd strips slots 8..15, even strips >= 2 slots 0..7.
constexpr size_t BF_LDS_BYTES_P = (size_t)(MF_WAVES * 4 + 2) * BF_TILE_F * 4 + (size_t)2 * MF_STRIP_TILES * 64 * 4;
// every tile of strip `bx` + its augmentation entries (wave 0): a FIXED number of DMA instructions per wave, so that the wait in
// front of the strip before it can name how many may stay in flight
// the augmentation entry of a column that is not a visible row
__device__ const float g_aug_inf[2] = {__builtin_inff(), 1.0f};
template <int M>
__device__ __forceinline__ void bf_request_strip(const float* __restrict__ vocab_bf, const float* __restrict__ row_norm, int n_rows, int bx,
                                                 int tiles_per_block, int n_tiles, int lane, int wave, int col, int half, float* slots, float* aug_dst) {
    constexpr int DPW = 8 / MF_WAVES;
    const int tile0 = bx * tiles_per_block;
    const int tile1 = min(tile0 + tiles_per_block, n_tiles);
#pragma unroll
    for (int i = 0; i < MF_STRIP_TILES; ++i) {
        const int t = min(tile0 + i, max(tile1 - 1, tile0));
        dma_tile_part<M == 1>(vocab_bf, n_rows, t, lane, slots + (size_t)i * BF_TILE_F, DPW * wave, DPW * wave + DPW);
    }
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < MF_STRIP_TILES; ++i) {
            const int t = min(tile0 + i, max(tile1 - 1, tile0));
            // (columns at or beyond n_rows read a constant {+inf, 1}: the table's own entry behind the last visible row is where the
            // appender of the previous frame -- a workgroup of this same launch on a pipelined handle -- writes its first row's norm)
            const float* src = t * 32 + col >= n_rows ? g_aug_inf + half : row_norm + 2 * (size_t)(t * 32 + col) + half;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(aug_dst + i * 64), 4, 0, 0);
        }
    }
}
// everything issued before the newest strip request has arrived (vector memory operations of gfx9 complete in issue order)
__device__ __forceinline__ void bf_wait_all_but_request(int wave) {
    constexpr int DPW = 8 / MF_WAVES;
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(MF_STRIP_TILES * DPW + MF_STRIP_TILES) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(MF_STRIP_TILES * DPW) : "memory");
}
// a barrier that waits for this wave's LDS traffic only (__syncthreads() would also wait for the strip in flight)
__device__ __forceinline__ void bf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int M>
__device__ __forceinline__ void knn_bf16_filter_body_p(float* s_dyn, int bid, const float* __restrict__ vocab_bf, const float* __restrict__ row_norm,
                                                       int n_rows, const float* __restrict__ queries, int nq, int qpad,
                                                       int tiles_per_block, int n_blocks, int px, uint64_t* __restrict__ partial_keys,
                                                       uint32_t* __restrict__ partial_bound, const SelfdistJob& sd,
                                                       const int32_t* __restrict__ n_lo = nullptr) {
    if (n_lo) n_rows = min(n_rows, max(n_lo[0], 1));                 // rows the search may see (see knn_bf16_filter_body); the sentinel entry follows them
    constexpr int NG = 4;
    constexpr int KH = 32;
    constexpr int NW = BF_QB / (NG * 32);
    constexpr int QW = NG * 32;
    constexpr int DPW = 8 / NW;
    const int n_fwg = px * ((nq + BF_QB - 1) / BF_QB);                // filter workgroups first, distance-matrix tiles behind them (see above)
    if (bid >= n_fwg) { selfdist_tile(sd, bid - n_fwg, s_dyn); return; }
    const int fb = bid;
    const int bx0 = fb % px, by = fb / px;
    const int n_my = (n_blocks - bx0 + px - 1) / px;                 // strips of this workgroup: bx0, bx0 + px, ...
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int q0 = by * BF_QB + wave * QW;
    float* s_q = s_dyn + (size_t)wave * NG * BF_TILE_F;
    float* s_tile = s_dyn + (size_t)NW * NG * BF_TILE_F;
    float* s_aug = s_tile + 2 * BF_TILE_F;                           // [2][MF_STRIP_TILES][64]
    const int n_tiles = (n_rows + 31) / 32;
    // ---- prologue: as the one-strip kernel, for strip bx0
    {
        const int tile0 = bx0 * tiles_per_block;
        const int tile1 = min(tile0 + tiles_per_block, n_tiles);
#pragma unroll
        for (int g = 0; g < NG; ++g) dma_a_tile<KH>(queries, nq, q0 / 32 + g, lane, s_q + g * BF_TILE_F);
        float augs[MF_STRIP_TILES];
#pragma unroll
        for (int i = 0; i < MF_STRIP_TILES; ++i) {
            const int t = min(tile0 + i, max(tile1 - 1, tile0));
            augs[i] = t * 32 + col >= n_rows ? g_aug_inf[half] : row_norm[2 * (size_t)(t * 32 + col) + half];
        }
        if (tile0 < tile1) {
            dma_tile_part<M == 1>(vocab_bf, n_rows, tile0, lane, s_tile, DPW * wave, DPW * wave + DPW);
            dma_tile_part<M == 1>(vocab_bf, n_rows, min(tile0 + 1, tile1 - 1), lane, s_tile + BF_TILE_F, DPW * wave, DPW * wave + DPW);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < MF_STRIP_TILES; ++i) s_aug[i * 64 + lane] = augs[i];
        }
    }
    uint4 bh[NG][4], bl[NG][4];
    float b_aug[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float x[KH];
        read_a_tile<KH>(s_q + g * BF_TILE_F, col, half, x);
        float part = 0.0f;
#pragma unroll
        for (int k = 0; k < KH; ++k) part = fmaf(x[k], x[k], part);
        const float qn = part + __shfl_xor(part, 32, 64);
        b_aug[g] = half == 0 ? 1.0f : qn;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            op_split2<M>(-2.0f * x[8 * s + 0], -2.0f * x[8 * s + 1], bh[g][s].x, bl[g][s].x);
            op_split2<M>(-2.0f * x[8 * s + 2], -2.0f * x[8 * s + 3], bh[g][s].y, bl[g][s].y);
            op_split2<M>(-2.0f * x[8 * s + 4], -2.0f * x[8 * s + 5], bh[g][s].z, bl[g][s].z);
            op_split2<M>(-2.0f * x[8 * s + 6], -2.0f * x[8 * s + 7], bh[g][s].w, bl[g][s].w);
        }
    }
    __syncthreads();                                                 // every wave has its queries in registers: the staging area is free
    {
        const int tile0 = bx0 * tiles_per_block;
        const int tile1 = min(tile0 + tiles_per_block, n_tiles);
        for (int t = tile0 + 2; t < tile1; ++t)
            dma_tile_part<M == 1>(vocab_bf, n_rows, t, lane, s_dyn + (size_t)(t - tile0 - 2) * BF_TILE_F, DPW * wave, DPW * wave + DPW);
    }
    if (n_my > 1) bf_request_strip<M>(vocab_bf, row_norm, n_rows, bx0 + px, tiles_per_block, n_tiles, lane, wave, col, half,
                                   s_dyn + (size_t)8 * BF_TILE_F, s_aug + MF_STRIP_TILES * 64);
    // ---- the strips
    for (int s = 0; s < n_my; ++s) {
        const int bx = bx0 + s * px;
        const int tile0 = bx * tiles_per_block;
        const int tile1 = min(tile0 + tiles_per_block, n_tiles);
        if (s > 0) {
            // strip s was requested one strip ago; behind it only the request of strip s + 1 (a fixed number of instructions) may
            // still be in flight
            if (s + 1 < n_my) bf_wait_all_but_request(wave); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bf_lds_barrier();
        }
        int32_t k0[NG], k1[NG], k2[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { k0[g] = MF_KEY_NONE; k1[g] = MF_KEY_NONE; k2[g] = MF_KEY_NONE; }
        f32x16 p0, p1;
        const float* aug_s = s_aug + (s & 1) * (MF_STRIP_TILES * 64);
        for (int t = tile0; t < tile1; ++t) {
            const int ti = t - tile0;
            if (s == 0 && ti == 2) {                                 // strip 0, tiles 2.. : one wait and one barrier for all of them
                if (n_my > 1) bf_wait_all_but_request(wave); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bf_lds_barrier();
            }
            uint4 ah[4], al[4];
            float aug;
            {
                const float* slot = s == 0 ? (ti < 2 ? s_tile + ti * BF_TILE_F : s_dyn + (size_t)(ti - 2) * BF_TILE_F)
                                           : s_dyn + (size_t)(((s & 1) ? 8 : 0) + ti) * BF_TILE_F;
                const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(slot + col * 64);
                uint32_t addr[8];
                uint4 av[8];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    addr[v] = base + ((((uint32_t)(4 * half + v)) ^ (uint32_t)(col & 15)) << 4);
                    addr[4 + v] = base + ((((uint32_t)(8 + 4 * half + v)) ^ (uint32_t)(col & 15)) << 4);
                }
                lds_read_ops<M>(addr, av, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(aug_s + ti * 64 + lane), aug);
#pragma unroll
                for (int v = 0; v < 4; ++v) { ah[v] = av[v]; al[v] = av[4 + v]; }
            }
            const uint32_t tl = (uint32_t)ti;
            f32x16 x0, x1;
            if (t == tile0) bf_pair<false, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], x0, x1, p0, p1, 0u, k0[2], k1[2], k2[2],
                                           k0[3], k1[3], k2[3]);
            else bf_pair<true, M>(ah, al, aug, bh[0], bl[0], b_aug[0], bh[1], bl[1], b_aug[1], x0, x1, p0, p1, tl - 1u, k0[2], k1[2], k2[2], k0[3],
                               k1[3], k2[3]);
            bf_pair<true, M>(ah, al, aug, bh[2], bl[2], b_aug[2], bh[3], bl[3], b_aug[3], p0, p1, x0, x1, tl, k0[0], k1[0], k2[0], k0[1], k1[1], k2[1]);
        }
        if (tile0 < tile1) {
            const uint32_t tlast = (uint32_t)(tile1 - 1 - tile0);
            push_group4(p0, tlast, k0[NG - 2], k1[NG - 2], k2[NG - 2]);
            push_group4(p1, tlast, k0[NG - 1], k1[NG - 1], k2[NG - 1]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint64_t a0 = widen_key(k0[g], tile0, half), a1 = widen_key(k1[g], tile0, half), a2 = widen_key(k2[g], tile0, half);
            const uint64_t b0 = shfl_xor_u64(a0, 32), b1 = shfl_xor_u64(a1, 32), b2 = shfl_xor_u64(a2, 32);
            const uint64_t m0 = a0 < b0 ? a0 : b0;
            const uint64_t hx = a0 < b0 ? b0 : a0, lx = a1 < b1 ? a1 : b1;
            const uint64_t m1 = hx < lx ? hx : lx;
            const uint64_t third = third_of_two_triples(a0, a1, a2, b0, b1, b2);
            const int qi = q0 + g * 32 + col;
            if (half == 0 && qi < qpad) {
                uint64_t* dst = partial_keys + ((size_t)bx * qpad + qi) * BF_KEEP;        // block-major: a wave's 32 queries make one 512-byte write
                dst[0] = m0;
                dst[1] = m1;
                partial_bound[(size_t)bx * qpad + qi] = (uint32_t)min(third >> 32, (uint64_t)0x7f800000u);
            }
        }
        if (s + 2 < n_my) {                                          // strip s is read by every wave: its slots take strip s + 2
            bf_lds_barrier();
            bf_request_strip<M>(vocab_bf, row_norm, n_rows, bx0 + (s + 2) * px, tiles_per_block, n_tiles, lane, wave, col, half,
                             s_dyn + (size_t)((s & 1) ? 8 : 0) * BF_TILE_F, s_aug + (s & 1) * (MF_STRIP_TILES * 64));
        }
    }
}
template <int M>
__global__ __launch_bounds__(256) void knn_bf16_filter_kernel_p(const float* __restrict__ vocab_bf, const float* __restrict__ row_norm, int n_rows,
                                                                const float* __restrict__ queries, int nq, int qpad, int tiles_per_block, int n_blocks,
                                                                int px, uint64_t* __restrict__ partial_keys, uint32_t* __restrict__ partial_bound,
                                                                SelfdistJob sd) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn_p[];
    knn_bf16_filter_body_p<M>(s_dyn_p, (int)blockIdx.x, vocab_bf, row_norm, n_rows, queries, nq, qpad, tiles_per_block, n_blocks, px, partial_keys,
                           partial_bound, sd);
}

// |bf16x3 filter score - reference distance| <= eps, u = 2^-24:
//   split: bf16 keeps 8 significant bits (unit roundoff 2^-8): x = hi + lo + d with |lo| <= 2^-8 |x|, |d| <= 2^-8 |lo| <= 2^-16 |x|;
//          the neglected ql.vl and the two d terms cost <= 3 * 2^-16 |q||v| (1 + 2^-7) on q.v, twice that on the score
//          -> 3.1 * 2^-16 (|v|^2 + |q|^2)   (tests/test_bf16_split_bound.py emulates the split on the CPU: adversarial inputs reach
//          a third of it)
//   accumulation: 2 + 3 dim products summed in fp32 by the matrix pipe; each addition is charged 2u (round-to-nearest or
//          truncation) of the running magnitude <= |v|^2 + |q|^2 + 2 * 3 |q||v| <= 4 (|v|^2 + |q|^2)  -> (3 dim + 4) * 2u * 4 (..)
//   norms (dim-term FMA chains) and the reference's own rounding, as in eps_for()                     -> (dim + 2 (dim/4 + 6)) u (..)
// a quarter more is added for slack; the measured worst case is reported by the tests (knn_max_err_ratio).
__device__ __forceinline__ float eps_bf16(int dim, float qn, float vn_max) {
    const float u = 5.9604645e-8f;
    return (3.1f * 1.5258789e-5f + ((3.0f * (float)dim + 4.0f) * 8.0f + 1.5f * (float)dim + 12.0f) * u) * 1.25f * (qn + vn_max);
}
// |fp16 one-product filter score - reference distance| <= eps: both operands rounded to half (u16 = 2^-11, relative, inside half's normal
// range): |q16.v16 - q.v| <= (2 u16 + u16^2) |q||v|, i.e. (2 u16 + u16^2)(|q|^2 + |v|^2) on the score -2 q.v (tests/test_fp16_split_bound.py);
// components below half's normal range (2^-14) are rounded with an ABSOLUTE error <= 2^-25 each: 2 * dim * 2^-25 (|q| + |v|) more on the
// score, charged with |x| <= 1 + |x|^2; the accumulation (dim products in fp32 by the matrix pipe) and norm terms as in eps_bf16.
__device__ __forceinline__ float eps_f16(int dim, float qn, float vn_max) {
    const float u = 5.9604645e-8f, u16 = 4.8828125e-4f;
    return ((2.0f * u16 + u16 * u16) + (((float)dim + 4.0f) * 8.0f + 1.5f * (float)dim + 12.0f) * u) * 1.25f * (qn + vn_max) +
           2.0f * (float)dim * 2.9802322e-8f * (2.0f + qn + vn_max);
}

// ------------------------------------------------------------------------------------------------ re-rank + certificate
// |filter score - reference distance| <= eps: both are fp32 evaluations of the same real number d = |v - q|^2 <= 2 (|v|^2 + |q|^2).
// With u = 2^-24 and gamma_n ~ n u:
//   filter: a chain of dim + 2 FMAs over terms whose magnitudes sum to <= 2 (|v|^2 + |q|^2)        -> 2 (dim + 2) u (|v|^2 + |q|^2)
//           the two norms are themselves dim-term FMA chains                                          ->       dim u (|v|^2 + |q|^2)
//   reference (dist.h:150-177): every term (v_k - q_k)^2 carries 3 roundings, then dim/4 + 3 additions
//           of non-negative numbers                                                                   -> 2 (dim/4 + 6) u (|v|^2 + |q|^2)
// total (3.5 dim + 16) u (|v|^2 + |q|^2); a quarter more is added for slack.  |v|^2 is replaced by the vocabulary maximum.
__device__ __forceinline__ float eps_for(int dim, float qn, float vn_max) {
    return (3.5f * (float)dim + 16.0f) * 5.9604645e-8f * 1.25f * (qn + vn_max);
}

template <int NG, int M>
__global__ __launch_bounds__((BF_QB / (NG * 32)) * 64) void knn_bf16_filter_kernel(const float* __restrict__ vocab_bf, const float* __restrict__ row_norm,
                                                                      int n_rows, const float* __restrict__ queries, int nq, int qpad,
                                                                      int tiles_per_block, int n_blocks, uint64_t* __restrict__ partial_keys,
                                                                      uint32_t* __restrict__ partial_bound, SelfdistJob sd) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn_f[];
    knn_bf16_filter_body<NG, M>(s_dyn_f, (int)blockIdx.x, vocab_bf, row_norm, n_rows, queries, nq, qpad, tiles_per_block, n_blocks, partial_keys,
                             partial_bound, sd);
}


// One workgroup per query (the kernel is a chain of dependent memory round trips: the more lanes share them, the shorter).
// Pass 1 finds tau = the second smallest filter score among the kept keys; a kept row whose score exceeds
// tau (1 + 2^-15) + 2 eps is strictly farther than the two rows that define tau (|score - distance| <= eps, keys are truncated by
// < 2^-16 relative), so only the few keys at or below that threshold are re-computed exactly in pass 2 -- each by 16 lanes:
// lane i of the group holds the term of floats [4i, 4i + 4) (one coalesced 256-byte row read) and the sixteen terms are added in
// the reference's order.  Nothing is dropped at this stage (more than RR_MAX_CAND keys under the threshold -- a cluster of
// near-identical rows -- sends the query to the exact scan): the bound on dropped rows comes from the filter alone.
// KEEP keys per (row block, query); LAST_KEY_BOUNDS: the block's last kept key also bounds what its merge dropped (f32 filter);
// BF16: the keys come from the bf16x3 filter (eps_bf16).  fail_count[2] collects max |score - distance| / eps (diagnostics).
constexpr int RR_MAX_CAND = 128;
#ifdef LCD_B_TIMING   // timing experiment only: phases of the re-rank workgroups of launch B (100 MHz), without extra barriers
__device__ unsigned long long g_rr_timing[8 * 512];
#define RR_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_rr_timing[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RR_STAMP(i) do { } while (0)
#endif
// HALVES = 2: a workgroup of 2 x MF_BLOCK threads re-ranks TWO queries (qi_first, qi_first + 1), one per half -- launch B of a
// pipelined frame runs with 512-thread workgroups because its scoring half needs eight waves per bucket (a 4-wave scoring workgroup
// takes twice as long), and a re-rank workgroup that used only half of its threads idled the other.  The halves share nothing but the
// barriers (every barrier of the body is reached by all threads: the conditions around them are launch-uniform).
template <int DIM, int KEEP, bool LAST_KEY_BOUNDS, bool BF16, int HALVES = 1>
__device__ __forceinline__ void knn_mfma_rerank_body(int qi_first, const uint64_t* __restrict__ partial_keys,
                                                     const uint32_t* __restrict__ partial_lmin, int n_blocks, int nq,
                                                     const float* __restrict__ vocab, const float* __restrict__ queries,
                                                     const int32_t* __restrict__ row_id,
                                                     const uint32_t* __restrict__ norm_max_bits,
                                                     int32_t* __restrict__ out_row, int32_t* __restrict__ out_word,
                                                     float* __restrict__ out_dist, int32_t* __restrict__ fail_list,
                                                     int32_t* __restrict__ fail_count, const CandBits& cb,
                                                     const int32_t* __restrict__ pend_lo = nullptr, const int32_t* __restrict__ pend_hi = nullptr,
                                                     int pend_cap = 0x7fffffff /* rows the filter's launch plan covered */,
                                                     float* stage = nullptr, int stage_rows = 0 /* LDS staging area of the pending rows (256 B each,
                                                     a multiple of 4), shared by the halves of the workgroup */,
                                                     int f16 = 0 /* the keys come from the one-product fp16 filter (eps_f16) */,
                                                     const float* __restrict__ pend_desc = nullptr, const uint32_t* __restrict__ pend_list = nullptr,
                                                     int32_t pend_first_id = 0
                                                     /* rows at or beyond pend_lo[0] are not vocabulary rows yet (a deferred append writes them in this
                                                        very launch): row pend_lo[0] + j is descriptor pend_list[j] of pend_desc, word pend_first_id + j */
                                                     , const float* __restrict__ cross = nullptr, int cross_ld = 0
                                                     /* cross[qi * cross_ld + c] = |query qi - descriptor c of pend_desc|^2, from the cross-frame tiles of
                                                        launch A of this pair (selfdist_tile): the pending rows' distances are read, not computed; NULL: the
                                                        rows are staged and their distances computed here */
                                // The re-rank workgroups WRITE the rows of the deferred append from the copy they have staged anyway (round 5: the
                                // default since it passed the GPU suite; launch B 15.3 -> 13.8 us at the headline, profiles/r05_first_call.txt);
                                // workgroup wr_index of wr_n
                                                     , const AppendRowsArgs& wr = AppendRowsArgs(), bool wr_on = false, int wr_index = 0, int wr_n = 1
                                // rows_only (round 6, PipeOpts::row_writer_wgs): this workgroup re-ranks nothing -- it is one of wr_n extra workgroups of
                                // the re-rank ROLE that only write the appended rows (wr_index, wr_index + wr_n, ...), so that no re-rank workgroup has
                                // the row stores at the end of its chain (the workgroups that wrote a row ended 3-4 us after those that did not) and
                                // the kernel gets no third branch (whose register demand made the scoring branch spill)
                                                     , bool rows_only = false
                                // shadow scores (round 6): sh_x[qi * sh_ld + j] is the filter's score of this query against descriptor j of the frame before
                                // (sh_q of them; shadow_scores_body of launch A).  Descriptor j is a row of the vocabulary iff bit j of sh_mask is set (the
                                // final new-word mask of that frame's decision loop, mw words, followed by its mw + 1 word prefix sums): row n_lo0 + rank(j),
                                // word pend_first_id + rank(j), read from pend_desc.  The words whose score is at or below the threshold join the candidates
                                // as keys of ONE row (SHADOW_ROW_BASE + j) and are evaluated exactly in the same round trip as the others; a word above the
                                // threshold cannot be among the two nearest, by the argument that covers every kept key above it.  The rows [n_lo0, p_hi)
                                // then need no scan of their own (and are written by the rows_only workgroups).
                                                     , const uint32_t* __restrict__ sh_mask = nullptr, int sh_q = 0, const float* __restrict__ sh_x = nullptr, int sh_ld = 0
                                                     ) {
    static_assert(DIM == 64, "16 lanes x 4 floats per candidate row");
    // rows [pend_lo[0], pend_hi[0]): words the previous frame created, appended on the device after this frame's filter took its
    // snapshot of the vocabulary (VWDictionary::update() of a pipelined handle).  They are scanned exactly here, so the result is
    // the 2-NN over the vocabulary as update() leaves it before this frame.
    // (the plan is made for an ESTIMATE of the row count: what lies between the rows it covered and the device's count is scanned here too)
    // (both counters through the scalar cache, requested together: as two vector loads the compiler made each uniform right behind its request --
    // two round trips in a row at the head of every re-rank workgroup, round 6's ISA)
    int n_lo0 = 0, p_hi_ld = 0;
    if (pend_lo && pend_hi) asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(n_lo0), "=&s"(p_hi_ld) : "s"(pend_lo), "s"(pend_hi) : "memory");
    else { if (pend_lo) n_lo0 = pend_lo[0]; if (pend_hi) p_hi_ld = pend_hi[0]; }
    const int p_lo = pend_lo ? min(n_lo0, pend_cap) : 0, p_hi = p_hi_ld;
    const int32_t pend_id0 = pend_first_id > 0 ? pend_first_id : n_lo0 - pend_first_id;   // word id of the first pending row (<= 0: ids that follow the rows, AppendArgs::first_id)
    // with shadow rows the pending scan only has to cover vocabulary rows the filter's plan did not reach ([p_lo, n_lo0): rare)
    const int p_hi_s = sh_q > 0 ? min(p_hi, n_lo0) : p_hi;
    __shared__ uint32_t s_plist[HALVES * MF_BLOCK];                    // the first entries of pend_list (one per thread: read with the keys)
    const uint32_t plreg = pend_list ? pend_list[threadIdx.x] : 0u;    // (the list buffer holds at least HALVES * MF_BLOCK entries)
    constexpr int SH_MW_MAX = 128;                                     // mask words of a frame of 4 096 descriptors
    __shared__ uint32_t s_shmask[2 * SH_MW_MAX + 1];
    const int sh_mw = sh_q > 0 ? (sh_q + 63) / 64 * 2 : 0;
    const uint32_t shreg = (sh_q > 0 && (int)threadIdx.x < 2 * sh_mw + 1) ? sh_mask[threadIdx.x] : 0u;
    auto sh_isword = [&](uint32_t j) -> bool { return (s_shmask[j >> 5] >> (j & 31)) & 1u; };
    auto sh_rank = [&](uint32_t j) -> int { return (int)(s_shmask[sh_mw + (j >> 5)] + (uint32_t)__popc(s_shmask[j >> 5] & ((1u << (j & 31)) - 1u))); };
    const int hf = HALVES == 2 ? (int)threadIdx.x / MF_BLOCK : 0;
    const int tid = HALVES == 2 ? (int)threadIdx.x % MF_BLOCK : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The pending rows are the same for every query: with a staging area they come in by LDS-DMA -- no registers, requested HERE, a
    // whole chunk in one round trip that runs under the two passes below -- instead of four rows per 16-lane group and trip
    // (three dependent round trips for the ~150 words a frame creates: +7 us on launch B, measured).
    // LDS slot (row r, position s) holds the row's 16-byte chunk s ^ (r & 15): sixteen consecutive lanes that read the same chunk of
    // sixteen consecutive rows touch sixteen different positions (no bank conflict)
    // (list_in_lds: every list entry the chunk can need lies in s_plist.  The other case -- a frame that created more than
    // HALVES * MF_BLOCK words -- reads the list from memory inside the issue loop, and a read there makes every trip wait for the
    // requests of the trip before it (one in-order counter): five serial round trips instead of one, 2.6 us on the ~150 rows a frame
    // creates.  So the common case gets a loop of its own without any read.)
    auto stage_chunk = [&](int first, int n_chunk) {
        const int wv = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63;
        // every re-rank workgroup of the launch wants the SAME rows at the same moment: each starts at another row (rotation by workgroup
        // index), so that at any time the requests spread over all the L2 channels instead of queueing at one
        const int n_inst = (n_chunk + 3) / 4;
        const int rot = (int)(((unsigned)blockIdx.x * 13u) % (unsigned)n_inst);
        const bool list_in_lds = !pend_list || first + n_chunk - n_lo0 <= HALVES * MF_BLOCK;
        if (list_in_lds) {
            for (int i0 = wv; i0 < n_inst; i0 += HALVES * MF_WAVES) {    // one instruction = four rows = 1 KB of LDS
                const int i = i0 + rot < n_inst ? i0 + rot : i0 + rot - n_inst;
                const int rl = min(i * 4 + (ln >> 4), n_chunk - 1);      // (the rows of a partial last group repeat the chunk's last row)
                const int chunk = (ln & 15) ^ ((i * 4 + (ln >> 4)) & 15);
                const int r = first + rl;
                const float* src = vocab + (size_t)r * DIM;
                if (pend_list && r >= n_lo0) src = pend_desc + (size_t)s_plist[r - n_lo0] * DIM;   // a row that is being written in this launch: its descriptor
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + chunk * 4),
                                                 (__attribute__((address_space(3))) void*)(stage + (size_t)i * 256), 16, 0, 0);
            }
            return;
        }
        for (int i0 = wv; i0 < n_inst; i0 += HALVES * MF_WAVES) {
            const int i = i0 + rot < n_inst ? i0 + rot : i0 + rot - n_inst;
            const int rl = min(i * 4 + (ln >> 4), n_chunk - 1);
            const int chunk = (ln & 15) ^ ((i * 4 + (ln >> 4)) & 15);
            const int r = first + rl;
            const float* src = vocab + (size_t)r * DIM;
            if (pend_list && r >= n_lo0) {
                const int j = r - n_lo0;
                src = pend_desc + (size_t)(j < HALVES * MF_BLOCK ? s_plist[j] : pend_list[j]) * DIM;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + chunk * 4),
                                             (__attribute__((address_space(3))) void*)(stage + (size_t)i * 256), 16, 0, 0);
        }
    };
    // With the cross-frame matrix the pending rows need no staging: row n_lo0 + j is descriptor pend_list[j] of the frame before, and the
    // query's distance to it is cross[qi][pend_list[j]] -- in the reference's arithmetic, computed by launch A.  Each workgroup stages only the
    // rows it WRITES (row wr_index + m wr_n at place m), and the distances are gathered into the staging area by 4-byte LDS-DMA (no registers):
    // half hf's value j at xd[hf * XD_MAX + j].  Not when rows of the vocabulary itself are pending (the plan covered fewer rows than exist).
    constexpr int XD_MAX = 2048, XD_OWN = 32;                           // pending rows / rows of its own a workgroup can take this way (40 KB of LDS)
    const int n_own = (wr_on && p_hi - n_lo0 > wr_index) ? (p_hi - n_lo0 - wr_index + wr_n - 1) / wr_n : 0;
    const bool use_cross = sh_q == 0 && cross != nullptr && pend_list != nullptr && stage != nullptr && stage_rows >= XD_OWN + HALVES * XD_MAX / DIM && p_hi > p_lo && p_lo == n_lo0 &&
                           p_hi - p_lo <= XD_MAX && n_own <= XD_OWN;
    float* const xd = stage + XD_OWN * DIM;
    const bool staged = stage != nullptr && stage_rows >= 4 && p_hi_s > p_lo && !use_cross;
    // rows [n_lo0, p_hi) ARE the rows of the deferred append, and every re-rank workgroup has them in its staging area: workgroup wr_index
    // of wr_n writes rows wr_index, wr_index + wr_n, ... (16 lanes per row) -- no workgroups of their own, no third branch in the kernel (whose
    // presence makes the scoring branch spill).  Stores only, at the END of the body: a read behind them would wait for them.
    auto write_rows = [&](int c0, int n_chunk, bool own = false) {      // own: the staging area holds this workgroup's rows only (use_cross)
        const AppendArgs& ap = wr.ap;
        const int n_new = p_hi - n_lo0, c16 = (int)threadIdx.x & 15;
        float nmax = 0.0f;
        for (int j = wr_index + wr_n * ((int)threadIdx.x >> 4), m = (int)threadIdx.x >> 4; j < n_new; j += wr_n * (HALVES * MF_BLOCK / 16), m += HALVES * MF_BLOCK / 16) {
            const int rl = own ? m : n_lo0 + j - c0;                   // the row's place in the staged chunk (uniform over its 16 lanes)
            if (rl < 0 || rl >= n_chunk) continue;
            const int32_t key = (c16 == 0 && wr.new_ws.n > 0) ? ws_runs_at_dev(wr.new_ws, j) : -1;      // (looked up in front of the row's stores)
            const uint4 x = *reinterpret_cast<const uint4*>(stage + (size_t)rl * DIM + ((c16 ^ (rl & 15)) * 4));
            append_write_row(ap, (size_t)n_lo0 + (size_t)j, c16, x, nmax);
            if (c16 == 0) {
                const size_t row = (size_t)n_lo0 + (size_t)j;
                ap.row_id[row] = ap.first_id > 0 ? ap.first_id + j : (int32_t)row - ap.first_id;    // (first_id <= 0: the id follows the ROW, AppendArgs)
                ap.row_wslot[row] = key;
                if (key >= 0 && ap.wrow) ap.wrow[key] = (uint32_t)row + 1u;
            }
        }
        append_norm_max(ap, nmax);
    };
    // the pinned row-count mirror of the frame whose rows this launch writes, when its decision loop left it to this launch (AppendArgs::mirror_later):
    // one thread of the launch, at the END of its workgroup's body (a store to host memory in front of a load would hold that load for its acknowledgement)
    auto store_mirror = [&]() {
        if (wr.ap.mirror_later && wr.ap.host_mirror && wr_on && wr_index == 0 && threadIdx.x == 0)
            __hip_atomic_store(wr.ap.host_mirror, ((unsigned long long)wr.ap.tag << 32) | (unsigned long long)(uint32_t)p_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    if (rows_only) {                                                   // (uniform over the workgroup)
        const int n_new = p_hi - n_lo0;
        if (!wr_on || stage == nullptr || stage_rows < 4 || pend_list == nullptr || n_new <= wr_index) { store_mirror(); return; }
        s_plist[threadIdx.x] = plreg;
        lds_barrier();
        const int wv = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63;
        const int n_mine = (n_new - wr_index + wr_n - 1) / wr_n;       // rows wr_index + wr_n m, m < n_mine
        const int cap = stage_rows & ~3;
        for (int m0 = 0; m0 < n_mine; m0 += cap) {                     // (one chunk unless a frame creates more than wr_n x 160 words)
            const int n_chunk = min(cap, n_mine - m0);
            if (m0 > 0) __syncthreads();
            for (int i = wv; i * 4 < n_chunk; i += HALVES * MF_WAVES) { // four rows per instruction, laid out as stage_chunk does
                const int rl = min(i * 4 + (ln >> 4), n_chunk - 1);
                const int chunk = (ln & 15) ^ ((i * 4 + (ln >> 4)) & 15);
                const int j = wr_index + wr_n * (m0 + rl);
                const float* src = pend_desc + (size_t)(j < HALVES * MF_BLOCK ? s_plist[j] : pend_list[j]) * DIM;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + chunk * 4),
                                                 (__attribute__((address_space(3))) void*)(stage + (size_t)i * 256), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            {   // write_rows(own = true) over this chunk: staged row m is new word wr_index + wr_n (m0 + m)
                const AppendArgs& ap = wr.ap;
                const int c16 = (int)threadIdx.x & 15;
                float nmax = 0.0f;
                for (int m = (int)threadIdx.x >> 4; m < n_chunk; m += HALVES * MF_BLOCK / 16) {
                    const int j = wr_index + wr_n * (m0 + m);
                    const int32_t key = (c16 == 0 && wr.new_ws.n > 0) ? ws_runs_at_dev(wr.new_ws, j) : -1;
                    const uint4 x = *reinterpret_cast<const uint4*>(stage + (size_t)m * DIM + ((c16 ^ (m & 15)) * 4));
                    append_write_row(ap, (size_t)n_lo0 + (size_t)j, c16, x, nmax);
                    if (c16 == 0) {
                        const size_t row = (size_t)n_lo0 + (size_t)j;
                        ap.row_id[row] = ap.first_id > 0 ? ap.first_id + j : (int32_t)row - ap.first_id;    // (first_id <= 0: the id follows the ROW, AppendArgs)
                        ap.row_wslot[row] = key;
                        if (key >= 0 && ap.wrow) ap.wrow[key] = (uint32_t)row + 1u;
                    }
                }
                append_norm_max(ap, nmax);
            }
        }
        store_mirror();
        return;
    }
    const bool valid = qi_first + hf < nq;                             // the odd query out: its half walks the last query again, writes nothing
    const int qi = valid ? qi_first + hf : nq - 1;
    __shared__ float s_thr_all[HALVES];
    float& s_thr = s_thr_all[hf];
    const int n_keys = n_blocks * KEEP;
    constexpr int GS = BF16 ? 4 : 1;                                   // rows a key stands for (see the exact phase)
    constexpr int RR_KEYS = RR_MAX_CAND / GS;                          // keys under the threshold a query may have before it goes to the exact redo
    // key c of the query: block c / KEEP, entry c % KEEP.  The bf16 filter's records are block-major ([block][query][KEEP], see
    // knn_bf16_filter_body), the f32 filter's query-major
    const int qpad_t = (nq + 63) / 64 * 64;
    auto key_at = [&](int c) -> uint64_t {
        return BF16 ? partial_keys[((size_t)(c / KEEP) * qpad_t + qi) * KEEP + (c % KEEP)] : partial_keys[(size_t)qi * n_keys + c];
    };
    auto bound_at = [&](int b) -> uint32_t { return BF16 ? partial_lmin[(size_t)b * qpad_t + qi] : partial_lmin[(size_t)qi * n_blocks + b]; };
    constexpr uint32_t INF = 0x7f800000u;
    __shared__ uint32_t s_a0_all[HALVES][MF_WAVES], s_a1_all[HALVES][MF_WAVES], s_bound_all[HALVES][MF_WAVES];
    __shared__ int s_ncand_all[HALVES];
    __shared__ uint64_t s_cand_all[HALVES][RR_MAX_CAND], s_exact_all[HALVES][RR_MAX_CAND];
    __shared__ int32_t s_word_all[HALVES][RR_MAX_CAND];
    __shared__ float s_err_all[HALVES][MF_WAVES];
    __shared__ uint64_t s_pend_all[HALVES][MF_WAVES][2];
    uint32_t (&s_a0)[MF_WAVES] = s_a0_all[hf]; uint32_t (&s_a1)[MF_WAVES] = s_a1_all[hf]; uint32_t (&s_bound)[MF_WAVES] = s_bound_all[hf];
    int& s_ncand = s_ncand_all[hf];
    uint64_t (&s_cand)[RR_MAX_CAND] = s_cand_all[hf]; uint64_t (&s_exact)[RR_MAX_CAND] = s_exact_all[hf];
    int32_t (&s_word)[RR_MAX_CAND] = s_word_all[hf];
    float (&s_err)[MF_WAVES] = s_err_all[hf];
    uint64_t (&s_pend)[MF_WAVES][2] = s_pend_all[hf];
    // Everything that does not depend on other loads is requested up front (the kernel is a chain of round trips): the first two
    // keys and the first bound of every thread, the query slice, the vocabulary norm bound and -- for the candidate bits -- the
    // thread's two entries of the query's row of the same-frame distance matrix.
    RR_STAMP(0);
    const uint64_t kreg0 = tid < n_keys ? key_at(tid) : KEY_NONE;
    const uint64_t kreg1 = tid + MF_BLOCK < n_keys ? key_at(tid + MF_BLOCK) : KEY_NONE;
    const uint64_t kreg2 = tid + 2 * MF_BLOCK < n_keys ? key_at(tid + 2 * MF_BLOCK) : KEY_NONE;   // (219 strips x 3 keys: a third of the threads have a third key)
    const uint32_t breg0 = tid < n_blocks ? bound_at(tid) : INF;
    const float4 q4 = reinterpret_cast<const float4*>(queries + (size_t)qi * DIM)[lane & 15];
    const float vn_max = __uint_as_float(norm_max_bits[0]);
    float dreg0 = __int_as_float(0x7f800000), dreg1 = __int_as_float(0x7f800000);
    if (cb.bits) {
        if (tid < cb.nq) dreg0 = cb.selfdist[(size_t)qi * cb.ld + tid];
        if (tid + MF_BLOCK < cb.nq) dreg1 = cb.selfdist[(size_t)qi * cb.ld + tid + MF_BLOCK];
    }
    float2 xsh = make_float2(__int_as_float(0x7f800000), __int_as_float(0x7f800000));   // the query's scores against descriptors 2 tid, 2 tid + 1 of the frame before
    if (sh_q > 0 && 2 * tid < sh_ld) xsh = *reinterpret_cast<const float2*>(sh_x + (size_t)qi * sh_ld + 2 * tid);
    float qn = fmaf(q4.w, q4.w, fmaf(q4.z, q4.z, fmaf(q4.y, q4.y, q4.x * q4.x)));
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) qn += __shfl_xor(qn, m, 64);

    // ---- pass 1: tau and the bound on dropped rows
    uint32_t a0 = INF, a1 = INF, bound = breg0;
    auto see = [&](uint64_t k, int c) {
        const uint32_t sc = min((uint32_t)(k >> 32), INF);                  // KEY_NONE -> +inf
        if (LAST_KEY_BOUNDS && (c % KEEP) == KEEP - 1) bound = min(bound, sc);   // rows the block merge dropped are no better than its last key
        const uint32_t h = max(a0, sc);
        a0 = min(a0, sc);
        a1 = min(a1, h);
    };
    see(kreg0, tid);
    see(kreg1, tid + MF_BLOCK);
    see(kreg2, tid + 2 * MF_BLOCK);
    for (int c = tid + 3 * MF_BLOCK; c < n_keys; c += MF_BLOCK) see(key_at(c), c);
    for (int c = tid + MF_BLOCK; c < n_blocks; c += MF_BLOCK) bound = min(bound, bound_at(c));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o0 = (uint32_t)__shfl_xor((int)a0, m, 64), o1 = (uint32_t)__shfl_xor((int)a1, m, 64);
        a1 = min(max(a0, o0), min(a1, o1));
        a0 = min(a0, o0);
        bound = min(bound, (uint32_t)__shfl_xor((int)bound, m, 64));
    }
    if (lane == 0) { s_a0[wave] = a0; s_a1[wave] = a1; s_bound[wave] = bound; }
    if (tid == 0) s_ncand = 0;
    // the pending rows are requested here -- behind the keys, which have arrived, and in front of pass 2's row reads, whose round trip
    // they share (a request in front of the keys would make the first use of a key wait for the whole chunk: the counter is in-order)
    // (measured in round 5, profiles/r05_ab_notes.txt 8: requested in FRONT of the keys instead, the driver's 20 steps take 0.0392-0.0401 ms
    // per frame against 0.0384-0.0390)
    if (pend_list || sh_q > 0) {
        s_plist[threadIdx.x] = plreg;
        if ((int)threadIdx.x < 2 * sh_mw + 1) s_shmask[threadIdx.x] = shreg;
        lds_barrier();
    }
    if (staged) stage_chunk(p_lo, min(stage_rows, p_hi_s - p_lo));
    if (use_cross) {
        const int wv = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63;
        auto plist_at = [&](int j) -> uint32_t { return j < HALVES * MF_BLOCK ? s_plist[j] : pend_list[j]; };
        for (int i = wv; i * 4 < n_own; i += HALVES * MF_WAVES) {      // the rows this workgroup writes: four per instruction, as stage_chunk lays them out
            const int rl = min(i * 4 + (ln >> 4), n_own - 1);
            const int chunk = (ln & 15) ^ ((i * 4 + (ln >> 4)) & 15);
            const float* src = pend_desc + (size_t)plist_at(wr_index + wr_n * rl) * DIM;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + chunk * 4),
                                             (__attribute__((address_space(3))) void*)(stage + (size_t)i * 256), 16, 0, 0);
        }
        const int n_pend = p_hi - p_lo;
        const float* crow = cross + (size_t)qi * cross_ld;
        for (int j0 = 0; j0 < n_pend; j0 += MF_BLOCK) {                // 64 values per instruction, lane l's at the instruction's base + 4 l
            const float* src = crow + plist_at(min(j0 + tid, n_pend - 1));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xd + hf * XD_MAX + j0 + wave * 64), 4, 0, 0);
        }
    }
    if (staged && lane < 16 && wave == 0) reinterpret_cast<float4*>(stage + (size_t)stage_rows * DIM)[hf * 16 + lane] = q4;   // the query, for every lane
    lds_barrier();                                                     // (LDS traffic only: __syncthreads() would also wait for the rows just requested)
    RR_STAMP(1);
    a0 = s_a0[0]; a1 = s_a1[0]; bound = s_bound[0];
#pragma unroll
    for (int w = 1; w < MF_WAVES; ++w) {
        const uint32_t o0 = s_a0[w], o1 = s_a1[w];
        a1 = min(max(a0, o0), min(a1, o1));
        a0 = min(a0, o0);
        bound = min(bound, s_bound[w]);
    }
    const float eps = BF16 ? (f16 ? eps_f16(DIM, qn, vn_max) : eps_bf16(DIM, qn, vn_max)) : eps_for(DIM, qn, vn_max);
    const float tau = __uint_as_float(a1);
    const float thr = tau + (2.0f * eps + tau * 3.0517578e-5f);               // +inf when fewer than two finite keys exist

    // ---- pass 2: the keys at or below the threshold (+inf: tombstone / padding row) ...
    auto take = [&](uint64_t k) {
        const uint32_t sc = (uint32_t)(k >> 32);
        if (k != KEY_NONE && sc < INF && __uint_as_float(sc) <= thr) {
            const int slot = atomicAdd(&s_ncand, 1);
            if (slot < RR_KEYS) s_cand[slot] = k;
        }
    };
    take(kreg0);
    take(kreg1);
    take(kreg2);
    for (int c = tid + 3 * MF_BLOCK; c < n_keys; c += MF_BLOCK) take(key_at(c));
    if (sh_q > 0) {                                                    // ... and the descriptors of the frame before that became words
        auto take_sh = [&](uint32_t j, float xs) {
            if (j < (uint32_t)sh_q && xs <= thr && sh_isword(j)) {
                const int slot = atomicAdd(&s_ncand, 1);
                if (slot < RR_KEYS) s_cand[slot] = ((uint64_t)(xs < 0.0f ? 0u : __float_as_uint(xs)) << 32) | (uint64_t)(SHADOW_ROW_BASE + j);
            }
        };
        take_sh(2u * (uint32_t)tid, xsh.x);
        take_sh(2u * (uint32_t)tid + 1u, xsh.y);
        for (int j = 2 * MF_BLOCK + tid; j < sh_q; j += MF_BLOCK) take_sh((uint32_t)j, sh_x[(size_t)qi * sh_ld + j]);   // (frames of more than 512 descriptors)
    }
    lds_barrier();
    RR_STAMP(2);
    const int n_keys_in = s_ncand;
    const bool overflow = n_keys_in > RR_KEYS;
    // A key of the bf16 / fp16 filters stands for a GROUP of four consecutive rows (push_group4: the key's score is the group's minimum, its
    // row the group's first): candidate slot i is row (i & 3) of key i >> 2 -- the four rows of a key are the four 16-lane groups of ONE
    // wave and trip.  A row of the group that is a tombstone, does not exist (yet), or lies at / behind the rows the pending scan covers is
    // no candidate: its exact distance reads +inf.  row_limit: the rows the keys may name (the filter's own row limit when rows are
    // appended on the device -- what lies behind is scanned exactly below --, else the plan's row count).
    const int n_cand = overflow ? n_keys_in : n_keys_in * GS;           // candidate ROWS
    const uint32_t row_limit = (uint32_t)(pend_lo ? p_lo : pend_cap);
    auto cand_row = [&](int i) -> uint32_t { return (uint32_t)s_cand[GS == 4 ? (i >> 2) : i] + (GS == 4 ? (uint32_t)(i & 3) : 0u); };
    // the vocabulary row a LIVE candidate stands for (the distance tie-break is by row): a shadow candidate is the row its descriptor is being
    // written to in this very launch -- behind every row the filter saw, in word order
    auto real_row = [&](int i) -> uint32_t {
        const uint32_t r = cand_row(i);                                   // (a live shadow slot is slot 0 of its key: r = SHADOW_ROW_BASE + j)
        return (sh_q > 0 && r >= SHADOW_ROW_BASE) ? (uint32_t)(n_lo0 + sh_rank(r - SHADOW_ROW_BASE)) : r;
    };
    // ... get their exact distances (reference arithmetic, dist.h:150-177), one candidate per 16-lane group and trip; the word
    // id of the row is fetched in the same round trip
    float err_ratio = 0.0f;
    if (!overflow) {
        // Two trips in flight (round 6): the rows of trip t + 1 are requested in front of trip t's arithmetic.  A query whose second neighbour is
        // far (a descriptor that will become a word) has dozens of keys under its threshold -- four, five trips of sixteen rows -- and with one
        // trip in flight each was a round trip of its own: those queries' workgroups were the tail of launch B (11.2 us median, 15.5 max).
        auto request = [&](int i, float4& v4, int32_t& wid) {
            const uint64_t k = s_cand[GS == 4 ? (i >> 2) : i];
            const uint32_t row = cand_row(i);
            const bool sh = sh_q > 0 && (uint32_t)k >= SHADOW_ROW_BASE;   // (uniform over the wave: the slots of ONE key) -- a key of ONE row, a word
            const uint32_t sj = (uint32_t)k - SHADOW_ROW_BASE;
            const bool in_range = sh ? (GS == 1 || (i & 3) == 0) : (GS == 1 || row < row_limit);
            const uint32_t rrow = in_range ? row : (uint32_t)k;            // (an address that exists: the group's first row)
            const float* src = sh ? pend_desc + (size_t)sj * DIM : vocab + (size_t)rrow * DIM;
            v4 = reinterpret_cast<const float4*>(src)[lane & 15];
            wid = 0;
            if ((lane & 15) == 0) wid = sh ? pend_id0 + sh_rank(sj) : row_id[rrow];
        };
        auto finish = [&](int i, const float4& v4, int32_t wid) {
            const uint64_t k = s_cand[GS == 4 ? (i >> 2) : i];
            const uint32_t row = cand_row(i);
            const bool sh = sh_q > 0 && (uint32_t)k >= SHADOW_ROW_BASE;
            const bool in_range = sh ? (GS == 1 || (i & 3) == 0) : (GS == 1 || row < row_limit);
            const float d0 = __fsub_rn(v4.x, q4.x), d1 = __fsub_rn(v4.y, q4.y), d2 = __fsub_rn(v4.z, q4.z), d3 = __fsub_rn(v4.w, q4.w);
            float t = __fmul_rn(d0, d0);
            t = __fadd_rn(t, __fmul_rn(d1, d1));
            t = __fadd_rn(t, __fmul_rn(d2, d2));
            t = __fadd_rn(t, __fmul_rn(d3, d3));
            float res = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) res = __fadd_rn(res, __shfl(t, (lane & 48) + j, 64));
            const bool live = in_range && (GS == 1 || __shfl(wid, lane & 48, 64) != 0);
            if (!live) res = __int_as_float(0x7f800000);
            float gmin = res;                                            // the filter's score of a key is its group's minimum (a shadow key: its one row's score)
            if (GS == 4) { gmin = fminf(gmin, __shfl_xor(gmin, 16, 64)); gmin = fminf(gmin, __shfl_xor(gmin, 32, 64)); }
            if ((lane & 15) == 0) {
                s_exact[i] = live ? (((uint64_t)__float_as_uint(res) << 32) | (uint32_t)i) : KEY_NONE;   // the slot stands in for the row: see below
                s_word[i] = wid;
                if (gmin < __int_as_float(0x7f800000)) err_ratio = fmaxf(err_ratio, fabsf(__uint_as_float((uint32_t)(k >> 32)) - gmin) / eps);
            }
        };
        constexpr int STEP = MF_BLOCK / 16;
        // (the trip count is uniform over the WAVE -- the four 16-lane groups of a wave hold the four slots of one key, and n_cand is a multiple
        // of four -- so the shuffles inside finish() always find their lanes)
        int i = tid >> 4;
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f); int32_t wa = 0;
        if (i < n_cand) request(i, va, wa);
        for (; i < n_cand; i += STEP) {
            float4 vb = make_float4(0.f, 0.f, 0.f, 0.f); int32_t wb = 0;
            if (i + STEP < n_cand) request(i + STEP, vb, wb);
            finish(i, va, wa);
            va = vb; wa = wb;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) err_ratio = fmaxf(err_ratio, __shfl_xor(err_ratio, m, 64));
    if (lane == 0) s_err[wave] = err_ratio;
    RR_STAMP(3);
    {   // the pending rows, sixteen lanes each, in the reference's arithmetic
        uint64_t pb = KEY_NONE, ps = KEY_NONE;
        constexpr int PU = 4;                                          // rows per 16-lane group and trip: their loads are in flight together (more
                                                                       // would cost the whole launch -- the scoring workgroups too -- occupancy)
        if (use_cross) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (each lane reads what its own request brought: no barrier)
            RR_STAMP(4);
            for (int j = tid; j < p_hi - p_lo; j += MF_BLOCK)
                top2_push(pb, ps, ((uint64_t)__float_as_uint(xd[hf * XD_MAX + j]) << 32) | (uint32_t)(p_lo + j));
        } else if (staged) {
            for (int c0 = p_lo; c0 < p_hi_s; c0 += stage_rows) {
                const int n_chunk = min(stage_rows, p_hi_s - c0);
                if (c0 > p_lo) { __syncthreads(); stage_chunk(c0, n_chunk); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                RR_STAMP(4);
                // ONE LANE PER ROW: the sixteen 4-float terms are formed and added by the same lane in the reference's order (dist.h:150-177),
                // the row's chunks from LDS, the query's as a broadcast read -- no cross-lane traffic (sixteen lanes per row gathered the
                // terms with sixteen ds_bpermute per row: at ~150 pending rows per query that was the whole cost of the scan)
                const float4* sq = reinterpret_cast<const float4*>(stage + (size_t)stage_rows * DIM) + hf * 16;
                for (int r = tid; r < n_chunk; r += MF_BLOCK) {
                    const float4* sv = reinterpret_cast<const float4*>(stage + (size_t)r * DIM);
                    float res = 0.0f;
#pragma unroll 4
                    for (int c = 0; c < 16; ++c) {
                        const float4 v = sv[c ^ (r & 15)], qq = sq[c];
                        const float d0 = __fsub_rn(v.x, qq.x), d1 = __fsub_rn(v.y, qq.y), d2 = __fsub_rn(v.z, qq.z), d3 = __fsub_rn(v.w, qq.w);
                        float t = __fmul_rn(d0, d0);
                        t = __fadd_rn(t, __fmul_rn(d1, d1));
                        t = __fadd_rn(t, __fmul_rn(d2, d2));
                        t = __fadd_rn(t, __fmul_rn(d3, d3));
                        res = __fadd_rn(res, t);
                    }
                    top2_push(pb, ps, ((uint64_t)__float_as_uint(res) << 32) | (uint32_t)(c0 + r));
                }
                if (wr_on && sh_q == 0 && c0 + stage_rows < p_hi) write_rows(c0, n_chunk);   // (more than one chunk: the staging area is about to be reused;
                                                                                // the LAST chunk's rows are written at the very end of the body)
            }
        } else
        for (int base = p_lo; base < (pend_list ? min(p_hi_s, n_lo0) : p_hi_s); base += PU * (MF_BLOCK / 16)) {   // (rows of a deferred append need the staged path)
            float4 v4[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int r0 = base + u * (MF_BLOCK / 16) + (tid >> 4);
                v4[u] = reinterpret_cast<const float4*>(vocab + (size_t)min(r0, p_hi_s - 1) * DIM)[lane & 15];
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int r0 = base + u * (MF_BLOCK / 16) + (tid >> 4);
                const float d0 = __fsub_rn(v4[u].x, q4.x), d1 = __fsub_rn(v4[u].y, q4.y), d2 = __fsub_rn(v4[u].z, q4.z), d3 = __fsub_rn(v4[u].w, q4.w);
                float t = __fmul_rn(d0, d0);
                t = __fadd_rn(t, __fmul_rn(d1, d1));
                t = __fadd_rn(t, __fmul_rn(d2, d2));
                t = __fadd_rn(t, __fmul_rn(d3, d3));
                float res = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) res = __fadd_rn(res, __shfl(t, (lane & 48) + j, 64));
                if ((lane & 15) == 0 && r0 < p_hi_s) top2_push(pb, ps, ((uint64_t)__float_as_uint(res) << 32) | (uint32_t)r0);
            }
        }
        if (p_hi_s > p_lo) {                                           // uniform
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const uint64_t ob = shfl_xor_u64(pb, m), os = shfl_xor_u64(ps, m);
                top2_push(pb, ps, ob);
                top2_push(pb, ps, os);
            }
            if (lane == 0) { s_pend[wave][0] = pb; s_pend[wave][1] = ps; }
        }
    }
    RR_STAMP(5);
    lds_barrier();                                                     // (what the halves hand over lies in LDS; no store is outstanding here)
    RR_STAMP(6);
    asm volatile("" : "+v"(dreg0), "+v"(dreg1));                       // (requested at the top, long since here: no wait for them behind the stores below)
    // the two best (distance, row) keys: ties go to the lower ROW (result_set.h:151-171), so the comparison key carries the row
    uint64_t best = KEY_NONE, second = KEY_NONE;
    int sbest = -1, ssecond = -1;
    if (wave == 0) {
        if (!overflow)
            for (int i = lane; i < n_cand; i += 64) {
                const uint64_t e = s_exact[i];
                if (e == KEY_NONE) continue;                              // (a row of a key's group that is no candidate)
                top2_push(best, second, (e & 0xFFFFFFFF00000000ull) | real_row((int)(uint32_t)e));
            }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint64_t ob = shfl_xor_u64(best, m), os = shfl_xor_u64(second, m);
            top2_push(best, second, ob);
            top2_push(best, second, os);
        }
#ifdef LCD_RR_SUBSTAMP
        RR_STAMP(4);
#endif
        if (p_hi_s > p_lo) {                                           // the pending rows' two best join (their rows differ from every kept key's)
#pragma unroll
            for (int w = 0; w < MF_WAVES; ++w) { top2_push(best, second, s_pend[w][0]); top2_push(best, second, s_pend[w][1]); }
        }
        // which candidate slots won (for their word ids; a pending row that won has no slot: -1)
        if (!overflow)
            for (int i = lane; i < n_cand; i += 64) {
                if (s_exact[i] == KEY_NONE) continue;
                const uint64_t key = (s_exact[i] & 0xFFFFFFFF00000000ull) | real_row(i);
                if (key == best) sbest = i;
                if (key == second) ssecond = i;
            }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            sbest = max(sbest, __shfl_xor(sbest, m, 64));
            ssecond = max(ssecond, __shfl_xor(ssecond, m, 64));
        }
#ifdef LCD_RR_SUBSTAMP
        RR_STAMP(5);
#endif
    }
    if (tid == 0) s_thr = (cb.have_index && second != KEY_NONE) ? __uint_as_float((uint32_t)(second >> 32)) : __int_as_float(0x7f800000);
    // The tail of the chain: thread 0 writes the query's results.  Everything that READS memory comes first -- the word ids of winners
    // that have no candidate slot, the certificate's operands -- and the stores last: the wait counter is one in-order counter for loads
    // and stores, so a load (or the reload of a spilled register) behind a store waits for the store's acknowledgement, a full round
    // trip each time (three of them in the first version of this block: 3.6 us of an 8 us chain, measured with in-kernel stamps).
    if (tid == 0 && valid) {
        err_ratio = fmaxf(fmaxf(s_err[0], s_err[1]), fmaxf(s_err[2], s_err[3]));
        const uint64_t k[2] = {best, second};
        const int sl[2] = {sbest, ssecond};
        int32_t wout[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (k[j] == KEY_NONE) continue;
            const int32_t rw = (int32_t)(uint32_t)k[j];
            wout[j] = sl[j] >= 0 ? s_word[sl[j]] : ((pend_list && rw >= n_lo0) ? pend_id0 + (rw - n_lo0) : row_id[rw]);
        }
        // certificate: every row the filter dropped is strictly farther than the exact second neighbour
        bool ok = !overflow;
        if (ok && bound < INF) {                                      // something finite was dropped
            if (second == KEY_NONE) ok = false;                       // fewer than two exact candidates but rows were dropped
            else ok = __uint_as_float(bound) - eps > __uint_as_float((uint32_t)(second >> 32));
        }
        // the certificate's premise is |filter score - exact distance| <= eps for every row; on the re-ranked candidates that error was
        // just measured: half the budget used up anywhere means the bound is no longer trusted for this query -> exact redo
        if (err_ratio >= 0.5f) ok = false;
        // fp16 operands hold magnitudes up to 65504: descriptors far outside that (the filter multiplies -2 q) are not this filter's
        // business -- the exact scan takes the query
        if (f16 && !(qn < 1.0e8f && vn_max < 1.0e8f)) ok = false;
        const bool report = err_ratio > 0.0f && eps > 0.0f;
        int reject = ok ? 0 : 1;
        asm volatile("" : "+v"(wout[0]), "+v"(wout[1]), "+v"(reject) :: "memory");   // every load of this thread has arrived: stores only from here
        ok = reject == 0;
        if (report) atomicMax(reinterpret_cast<uint32_t*>(fail_count) + 2, __float_as_uint(err_ratio));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            out_row[2 * qi + j] = k[j] == KEY_NONE ? -1 : (int32_t)(uint32_t)k[j];
            out_word[2 * qi + j] = wout[j];
            out_dist[2 * qi + j] = k[j] == KEY_NONE ? -1.0f : __uint_as_float((uint32_t)(k[j] >> 32));
        }
        if (!ok) fail_list[atomicAdd(fail_count, 1)] = qi;
    }
    RR_STAMP(7);
    if (cb.bits) {                                                    // the query's row of the candidate bit matrix (uniform branch)
        __shared__ int s_below_all[HALVES];                           // + the compact list of the set bits below qi (CandBits::list)
        int& s_below = s_below_all[hf];
        if (tid == 0) s_below = 0;
        // LDS-only barriers from here on: thread 0 has just stored the query's results, and __syncthreads() would hold the whole
        // workgroup until those stores are acknowledged -- a memory round trip per barrier, twice, at the end of a latency chain
        lds_barrier();
        const float thr2 = s_thr;
        for (int base = wave * 64; base < cb.ld; base += MF_BLOCK) {
            const int r = base + lane;
            float d = r == tid ? dreg0 : (r == tid + MF_BLOCK ? dreg1 : __int_as_float(0x7f800000));
            if (r >= 2 * MF_BLOCK && r < cb.nq) d = cb.selfdist[(size_t)qi * cb.ld + r];
            const unsigned long long m = __ballot(d < thr2);
            if (lane == 0 && valid) *reinterpret_cast<unsigned long long*>(cb.bits + (size_t)qi * cb.bw + (base >> 5)) = m;
            if (cb.cnt && d < thr2 && r < qi) {
                const int pos = atomicAdd(&s_below, 1);               // any order: the decision loop takes the two smallest (distance, j)
                if (pos < CAND_LIST && valid) cb.list[(size_t)qi * CAND_LIST + pos] = make_uint2((uint32_t)r, __float_as_uint(d));
            }
        }
        if (cb.cnt) {
            lds_barrier();
            if (tid == 0 && valid) cb.cnt[qi] = s_below;
        }
    }
    if (n_own > 0 && use_cross) {                                      // (every wave has waited for its requests; the rows came with wave 0's, ...)
        __syncthreads();
        write_rows(0, n_own, true);
    }
    if (wr_on && staged && sh_q == 0) {                                // the last (usually the only) chunk is still in the staging area
        const int c0_last = p_lo + ((p_hi - p_lo - 1) / stage_rows) * stage_rows;
        write_rows(c0_last, min(stage_rows, p_hi - c0_last));
    }
    store_mirror();
}

template <int DIM, int KEEP, bool LAST_KEY_BOUNDS, bool BF16>
__global__ __launch_bounds__(MF_BLOCK) void knn_mfma_rerank_kernel(const uint64_t* __restrict__ partial_keys,
                                                                   const uint32_t* __restrict__ partial_lmin, int n_blocks, int nq,
                                                                   const float* __restrict__ vocab, const float* __restrict__ queries,
                                                                   const int32_t* __restrict__ row_id,
                                                                   const uint32_t* __restrict__ norm_max_bits,
                                                                   int32_t* __restrict__ out_row, int32_t* __restrict__ out_word,
                                                                   float* __restrict__ out_dist, int32_t* __restrict__ fail_list,
                                                                   int32_t* __restrict__ fail_count, CandBits cb, int f16, int n_rows) {
    knn_mfma_rerank_body<DIM, KEEP, LAST_KEY_BOUNDS, BF16>((int)blockIdx.x, partial_keys, partial_lmin, n_blocks, nq, vocab, queries, row_id,
                                                          norm_max_bits, out_row, out_word, out_dist, fail_list, fail_count, cb, nullptr, nullptr,
                                                          n_rows, nullptr, 0, f16);
}

// ------------------------------------------------------------------------------------------------ software-pipelined frames
// Consecutive frames of a device-resident stream overlap INSIDE two launches instead of across streams: the 2-NN stage of frame t
// does not depend on the index stage of frame t - 1 (lcd_frame_dev never changes the vocabulary), so
//   launch A(t) = matrix-core filter of frame t  +  ONE workgroup running the whole tail of frame t - 1 (decision loop, retirements,
//                 registration, idf) + the redo workgroups of frame t - 1: the tail is a single-workgroup latency chain that used
//                 to sit alone on the critical path of every frame; here it hides behind the filter;
//   launch B(t) = exact re-rank of frame t  +  TF-IDF scoring of frame t - 1: two groups of small latency-bound workgroups that fill
//                 each other's stalls.
// Two dependent launches per frame on ONE stream, no events, no second host thread; the data each part reads was written by the
// previous launch (A -> B -> A ...).  The per-frame scratch exists twice (see engine.h).
struct FilterArgs {
    const float* vocab_bf; const float* row_norm; int n_rows; const float* queries; int nq, qpad, tiles_per_block, n_blocks;
    uint64_t* pk; uint32_t* pl; SelfdistJob sd; const int32_t* n_lo;
    const uint4* qsplit; const float* qnorm;                           // pre-split queries (non-persistent pipelined launch)
    int delay;                                                         // PipeOpts::filter_delay (timing experiments)
    const float* sh_bf; const float* sh_norm; int sh_rows; float* sh_x; int sh_ld;   // shadow scores (shadow_scores_body): the operand rows of the frame before, the score matrix
};
struct RerankArgs {
    const uint64_t* pk; const uint32_t* pl; int n_blocks, nq; const float* vocab; const float* queries; const int32_t* row_id;
    const uint32_t* norm_max_bits; int32_t* out_row; int32_t* out_word; float* out_dist; int32_t* fail_list; int32_t* fail_count; CandBits cb;
    const int32_t* n_lo; const int32_t* n_hi; int plan_rows;
    int stage_rows;                                                    // rows the launch's dynamic LDS stages (0: none)
    int f16;                                                           // the filter multiplied fp16 operands (one product): eps_f16
    const float* pend_desc; const uint32_t* pend_list; int32_t pend_first_id;   // the rows a deferred append writes in this launch, as descriptors
    const float* cross; int cross_ld;                                  // this frame's distances to every descriptor of pend_desc (CrossJob of the previous pair), or NULL
    const uint32_t* sh_mask; int sh_q; const float* sh_x; int sh_ld;   // shadow scores (knn_mfma_rerank_body): sh_q == 0: none
};
constexpr int PIPE_BLOCK = 256;     // workgroup size of both fused launches (the filter's and the re-rank's)

// workgroup 0 is the decision loop of the previous frame, workgroup 1 the retirement + registration of the frame before that (dispatched
// first: they are the longest single workgroups of the launch); the redo helpers of the decision loop come LAST -- they have nothing
// to do unless the certificate rejected a query, and in front they would each hold a compute unit's LDS while they find out
struct TailRoles { int has_resolve, has_register, n_redo, n_filter_wgs, n_q_wgs, n_sh_wgs; };
#ifdef LCD_B_TIMING   // timing experiment only: start / end of every workgroup of launch A (100 MHz)
__device__ unsigned long long g_a_timing[2 * 4096];
#define A_STAMP(i) do { __builtin_amdgcn_s_barrier(); if (threadIdx.x == 0 && blockIdx.x < 4096) g_a_timing[2 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define A_STAMP(i) do { } while (0)
#endif
// grid order: decision loop, registration, filter workgroups (+ distance tiles), query pre-split workgroups, redo helpers
template <bool PERSISTENT, int M>
__device__ __forceinline__ void frame_a_body(float* s_dyn, const FilterArgs& f, int px, const TailRoles& tr, const ResolveArgs& r, const FwArgs& a,
                                             const RetireArgs& ret, const QSplitArgs& qs) {
    const int bid = (int)blockIdx.x;
    const int n_front = tr.has_resolve + tr.has_register;
    A_STAMP(0);
    if (bid < tr.has_resolve) { frame_resolve_part<PIPE_BLOCK>((uint32_t*)s_dyn, r, 0, 1 + tr.n_redo); A_STAMP(1); return; }
    if (bid < n_front) { frame_register_part<PIPE_BLOCK>((uint32_t*)s_dyn, a, ret); A_STAMP(1); return; }
    const int after_strips = n_front + tr.n_filter_wgs;
    if (bid >= after_strips && bid < after_strips + tr.n_sh_wgs) {      // the shadow scores of this frame against the frame before it (non-persistent launches)
        if constexpr (!PERSISTENT) shadow_scores_body<M>(s_dyn, bid - after_strips, f.sh_bf, f.sh_norm, f.sh_rows, f.qsplit, f.qnorm, f.nq, f.qpad, f.sh_x, f.sh_ld);
        A_STAMP(1);
        return;
    }
    const int after_filter = after_strips + tr.n_sh_wgs;
    if (bid >= after_filter && bid < after_filter + tr.n_q_wgs) { qsplit_body(qs, bid - after_filter); A_STAMP(1); return; }
    if (bid >= after_filter + tr.n_q_wgs) { frame_resolve_part<PIPE_BLOCK>((uint32_t*)s_dyn, r, bid - after_filter - tr.n_q_wgs + 1, 1 + tr.n_redo); A_STAMP(1); return; }
    for (int i = 0; i < f.delay; ++i) __builtin_amdgcn_s_sleep(1);      // (0 unless "filter_delay" is set)
    if constexpr (PERSISTENT)
        knn_bf16_filter_body_p<M>(s_dyn, bid - n_front, f.vocab_bf, f.row_norm, f.n_rows, f.queries, f.nq, f.qpad, f.tiles_per_block, f.n_blocks, px, f.pk,
                               f.pl, f.sd, f.n_lo);
    else
        knn_bf16_filter_body_q<M>(s_dyn, bid - n_front, f.vocab_bf, f.row_norm, f.n_rows, f.qsplit, f.qnorm, f.nq, f.qpad, f.tiles_per_block, f.n_blocks, f.pk,
                               f.pl, f.sd, f.n_lo);
    A_STAMP(1);
}
// two workgroups per compute unit: 66 KB of LDS each, and a register budget of two waves per SIMD
template <int M>
__global__ __launch_bounds__(PIPE_BLOCK, 2) void frame_a_kernel(FilterArgs f, TailRoles tr, ResolveArgs r, FwArgs a, RetireArgs ret, QSplitArgs qs) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn_a[];
    frame_a_body<false, M>(s_dyn_a, f, 0, tr, r, a, ret, qs);
}
// the same launch over a vocabulary of more strips than compute units: persistent filter workgroups (knn_bf16_filter_body_p)
template <int M>
__global__ __launch_bounds__(PIPE_BLOCK) void frame_a_kernel_p(FilterArgs f, int px, TailRoles tr, ResolveArgs r, FwArgs a, RetireArgs ret, QSplitArgs qs) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn_ap[];
    frame_a_body<true, M>(s_dyn_ap, f, px, tr, r, a, ret, qs);
}
#ifdef LCD_B_TIMING   // timing experiment only: start / end of every workgroup of launch B (100 MHz)
__device__ unsigned long long g_b_timing[2 * 4096];
#define B_STAMP(i) do { __builtin_amdgcn_s_barrier(); if (threadIdx.x == 0 && blockIdx.x < 4096) g_b_timing[2 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define B_STAMP(i) do { } while (0)
#endif
constexpr int PIPE_B_BLOCK = 512;   // workgroup size of launch B: eight waves per sealed bucket, two queries per re-rank workgroup
constexpr uint32_t PIPE_B_STAGE_ROWS = 160;   // pending rows a re-rank workgroup stages in LDS at a time
constexpr int APPEND_SPLIT_BUCKETS = 1024;    // sealed buckets (x 256 signatures) from which a deferred append's row writers get a launch of their own
static_assert(PIPE_B_BLOCK == 2 * MF_BLOCK, "the re-rank halves");
// WITH_APPEND: the workgroups that write a deferred append's rows ride in this launch.  Their mere presence changes the register
// allocation of the whole kernel: the scoring branch, which holds everything in registers without them, then parks ~14 values in scratch
// memory and reloads them inside its latency chain (launch B at 10^6 signatures: 47 -> 62 us; compile-time evidence: the kernel's scratch
// accesses by source file, tools/store_wait_audit.py's sibling in DESIGN 7a).  A memory of >= APPEND_SPLIT_BUCKETS sealed buckets
// therefore launches the row writers as a kernel of their own behind launch B (one more launch, ~4 us, against ~15 us of scoring).
// n_wr > 0: workgroups [n_rerank_wgs, n_rerank_wgs + n_wr) belong to the re-rank role but only write the appended rows (rows_only above);
// the re-rank workgroups proper then write none
template <bool WITH_APPEND>
__global__ __launch_bounds__(PIPE_B_BLOCK, 6) void frame_b_kernel(RerankArgs k, int n_rerank_wgs, ScoreArgs A, int n_score_wgs, AppendRowsArgs app, int n_wr) {
    int bid = (int)blockIdx.x;
    B_STAMP(0);
    const bool rows_only = !WITH_APPEND && n_wr > 0 && bid >= n_rerank_wgs && bid < n_rerank_wgs + n_wr;
    const int wr_index = bid - n_rerank_wgs;
    if (!WITH_APPEND && n_wr > 0 && bid >= n_rerank_wgs + n_wr) bid -= n_wr;            // the scoring workgroups follow
    if (WITH_APPEND && bid >= n_rerank_wgs + n_score_wgs) {              // the rows the decision loop of launch A published (deferred append)
        extern __shared__ __attribute__((aligned(16))) float s_dyn_b2[];
        append_rows_body<PIPE_B_BLOCK>(app, bid - n_rerank_wgs - n_score_wgs, app.ap.lds_bytes >= 1024 ? s_dyn_b2 : nullptr, (app.ap.lds_bytes / 256) & ~3);
        B_STAMP(1);
        return;
    }
    if (bid < n_rerank_wgs || rows_only) {                               // (a multiple of 8: see launch_frame_b)
        // consecutive query pairs on one XCD: eight queries share a 128-byte line of the block-major candidate records
        const int pair = rows_only ? 0 : (bid & 7) * (n_rerank_wgs >> 3) + (bid >> 3);
        if (2 * pair >= k.nq) return;
        extern __shared__ __attribute__((aligned(16))) float s_dyn_b[];
        const bool wr_any = !WITH_APPEND && app.ap.enabled && app.ap.defer_rows;
        knn_mfma_rerank_body<64, BF_KEEP, false, true, 2>(2 * pair, k.pk, k.pl, k.n_blocks, k.nq, k.vocab, k.queries, k.row_id, k.norm_max_bits, k.out_row,
                                                          k.out_word, k.out_dist, k.fail_list, k.fail_count, k.cb, k.n_lo, k.n_hi, k.plan_rows,
                                                          s_dyn_b, k.stage_rows, k.f16, k.pend_desc, k.pend_list, k.pend_first_id, k.cross, k.cross_ld
                                                          , app, wr_any && (n_wr == 0 || rows_only), rows_only ? wr_index : pair, n_wr > 0 ? n_wr : (k.nq + 1) / 2
                                                          , rows_only, k.sh_mask, k.sh_q, k.sh_x, k.sh_ld);
        B_STAMP(1);
        return;
    }
    const int g = bid - n_rerank_wgs;
    if (g < A.n_closed_pad) {                                            // consecutive buckets on one XCD: they share directory lines
        const int b = (g & 7) * (A.n_closed_pad >> 3) + (g >> 3);
        if (b < A.n_closed) score_sealed_body<PIPE_B_BLOCK>(A, b);
    } else score_open_body<PIPE_B_BLOCK>(A, g - A.n_closed_pad);
    B_STAMP(1);
}

__global__ __launch_bounds__(PIPE_B_BLOCK) void append_rows_kernel(AppendRowsArgs app) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn_ar[];
    append_rows_body<PIPE_B_BLOCK>(app, (int)blockIdx.x, app.ap.lds_bytes >= 1024 ? s_dyn_ar : nullptr, (app.ap.lds_bytes / 256) & ~3);
}

// ------------------------------------------------------------------------------------------------ row-parallel exact scan
// rowpar_body.cuh; stand-alone launch for the paths without a fused frame tail
__global__ __launch_bounds__(MF_BLOCK) void knn_rowpar_kernel(RowparArgs a, int32_t* __restrict__ fail_count) {
    rowpar_body<64, MF_BLOCK>(a, (int)blockIdx.x, (int)gridDim.x, fail_count);
}
// shard_merge_kernel + the bit rows of selfdist_l2_kernel for a frame whose distance matrix exists already (lcd_kernels.h): wave w of a workgroup
// owns query 4 * blockIdx.x + w; its lanes read the same records (one request), then the query's row of the (symmetric) matrix
__global__ __launch_bounds__(256) void shard_merge_bits_kernel(ShardMergeJob mj, CandBits cb) {
    const int lane = threadIdx.x & 63;
    const int qi = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (qi >= cb.nq) return;                                         // (wave-uniform)
    const ShardMerged m = shard_merge_one(mj.cand, mj.world, mj.rank, cb.nq, qi, mj.by_word);
    const bool v0 = m.dist[0] >= 0.0f && m.word[0] != 0, v1 = m.dist[1] >= 0.0f && m.word[1] != 0;   // cand_threshold(), knn2_kernels.hip
    const float thr = (cb.have_index && v0 && v1) ? m.dist[1] : __int_as_float(0x7f800000);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { mj.out_word[2 * qi + j] = m.word[j]; mj.out_dist[2 * qi + j] = m.dist[j]; mj.out_wslot[2 * qi + j] = m.wslot[j]; }
    }
    cand_bits_row(cb, qi, thr, lane, 64);
}
// A sharded search's last launch: the exact redo, then the rank's candidate records (shard_pack_kernel's work -- one launch of ~4.7 us less per
// frame and rank).  Nothing to redo (the usual frame): every workgroup sees that and packs its stride of the records; the counters are zero
// already.  A redo: the last workgroup to arrive, which merged and knows every result final, packs all records and zeroes the counters
// ([0] rejected queries, [1] arrivals, [3] done) -- every other workgroup has read [0] and taken its ticket by then.
__global__ __launch_bounds__(MF_BLOCK) void knn_rowpar_pack_kernel(RowparArgs a, int32_t* __restrict__ fail_count, ShardPackArgs p) {
    const int st = rowpar_body<64, MF_BLOCK>(a, (int)blockIdx.x, (int)gridDim.x, fail_count);
    if (st == 0) {
        for (int i = (int)blockIdx.x * MF_BLOCK + (int)threadIdx.x; i < p.q2; i += (int)gridDim.x * MF_BLOCK) shard_pack_one(p, i);
    } else if (st == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // (the redone queries' slots were written by this workgroup's other waves)
        for (int i = (int)threadIdx.x; i < p.q2; i += MF_BLOCK) shard_pack_one(p, i);
        if (threadIdx.x < 4 && threadIdx.x != 2) fail_count[threadIdx.x] = 0;
    }
}

}  // namespace
}  // namespace lcd
#ifdef LCD_B_TIMING
extern "C" int lcd_debug_a_timing(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_a_timing), (size_t)n_words * 8);
}
extern "C" int lcd_debug_rr_timing(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_rr_timing), (size_t)n_words * 8);
}
extern "C" int lcd_debug_b_timing(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_b_timing), (size_t)n_words * 8);
}
#endif
#ifdef LCD_MFMA_TIMING
extern "C" int lcd_debug_mfma_timing(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_mf_timing), (size_t)n_words * 8);
}
extern "C" int lcd_debug_mfma_timing2(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_mf_timing2), (size_t)n_words * 8);
}
#endif
namespace lcd {

// ================================================================================================ host side
bool knn_mfma_supported(int dtype, int dim) { return dtype == 0 && dim == 64; }

static int mfma_ng() {   // 32-query groups per wave: 4 (one wave per SIMD, default) or 2 (two waves per SIMD)
    return 4;            // (the two-waves-per-SIMD variant, 2, measured the same)
}

MfmaPlan knn_mfma_plan(int q, int n_rows) {
    MfmaPlan p;
    p.q = q;
    p.qpad = (q + 63) / 64 * 64;
    p.n_rows = n_rows;
    const int n_tiles = (n_rows + 31) / 32;
    const int qw = mfma_ng() * 32;
    const int qgroups = (q + qw - 1) / qw;
    // waves to fill the chip: 256 CUs x 4 SIMDs x (2 waves for NG = 2, 1 wave for NG = 4) -> workgroups of 4 waves
    int nb = ((mfma_ng() == 2 ? 512 : 256) + qgroups - 1) / qgroups;
    if (nb > (n_tiles + MF_WAVES - 1) / MF_WAVES) nb = (n_tiles + MF_WAVES - 1) / MF_WAVES;   // at least one tile per wave
    if (nb < 1) nb = 1;
    nb = (nb + 7) / 8 * 8;
    int tpb = (n_tiles + nb - 1) / nb;
    tpb = (tpb + MF_WAVES - 1) / MF_WAVES * MF_WAVES;               // equal strips for the 4 waves
    if (tpb < MF_WAVES) tpb = MF_WAVES;
    if (tpb > MF_STRIP_TILES * MF_WAVES) tpb = MF_STRIP_TILES * MF_WAVES;   // the in-loop keys index at most 8 tiles per wave
    p.tiles_per_block = tpb;
    p.n_blocks = n_tiles > 0 ? (n_tiles + tpb - 1) / tpb : 0;
    return p;
}
size_t knn_mfma_partial_bytes(const MfmaPlan& p) {
    const size_t nb = (size_t)(p.n_blocks > 0 ? p.n_blocks : 1);
    return nb * MF_KEEP * p.qpad * sizeof(uint64_t) + nb * p.qpad * sizeof(uint32_t);
}

hipError_t launch_row_norms(const void* vocab, const int32_t* row_id, int first, int n, int dim, float* norm, uint32_t* norm_max_bits,
                            hipStream_t s) {
    if (n <= 0) return hipSuccess;
    row_norm_kernel<<<(n + 255) / 256, 256, 0, s>>>((const float*)vocab, row_id, first, n, dim, norm, norm_max_bits);
    return hipGetLastError();
}
hipError_t launch_vocab_tail(float* norm, void* bf, long long first, long long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    vocab_tail_kernel<<<(unsigned)((n * 16 + 255) / 256), 256, 0, s>>>(norm, (uint32_t*)bf, first, n);
    return hipGetLastError();
}
hipError_t launch_norm_tombstone(float* norm, const int32_t* rows, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    norm_tombstone_kernel<<<(n + 255) / 256, 256, 0, s>>>(norm, rows, n);
    return hipGetLastError();
}

__global__ void fail_count_reset_kernel(int32_t* __restrict__ fail_count) {
    if (threadIdx.x < 4 && threadIdx.x != 2) fail_count[threadIdx.x] = 0;
}

hipError_t launch_knn_mfma(int dim, const void* vocab, const float* row_norm, const uint32_t* norm_max_bits, const int32_t* row_id,
                           const void* queries, const MfmaPlan& p, void* partial, int32_t* out_row, int32_t* out_word, float* out_dist,
                           int32_t* fail_list, int32_t* fail_count, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, bool reset_count,
                           const CandBits* cb) {
    if (p.q == 0) return hipSuccess;
    uint64_t* pk = (uint64_t*)partial;
    uint32_t* pl = (uint32_t*)(pk + (size_t)(p.n_blocks > 0 ? p.n_blocks : 1) * MF_KEEP * p.qpad);
    hipError_t e = hipSuccess;
    if (reset_count) {   // [0] rejected queries, [1] arrival counter of the row-parallel redo (the fused frame tail leaves them zeroed)
        // ... and [3], the "redo done" flag: a stand-alone redo (knn_rowpar_kernel) leaves it raised, and the decision workgroup of a fused frame
        // tail that found it raised would not wait for ITS redo ([2], the running maximum of the error ratio, stays).  One launch for the
        // three words (two hipMemsetAsync calls were two fill kernels in front of every stand-alone search)
        fail_count_reset_kernel<<<1, 64, 0, s>>>(fail_count);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (p.n_blocks > 0) {
        const int ng = mfma_ng();
        dim3 grid(p.n_blocks, (p.q + ng * 32 - 1) / (ng * 32));
        if (ev_begin) { e = hipEventRecord(ev_begin, s); if (e != hipSuccess) return e; }
        if (ng == 2)
            knn_mfma_filter_kernel<64, 2><<<grid, MF_BLOCK, 0, s>>>((const float*)vocab, row_norm, p.n_rows, (const float*)queries, p.q, p.qpad,
                                                                     p.tiles_per_block, pk, pl);
        else
            knn_mfma_filter_kernel<64, 4><<<grid, MF_BLOCK, 0, s>>>((const float*)vocab, row_norm, p.n_rows, (const float*)queries, p.q, p.qpad,
                                                                     p.tiles_per_block, pk, pl);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (ev_end) { e = hipEventRecord(ev_end, s); if (e != hipSuccess) return e; }
    }
    if (dim != 64) return hipErrorInvalidValue;
    knn_mfma_rerank_kernel<64, MF_KEEP, true, false><<<p.q, MF_BLOCK, 0, s>>>(pk, pl, p.n_blocks, p.q, (const float*)vocab,
                                                                                   (const float*)queries, row_id, norm_max_bits, out_row,
                                                                                   out_word, out_dist, fail_list, fail_count,
                                                                                   cb ? *cb : CandBits{}, 0, p.n_rows);
    return hipGetLastError();
}

// ---- bf16x3 filter
// compute units the launch plans are made for: the device's own count (lcd_create asks hipDeviceProp_t; 256 on MI355X, also the value the
// plans are tested with on a machine without a device)
static int g_plan_cus = 256;
void knn_set_compute_units(int cus) { if (cus >= 16 && cus <= 4096) g_plan_cus = cus; }
int knn_selfdist_wgs(int q) { return selfdist_tiles(q); }
// other_wgs: workgroups of the same launch that run for about as long as a filter workgroup (distance-matrix tiles, the frame tail's
// two workgroups).  Every workgroup of the launch holds a whole compute unit's LDS: 256 strips + 2 tail workgroups used to leave two
// strips for a second round -- a 12 us workgroup each, the launch took twice as long as its filter.
MfmaPlan knn_bf16_plan(int q, int n_rows, int other_wgs) {
    MfmaPlan p;
    p.q = q;
    p.qpad = (q + 63) / 64 * 64;
    p.n_rows = n_rows;
    p.other_wgs = other_wgs > 0 && other_wgs < g_plan_cus / 2 ? other_wgs : 0;
    const int n_tiles = (n_rows + 31) / 32;
    const int qchunks = (q + BF_QB - 1) / BF_QB;
    const int cus = g_plan_cus - p.other_wgs;
    const int per_q = cus / qchunks > 0 ? cus / qchunks : 1;         // workgroups (4 waves, one per SIMD, a whole compute unit's LDS) per block of 512 queries
    // tiles per workgroup when every compute unit gets one; more than a strip holds (the in-loop keys index 8 tiles): the workgroups
    // are persistent and walk `rounds` equal strips each -- so that a vocabulary a little larger than 256 x 8 tiles does not run as
    // one full round plus a nearly empty one
    int w = (n_tiles + per_q - 1) / per_q;
    if (w < 1) w = 1;
    const int rounds = (w + MF_STRIP_TILES - 1) / MF_STRIP_TILES;
    int tpb = (w + rounds - 1) / rounds;
    if (tpb < 1) tpb = 1;
    if (tpb > MF_STRIP_TILES) tpb = MF_STRIP_TILES;
    p.tiles_per_block = tpb;
    p.n_blocks = n_tiles > 0 ? (n_tiles + tpb - 1) / tpb : 0;
    return p;
}
size_t knn_bf16_partial_bytes(const MfmaPlan& p) {
    const size_t nb = (size_t)(p.n_blocks > 0 ? p.n_blocks : 1);
    return nb * BF_KEEP * p.qpad * sizeof(uint64_t) + nb * p.qpad * sizeof(uint32_t);
}
hipError_t launch_vocab_bf16(const void* vocab, int first, int n, int dim, void* bf, hipStream_t s, int f16) {
    if (n <= 0) return hipSuccess;
    if (dim != 64) return hipErrorInvalidValue;
    vocab_bf16_kernel<<<(n * 16 + 255) / 256, 256, 0, s>>>((const float*)vocab, first, n, (uint32_t*)bf, f16);
    return hipGetLastError();
}
// Persistent filter workgroups per block of 512 queries, or 0: the one-strip kernel (every strip gets its own workgroup; up to one
// workgroup per compute unit that is the faster launch).  MfmaPlan::filter_units (lcd_set_option "filter_units") overrides the
// number of compute units to plan for: tests shorten it to walk many strips per workgroup.
static int bf16_persistent_px(const MfmaPlan& p) {
    const int cus = p.filter_units >= 0 ? p.filter_units : g_plan_cus - p.other_wgs;
    const int qchunks = (p.q + BF_QB - 1) / BF_QB;
    if (p.one_strip || cus <= 0 || qchunks <= 0 || p.n_blocks * qchunks <= (p.filter_units >= 0 ? g_plan_cus : cus)) return 0;
    const int px_max = cus / qchunks > 0 ? cus / qchunks : 1;
    const int rounds = (p.n_blocks + px_max - 1) / px_max;
    return (p.n_blocks + rounds - 1) / rounds;                          // equal shares: ceil(strips / rounds) workgroups of <= rounds strips
}
bool knn_bf16_persistent(const MfmaPlan& p) { return p.q > 0 && bf16_persistent_px(p) > 0; }   // which kernel the launch will be (profile labels)
// The plan of a pipelined frame's filter (launch A, pre-split queries: 66 KB of LDS per workgroup, so two share a compute unit).
//   up to one strip per compute unit the distance tiles leave free: one workgroup per strip, alone on its compute unit;
//   up to TWO strips per compute unit: still one workgroup per strip, in pairs -- a pair runs its tiles at 2.2 us each instead of
//     1.24, but starts and ends once per strip instead of walking two strips one after the other (59 000 words: 34.1 us per
//     frame against 38.0 with persistent workgroups, 80 000 words: 39.0 against 41.6);
//   beyond: persistent workgroups (146 KB of LDS, one per compute unit; the two tail workgroups need units of their own then).
MfmaPlan knn_bf16_plan_pipelined(int q, int n_rows, int n_tile_wgs, int filter_units) {
    MfmaPlan p = knn_bf16_plan(q, n_rows, n_tile_wgs);
    p.filter_units = filter_units;
    if (bf16_persistent_px(p) == 0) return p;
    if (filter_units >= 0) {                                            // (tests: a fixed number of persistent workgroups)
        p = knn_bf16_plan(q, n_rows, 2 + n_tile_wgs);
        p.filter_units = filter_units;
        return p;
    }
    const int n_tiles = (n_rows + 31) / 32;
    const int qchunks = (q + BF_QB - 1) / BF_QB;
    const int slots = (2 * (g_plan_cus - p.other_wgs) - 16) / qchunks;         // strips that can be resident at once (16: the two tails, the pre-split, redo helpers)
    if (slots > 0 && n_tiles <= slots * MF_STRIP_TILES) {
        int tpb = (n_tiles + slots - 1) / slots;
        if (tpb < 1) tpb = 1;
        p.tiles_per_block = tpb;
        p.n_blocks = (n_tiles + tpb - 1) / tpb;
        p.one_strip = 1;
        return p;
    }
    p = knn_bf16_plan(q, n_rows, 2 + n_tile_wgs);
    p.filter_units = filter_units;
    return p;
}
hipError_t launch_knn_bf16(int dim, const void* vocab, const void* vocab_bf, const float* row_norm, const uint32_t* norm_max_bits,
                           const int32_t* row_id, const void* queries, const MfmaPlan& p, void* partial, int32_t* out_row, int32_t* out_word,
                           float* out_dist, int32_t* fail_list, int32_t* fail_count, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end,
                           bool reset_count, const CandBits* cb, bool with_selfdist, hipStream_t s_rerank, hipEvent_t ev_bridge) {
    if (p.q == 0) return hipSuccess;
    if (dim != 64) return hipErrorInvalidValue;
    if (!s_rerank || !ev_bridge) s_rerank = s;
    uint64_t* pk = (uint64_t*)partial;
    uint32_t* pl = (uint32_t*)(pk + (size_t)(p.n_blocks > 0 ? p.n_blocks : 1) * BF_KEEP * p.qpad);
    hipError_t e = hipSuccess;
    if (reset_count) {
        // ... and [3], the "redo done" flag: a stand-alone redo (knn_rowpar_kernel) leaves it raised, and the decision workgroup of a fused frame
        // tail that found it raised would not wait for ITS redo ([2], the running maximum of the error ratio, stays).  One launch for the
        // three words (two hipMemsetAsync calls were two fill kernels in front of every stand-alone search)
        fail_count_reset_kernel<<<1, 64, 0, s>>>(fail_count);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (p.n_blocks > 0) {
        // (one wave per SIMD, NG = 4; the two-waves-per-SIMD variant measured equal and is no longer instantiated)
        static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_bf16_filter_kernel<4, 0>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES);
        static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_bf16_filter_kernel<4, 1>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES);
        (void)attr0; (void)attr1;
        SelfdistJob sd;
        if (with_selfdist && cb) {                                    // the same-frame distance matrix rides along
            sd.queries = (const float*)queries; sd.nq = p.q; sd.out = const_cast<float*>(cb->selfdist); sd.ld = cb->ld; sd.n_tiles = sd.n_self = selfdist_tiles(p.q);
        }
        const int grid = sd.n_tiles + p.n_blocks * ((p.q + BF_QB - 1) / BF_QB);
        const int px = bf16_persistent_px(p);
        if (ev_begin) { e = hipEventRecord(ev_begin, s); if (e != hipSuccess) return e; }
        if (px > 0) {
            static const hipError_t attrp0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_bf16_filter_kernel_p<0>),
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES_P);
            static const hipError_t attrp1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_bf16_filter_kernel_p<1>),
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES_P);
            (void)attrp0; (void)attrp1;
            const int gp = sd.n_tiles + px * ((p.q + BF_QB - 1) / BF_QB);
            if (p.f16) knn_bf16_filter_kernel_p<1><<<gp, 256, BF_LDS_BYTES_P, s>>>((const float*)vocab_bf, row_norm, p.n_rows, (const float*)queries, p.q, p.qpad,
                                                                                   p.tiles_per_block, p.n_blocks, px, pk, pl, sd);
            else knn_bf16_filter_kernel_p<0><<<gp, 256, BF_LDS_BYTES_P, s>>>((const float*)vocab_bf, row_norm, p.n_rows, (const float*)queries, p.q, p.qpad,
                                                                             p.tiles_per_block, p.n_blocks, px, pk, pl, sd);
        } else if (p.f16)
            knn_bf16_filter_kernel<4, 1><<<grid, 256, BF_LDS_BYTES, s>>>((const float*)vocab_bf, row_norm, p.n_rows, (const float*)queries, p.q,
                                                                          p.qpad, p.tiles_per_block, p.n_blocks, pk, pl, sd);
        else
            knn_bf16_filter_kernel<4, 0><<<grid, 256, BF_LDS_BYTES, s>>>((const float*)vocab_bf, row_norm, p.n_rows, (const float*)queries, p.q,
                                                                          p.qpad, p.tiles_per_block, p.n_blocks, pk, pl, sd);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (ev_end) { e = hipEventRecord(ev_end, s); if (e != hipSuccess) return e; }
    }
    if (s_rerank != s) {                                             // the re-rank of this frame on its own stream: the next frame's
        e = hipEventRecord(ev_bridge, s);                             // filter does not have to wait for it
        if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(s_rerank, ev_bridge, 0);
        if (e != hipSuccess) return e;
    }
    knn_mfma_rerank_kernel<64, BF_KEEP, false, true><<<p.q, MF_BLOCK, 0, s_rerank>>>(
        pk, pl, p.n_blocks, p.q, (const float*)vocab, (const float*)queries, row_id, norm_max_bits, out_row, out_word, out_dist, fail_list,
        fail_count, cb ? *cb : CandBits{}, p.f16, p.n_rows);
    return hipGetLastError();
}

hipError_t launch_shard_merge_bits(const ShardMergeJob& mj, const CandBits& cb, hipStream_t s) {
    if (cb.nq <= 0) return hipSuccess;
    if (!mj.cand || !cb.selfdist || !cb.bits || (cb.bw & 1) || cb.ld % 64 != 0) return hipErrorInvalidValue;
    shard_merge_bits_kernel<<<(cb.nq + 3) / 4, 256, 0, s>>>(mj, cb);
    return hipGetLastError();
}

size_t knn_rowpar_partial_bytes(int n_rows, int q) { return (size_t)(q > 0 ? q : 1) * ((n_rows + MF_BLOCK - 1) / MF_BLOCK + 1) * 2 * sizeof(uint64_t); }

hipError_t launch_knn_rowpar(int dim, const void* vocab, const int32_t* row_id, int n_rows, const void* queries, const int32_t* fail_list,
                             int32_t* fail_count, void* partial, int32_t* out_row, int32_t* out_word, float* out_dist, hipStream_t s,
                             const CandBits* cb, const ShardPackArgs* pack) {
    if (dim != 64 || n_rows <= 0) return hipErrorInvalidValue;
    RowparArgs a;
    a.enabled = 1; a.vocab = (const float*)vocab; a.row_id = row_id; a.n_rows = n_rows; a.queries = (const float*)queries;
    a.fail_list = fail_list; a.partial = (unsigned long long*)partial; a.out_row = out_row; a.out_word = out_word; a.out_dist = out_dist;
    if (cb) a.cb = *cb;
    const int nb = (n_rows + MF_BLOCK - 1) / MF_BLOCK;
    if (pack) knn_rowpar_pack_kernel<<<nb, MF_BLOCK, 0, s>>>(a, fail_count, *pack);
    else knn_rowpar_kernel<<<nb, MF_BLOCK, 0, s>>>(a, fail_count);
    return hipGetLastError();
}

// ---- software-pipelined frames (see frame_a_kernel / frame_b_kernel)
int pipe_block_size() { return PIPE_BLOCK; }
int pipe_b_block_size() { return PIPE_B_BLOCK; }

size_t knn_qsplit_bytes(int q) { return (size_t)((q + 63) / 64 * 64) * 256; }

// lcd_set_option "cross_frame_tiles" = 1: launch A also computes the frame's distances to the frame before it, and the re-rank reads its
// pending rows' distances there instead of staging the rows.  Built, bit-identical (tests/test_gpu_append_dev.py), and NOT the default:
// measured on one box with the kernel trace of the driver's command (profiles/r05_ab_notes.txt 9), launch B 19.7 -> 17.5 us while every
// frame creates ~150 words, but the 64 extra tiles share compute units with the filter strips -- launch A 14.2 -> 15.3 us there and
// 13.3 -> 15.0 us once frames mostly revisit (where launch B gains nothing): 0.0417 -> 0.0412 ms per frame over the driver's 20 steps,
// 0.0409 -> 0.0412 over 200, and the filter launch is the one the roofline is quoted on.

hipError_t launch_frame_a(const PipeKnn* kp, const QSplitArgs* qsp, const TailLaunch* resolve, const TailLaunch* reg, hipStream_t s, hipEvent_t ev_begin,
                          hipEvent_t ev_end, const PipeOpts& opt) {
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&frame_a_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)BF_LDS_BYTES_Q);
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&frame_a_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)BF_LDS_BYTES_Q);
    (void)attr0; (void)attr1;
    FilterArgs f{};
    MfmaPlan p;
    p.q = 0; p.qpad = 0; p.n_rows = 0; p.tiles_per_block = 1; p.n_blocks = 0;
    if (kp) {
        const PipeKnn& k = *kp;
        p = k.plan;
        uint64_t* pk = (uint64_t*)k.partial;
        uint32_t* pl = (uint32_t*)(pk + (size_t)(p.n_blocks > 0 ? p.n_blocks : 1) * BF_KEEP * p.qpad);
        if (p.n_shadow > 0 && k.sh_bf && k.sh_norm && k.sh_rows > 0 && k.sh_x) { f.sh_bf = (const float*)k.sh_bf; f.sh_norm = k.sh_norm; f.sh_rows = k.sh_rows; f.sh_x = k.sh_x; f.sh_ld = k.sh_ld; }
        f.vocab_bf = (const float*)k.vocab_bf; f.row_norm = k.row_norm; f.n_rows = p.n_rows; f.queries = (const float*)k.queries; f.nq = p.q; f.qpad = p.qpad;
        f.tiles_per_block = p.tiles_per_block; f.n_blocks = p.n_blocks; f.pk = pk; f.pl = pl; f.n_lo = k.n_lo;
        f.qsplit = (const uint4*)k.qsplit; f.qnorm = k.qnorm;
        if (k.cb.selfdist) {                                          // the same-frame distance matrix rides along
            f.sd.queries = (const float*)k.queries; f.sd.nq = p.q; f.sd.out = const_cast<float*>(k.cb.selfdist); f.sd.ld = k.cb.ld; f.sd.n_tiles = f.sd.n_self = selfdist_tiles(p.q);
        }
        if (opt.cross_frames && k.cross && k.cross_cols && k.cross_ncols > 0 && p.q > 0) {   // ... and so do this frame's distances to the frame before it
            f.sd.queries = (const float*)k.queries; f.sd.nq = p.q;
            f.sd.other = (const float*)k.cross_cols; f.sd.n_other = k.cross_ncols; f.sd.xout = k.cross; f.sd.xld = k.cross_ld;
            f.sd.n_tiles += ((p.q + 63) / 64) * ((k.cross_ncols + 63) / 64);
        }
    }
    f.delay = opt.filter_delay;
    const int px = p.q > 0 ? bf16_persistent_px(p) : 0;
    if (px == 0 && p.q > 0 && (!f.qsplit || !f.qnorm)) return hipErrorInvalidValue;      // the one-strip launch reads pre-split queries
    if (p.n_shadow > 0 && (px > 0 || !f.sh_x)) return hipErrorInvalidValue;   // shadow scores: one-strip launches with the rows at hand only (the engine plans them so)
    TailRoles tr;
    tr.n_filter_wgs = p.q > 0 ? f.sd.n_tiles + (px > 0 ? px : p.n_blocks) * ((p.q + BF_QB - 1) / BF_QB) : 0;
    tr.n_sh_wgs = (p.q > 0 && p.n_shadow > 0) ? ((f.sh_rows + 31) / 32) * ((p.q + BF_QB - 1) / BF_QB) : 0;
    tr.has_resolve = resolve ? 1 : 0; tr.has_register = reg ? 1 : 0; tr.n_redo = resolve ? resolve->n_redo : 0;
    QSplitArgs qs{};
    if (qsp) { qs = *qsp; qs.n_wgs = std::min((qs.qpad * 8 + PIPE_BLOCK - 1) / PIPE_BLOCK, 32); if (qs.n_wgs < 1) qs.n_wgs = 1; }   // one item per thread up to 1 024 descriptors
    tr.n_q_wgs = qsp ? qs.n_wgs : 0;
    const int grid = tr.n_filter_wgs + tr.n_sh_wgs + tr.has_resolve + tr.has_register + tr.n_redo + tr.n_q_wgs;
    if (grid == 0) return hipSuccess;
    const size_t lds = px > 0 ? BF_LDS_BYTES_P : BF_LDS_BYTES_Q;
    if ((resolve && resolve->shmem_resolve > lds) || (reg && reg->shmem + (size_t)reg->a.n * 4 > lds)) return hipErrorInvalidValue;   // (+ the word slots parked in LDS)
    ResolveArgs r{}; FwArgs a{}; RetireArgs ret{};
    if (resolve) { r = resolve->r; r.ap.lds_bytes = (int)lds; }        // what the decision loop's tables leave of it stages the frame's new rows
    if (reg) { a = reg->a; ret = reg->ret; }
    // ev_begin / ev_end: the launch's own start and end time stamps (hipExtLaunchKernel attaches the two events to the dispatch; a pair
    // of hipEventRecord around it costs the stream ~10 us of barrier packets -- and measures the gap in front of the kernel with it)
    const bool timed = ev_begin != nullptr && ev_end != nullptr;
    const bool f16 = kp ? p.f16 != 0 : opt.f16 != 0;                    // (no filter: the variant the handle's filter launches keep hot)
    if (px > 0) {
        static const hipError_t attrp0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&frame_a_kernel_p<0>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES_P);
        static const hipError_t attrp1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&frame_a_kernel_p<1>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES_P);
        (void)attrp0; (void)attrp1;
        if (f16) {
            if (timed) hipExtLaunchKernelGGL(frame_a_kernel_p<1>, dim3(grid), dim3(PIPE_BLOCK), (uint32_t)BF_LDS_BYTES_P, s, ev_begin, ev_end, 0u, f, px, tr, r, a, ret, qs);
            else frame_a_kernel_p<1><<<grid, PIPE_BLOCK, BF_LDS_BYTES_P, s>>>(f, px, tr, r, a, ret, qs);
        } else {
            if (timed) hipExtLaunchKernelGGL(frame_a_kernel_p<0>, dim3(grid), dim3(PIPE_BLOCK), (uint32_t)BF_LDS_BYTES_P, s, ev_begin, ev_end, 0u, f, px, tr, r, a, ret, qs);
            else frame_a_kernel_p<0><<<grid, PIPE_BLOCK, BF_LDS_BYTES_P, s>>>(f, px, tr, r, a, ret, qs);
        }
    } else if (f16) {
        if (timed) hipExtLaunchKernelGGL(frame_a_kernel<1>, dim3(grid), dim3(PIPE_BLOCK), (uint32_t)BF_LDS_BYTES_Q, s, ev_begin, ev_end, 0u, f, tr, r, a, ret, qs);
        else frame_a_kernel<1><<<grid, PIPE_BLOCK, BF_LDS_BYTES_Q, s>>>(f, tr, r, a, ret, qs);
    } else {
        if (timed) hipExtLaunchKernelGGL(frame_a_kernel<0>, dim3(grid), dim3(PIPE_BLOCK), (uint32_t)BF_LDS_BYTES_Q, s, ev_begin, ev_end, 0u, f, tr, r, a, ret, qs);
        else frame_a_kernel<0><<<grid, PIPE_BLOCK, BF_LDS_BYTES_Q, s>>>(f, tr, r, a, ret, qs);
    }
    return hipGetLastError();
}


hipError_t launch_frame_b(const PipeKnn* k, const ScoreArgs* score, int score_wgs, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end,
                          const AppendRowsArgs* app, const PipeOpts& opt) {
    RerankArgs rk{};
    int n_rerank = 0;
    if (k) {
        const MfmaPlan& p = k->plan;
        uint64_t* pk = (uint64_t*)k->partial;
        rk.pk = pk; rk.pl = (uint32_t*)(pk + (size_t)(p.n_blocks > 0 ? p.n_blocks : 1) * BF_KEEP * p.qpad);
        rk.n_blocks = p.n_blocks; rk.nq = p.q; rk.vocab = (const float*)k->vocab; rk.queries = (const float*)k->queries; rk.row_id = k->row_id;
        rk.norm_max_bits = k->norm_max_bits; rk.out_row = k->out_row; rk.out_word = k->out_word; rk.out_dist = k->out_dist;
        rk.fail_list = k->fail_list; rk.fail_count = k->fail_count; rk.cb = k->cb; rk.n_lo = k->n_lo; rk.n_hi = k->n_hi; rk.plan_rows = p.n_rows; rk.f16 = p.f16;
        n_rerank = ((p.q + 1) / 2 + 7) & ~7;                          // two queries per workgroup; padded to the XCD count (frame_b_kernel)
    }
    ScoreArgs A{};
    if (score) A = *score; else score_wgs = 0;
    AppendRowsArgs ar{};
    int n_app = 0;
    if (app && app->ap.enabled && app->ap.defer_rows) { ar = *app; n_app = ar.n_wgs = APPEND_ROW_WGS; }
    if (n_rerank + score_wgs + n_app == 0) return hipSuccess;
    // frames that append their words on the device: 40 KB of dynamic LDS stage 160 pending rows per re-rank workgroup (three workgroups
    // of launch B share a compute unit: 3 x (40 + 10) KB of its 160 KB); the workgroups that write appended rows stage them there too
    const uint32_t dyn = ((k && k->n_hi) || n_app) ? PIPE_B_STAGE_ROWS * 256u + 512u : 0u;   // + the workgroup's two queries
    rk.stage_rows = (dyn && k && k->n_hi) ? (int)PIPE_B_STAGE_ROWS : 0;
    ar.ap.lds_bytes = (int)(PIPE_B_STAGE_ROWS * 256u);
    if (k && n_app) { rk.pend_desc = ar.ap.descriptors; rk.pend_list = ar.ap.list_out; rk.pend_first_id = ar.ap.first_id; }   // k's pending rows ARE the rows being written
    // ... and their distances to k's queries were computed by launch A of this pair (selfdist_tile's cross-frame tiles), if it knew both frames
    if (k && n_app && opt.cross_frames && k->cross && k->cross_cols == (const void*)ar.ap.descriptors) { rk.cross = k->cross; rk.cross_ld = k->cross_ld; }
    // Who writes the rows of the deferred append: the re-rank workgroups (they hold the rows in their staging area) -- no third branch in the
    // kernel, whose presence makes the scoring branch spill: launch B 15.4 -> 13.8 us once frames create few words.  While frames create
    // ~150 words each (the driver's 20 steps) the two ways are within box-to-box noise of each other (0.0387 against 0.0380 ms per frame
    // on one box, 0.0418 against 0.0427 on another), and choosing per launch by the expected number of new rows lost to both (the second
    // kernel variant is loaded in the middle of the stream): profiles/r05_ab_notes.txt 6.  "append_from_rerank" = 0 keeps round 4's eight
    // row-writer workgroups for A/B runs.
    const bool writers = opt.append_from_rerank == 0;
    if (!writers && k && n_app > 0 && k->n_hi && k->plan.q > 0 && rk.stage_rows >= 4) n_app = 0;   // the re-rank workgroups write the rows (`ar` stays filled in: they get it);
                                                                                      // without re-rank workgroups or a staging area the writers stay
    // shadow rows: the filter of launch A ranked the descriptors of the frame whose rows are written here; the re-rank needs that frame's new-word mask
    const bool shadow = k && k->plan.n_shadow > 0 && k->sh_mask && k->sh_x && k->sh_q > 0 && app && app->ap.enabled && app->ap.defer_rows && rk.pend_desc && !rk.cross;
    if (k && k->plan.n_shadow > 0 && !shadow) return hipErrorInvalidValue;          // (launch A wrote scores nobody reads: the engine plans the two together)
    if (shadow) { rk.sh_mask = k->sh_mask; rk.sh_q = k->sh_q; rk.sh_x = k->sh_x; rk.sh_ld = k->sh_ld; }
    // round 6: rows written by n_wr extra workgroups of the re-rank role instead of by the re-rank workgroups themselves ("row_writer_wgs");
    // with shadow rows the re-rank workgroups stage nothing they could write, so the writers are not optional
    const int n_wr = (!writers && n_app == 0 && k && app && app->ap.enabled && app->ap.defer_rows && k->n_hi && k->plan.q > 0 && rk.stage_rows >= 4 && !rk.cross &&
                      (opt.row_writer_wgs > 0 || shadow)) ? (opt.row_writer_wgs > 0 ? opt.row_writer_wgs : 16) : 0;
    if (shadow && n_wr == 0) return hipErrorInvalidValue;
    const bool split = n_app > 0 && score && score->n_closed >= (opt.append_split_buckets >= 0 ? opt.append_split_buckets : APPEND_SPLIT_BUCKETS);   // (see frame_b_kernel)
    if (split || n_app == 0) {
        if (n_rerank + score_wgs > 0) {
            if (ev_begin != nullptr && ev_end != nullptr)
                hipExtLaunchKernelGGL(frame_b_kernel<false>, dim3(n_rerank + n_wr + score_wgs), dim3(PIPE_B_BLOCK), dyn, s, ev_begin, ev_end, 0u, rk, n_rerank, A, score_wgs, ar, n_wr);
            else frame_b_kernel<false><<<n_rerank + n_wr + score_wgs, PIPE_B_BLOCK, dyn, s>>>(rk, n_rerank, A, score_wgs, ar, n_wr);
        }
        if (n_app > 0) append_rows_kernel<<<n_app, PIPE_B_BLOCK, PIPE_B_STAGE_ROWS * 256u, s>>>(ar);
        return hipGetLastError();
    }
    if (ev_begin != nullptr && ev_end != nullptr)
        hipExtLaunchKernelGGL(frame_b_kernel<true>, dim3(n_rerank + score_wgs + n_app), dim3(PIPE_B_BLOCK), dyn, s, ev_begin, ev_end, 0u, rk, n_rerank, A, score_wgs, ar, 0);
    else frame_b_kernel<true><<<n_rerank + score_wgs + n_app, PIPE_B_BLOCK, dyn, s>>>(rk, n_rerank, A, score_wgs, ar, 0);
    return hipGetLastError();
}

}  // namespace lcd

// The launch plan of a pipelined frame's filter for a vocabulary of n_rows and q descriptors, as the engine makes it (tests; no device
// needed): out[0] tiles per workgroup, [1] strips, [2] persistent workgroups per block of 512 queries (0: one workgroup per strip),
// [3] distance-tile workgroups riding in the launch, [4] query blocks
extern "C" int lcd_debug_frame_plan(int q, int n_rows, int new_words_compared, int* out) {
    if (q <= 0 || n_rows <= 0 || !out) return -1;
    const int tiles = new_words_compared ? lcd::knn_selfdist_wgs(q) : 0;
    const lcd::MfmaPlan p = lcd::knn_bf16_plan_pipelined(q, n_rows, tiles, -1);
    out[0] = p.tiles_per_block; out[1] = p.n_blocks; out[2] = lcd::bf16_persistent_px(p); out[3] = tiles; out[4] = (q + lcd::BF_QB - 1) / lcd::BF_QB;
    return 0;
}

#ifdef LCD_SCORE_TIMING   // timing experiment only: the phase stamps of the scoring workgroups of launch B (this translation unit's copy)
extern "C" int lcd_debug_score_timing_pipe(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_score_timing), (size_t)n_words * 8);
}
#endif

#ifdef LCD_TAIL_TIMING   // timing experiment only: the stamps of the tail that ran inside the fused filter launch (this translation unit's copy)
extern "C" int lcd_debug_tail_timing_pipe(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_tail_timing), 64) != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(lcd::g_resolve_timing), 64) != hipSuccess) return -3;
    return (int)hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(lcd::g_sweep_timing), 256);
}
#endif
