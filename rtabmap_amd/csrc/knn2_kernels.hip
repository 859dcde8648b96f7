// knn2_kernels.hip -- exact brute-force 2-NN (squared L2 / Hamming) over the device-resident vocabulary, gfx950.
//
// Replaces, for the quantisation hot loop of VWDictionary::addNewWords / findNN (reference VWDictionary.cpp:1015-1086,
// 1347-1404): rtflann LinearIndex::findNeighbors (linear_index.h:129-144) with KNNSimpleResultSet (result_set.h:151-171)
// and the cv::BFMatcher / cv::cuda brute-force matchers (VWDictionary.cpp:1027-1028, 1053-1066).
//
// Mapping (wave64, no LDS staging needed):
//   * one LANE owns one query descriptor and keeps it in VGPRs for the whole kernel (64 floats / 8 dwords);
//   * a WAVE walks a contiguous strip of vocabulary rows; the row address is wave-uniform, so the row is fetched
//     with scalar loads (s_load_dwordx8/x16 through the scalar cache) and fed to the VALU as SGPR operands --
//     the descriptor matrix is read from HBM/L2 exactly once per 64 queries, perfectly coalesced, zero VGPR cost;
//   * each lane keeps a running (best, second) pair of packed keys (distance << 32 | row): "lower row wins ties"
//     is part of the integer comparison, no cross-lane traffic in the loop;
//   * the 4 waves of a workgroup cover 4 strips for the same 64 queries and merge through LDS once; the per-workgroup
//     partials [n_blocks][2][qpad] are merged by knn2_merge_kernel (one wave per query, __shfl_xor butterfly).
//   * grid.x = row blocks (a multiple of 8 so that the blocks sharing a row range land on one XCD/L2), grid.y = 64-query
//     groups.
//
// Arithmetic is the reference's own, bit for bit:
//   L2: ((d0*d0 + d1*d1) + d2*d2) + d3*d3 per group of four, added to one running float (dist.h:158-166), every product
//       and sum individually rounded (__fmul_rn/__fadd_rn: no FMA contraction);
//   Hamming: popcount(a ^ b) (dist.h:555-579).
// Bound: VALU issue (3 VALU ops per float element, 2+ per dword for Hamming), not HBM: see DESIGN.md.
#include "lcd_kernels.h"
#include "shard_body.cuh"

namespace lcd {
namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = 4;
constexpr int HSHIFT = 21;                       // Hamming packed key: (dist << 21) | row-in-block
constexpr uint32_t HROWMASK = (1u << HSHIFT) - 1;

__device__ __forceinline__ void top2_push(uint64_t& best, uint64_t& second, uint64_t k) {
    const uint64_t hi = best > k ? best : k;
    best = best < k ? best : k;
    second = second < hi ? second : hi;
}
__device__ __forceinline__ void top2_push32(uint32_t& best, uint32_t& second, uint32_t k) {
    const uint32_t hi = max(best, k);
    best = min(best, k);
    second = min(second, hi);
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((uint64_t)hi << 32) | lo;
}

// rtflann::L2<float>::operator() (dist.h:150-177), a = vocabulary row (wave-uniform), b = the lane's query
template <int DIM>
__device__ __forceinline__ float l2_ref(const float* __restrict__ row, const float (&q)[DIM]) {
    float res = 0.0f;
#pragma unroll
    for (int g = 0; g + 3 < DIM; g += 4) {
        const float d0 = __fsub_rn(row[g + 0], q[g + 0]);
        const float d1 = __fsub_rn(row[g + 1], q[g + 1]);
        const float d2 = __fsub_rn(row[g + 2], q[g + 2]);
        const float d3 = __fsub_rn(row[g + 3], q[g + 3]);
        float t = __fmul_rn(d0, d0);
        t = __fadd_rn(t, __fmul_rn(d1, d1));
        t = __fadd_rn(t, __fmul_rn(d2, d2));
        t = __fadd_rn(t, __fmul_rn(d3, d3));
        res = __fadd_rn(res, t);
    }
#pragma unroll
    for (int g = DIM & ~3; g < DIM; ++g) {
        const float d0 = __fsub_rn(row[g], q[g]);
        res = __fadd_rn(res, __fmul_rn(d0, d0));
    }
    return res;
}
// any dimension: the query is re-read from memory (L1-resident) -- correctness path for unusual descriptor sizes
__device__ __forceinline__ float l2_ref_dyn(const float* __restrict__ row, const float* __restrict__ q, int dim) {
    float res = 0.0f;
    int g = 0;
    for (; g + 3 < dim; g += 4) {
        const float d0 = __fsub_rn(row[g + 0], q[g + 0]);
        const float d1 = __fsub_rn(row[g + 1], q[g + 1]);
        const float d2 = __fsub_rn(row[g + 2], q[g + 2]);
        const float d3 = __fsub_rn(row[g + 3], q[g + 3]);
        float t = __fmul_rn(d0, d0);
        t = __fadd_rn(t, __fmul_rn(d1, d1));
        t = __fadd_rn(t, __fmul_rn(d2, d2));
        t = __fadd_rn(t, __fmul_rn(d3, d3));
        res = __fadd_rn(res, t);
    }
    for (; g < dim; ++g) {
        const float d0 = __fsub_rn(row[g], q[g]);
        res = __fadd_rn(res, __fmul_rn(d0, d0));
    }
    return res;
}

template <int W>
__device__ __forceinline__ uint32_t hamming_ref(const uint32_t* __restrict__ row, const uint32_t (&q)[W]) {
    uint32_t d = 0;
    // The bit counts of a row accumulate in the instruction itself (v_bcnt_u32_b32 d, x, d = popcount(x) + d), one chain per row.  Left
    // to the compiler the eight counts of a 256-bit row are summed as a tree -- eight v_bcnt_u32_b32 + three v_add3_u32
    // (tools/isa_loop_histogram.py: 94 VALU per trip of four rows, 82 this way; SURVEY.md 8d counts 16 per row: 8 xor + 8 counts); the
    // four rows of a trip keep four chains in flight.  An integer sum: the same number in any order.  Measured (round 5, same box,
    // 200 000 words x 500 descriptors): scan 77.6 -> 67.3 us, -13 % (profiles/r05_first_call.txt).
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t x = row[w] ^ q[w];
        asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(d));
    }
    return d;
}
__device__ __forceinline__ uint32_t hamming_dyn(const uint32_t* __restrict__ row, const uint32_t* __restrict__ q, int w32) {
    uint32_t d = 0;
    for (int w = 0; w < w32; ++w) d += __popc(row[w] ^ q[w]);
    return d;
}

struct Strip { int begin, end; };
// rows [row0, row1) of the workgroup split into WAVES contiguous strips
__device__ __forceinline__ Strip wave_strip(int row0, int row1, int wave) {
    const int per = (row1 - row0 + WAVES - 1) / WAVES;
    Strip s;
    s.begin = min(row0 + wave * per, row1);
    s.end = min(s.begin + per, row1);
    return s;
}

// cross-wave merge + partial store.  partial layout: [block][slot][qpad]
__device__ __forceinline__ void block_merge_store(uint64_t best, uint64_t second, int wave, int lane, int qi, int qpad,
                                                  uint64_t* __restrict__ partial) {
    __shared__ uint64_t s_key[WAVES][2][64];
    s_key[wave][0][lane] = best;
    s_key[wave][1][lane] = second;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            top2_push(best, second, s_key[w][0][lane]);
            top2_push(best, second, s_key[w][1][lane]);
        }
        partial[((size_t)blockIdx.x * 2 + 0) * qpad + qi] = best;
        partial[((size_t)blockIdx.x * 2 + 1) * qpad + qi] = second;
    }
}

// ------------------------------------------------------------------------------------------------ L2, fixed DIM
template <int DIM>
__global__ __launch_bounds__(BLOCK) void knn2_l2_kernel(const float* __restrict__ vocab, const int32_t* __restrict__ row_id,
                                                        int n_rows, const float* __restrict__ queries, int nq, int qpad,
                                                        int rows_per_block, uint64_t* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {
        const int qi = blockIdx.y * 64 + lane;
        const int qsrc = qi < nq ? qi : nq - 1;        // tail lanes repeat the last query; their results are never read
        float q[DIM];
        {
            const float4* src = reinterpret_cast<const float4*>(queries + (size_t)qsrc * DIM);
#pragma unroll
            for (int g = 0; g < DIM / 4; ++g) {
                const float4 v = src[g];
                q[4 * g + 0] = v.x; q[4 * g + 1] = v.y; q[4 * g + 2] = v.z; q[4 * g + 3] = v.w;
            }
        }
        const int row0 = blockIdx.x * rows_per_block;
        const int row1 = min(row0 + rows_per_block, n_rows);
        const Strip s = wave_strip(row0, row1, wave);
        uint64_t best = KEY_NONE, second = KEY_NONE;
        for (int r = s.begin; r < s.end; ++r) {
            if (row_id[r] == 0) continue;               // tombstone (wave-uniform branch)
            const float d = l2_ref<DIM>(vocab + (size_t)r * DIM, q);
            top2_push(best, second, ((uint64_t)__float_as_uint(d) << 32) | (uint32_t)r);
        }
        block_merge_store(best, second, wave, lane, qi, qpad, partial);
    }
}

__global__ __launch_bounds__(BLOCK) void knn2_l2_dyn_kernel(const float* __restrict__ vocab, const int32_t* __restrict__ row_id,
                                                            int n_rows, int dim, const float* __restrict__ queries, int nq,
                                                            int qpad, int rows_per_block, uint64_t* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.y * 64 + lane;
    const float* q = queries + (size_t)(qi < nq ? qi : nq - 1) * dim;
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(row0 + rows_per_block, n_rows);
    const Strip s = wave_strip(row0, row1, wave);
    uint64_t best = KEY_NONE, second = KEY_NONE;
    for (int r = s.begin; r < s.end; ++r) {
        if (row_id[r] == 0) continue;
        const float d = l2_ref_dyn(vocab + (size_t)r * dim, q, dim);
        top2_push(best, second, ((uint64_t)__float_as_uint(d) << 32) | (uint32_t)r);
    }
    block_merge_store(best, second, wave, lane, qi, qpad, partial);
}

// ------------------------------------------------------------------------------------------------ Hamming, W dwords
template <int W>
__global__ __launch_bounds__(BLOCK) void knn2_hamming_kernel(const uint32_t* __restrict__ vocab, const int32_t* __restrict__ row_id,
                                                             int n_rows, const uint32_t* __restrict__ queries, int nq, int qpad,
                                                             int rows_per_block, uint64_t* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.y * 64 + lane;
    const int qsrc = qi < nq ? qi : nq - 1;
    uint32_t q[W];
#pragma unroll
    for (int w = 0; w < W; ++w) q[w] = queries[(size_t)qsrc * W + w];
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(row0 + rows_per_block, n_rows);
    const Strip s = wave_strip(row0, row1, wave);
    uint32_t best = ~0u, second = ~0u;              // (dist << 21) | (row - row0): one v_min/v_max each per candidate
    int r = s.begin;
    // 4 rows per trip: the 4 scalar row loads (and the 4 tombstone flags) are issued back to back, so one wave has
    // 128 B of vocabulary in flight while it works; tombstones are masked by a wave-uniform select, not a branch.
    // the row and flag addresses are two pointers that advance, not recomputed from the row index on every trip (46 -> 25 scalar instructions)
    const uint32_t* vp = vocab + (size_t)r * W;
    const int32_t* ip = row_id + r;
    for (; r + 4 <= s.end; r += 4, vp += 4 * W, ip += 4) {
        uint32_t key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t d = hamming_ref<W>(vp + u * W, q);
            key[u] = ip[u] != 0 ? ((d << HSHIFT) | (uint32_t)(r + u - row0)) : ~0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) top2_push32(best, second, key[u]);
    }
    for (; r < s.end; ++r) {
        if (row_id[r] == 0) continue;
        const uint32_t d = hamming_ref<W>(vocab + (size_t)r * W, q);
        top2_push32(best, second, (d << HSHIFT) | (uint32_t)(r - row0));
    }
    const uint64_t b64 = best == ~0u ? KEY_NONE : (((uint64_t)(best >> HSHIFT) << 32) | (uint32_t)(row0 + (best & HROWMASK)));
    const uint64_t s64 = second == ~0u ? KEY_NONE : (((uint64_t)(second >> HSHIFT) << 32) | (uint32_t)(row0 + (second & HROWMASK)));
    block_merge_store(b64, s64, wave, lane, qi, qpad, partial);
}

__global__ __launch_bounds__(BLOCK) void knn2_hamming_dyn_kernel(const uint32_t* __restrict__ vocab, const int32_t* __restrict__ row_id,
                                                                 int n_rows, int w32, const uint32_t* __restrict__ queries, int nq,
                                                                 int qpad, int rows_per_block, uint64_t* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.y * 64 + lane;
    const uint32_t* q = queries + (size_t)(qi < nq ? qi : nq - 1) * w32;
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(row0 + rows_per_block, n_rows);
    const Strip s = wave_strip(row0, row1, wave);
    uint64_t best = KEY_NONE, second = KEY_NONE;
    for (int r = s.begin; r < s.end; ++r) {
        if (row_id[r] == 0) continue;
        const uint32_t d = hamming_dyn(vocab + (size_t)r * w32, q, w32);
        top2_push(best, second, ((uint64_t)d << 32) | (uint32_t)r);
    }
    block_merge_store(best, second, wave, lane, qi, qpad, partial);
}

// ------------------------------------------------------------------------------------------------ merge
// one wave per query: lanes stride over the [n_blocks*2] partial keys, then a 6-step butterfly
__global__ __launch_bounds__(BLOCK) void knn2_merge_kernel(int dtype, const uint64_t* __restrict__ partial, int n_keys, int qpad,
                                                           int nq, const int32_t* __restrict__ row_id,
                                                           int32_t* __restrict__ out_row, int32_t* __restrict__ out_word,
                                                           float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (qi >= nq) return;
    uint64_t best = KEY_NONE, second = KEY_NONE;
    for (int c = lane; c < n_keys; c += 64) top2_push(best, second, partial[(size_t)c * qpad + qi]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t ob = shfl_xor_u64(best, m), os = shfl_xor_u64(second, m);
        top2_push(best, second, ob);
        top2_push(best, second, os);
    }
    if (lane == 0) {
        const int qo = qi;
        const uint64_t k[2] = {best, second};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (k[j] == KEY_NONE) {
                out_row[2 * qo + j] = -1; out_word[2 * qo + j] = 0; out_dist[2 * qo + j] = -1.0f;
            } else {
                const uint32_t row = (uint32_t)k[j], hi = (uint32_t)(k[j] >> 32);
                out_row[2 * qo + j] = (int32_t)row;
                out_word[2 * qo + j] = row_id[row];
                out_dist[2 * qo + j] = dtype == 0 ? __uint_as_float(hi) : (float)hi;   // VWDictionary.cpp:1078-1083
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ self distances
// out[r*ld + qi] = dist(query r, query qi): lane = qi (coalesced stores), each wave walks exactly 8 rows r of the
// same matrix (one byte of the bit row below).  Optionally also emits, for the addNewWords resolution (resolve_kernels.hip), the candidate bit matrix
//     bits[qi][r / 32] bit (r & 31)  =  dist(r, qi) < thr(qi)
// where thr(qi) is the distance of qi's second indexed neighbour (+inf when it has fewer than two): a same-frame new
// word r can only enter the two best candidates of descriptor qi if it is strictly closer than that neighbour
// (std::multimap keeps the indexed entries first on equal keys, VWDictionary.cpp:1091-1160).
constexpr int SD_WROWS = 8;                  // rows per wave = bits per stored byte
constexpr int SD_ROWS = SD_WROWS * WAVES;

__device__ __forceinline__ float cand_threshold(int have_index, const int32_t* __restrict__ knn_word,
                                                const float* __restrict__ knn_dist, int qi) {
    if (!have_index) return __int_as_float(0x7f800000);
    const bool v0 = knn_dist[2 * qi] >= 0.0f && knn_word[2 * qi] != 0;
    const bool v1 = knn_dist[2 * qi + 1] >= 0.0f && knn_word[2 * qi + 1] != 0;
    return (v0 && v1) ? knn_dist[2 * qi + 1] : __int_as_float(0x7f800000);
}

template <int DIM>
__global__ __launch_bounds__(BLOCK) void selfdist_l2_kernel(const float* __restrict__ queries, int nq, float* __restrict__ out, int ld,
                                                            int have_index, const int32_t* __restrict__ knn_word,
                                                            const float* __restrict__ knn_dist, uint32_t* __restrict__ bits, int bw,
                                                            ShardMergeJob mj) {
    // The workgroup's SD_ROWS rows are staged through LDS in the round trip that brings every lane its own query (round 6: the rows used to be
    // read from memory one after the other inside the loop, eight dependent round trips per wave: 10.8 us for 500 x 500 distances).  The
    // arithmetic is l2_ref's, operand for operand: the distances are the reference's bits either way.
    __shared__ float s_rows[SD_ROWS * DIM];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.y * 64 + lane;
    const int qsrc = qi < nq ? qi : nq - 1;
    constexpr int V = SD_ROWS * DIM / 4;                             // float4s of the staged rows
    const int row_first = blockIdx.x * SD_ROWS;
    float4 stage[(V + BLOCK - 1) / BLOCK];
#pragma unroll
    for (int u = 0; u < (V + BLOCK - 1) / BLOCK; ++u) {
        const int v = (int)threadIdx.x + u * BLOCK;
        const int r = min(row_first + v / (DIM / 4), nq - 1);        // rows behind the frame's last: clamped, never used
        stage[u] = v < V ? reinterpret_cast<const float4*>(queries + (size_t)r * DIM)[v % (DIM / 4)] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    float q[DIM];
    const float4* src = reinterpret_cast<const float4*>(queries + (size_t)qsrc * DIM);
#pragma unroll
    for (int g = 0; g < DIM / 4; ++g) {
        const float4 v = src[g];
        q[4 * g + 0] = v.x; q[4 * g + 1] = v.y; q[4 * g + 2] = v.z; q[4 * g + 3] = v.w;
    }
    float thr = 0.0f;
    if (mj.cand) {
        // a sharded frame: the query's global 2-NN is merged here from the ranks' records (shard_merge_kernel's work: one launch less per frame
        // and rank) -- every workgroup needs its queries' thresholds, the first row block's first wave also writes the result for the decision loop
        const ShardMerged m = shard_merge_one(mj.cand, mj.world, mj.rank, nq, qsrc, mj.by_word);
        const bool v0 = m.dist[0] >= 0.0f && m.word[0] != 0, v1 = m.dist[1] >= 0.0f && m.word[1] != 0;   // cand_threshold()
        thr = (have_index && v0 && v1) ? m.dist[1] : __int_as_float(0x7f800000);
        if (blockIdx.x == 0 && wave == 0 && qi < nq) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { mj.out_word[2 * qi + j] = m.word[j]; mj.out_dist[2 * qi + j] = m.dist[j]; mj.out_wslot[2 * qi + j] = m.wslot[j]; }
        }
    } else if (bits) thr = cand_threshold(have_index, knn_word, knn_dist, qsrc);
#pragma unroll
    for (int u = 0; u < (V + BLOCK - 1) / BLOCK; ++u) {
        const int v = (int)threadIdx.x + u * BLOCK;
        if (v < V) reinterpret_cast<float4*>(s_rows)[v] = stage[u];
    }
    __syncthreads();
    const int r0 = row_first + wave * SD_WROWS;
    const int r1 = min(r0 + SD_WROWS, nq);
    uint32_t word = 0;
    for (int r = r0; r < r1; ++r) {
        const float d = l2_ref<DIM>(s_rows + (size_t)(r - row_first) * DIM, q);
        if (qi < nq) out[(size_t)r * ld + qi] = d;
        word |= (d < thr ? 1u : 0u) << (r - r0);
    }
    if (bits && qi < nq && r0 < nq) reinterpret_cast<unsigned char*>(bits)[(size_t)qi * bw * 4 + (r0 >> 3)] = (unsigned char)word;
}
__global__ __launch_bounds__(BLOCK) void selfdist_l2_dyn_kernel(const float* __restrict__ queries, int nq, int dim, float* __restrict__ out,
                                                                int ld, int have_index, const int32_t* __restrict__ knn_word,
                                                                const float* __restrict__ knn_dist, uint32_t* __restrict__ bits, int bw) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.y * 64 + lane;
    const int qsrc = qi < nq ? qi : nq - 1;
    const float* q = queries + (size_t)qsrc * dim;
    const float thr = bits ? cand_threshold(have_index, knn_word, knn_dist, qsrc) : 0.0f;
    const int r0 = blockIdx.x * SD_ROWS + wave * SD_WROWS;
    const int r1 = min(r0 + SD_WROWS, nq);
    uint32_t word = 0;
    for (int r = r0; r < r1; ++r) {
        const float d = l2_ref_dyn(queries + (size_t)r * dim, q, dim);
        if (qi < nq) out[(size_t)r * ld + qi] = d;
        word |= (d < thr ? 1u : 0u) << (r - r0);
    }
    if (bits && qi < nq && r0 < nq) reinterpret_cast<unsigned char*>(bits)[(size_t)qi * bw * 4 + (r0 >> 3)] = (unsigned char)word;
}
__global__ __launch_bounds__(BLOCK) void selfdist_hamming_dyn_kernel(const uint32_t* __restrict__ queries, int nq, int w32,
                                                                     float* __restrict__ out, int ld, int have_index,
                                                                     const int32_t* __restrict__ knn_word,
                                                                     const float* __restrict__ knn_dist, uint32_t* __restrict__ bits, int bw) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.y * 64 + lane;
    const int qsrc = qi < nq ? qi : nq - 1;
    const uint32_t* q = queries + (size_t)qsrc * w32;
    const float thr = bits ? cand_threshold(have_index, knn_word, knn_dist, qsrc) : 0.0f;
    const int r0 = blockIdx.x * SD_ROWS + wave * SD_WROWS;
    const int r1 = min(r0 + SD_WROWS, nq);
    uint32_t word = 0;
    for (int r = r0; r < r1; ++r) {
        const float d = (float)hamming_dyn(queries + (size_t)r * w32, q, w32);
        if (qi < nq) out[(size_t)r * ld + qi] = d;
        word |= (d < thr ? 1u : 0u) << (r - r0);
    }
    if (bits && qi < nq && r0 < nq) reinterpret_cast<unsigned char*>(bits)[(size_t)qi * bw * 4 + (r0 >> 3)] = (unsigned char)word;
}

// ------------------------------------------------------------------------------------------------ merge + same-frame distances, one launch (round 6)
// The Hamming frame (config 3) ran the merge of the scan's partial keys and the same-frame distance matrix as two dependent launches of ~5 us each
// -- the second only because the candidate bits need each query's second-neighbour distance.  One wave per query does both: the merge as
// knn2_merge_kernel, then the query's ROW of the (symmetric) distance matrix -- lane l takes descriptors l, l + 64, ... -- whose distances are computed
// while the winner's word id is on its way, and the bit row from ballots once the threshold is known.
__global__ __launch_bounds__(BLOCK) void knn2_merge_selfdist_hamming_kernel(const uint64_t* __restrict__ partial, int n_keys, int qpad, int nq,
                                                                            const int32_t* __restrict__ row_id, int32_t* __restrict__ out_row,
                                                                            int32_t* __restrict__ out_word, float* __restrict__ out_dist,
                                                                            const uint32_t* __restrict__ queries, int w32, float* __restrict__ sd, int ld,
                                                                            int have_index, uint32_t* __restrict__ bits, int bw) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (qi >= nq) return;
    // the query's row of the distance matrix first: it needs nothing from the merge, and its reads travel with the partial keys'
    const uint32_t* q = queries + (size_t)qi * w32;
    constexpr int MAXK = 8;                                          // frames of up to 512 descriptors keep their row in registers; longer ones loop again
    uint32_t dreg[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int r = 64 * k + lane;
        dreg[k] = (r < nq) ? hamming_dyn(queries + (size_t)r * w32, q, w32) : 0xFFFFFFFFu;
    }
    uint64_t best = KEY_NONE, second = KEY_NONE;
    for (int c = lane; c < n_keys; c += 64) top2_push(best, second, partial[(size_t)c * qpad + qi]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t ob = shfl_xor_u64(best, m), os = shfl_xor_u64(second, m);
        top2_push(best, second, ob);
        top2_push(best, second, os);
    }
    // (every lane holds the merged pair) the two word ids, needed for the threshold below
    int32_t w0 = 0, w1 = 0;
    if (best != KEY_NONE) w0 = row_id[(uint32_t)best];
    if (second != KEY_NONE) w1 = row_id[(uint32_t)second];
    const float d0 = best == KEY_NONE ? -1.0f : (float)(uint32_t)(best >> 32), d1 = second == KEY_NONE ? -1.0f : (float)(uint32_t)(second >> 32);
    if (lane == 0) {
        out_row[2 * qi] = best == KEY_NONE ? -1 : (int32_t)(uint32_t)best;       out_word[2 * qi] = w0;     out_dist[2 * qi] = d0;       // VWDictionary.cpp:1078-1083
        out_row[2 * qi + 1] = second == KEY_NONE ? -1 : (int32_t)(uint32_t)second; out_word[2 * qi + 1] = w1; out_dist[2 * qi + 1] = d1;
    }
    // cand_threshold(): the second indexed neighbour's distance when both neighbours are valid, else +inf
    const bool v0 = d0 >= 0.0f && w0 != 0, v1 = d1 >= 0.0f && w1 != 0;
    const float thr = (have_index && v0 && v1) ? d1 : __int_as_float(0x7f800000);
    for (int k = 0; 64 * k < ld; ++k) {
        const int r = 64 * k + lane;
        uint32_t du = k < MAXK ? dreg[k < MAXK ? k : 0] : 0xFFFFFFFFu;
        if (k >= MAXK) du = (r < nq) ? hamming_dyn(queries + (size_t)r * w32, q, w32) : 0xFFFFFFFFu;
        const float d = (float)du;
        if (r < nq) sd[(size_t)qi * ld + r] = d;                      // row qi of the symmetric matrix: D[qi][r] == D[r][qi]
        const unsigned long long bal = __ballot(r < nq && d < thr);
        if (bits && lane == 0) { bits[(size_t)qi * bw + 2 * k] = (uint32_t)bal; bits[(size_t)qi * bw + 2 * k + 1] = (uint32_t)(bal >> 32); }
    }
}

}  // namespace

// ================================================================================================ host side
KnnPlan knn_plan(int q, int n_rows, int dim_bytes) {
    (void)dim_bytes;
    KnnPlan p;
    p.q = q;
    p.qpad = (q + 63) / 64 * 64;
    p.n_rows = n_rows;
    const int qgroups = p.qpad / 64;
    // aim at ~6 waves per SIMD over the chip (256 CUs x 4 SIMDs) so that scalar-load latency is covered by other waves
    const int target_blocks = (256 * 4 * 6 + WAVES * qgroups - 1) / (WAVES * qgroups);
    int nb = target_blocks;
    const int min_rows_per_block = 4 * WAVES;              // do not shred the vocabulary below 4 rows per wave
    if ((long long)nb * min_rows_per_block > n_rows) nb = (n_rows + min_rows_per_block - 1) / min_rows_per_block;
    if (nb < 1) nb = 1;
    nb = (nb + 7) / 8 * 8;                                 // blocks that share a row range stay on one XCD (b % 8)
    int rpb = (n_rows + nb - 1) / nb;
    if (rpb < 1) rpb = 1;
    if (rpb > (int)HROWMASK) rpb = (int)HROWMASK;          // Hamming packed key holds 21 bits of row-in-block
    p.rows_per_block = rpb;
    p.n_blocks = n_rows > 0 ? (n_rows + rpb - 1) / rpb : 0;
    return p;
}
size_t knn_partial_bytes(const KnnPlan& p) { return (size_t)(p.n_blocks > 0 ? p.n_blocks : 1) * 2 * p.qpad * sizeof(uint64_t); }

hipError_t launch_knn2_partial(int dtype, int dim, const void* vocab, const int32_t* row_id, const void* queries,
                               const KnnPlan& p, uint64_t* partial, hipStream_t s) {
    if (p.n_blocks == 0 || p.q == 0) return hipSuccess;
    dim3 grid(p.n_blocks, p.qpad / 64), block(BLOCK);
    if (dtype == 0) {
        const float* v = (const float*)vocab; const float* qq = (const float*)queries;
        if (dim == 64) knn2_l2_kernel<64><<<grid, block, 0, s>>>(v, row_id, p.n_rows, qq, p.q, p.qpad, p.rows_per_block, partial);
        else if (dim == 128) knn2_l2_kernel<128><<<grid, block, 0, s>>>(v, row_id, p.n_rows, qq, p.q, p.qpad, p.rows_per_block, partial);
        else knn2_l2_dyn_kernel<<<grid, block, 0, s>>>(v, row_id, p.n_rows, dim, qq, p.q, p.qpad, p.rows_per_block, partial);
    } else {
        const uint32_t* v = (const uint32_t*)vocab; const uint32_t* qq = (const uint32_t*)queries;
        const int w32 = dim / 4;
        if (w32 == 8) knn2_hamming_kernel<8><<<grid, block, 0, s>>>(v, row_id, p.n_rows, qq, p.q, p.qpad, p.rows_per_block, partial);
        else if (w32 == 16) knn2_hamming_kernel<16><<<grid, block, 0, s>>>(v, row_id, p.n_rows, qq, p.q, p.qpad, p.rows_per_block, partial);
        else if (w32 == 4) knn2_hamming_kernel<4><<<grid, block, 0, s>>>(v, row_id, p.n_rows, qq, p.q, p.qpad, p.rows_per_block, partial);
        else knn2_hamming_dyn_kernel<<<grid, block, 0, s>>>(v, row_id, p.n_rows, w32, qq, p.q, p.qpad, p.rows_per_block, partial);
    }
    return hipGetLastError();
}

hipError_t launch_knn2_merge(int dtype, const KnnPlan& p, const uint64_t* partial, const int32_t* row_id,
                             int32_t* out_row, int32_t* out_word, float* out_dist, hipStream_t s) {
    if (p.q == 0) return hipSuccess;
    knn2_merge_kernel<<<(p.q + WAVES - 1) / WAVES, BLOCK, 0, s>>>(dtype, partial, p.n_blocks * 2, p.qpad, p.q, row_id,
                                                                   out_row, out_word, out_dist);
    return hipGetLastError();
}

hipError_t launch_knn2_merge_selfdist_hamming(const KnnPlan& p, const uint64_t* partial, const int32_t* row_id, int32_t* out_row, int32_t* out_word,
                                              float* out_dist, const void* queries, int dim_bytes, float* selfdist, int ld, int have_index, uint32_t* bits,
                                              int bw, hipStream_t s) {
    if (p.q == 0) return hipSuccess;
    knn2_merge_selfdist_hamming_kernel<<<(p.q + WAVES - 1) / WAVES, BLOCK, 0, s>>>(partial, p.n_blocks * 2, p.qpad, p.q, row_id, out_row, out_word, out_dist,
                                                                                    (const uint32_t*)queries, dim_bytes / 4, selfdist, ld, have_index, bits, bw);
    return hipGetLastError();
}

bool selfdist_can_merge(int dtype, int dim) { return dtype == 0 && (dim == 64 || dim == 128); }

hipError_t launch_selfdist(int dtype, int dim, const void* queries, int q, float* out, int ld, hipStream_t s, int have_index,
                           const int32_t* knn_word, const float* knn_dist, uint32_t* bits, int bw, const ShardMergeJob* merge) {
    if (q == 0) return hipSuccess;
    if (merge && !selfdist_can_merge(dtype, dim)) return hipErrorInvalidValue;
    dim3 grid((q + SD_ROWS - 1) / SD_ROWS, (q + 63) / 64), block(BLOCK);
    if (dtype == 0) {
        const float* qq = (const float*)queries;
        const ShardMergeJob mj = merge ? *merge : ShardMergeJob{};
        if (dim == 64) selfdist_l2_kernel<64><<<grid, block, 0, s>>>(qq, q, out, ld, have_index, knn_word, knn_dist, bits, bw, mj);
        else if (dim == 128) selfdist_l2_kernel<128><<<grid, block, 0, s>>>(qq, q, out, ld, have_index, knn_word, knn_dist, bits, bw, mj);
        else if (merge) return hipErrorInvalidValue;
        else selfdist_l2_dyn_kernel<<<grid, block, 0, s>>>(qq, q, dim, out, ld, have_index, knn_word, knn_dist, bits, bw);
    } else {
        selfdist_hamming_dyn_kernel<<<grid, block, 0, s>>>((const uint32_t*)queries, q, dim / 4, out, ld, have_index, knn_word, knn_dist,
                                                           bits, bw);
    }
    return hipGetLastError();
}

}  // namespace lcd
