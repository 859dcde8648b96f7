// tfidf.hip -- kernels and host-side management of the blocked inverted index (see tfidf.h for the layout).
//
// Reference behaviour reproduced (Memory.cpp:2215-2291): for every UNIQUE word id w > 0 of the query,
//   nw = refs(w).size(); logNnw = log10(N / nw) (float); skipped when nw == 0 or logNnw == 0;
//   for every (signature s, count nwi) in refs(w): ni = getNi(s); if ni != 0: L[s] += (nwi * logNnw) / ni   (all fp32).
// Here logNnw is computed with the same fp32 operations, rounded once to Q5.26, and L[s] = (sum of nwi * logNnw as exact 64-bit
// integers) / ni: one float rounding and one float division per signature instead of one of each per posting.  The two agree to
// rounding (~1e-7 relative; the tests bound 1e-4 with an absolute floor of 1e-7).
#include "tfidf.h"
#include "frame_tail_body.cuh"
#include "score_body.cuh"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iterator>

namespace lcd {
namespace {

#ifndef LCD_FW_BLOCK
#define LCD_FW_BLOCK 1024
#endif
constexpr int FW_BLOCK = LCD_FW_BLOCK;   // frame_words_kernel / frame_tail_kernel (timing experiments build it with 256 to compare with the fused launch's tail)
constexpr int BR_BLOCK = 256;    // bulk_register_kernel (one workgroup per signature)
constexpr int SEAL_BLOCK = 256;
constexpr int SEAL_TILE = SEAL_BLOCK * 32;   // wslots per workgroup of the sealing scan (one 32-wslot directory block per thread)

// exclusive scan, in place, of data[0..n) (LDS) by a whole workgroup of NT threads; returns the total.  scratch[NT / 64 + 1] in
// LDS.  data[] must be complete (barrier) before the call; three barriers inside.
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t* data, int n, uint32_t* scratch) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n + NT - 1) / NT;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += data[i];
    uint32_t x = sum;                                   // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
    if (lane == 63) scratch[wv] = x;
    __syncthreads();
    if (wv == 0) {
        const uint32_t t = lane < NW ? scratch[lane] : 0u;
        uint32_t s = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(s, off, 64); if (lane >= off) s += y; }
        if (lane < NW) scratch[lane] = s - t;
        if (lane == NW - 1) scratch[NW] = s;
    }
    __syncthreads();
    uint32_t run = scratch[wv] + x - sum;               // exclusive prefix of this thread's chunk
    const uint32_t total = scratch[NW];
    for (int i = lo; i < hi; ++i) { const uint32_t v = data[i]; data[i] = run; run += v; }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(FW_BLOCK) void frame_words_kernel(FwArgs a, RetireArgs retire) {
    extern __shared__ uint32_t fw_dyn_smem[];
    retire_body(retire, a.slot_begin, a.slot_cnt, a.nw, a.slot_ni, a.slot_sig);
    __syncthreads();
    frame_words_body<FW_BLOCK, true>(fw_dyn_smem, a);
}

// The single-workgroup tail of a frame in ONE launch (frame_tail_body.cuh); workgroups 1.. are the exact redo of rejected queries
__global__ __launch_bounds__(FW_BLOCK) void frame_tail_kernel(ResolveArgs r, FwArgs a, RetireArgs retire) {
    extern __shared__ uint32_t ft_dyn_smem[];
    frame_tail_body<FW_BLOCK>(ft_dyn_smem, r, a, retire, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------- bulk registration
// Memory::loadDataFromDb replay (Memory.cpp:447-480): one workgroup per signature, all signatures of the call in ONE launch.
// Signature s gets slot slot0 + s; its (unique word, count) pairs are appended to its bucket's log at a position reserved with
// one atomic on the bucket's entry counter (the order of the signatures inside a bucket's log is irrelevant).
__global__ __launch_bounds__(BR_BLOCK) void bulk_register_kernel(const int32_t* __restrict__ ids, const long long* __restrict__ offsets,
                                                                 const int32_t* __restrict__ sig_ids, const int32_t* __restrict__ ni,
                                                                 long long slot0, const int32_t* __restrict__ xlate, long long xlate_n,
                                                                 const BucketDev* __restrict__ tab, uint32_t* __restrict__ bkt_ne,
                                                                 uint32_t* __restrict__ nw, int32_t* __restrict__ slot_sig,
                                                                 uint32_t* __restrict__ slot_ni, uint32_t* __restrict__ slot_begin,
                                                                 uint32_t* __restrict__ slot_cnt) {
    extern __shared__ uint32_t br_dyn_smem[];
    const long long s = blockIdx.x;
    const long long o0 = offsets[s] - offsets[0], o1 = offsets[s + 1] - offsets[0];
    const int n = (int)(o1 - o0);
    int H = 64;
    while (H < 2 * n) H <<= 1;
    const long long slot = slot0 + s;
    const int b = (int)(slot / TF_R);
    FwArgs a;
    a.src = ids + o0; a.n = n; a.xlate = xlate; a.xlate_n = xlate_n; a.H = H; a.do_register = 1; a.want_q = 0;
    a.sig_id = sig_ids[s]; a.slot = slot; a.slot_local = (uint32_t)(slot % TF_R); a.ni = ni ? (uint32_t)ni[s] : (uint32_t)n;
    a.N = 0.0f; a.stamp = 0u; a.nw = nw; a.did = nullptr;
    a.coo_w = const_cast<uint32_t*>(tab[b].coo_w); a.coo_pc = const_cast<uint32_t*>(tab[b].coo_pc); a.ne_counter = bkt_ne + b;
    a.slot_sig = slot_sig; a.slot_ni = slot_ni; a.slot_begin = slot_begin; a.slot_cnt = slot_cnt;
    a.q_w = nullptr; a.q_idf = nullptr; a.q_did = nullptr; a.qd_did = nullptr; a.qd_idf = nullptr; a.q_meta = nullptr; a.idf_tab = nullptr;
    a.wrow = nullptr; a.row_wslot = nullptr;
    frame_words_body<BR_BLOCK, false>(br_dyn_smem, a);
}

// every closed bucket (dead ones write zeros) and the open bucket in ONE launch, likelihood (or the integer sums) written directly
template <int SCB>
__global__ __launch_bounds__(SCB) void score_kernel(ScoreArgs A) {
    const int g = (int)blockIdx.x;
    if (g < A.n_closed_pad) {                                            // consecutive buckets on one XCD: they share directory lines
        const int b = (g & 7) * (A.n_closed_pad >> 3) + (g >> 3);
        if (b < A.n_closed) score_sealed_body<SCB>(A, b);
    } else score_open_body<SCB>(A, g - A.n_closed_pad);
}

// what one scoring launch has to read for the frame in q_* (bench.py's algorithmic bytes): cnt[0] dense row bytes, [1] sparse postings,
// [2] directory lookups, [3] lookups that found the word, [4] entries of the open bucket's log, [5] postings of the frame's words
// over all live signatures (sum of nw), [6] unique words, [7] dense words of the frame
__global__ __launch_bounds__(256) void score_work_kernel(ScoreArgs A, const uint32_t* __restrict__ nw, unsigned long long* __restrict__ cnt) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int U = (int)A.q_meta[0], Ud = (int)A.q_meta[1];
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    if (b < A.n_closed) {
        const BucketDev B = A.tab[b];
        if (B.state != 1u) return;
        const uint32_t D = A.bkt_D[b], flags = A.bkt_flags[b];
        for (int j = tid; j < Ud; j += 256) if ((uint32_t)A.qd_did[j] < D) c0 += TF_R;
        for (int k = tid; k < U; k += 256) {
            const uint32_t w = A.q_w[k];
            const int32_t d = A.q_did[k];
            const bool dense_here = d >= 0 && (uint32_t)d < D;
            if (A.q_idf[k] != 0 && w < B.W && (!dense_here || (flags & 1u))) {
                c2 += 1;
                const uint2 blk = B.dirb[w >> 5];
                const uint32_t bit = 1u << (w & 31);
                if (blk.x & bit) {
                    const uint32_t r = blk.y + (uint32_t)__popc(blk.x & (bit - 1u));
                    c1 += B.sp_off[r + 1] - B.sp_off[r];
                    c3 += 1;
                }
            }
        }
    } else {
        for (int sl = tid; sl < A.n_open_slots; sl += 256) {
            const long long slot = (long long)A.n_closed * TF_R + sl;
            if (A.slot_ni[slot] != 0u) c4 += A.slot_cnt[slot];
        }
        for (int k = tid; k < U; k += 256) if (A.q_idf[k] != 0) c5 += nw[A.q_w[k]];
        if (tid == 0) { atomicAdd(&cnt[6], (unsigned long long)U); atomicAdd(&cnt[7], (unsigned long long)Ud); }
    }
    if (c0) atomicAdd(&cnt[0], c0);
    if (c1) atomicAdd(&cnt[1], c1);
    if (c2) atomicAdd(&cnt[2], c2);
    if (c3) atomicAdd(&cnt[3], c3);
    if (c4) atomicAdd(&cnt[4], c4);
    if (c5) atomicAdd(&cnt[5], c5);
}

// fixed point -> float after the cross-GPU sum
__global__ void finalize_kernel(const long long* __restrict__ lfix, long long n, const uint32_t* __restrict__ slot_ni, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fixed_to_like(lfix[i], slot_ni[i]);
}
__global__ void gather_f32_kernel(const float* __restrict__ dense, const long long* __restrict__ slots, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const long long s = slots[i]; out[i] = s >= 0 ? dense[s] : 0.0f; }
}

// ---------------------------------------------------------------------------------------------- sealing
// All sealing kernels take a batch of jobs (blockIdx.y); a single job travels by value in the kernel arguments.
#define SEAL_JOB() const SealJob J = jobs ? jobs[blockIdx.y] : job1

// (1) postings per word in this bucket
__global__ __launch_bounds__(SEAL_BLOCK) void seal_count_kernel(const SealJob* __restrict__ jobs, SealJob job1) {
    SEAL_JOB();
    const uint32_t ne = min(J.ne[0], J.ent_cap);
    for (uint32_t e = blockIdx.x * SEAL_BLOCK + threadIdx.x; e < ne; e += gridDim.x * SEAL_BLOCK) atomicAdd(&J.cntw[J.coo_w[e]], 1u);
}
// (2) words that reach TF_DENSE_T postings in one bucket get a dense id (once; several buckets of a batch may race for a word)
__global__ __launch_bounds__(SEAL_BLOCK) void seal_densify_kernel(const SealJob* __restrict__ jobs, SealJob job1, int32_t* __restrict__ did,
                                                                  uint32_t* __restrict__ n_dense) {
    SEAL_JOB();
    for (uint32_t w = blockIdx.x * SEAL_BLOCK + threadIdx.x; w < J.W; w += gridDim.x * SEAL_BLOCK) {
        if (J.cntw[w] < (uint32_t)TF_DENSE_T || did[w] != -1) continue;
        if (__hip_atomic_load(n_dense, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (uint32_t)TF_DENSE_MAX) continue;
        if (atomicCAS(&did[w], -1, -2) != -1) continue;                    // another bucket of the batch claimed it
        const uint32_t id = atomicAdd(n_dense, 1u);
        did[w] = id < (uint32_t)TF_DENSE_MAX ? (int32_t)id : -1;
    }
}
// (3) dense cells are written, the others stay counted in cntw (= sparse postings per word from here on)
__global__ __launch_bounds__(SEAL_BLOCK) void seal_classify_kernel(const SealJob* __restrict__ jobs, SealJob job1, const int32_t* __restrict__ did,
                                                                   const uint32_t* __restrict__ n_dense, uint32_t* __restrict__ bkt_D,
                                                                   uint32_t* __restrict__ bkt_flags, uint32_t* __restrict__ h_n_dense) {
    SEAL_JOB();
    const uint32_t nd = min(n_dense[0], (uint32_t)TF_DENSE_MAX);
    const uint32_t D = min(nd, J.D_alloc);
    if (blockIdx.x == 0 && threadIdx.x == 0) { bkt_D[J.bucket] = D; if (h_n_dense && blockIdx.y == 0) h_n_dense[0] = nd; }
    const uint32_t ne = min(J.ne[0], J.ent_cap);
    for (uint32_t e = blockIdx.x * SEAL_BLOCK + threadIdx.x; e < ne; e += gridDim.x * SEAL_BLOCK) {
        const uint32_t w = J.coo_w[e], pc = J.coo_pc[e];
        const int32_t d = did[w];
        if (d < 0 || (uint32_t)d >= D) continue;
        const uint32_t cnt = pc & TF_CNT_MASK, sl = pc >> TF_CNT_BITS;
        J.dense[(size_t)d * TF_R + sl] = (uint8_t)min(cnt, 255u);
        if (cnt <= 255u) atomicSub(&J.cntw[w], 1u);
        else atomicOr(&bkt_flags[J.bucket], 1u);                         // the excess stays a sparse posting
    }
}
// (4) per tile of SEAL_TILE wslots: number of present words and of sparse postings
__global__ __launch_bounds__(SEAL_BLOCK) void seal_tile_kernel(const SealJob* __restrict__ jobs, SealJob job1) {
    SEAL_JOB();
    __shared__ uint32_t s_p[SEAL_BLOCK / 64], s_a[SEAL_BLOCK / 64];
    const uint32_t w0 = (blockIdx.x * SEAL_BLOCK + threadIdx.x) * 32u;
    uint32_t p = 0, a = 0;
    for (uint32_t i = 0; i < 32u && w0 + i < J.W; ++i) { const uint32_t c = J.cntw[w0 + i]; p += c ? 1u : 0u; a += c; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { p += __shfl_xor(p, off, 64); a += __shfl_xor(a, off, 64); }
    if ((threadIdx.x & 63) == 0) { s_p[threadIdx.x >> 6] = p; s_a[threadIdx.x >> 6] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tp = 0, ta = 0;
        for (int i = 0; i < SEAL_BLOCK / 64; ++i) { tp += s_p[i]; ta += s_a[i]; }
        J.tile_sums[2 * blockIdx.x] = tp; J.tile_sums[2 * blockIdx.x + 1] = ta;
    }
}
// (5) directory blocks {presence bits, rank of the block's first word} and the offsets of the present words
__global__ __launch_bounds__(SEAL_BLOCK) void seal_scan_kernel(const SealJob* __restrict__ jobs, SealJob job1) {
    SEAL_JOB();
    __shared__ uint32_t s_p[SEAL_BLOCK], s_a[SEAL_BLOCK], scratch[SEAL_BLOCK / 64 + 1];
    uint32_t base_p = 0, base_a = 0;
    for (uint32_t t = 0; t < blockIdx.x; ++t) { base_p += J.tile_sums[2 * t]; base_a += J.tile_sums[2 * t + 1]; }
    const uint32_t blk = blockIdx.x * SEAL_BLOCK + threadIdx.x;
    const uint32_t w0 = blk * 32u;
    uint32_t bits = 0, p = 0, a = 0;
    uint32_t c[32];
#pragma unroll
    for (uint32_t i = 0; i < 32u; ++i) {
        c[i] = (w0 + i < J.W) ? J.cntw[w0 + i] : 0u;
        if (c[i]) { bits |= 1u << i; p += 1u; }
        a += c[i];
    }
    s_p[threadIdx.x] = p; s_a[threadIdx.x] = a;
    __syncthreads();
    const uint32_t tot_p = block_exclusive_scan<SEAL_BLOCK>(s_p, SEAL_BLOCK, scratch);
    const uint32_t tot_a = block_exclusive_scan<SEAL_BLOCK>(s_a, SEAL_BLOCK, scratch);
    uint32_t rp = base_p + s_p[threadIdx.x], ra = base_a + s_a[threadIdx.x];
    if (w0 < J.W) {
        J.dirb[blk] = make_uint2(bits, rp);
        if (J.dir2) {                                                    // the same block in the word-major directory
            uint32_t f[6] = {0u, 0u, 0u, 0u, 0u, 0u};
            bool sat = false;
#pragma unroll
            for (uint32_t i = 0; i < 32u; ++i) { sat = sat || c[i] > 30u; f[i / 6] |= (c[i] > 30u ? 31u : c[i]) << (5u * (i % 6)); }
            uint4* rec = reinterpret_cast<uint4*>(J.dir2 + ((size_t)blk * J.dir2_stride + (uint32_t)J.bucket) * TF_DIR2_DWORDS);
            rec[0] = make_uint4(ra, rp | (sat ? TF_DIR2_SAT : 0u), f[0], f[1]);
            rec[1] = make_uint4(f[2], f[3], f[4], f[5]);
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < 32u; ++i) if (c[i]) { J.sp_off[rp++] = ra; ra += c[i]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) J.sp_off[base_p + tot_p] = base_a + tot_a;   // end of the last segment
}
// (6) sparse postings into their segments (cntw counts down: no second cursor array)
__global__ __launch_bounds__(SEAL_BLOCK) void seal_scatter_kernel(const SealJob* __restrict__ jobs, SealJob job1, const int32_t* __restrict__ did,
                                                                  const uint32_t* __restrict__ bkt_D) {
    SEAL_JOB();
    const uint32_t D = bkt_D[J.bucket];
    const uint32_t ne = min(J.ne[0], J.ent_cap);
    for (uint32_t e = blockIdx.x * SEAL_BLOCK + threadIdx.x; e < ne; e += gridDim.x * SEAL_BLOCK) {
        const uint32_t w = J.coo_w[e];
        uint32_t pc = J.coo_pc[e];
        const int32_t d = did[w];
        if (d >= 0 && (uint32_t)d < D) {
            const uint32_t cnt = pc & TF_CNT_MASK;
            if (cnt <= 255u) continue;
            pc = (pc & ~TF_CNT_MASK) | (cnt - 255u);
        }
        const uint2 blk = J.dirb[w >> 5];
        const uint32_t r = blk.y + (uint32_t)__popc(blk.x & ((1u << (w & 31)) - 1u));
        const uint32_t pos = J.sp_off[r] + atomicSub(&J.cntw[w], 1u) - 1u;
        J.sp_ent[pos] = pc;
    }
}

__global__ void set_bucket_kernel(BucketDev* tab, int b, BucketDev v) { tab[b] = v; }

__global__ void retire_kernel(long long slot, const uint32_t* __restrict__ coo_w, const uint32_t* __restrict__ slot_begin,
                              const uint32_t* __restrict__ slot_cnt, uint32_t* __restrict__ nw, uint32_t* __restrict__ slot_ni,
                              int32_t* __restrict__ slot_sig) {
    const uint32_t begin = slot_begin[slot], cnt = slot_cnt[slot];
    for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) atomicSub(&nw[coo_w[begin + k]], 1u);
    if (threadIdx.x == 0) { slot_ni[slot] = 0u; slot_sig[slot] = 0; }
}

// the words left the dictionary: a wslot may be handed out again only if nothing references it (ok[i] tells the host)
// verdict: 1 = free (nothing references the key and no vocabulary row carries it), 2 = a live row's key (permanent, whatever its reference
// count: a word a frame appended on the device whose signature is gone, or that never had one), 0 = still referenced
__global__ void wslot_release_kernel(const int32_t* __restrict__ ws, int n, const uint32_t* __restrict__ nw, const uint32_t* __restrict__ wrow,
                                     int32_t* __restrict__ did, uint2* __restrict__ idf_tab, uint8_t* __restrict__ ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t w = ws[i];
    const bool is_row = wrow[w] != 0u;
    const bool free_now = !is_row && nw[w] == 0u;
    if (free_now) { did[w] = -1; idf_tab[w] = make_uint2(0u, 0u); }
    ok[i] = free_now ? 1 : (is_row ? 2 : 0);
}
// rows [first_row, first_row + n) carry the keys ws[0 .. n)
__global__ void wrow_set_kernel(const int32_t* __restrict__ ws, int n, long long first_row, uint32_t* __restrict__ wrow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ws[i] >= 0) wrow[ws[i]] = (uint32_t)(first_row + i) + 1u;
}
// the keys of logged removals ({row, key} pairs) leave their quarantine
__global__ void wrow_unlog_kernel(const int32_t* __restrict__ pairs, int n, uint32_t* __restrict__ wrow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t w = pairs[2 * i + 1];
    if (w >= 0 && wrow[w] == 0xFFFFFFFFu) wrow[w] = 0u;
}
// the rows rows[0 .. n) are gone: their keys belong to no row any more
__global__ void wrow_clear_kernel(const int32_t* __restrict__ row_wslot, const int32_t* __restrict__ rows, int n, uint32_t* __restrict__ wrow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t w = row_wslot[rows[i]];
    if (w >= 0) wrow[w] = 0u;
}
// table[pairs[2i]] = pairs[2i + 1]
__global__ void scatter_pairs_kernel(const int32_t* __restrict__ pairs, int n, int32_t* __restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) table[pairs[2 * i]] = pairs[2 * i + 1];
}
__global__ void iota_i32_kernel(int32_t* dst, int n, int32_t first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = first + i;
}

// ---------------------------------------------------------------------------------------------- adjustLikelihood
inline int next_pow2(int v) { int p = 2; while (p < v) p <<= 1; return p; }

}  // namespace

// VWDictionary::update()'s append branch for ONE RANK of a sharded vocabulary (lcd_shard_frame_dev with "shard_append"): the decision
// loop ran replicated on the merged candidates, so every rank holds the same codes; this rank turns the new words it OWNS into rows of
// its shard -- behind the rows it has, in word order -- without the host.  Ownership as WsRuns says (block-cyclic over the ranks, or all
// to the last rank); the count of owned ids below a given id has a closed form, so a word's row needs no scan.  One workgroup.
__device__ __forceinline__ int shard_owned_below(int32_t id, int32_t first, int32_t block, int rank, int world) {   // owned ids in [first, id)
    if (id <= first) return 0;
    const long long x = (long long)id - first, cyc = (long long)block * world;
    const long long rem = x % cyc - (long long)rank * block;
    return (int)((x / cyc) * block + (rem < 0 ? 0 : (rem > block ? block : rem)));
}
__device__ __forceinline__ void shard_append_body(int* s_first /* LDS [q]: the descriptor that created the k-th new word */, const AppendArgs& ap,
                                                  const WsRuns& new_ws, const int32_t* __restrict__ codes, int q, int rank, int world,
                                                  int32_t own_first, int32_t own_block) {
    __shared__ int s_n_new;
    const int tid = threadIdx.x;
    if (tid == 0) s_n_new = 0;
    for (int k = tid; k < q; k += blockDim.x) s_first[k] = 0x7fffffff;
    __syncthreads();
    for (int i = tid; i < q; i += blockDim.x) {
        const int c = codes[i];
        if (c < 0) { atomicMin(&s_first[-c - 1], i); atomicMax(&s_n_new, -c); }   // (a later descriptor that matched the new word carries the same code)
    }
    __syncthreads();
    const int n_new = s_n_new, n_in = ap.cnt_in[0];
    auto below = [&](int k) -> int {                                   // owned words among the frame's first k new words
        if (own_block > 0) return shard_owned_below(ap.first_id + k, own_first, own_block, rank, world) - shard_owned_below(ap.first_id, own_first, own_block, rank, world);
        return rank == world - 1 ? k : 0;
    };
    const int n_own = below(n_new);
    const int n_take = (long long)n_in + n_own <= ap.capacity ? n_own : 0;
    const int c = tid & 15;
    float norm_max = 0.0f;
    for (int k = tid >> 4; k < n_new && n_take > 0; k += (int)blockDim.x >> 4) {
        const int j = below(k);
        if (below(k + 1) == j) continue;                               // another rank's word (uniform over the row's 16 lanes)
        const size_t row = (size_t)n_in + (size_t)j;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(ap.descriptors) + (size_t)s_first[k] * ap.row_dwords;
        if (ap.is_f32_64) append_write_row(ap, row, c, reinterpret_cast<const uint4*>(src)[c], norm_max);
        else { uint32_t* dst = ap.vocab + row * ap.row_dwords; for (int d = c; d < ap.row_dwords; d += 16) dst[d] = src[d]; }
        if (c == 0) append_write_ids(ap, new_ws, row, k);
    }
    if (ap.is_f32_64) append_norm_max(ap, norm_max);
    if (tid == 0) {
        ap.cnt_out[0] = n_in + n_take;
        if (ap.log_slot) ap.log_slot[0] = n_take > 0 || n_own == 0 ? n_new : -1;   // the host derives the owned ids from the frame's total (-1: dropped, no room)
        if (ap.host_mirror) __hip_atomic_store(ap.host_mirror, ((unsigned long long)ap.tag << 32) | (unsigned long long)(uint32_t)(n_in + n_take), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(1024) void shard_append_kernel(AppendArgs ap, WsRuns new_ws, const int32_t* __restrict__ codes, int q, int rank, int world,
                                                           int32_t own_first, int32_t own_block) {
    extern __shared__ int s_first[];
    shard_append_body(s_first, ap, new_ws, codes, q, rank, world, own_first, own_block);
}
// the registration of a sharded frame and, beside it, this rank's share of the frame's append: two single-workgroup latency chains that need
// nothing of each other (the registration reads wrow[] only to COUNT references to words an enqueued clean tombstoned, 0xFFFFFFFF: a key the
// appender is claiming at that moment reads as 0 or as its row, never as that) -- one launch of two workgroups instead of two launches
__global__ __launch_bounds__(FW_BLOCK) void frame_words_append_kernel(FwArgs a, RetireArgs retire, ShardAppendJob j) {
    extern __shared__ uint32_t fwa_dyn_smem[];
    if (blockIdx.x == 1) {
        shard_append_body(reinterpret_cast<int*>(fwa_dyn_smem), j.ap, j.new_ws, j.codes, j.q, j.rank, j.world, j.own_first, j.own_block);
        return;
    }
    retire_body(retire, a.slot_begin, a.slot_cnt, a.nw, a.slot_ni, a.slot_sig);
    __syncthreads();
    frame_words_body<FW_BLOCK, true>(fwa_dyn_smem, a);
}
hipError_t launch_shard_append(const AppendArgs& ap, const WsRuns& new_ws, const int32_t* codes, int q, int rank, int world, int32_t own_first,
                               int32_t own_block, hipStream_t s) {
    if (q <= 0 || q > 8192) return hipErrorInvalidValue;
    shard_append_kernel<<<1, 1024, (size_t)q * 4, s>>>(ap, new_ws, codes, q, rank, world, own_first, own_block);
    return hipGetLastError();
}

hipError_t launch_gather_f32(const float* dense, const int64_t* slots, int n, float* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    gather_f32_kernel<<<(n + 255) / 256, 256, 0, s>>>(dense, (const long long*)slots, n, out);
    return hipGetLastError();
}

// ================================================================================================ host side
#define TF_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return e__; } while (0)

hipError_t BufPool::get(size_t bytes, DevBuf* out, int64_t* total) {
    // smallest free buffer that fits and is not more than 4x too large
    int best = -1;
    for (size_t i = 0; i < free_list.size(); ++i) {
        if (free_list[i].cap >= bytes && free_list[i].cap <= 4 * std::max<size_t>(bytes, 4096) && (best < 0 || free_list[i].cap < free_list[best].cap)) best = (int)i;
    }
    if (best >= 0) { *out = free_list[best]; free_list.erase(free_list.begin() + best); return hipSuccess; }
    size_t cap = 4096;
    while (cap < bytes) cap *= 2;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) return e;
    out->p = p; out->cap = cap;
    if (total) *total += (int64_t)cap;
    return hipSuccess;
}
void BufPool::put(DevBuf* b) {
    if (b->p) free_list.push_back(*b);
    b->p = nullptr; b->cap = 0;
}
void BufPool::destroy(int64_t* total) {
    for (DevBuf& b : free_list) b.release(total);
    free_list.clear();
}

// grow a table, new part filled with `byte`
static hipError_t grow_filled(DevBuf& buf, size_t bytes, int byte, hipStream_t s, int64_t* total) {
    const size_t old = buf.cap;
    if (bytes <= old) return hipSuccess;
    hipError_t e = buf.reserve(bytes, old, s, total);
    if (e != hipSuccess) return e;
    return hipMemsetAsync((char*)buf.p + old, byte, buf.cap - old, s);
}
static hipError_t grow_zeroed(DevBuf& buf, size_t bytes, hipStream_t s, int64_t* total) { return grow_filled(buf, bytes, 0, s, total); }

// frames / signatures of thousands of words need more than the default 64 KB of dynamic LDS: allow everything the CU has left
// after the kernel's static LDS.  A failure here is not fatal: only launches that actually ask for more than 64 KB would fail.
static hipError_t set_max_lds(const void* fn) {
    hipFuncAttributes attr;
    size_t fixed = 1024;
    if (hipFuncGetAttributes(&attr, fn) == hipSuccess) fixed = (attr.sharedSizeBytes + 255) & ~(size_t)255;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - fixed)) != hipSuccess) (void)hipGetLastError();
    return hipSuccess;
}

hipError_t Tfidf::init(hipStream_t s, int64_t* bytes, int64_t sig_capacity, int64_t vocab_capacity) {
    stream = s;
    bytes_device = bytes;
    TF_TRY(ensure_slots(sig_capacity > 0 ? sig_capacity : TF_R));
    TF_TRY(ensure_buckets((int)((sig_capacity > 0 ? sig_capacity : TF_R) / TF_R) + 64));
    TF_TRY(ensure_wslots((int32_t)std::min<int64_t>(std::max<int64_t>(65536, 4 * vocab_capacity), 1 << 27)));   // growing later means a stream sync
    // the word-major directory is sized for the same horizon (its growth is a stream sync and a copy of the whole table): blocks for
    // twice the vocabulary hint + the keys a stream of frames holds in reserve, buckets for the signature hint
    dir2_hint_blocks = (uint32_t)std::min<int64_t>((2 * std::max<int64_t>(vocab_capacity, 8192) + 65536) / 32, 1 << 22);
    dir2_hint_stride = (uint32_t)std::min<int64_t>((sig_capacity > 0 ? sig_capacity : TF_R) / TF_R + 64, 1 << 20);
    TF_TRY(q_w.reserve(TF_MAX_WORDS * 4, 0, stream, bytes_device));
    TF_TRY(q_idf.reserve(TF_MAX_WORDS * 4, 0, stream, bytes_device));
    TF_TRY(q_did.reserve(TF_MAX_WORDS * 4, 0, stream, bytes_device));
    TF_TRY(qd_did.reserve(TF_MAX_WORDS * 4, 0, stream, bytes_device));
    TF_TRY(qd_idf.reserve(TF_MAX_WORDS * 4, 0, stream, bytes_device));
    TF_TRY(q_meta.reserve(64, 0, stream, bytes_device));
    TF_TRY(hipMemsetAsync(q_meta.p, 0, 64, stream));
    TF_TRY(n_dense.reserve(64, 0, stream, bytes_device));
    TF_TRY(hipMemsetAsync(n_dense.p, 0, 64, stream));
    TF_TRY(hipHostMalloc((void**)&h_n_dense, 64, hipHostMallocDefault));
    h_n_dense[0] = 0;
    TF_TRY(set_max_lds(reinterpret_cast<const void*>(&frame_words_kernel)));
    TF_TRY(set_max_lds(reinterpret_cast<const void*>(&frame_tail_kernel)));
    TF_TRY(set_max_lds(reinterpret_cast<const void*>(&frame_words_append_kernel)));
    TF_TRY(set_max_lds(reinterpret_cast<const void*>(&bulk_register_kernel)));
    return hipSuccess;
}

void Tfidf::destroy() {
    harvest_released(true);
    for (PinBlock& b : pin_free) { (void)hipEventDestroy(b.ev); (void)hipHostFree(b.p); }
    pin_free.clear();
    for (Bucket& b : buckets) { b.coo_w.release(bytes_device); b.coo_pc.release(bytes_device); b.sealed.release(bytes_device); }
    buckets.clear();
    pool.destroy(bytes_device);
    DevBuf* all[] = {&slot_sig, &slot_ni, &slot_begin, &slot_cnt, &nw, &did, &wrow, &idf_tab, &d_id2ws, &bkt_tab, &bkt_ne, &bkt_D, &bkt_flags,
                     &n_dense, &dir2, &seal_cntw, &seal_tiles, &q_w, &q_idf, &q_did, &qd_did, &qd_idf, &q_meta, &d_stage, &d_pairs};
    for (DevBuf* d : all) d->release(bytes_device);
    h_stage.release();
    if (h_n_dense) { (void)hipHostFree(h_n_dense); h_n_dense = nullptr; }
}

hipError_t Tfidf::ensure_slots(int64_t n) {
    TF_TRY(grow_zeroed(slot_sig, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(slot_ni, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(slot_begin, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(slot_cnt, (size_t)n * 4, stream, bytes_device));
    return hipSuccess;
}

hipError_t Tfidf::ensure_wslots(int32_t n) {
    TF_TRY(grow_zeroed(nw, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_filled(did, (size_t)n * 4, 0xFF, stream, bytes_device));
    TF_TRY(grow_zeroed(wrow, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(idf_tab, (size_t)n * 8, stream, bytes_device));
    return hipSuccess;
}

hipError_t Tfidf::ensure_buckets(int n) {
    TF_TRY(grow_zeroed(bkt_tab, (size_t)n * sizeof(BucketDev), stream, bytes_device));
    TF_TRY(grow_zeroed(bkt_ne, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(bkt_D, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(bkt_flags, (size_t)n * 4, stream, bytes_device));
    return hipSuccess;
}

void Tfidf::free_wslot(int32_t w) {
    // insert w into the interval set, merging with its neighbours
    int32_t start = w, len = 1;
    auto next = ws_free.lower_bound(w);
    if (next != ws_free.begin()) {
        auto prev = std::prev(next);
        if (prev->first + prev->second > w) return;                   // already free (cannot happen)
        if (prev->first + prev->second == w) { start = prev->first; len += prev->second; ws_free.erase(prev); }
    }
    if (next != ws_free.end() && next->first == w + 1) { len += next->second; ws_free.erase(next); }
    ws_free[start] = len;
    ws_free_count += 1;
}

// the same for a run of consecutive keys [start, start + len): one interval operation for the whole run
void Tfidf::free_wslot_run(int32_t start, int32_t len) {
    if (len <= 0) return;
    auto next = ws_free.lower_bound(start);
    const bool clash_next = next != ws_free.end() && next->first < start + len;
    const bool clash_prev = next != ws_free.begin() && std::prev(next)->first + std::prev(next)->second > start;
    if (len == 1 || clash_next || clash_prev) {                          // (an overlap cannot happen; key by key it is at least ignored safely)
        for (int32_t k = 0; k < len; ++k) free_wslot(start + k);
        return;
    }
    int32_t s0 = start, l0 = len;
    if (next != ws_free.begin()) {
        auto prev = std::prev(next);
        if (prev->first + prev->second == start) { s0 = prev->first; l0 += prev->second; ws_free.erase(prev); }
    }
    if (next != ws_free.end() && next->first == start + len) { l0 += next->second; ws_free.erase(next); }
    ws_free[s0] = l0;
    ws_free_count += len;
}

int32_t Tfidf::take_wslot() {
    if (ws_free.empty()) return -1;
    auto it = std::prev(ws_free.end());
    const int32_t w = it->first + it->second - 1;
    if (--it->second == 0) ws_free.erase(it);
    ws_free_count -= 1;
    return w;
}

hipError_t Tfidf::wslot_of(int32_t word_id, bool create, int32_t* out) {
    *out = -1;
    if (word_id <= 0) return hipSuccess;
    if ((size_t)word_id < id2ws.size() && id2ws[word_id] >= 0) { *out = id2ws[word_id]; return hipSuccess; }
    int32_t w = -1;
    if (resv.n > 0 && resv.first_id > 0 && word_id >= resv.first_id && word_id < resv.first_id + resv.n) {
        // a new word of the last device-quantised frame: its wslot was reserved when the frame was enqueued
        w = ws_runs_at(resv.runs, word_id - resv.first_id);
    } else {
        // a new word of an earlier frame whose reservation is being checked by the device: the verdict decides whether it exists
        if (std::find(held_ids.begin(), held_ids.end(), word_id) != held_ids.end()) { TF_TRY(flush_held()); harvest_released(true); }
        for (size_t i = 0; i < releasing.size(); ++i) {
            const ReleaseBatch& r = releasing[i];
            if (std::find(r.ids.begin(), r.ids.end(), word_id) != r.ids.end()) { harvest_released(true); break; }
        }
        if ((size_t)word_id < id2ws.size() && id2ws[word_id] >= 0) { *out = id2ws[word_id]; return hipSuccess; }
        if (!create) return hipSuccess;
        if (word_id >= (1 << 28)) return hipErrorInvalidValue;        // the id -> wslot table is direct-indexed
        harvest_released(false);
        w = take_wslot();
        if (w < 0) { w = n_wslots++; TF_TRY(ensure_wslots(n_wslots)); }
    }
    if ((size_t)word_id >= id2ws.size()) id2ws.resize((size_t)word_id + 1 + id2ws.size() / 2, -1);
    id2ws[word_id] = w;
    id2ws_dirty.push_back(word_id);
    *out = w;
    return hipSuccess;
}

// bring the device copy of id2ws up to date: the entries changed since the last call travel as (id, wslot) pairs
hipError_t Tfidf::sync_id2ws() {
    TF_TRY(grow_filled(d_id2ws, std::max<size_t>(id2ws.size(), 1) * 4, 0xFF, stream, bytes_device));
    d_id2ws_n = (int64_t)(d_id2ws.cap / 4);
    if (id2ws_dirty.empty()) return hipSuccess;
    const size_t m = id2ws_dirty.size();
    TF_TRY(h_stage.reserve(m * 8));
    int32_t* st = h_stage.as<int32_t>();
    for (size_t i = 0; i < m; ++i) { st[2 * i] = id2ws_dirty[i]; st[2 * i + 1] = id2ws[id2ws_dirty[i]]; }
    TF_TRY(d_pairs.reserve(m * 8, 0, stream, bytes_device));
    TF_TRY(hipMemcpyAsync(d_pairs.p, st, m * 8, hipMemcpyHostToDevice, stream));
    scatter_pairs_kernel<<<(unsigned)((m + 255) / 256), 256, 0, stream>>>(d_pairs.as<int32_t>(), (int)m, d_id2ws.as<int32_t>());
    TF_TRY(hipGetLastError());
    TF_TRY(hipStreamSynchronize(stream));                             // staging buffers are reused
    id2ws_dirty.clear();
    return hipSuccess;
}

void Tfidf::harvest_released(bool wait) {
    for (size_t i = 0; i < releasing.size();) {
        ReleaseBatch& r = releasing[i];
        const hipError_t q = wait ? hipEventSynchronize(r.blk.ev) : hipEventQuery(r.blk.ev);
        if (q != hipSuccess) { ++i; continue; }
        // the verdicts of a batch are mostly "free" for long runs of consecutive keys (what a frame reserved and did not use): a run
        // goes back into the interval set with ONE operation -- key by key a batch of 16 384 keys kept the host busy for ~0.2 ms, a
        // pause of the enqueueing thread every ~40 frames
        int32_t run_start = 0, run_len = 0;
        for (size_t k = 0; k < r.ws.size(); ++k) {
            if (r.ok[k] == 1) {
                if (run_len > 0 && r.ws[k] == run_start + run_len) { run_len += 1; continue; }
                free_wslot_run(run_start, run_len);
                run_start = r.ws[k]; run_len = 1;
                continue;
            }
            // still referenced, or the key of a vocabulary row.  A wslot reserved for a frame's new word: the word exists (the frame
            // created it) and keeps it.
            const int32_t id = k < r.ids.size() ? r.ids[k] : 0;
            if (id > 0) {
                if ((size_t)id >= id2ws.size()) id2ws.resize((size_t)id + 1 + id2ws.size() / 2, -1);
                if (id2ws[id] < 0) { id2ws[id] = r.ws[k]; id2ws_dirty.push_back(id); }
            } else if (r.ok[k] == 0 && r.recheck) {
                // a key without a word: the word was removed from the dictionary while a frame in flight still matched it (its row is
                // tombstoned, the references of that frame's signature remain).  It comes back when those references are gone.
                ghost_ws.push_back(r.ws[k]);
            }
        }
        free_wslot_run(run_start, run_len);
        pin_free.push_back(r.blk);
        releasing.erase(releasing.begin() + i);
    }
}

// hand wslots back: a kernel checks each one (nw == 0) and reports through pinned memory; the host collects the verdicts of
// finished batches later (harvest_released), so nothing is synchronised here and a wslot that is still referenced is never reused
hipError_t Tfidf::release_wslots(const std::vector<int32_t>& ws, const std::vector<int32_t>* ids, bool recheck) {
    if (ws.empty()) return hipSuccess;
    const size_t m = ws.size();
    ReleaseBatch r;
    r.ws = ws;
    r.recheck = recheck;
    if (ids) r.ids = *ids;
    const size_t need = m * 5 + 16;                                          // [m wslots][m verdicts]
    for (size_t i = 0; i < pin_free.size(); ++i)
        if (pin_free[i].cap >= need) { r.blk = pin_free[i]; pin_free.erase(pin_free.begin() + i); break; }
    if (!r.blk.p) {
        r.blk.cap = 8192;
        while (r.blk.cap < need) r.blk.cap *= 2;
        TF_TRY(hipHostMalloc(&r.blk.p, r.blk.cap, hipHostMallocDefault));
        TF_TRY(hipEventCreateWithFlags(&r.blk.ev, hipEventDisableTiming));
    }
    int32_t* p_ws = (int32_t*)r.blk.p;
    uint8_t* p_ok = (uint8_t*)(p_ws + m);
    std::memcpy(p_ws, ws.data(), m * 4);
    std::memset(p_ok, 0, m);
    r.ok = p_ok;
    wslot_release_kernel<<<(unsigned)((m + 255) / 256), 256, 0, stream>>>(p_ws, (int)m, nw.as<uint32_t>(), wrow.as<uint32_t>(), did.as<int32_t>(),
                                                                           idf_tab.as<uint2>(), p_ok);
    TF_TRY(hipGetLastError());
    TF_TRY(hipEventRecord(r.blk.ev, stream));
    releasing.push_back(r);
    return hipSuccess;
}

hipError_t Tfidf::flush_held() {
    if (held_ws.empty()) return hipSuccess;
    std::vector<int32_t> ws, ids;
    ws.swap(held_ws); ids.swap(held_ids);
    if (!ghost_ws.empty() && (++flushes & 7u) == 0u) {                       // every 8th batch also asks about the keys that were still referenced
        ws.insert(ws.end(), ghost_ws.begin(), ghost_ws.end());
        ids.resize(ws.size(), 0);
        ghost_ws.clear();
    }
    return release_wslots(ws, &ids, true);
}

hipError_t Tfidf::rows_take_keys(const int32_t* d_ws, int n, int64_t first_row) {
    if (n <= 0) return hipSuccess;
    wrow_set_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_ws, n, (long long)first_row, wrow.as<uint32_t>());
    return hipGetLastError();
}
hipError_t Tfidf::rows_drop_keys(const int32_t* d_row_wslot, const int32_t* d_rows, int n) {
    if (n <= 0) return hipSuccess;
    wrow_clear_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_row_wslot, d_rows, n, wrow.as<uint32_t>());
    return hipGetLastError();
}
// a word the device numbered and keyed (LCD_NEW_WORD_IDS_AUTO): its row says which postings key it holds
void Tfidf::adopt_key(int32_t word_id, int32_t ws) {
    if (word_id <= 0 || ws < 0) return;
    if ((size_t)word_id >= id2ws.size()) id2ws.resize((size_t)word_id + 1 + id2ws.size() / 2, -1);
    if (id2ws[word_id] < 0) { id2ws[word_id] = ws; id2ws_dirty.push_back(word_id); }
}
void Tfidf::forget_word(int32_t word_id, int32_t ws) {
    if (word_id <= 0 || ws < 0 || (size_t)word_id >= id2ws.size() || id2ws[word_id] != ws) return;
    id2ws[word_id] = -1;
    id2ws_dirty.push_back(word_id);
    held_ws.push_back(ws); held_ids.push_back(0);
}
hipError_t Tfidf::rows_unlog_keys(const int32_t* d_pairs, int n) {
    if (n <= 0) return hipSuccess;
    wrow_unlog_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_pairs, n, wrow.as<uint32_t>());
    return hipGetLastError();
}
hipError_t Tfidf::rows_clear() {
    return (wrow.p && n_wslots > 0) ? hipMemsetAsync(wrow.p, 0, (size_t)n_wslots * 4, stream) : hipSuccess;
}

hipError_t Tfidf::release_words(const int32_t* word_ids, int n) {
    TF_TRY(flush_retire());               // retirements ride with the next frame otherwise: the check below would still see their references
    std::vector<int32_t> ws;
    for (int i = 0; i < n; ++i) {
        int32_t w = -1;
        TF_TRY(wslot_of(word_ids[i], false, &w));
        if (w < 0) continue;
        id2ws[word_ids[i]] = -1;
        id2ws_dirty.push_back(word_ids[i]);
        ws.push_back(w);
    }
    return release_wslots(ws);
}

// Postings keys for the words the coming frame may create (at most n).  The previous frame's reservation is handed to the device
// for checking: keys it did not use (nw == 0) are recycled, used ones become the permanent keys of those words.
hipError_t Tfidf::reserve_new_words(int32_t first_id, int n, WsRuns* runs, bool may_flush) {
    runs->n = 0;
    // first_id == -1 (LCD_NEW_WORD_IDS_AUTO): the device numbers the words; the host learns id and key of each from its row when it catches up (adopt_key)
    if ((first_id <= 0 && first_id != -1) || n <= 0) return hipSuccess;
    if (first_id > 0 && (int64_t)first_id + n >= (1 << 28)) return hipErrorInvalidValue;
    if (resv.n > 0) {
        int32_t k = 0;
        for (int i = 0; i < resv.runs.n; ++i) {
            for (int32_t j = 0; j < resv.runs.len[i]; ++j, ++k) {
                const int32_t id = resv.first_id > 0 ? resv.first_id + k : 0;   // (0: numbered on the device -- the check only decides whether the key is in use)
                if (id > 0 && (size_t)id < id2ws.size() && id2ws[id] >= 0) continue;    // already the word's permanent key
                held_ws.push_back(resv.runs.start[i] + j);
                held_ids.push_back((id > 0 && (first_id <= 0 || id < first_id)) ? id : 0);   // ids the caller is re-using now name other words
            }
        }
        resv.n = 0;
        // one check launch per ~32 frames, not per frame (the frame tail that may have used these keys is already enqueued)
        if (may_flush && held_ws.size() >= 16384) TF_TRY(flush_held());
    }
    harvest_released(false);
    int left = n;
    while (left > 0 && runs->n < 15 && !ws_free.empty()) {                    // recycled intervals first ...
        auto it = std::prev(ws_free.end());
        const int32_t take = std::min(it->second, left);
        const int32_t start = it->first + it->second - take;
        runs->start[runs->n] = start; runs->len[runs->n] = take; runs->n += 1;
        if ((it->second -= take) == 0) ws_free.erase(it);
        ws_free_count -= take;
        left -= take;
    }
    if (left > 0) {                                                           // ... the rest fresh
        runs->start[runs->n] = n_wslots; runs->len[runs->n] = left; runs->n += 1;
        n_wslots += left;
        TF_TRY(ensure_wslots(n_wslots));
    }
    resv.first_id = first_id; resv.n = n; resv.runs = *runs;
    return hipSuccess;
}

hipError_t Tfidf::set_bucket(int b) {
    const Bucket& k = buckets[b];
    BucketDev d;
    d.coo_w = k.coo_w.as<uint32_t>(); d.coo_pc = k.coo_pc.as<uint32_t>();
    d.dense = k.sealed.as<uint8_t>();
    d.dirb = k.sealed.p ? (const uint2*)((const char*)k.sealed.p + k.off_dirb) : nullptr;
    d.sp_off = k.sealed.p ? (const uint32_t*)((const char*)k.sealed.p + k.off_spoff) : nullptr;
    d.sp_ent = k.sealed.p ? (const uint32_t*)((const char*)k.sealed.p + k.off_spent) : nullptr;
    d.W = k.W; d.D_alloc = k.D_alloc; d.state = (uint32_t)k.state; d.pad = 0;
    set_bucket_kernel<<<1, 1, 0, stream>>>(bkt_tab.as<BucketDev>(), b, d);
    return hipGetLastError();
}

hipError_t Tfidf::new_bucket() {
    const int b = (int)buckets.size();
    buckets.emplace_back();
    TF_TRY(ensure_buckets(b + 1));
    return hipSuccess;
}

// log capacity of bucket b for `entries` postings (grows; the first allocation comes from the pool)
static hipError_t ensure_log(Tfidf& t, int b, int64_t entries) {
    Bucket& k = t.buckets[b];
    const size_t need = (size_t)std::max<int64_t>(entries, 1) * 4;
    if (need <= k.coo_w.cap) return hipSuccess;
    if (!k.coo_w.p) {
        TF_TRY(t.pool.get(std::max(need, (size_t)TF_R * 512 * 4), &k.coo_w, t.bytes_device));
        TF_TRY(t.pool.get(std::max(need, (size_t)TF_R * 512 * 4), &k.coo_pc, t.bytes_device));
    } else {
        TF_TRY(k.coo_w.reserve(need, (size_t)k.ub_entries * 4, t.stream, t.bytes_device));
        TF_TRY(k.coo_pc.reserve(need, (size_t)k.ub_entries * 4, t.stream, t.bytes_device));
    }
    return t.set_bucket(b);
}

// Seal full buckets on the device.  bulk: the caller is a synchronous bulk load -- the dense ids discovered in the first batch are
// read back once so that every bucket of the load gets exactly the rows it needs; otherwise nothing is read back and the number
// of rows is an upper estimate from the (possibly stale) pinned mirror of the dense-id counter.
// the word-major directory grows in both directions: more blocks of 32 wslots (rows appended), more buckets (wider rows)
hipError_t Tfidf::ensure_dir2(uint32_t blocks, uint32_t buckets_needed) {
    if (blocks <= dir2_blocks && buckets_needed <= dir2_stride) return hipSuccess;
    uint32_t nb = std::max<uint32_t>(dir2_blocks, 256), ns = std::max<uint32_t>(dir2_stride, 64);
    if (!dir2.p) { while (nb < dir2_hint_blocks) nb *= 2; while (ns < dir2_hint_stride) ns *= 2; }
    while (nb < blocks) nb *= 2;
    while (ns < buckets_needed) ns *= 2;
    DevBuf nd;
    const size_t rec = (size_t)TF_DIR2_DWORDS * 4;
    TF_TRY(nd.reserve((size_t)nb * ns * rec, 0, stream, bytes_device));
    TF_TRY(hipMemsetAsync(nd.p, 0, (size_t)nb * ns * rec, stream));
    if (dir2.p && dir2_blocks && dir2_stride)
        TF_TRY(hipMemcpy2DAsync(nd.p, (size_t)ns * rec, dir2.p, (size_t)dir2_stride * rec, (size_t)dir2_stride * rec, dir2_blocks, hipMemcpyDeviceToDevice, stream));
    if (dir2.p) { TF_TRY(hipStreamSynchronize(stream)); dir2.release(bytes_device); }
    dir2 = nd; dir2_blocks = nb; dir2_stride = ns;
    return hipSuccess;
}

hipError_t Tfidf::seal_batch(const std::vector<int>& ids, bool bulk) {
    if (ids.empty()) return hipSuccess;
    const uint32_t W = (uint32_t)n_wslots;
    TF_TRY(ensure_dir2((W + 31) / 32, (uint32_t)buckets.size() + 1));
    const int tiles = std::max(1, (int)((W + SEAL_TILE - 1) / SEAL_TILE));
    const size_t BATCH = 64;
    TF_TRY(seal_cntw.reserve(std::min(BATCH, ids.size()) * std::max<size_t>(W, 1) * 4, 0, stream, bytes_device));
    TF_TRY(seal_tiles.reserve(std::min(BATCH, ids.size()) * (size_t)tiles * 8, 0, stream, bytes_device));
    DevBuf d_jobs;                                                     // batch > 1: job table on the device
    bool have_dense_count = false;
    for (size_t i0 = 0; i0 < ids.size(); i0 += BATCH) {
        const size_t nb = std::min(BATCH, ids.size() - i0);
        std::vector<SealJob> jobs(nb);
        uint32_t max_e = 1;
        TF_TRY(hipMemsetAsync(seal_cntw.p, 0, nb * std::max<size_t>(W, 1) * 4, stream));
        for (size_t j = 0; j < nb; ++j) {
            Bucket& k = buckets[ids[i0 + j]];
            SealJob& J = jobs[j];
            J.bucket = ids[i0 + j];
            J.coo_w = k.coo_w.as<uint32_t>(); J.coo_pc = k.coo_pc.as<uint32_t>(); J.ne = bkt_ne.as<uint32_t>() + J.bucket;
            J.cntw = seal_cntw.as<uint32_t>() + j * (size_t)W; J.tile_sums = seal_tiles.as<uint32_t>() + j * (size_t)tiles * 2;
            J.W = W; J.ent_cap = (uint32_t)std::max<int64_t>(k.ub_entries, 1);
            J.dense = nullptr; J.D_alloc = 0; J.dirb = nullptr; J.sp_off = nullptr; J.sp_ent = nullptr;
            J.dir2 = dir2.as<uint32_t>(); J.dir2_stride = dir2_stride;
            max_e = std::max(max_e, J.ent_cap);
        }
        const SealJob* dj = nullptr;
        auto upload_jobs = [&]() -> hipError_t {
            if (nb == 1) return hipSuccess;
            TF_TRY(d_jobs.reserve(nb * sizeof(SealJob), 0, stream, bytes_device));
            TF_TRY(hipMemcpyAsync(d_jobs.p, jobs.data(), nb * sizeof(SealJob), hipMemcpyHostToDevice, stream));
            TF_TRY(hipStreamSynchronize(stream));                      // pageable source; only bulk loads come here
            dj = (const SealJob*)d_jobs.p;
            return hipSuccess;
        };
        TF_TRY(upload_jobs());
        const dim3 ge((unsigned)std::min<uint32_t>((max_e + SEAL_BLOCK - 1) / SEAL_BLOCK, 512), (unsigned)nb);
        const dim3 gw((unsigned)std::min<uint32_t>((W + SEAL_BLOCK - 1) / SEAL_BLOCK + 1, 512), (unsigned)nb);
        seal_count_kernel<<<ge, SEAL_BLOCK, 0, stream>>>(dj, jobs[0]);
        seal_densify_kernel<<<gw, SEAL_BLOCK, 0, stream>>>(dj, jobs[0], did.as<int32_t>(), n_dense.as<uint32_t>());
        TF_TRY(hipGetLastError());
        uint32_t D_alloc;
        if (bulk) {
            if (!have_dense_count) {
                uint32_t nd = 0;
                TF_TRY(hipMemcpyAsync(&nd, n_dense.p, 4, hipMemcpyDeviceToHost, stream));
                TF_TRY(hipStreamSynchronize(stream));
                h_n_dense[0] = std::min<uint32_t>(nd, TF_DENSE_MAX);
                have_dense_count = true;
            }
            D_alloc = std::min<uint32_t>(TF_DENSE_MAX, h_n_dense[0] + 32);
        } else {
            const uint32_t hv = *(volatile uint32_t*)h_n_dense;
            D_alloc = std::min<uint32_t>(TF_DENSE_MAX, hv + (hv ? 128u : 1024u));
        }
        for (size_t j = 0; j < nb; ++j) {
            Bucket& k = buckets[ids[i0 + j]];
            SealJob& J = jobs[j];
            const size_t a256 = 255;
            const size_t sz_dense = ((size_t)D_alloc * TF_R + a256) & ~a256;
            const size_t sz_dirb = ((((size_t)W + 31) / 32) * 8 + a256) & ~a256;
            const size_t sz_off = ((std::min<size_t>(J.ent_cap, W) + 1) * 4 + a256) & ~a256;
            const size_t sz_ent = ((size_t)J.ent_cap * 4 + a256) & ~a256;
            TF_TRY(pool.get(sz_dense + sz_dirb + sz_off + sz_ent, &k.sealed, bytes_device));
            k.off_dirb = sz_dense; k.off_spoff = sz_dense + sz_dirb; k.off_spent = sz_dense + sz_dirb + sz_off;
            k.W = W; k.D_alloc = D_alloc;
            J.dense = k.sealed.as<uint8_t>(); J.D_alloc = D_alloc;
            J.dirb = (uint2*)((char*)k.sealed.p + k.off_dirb);
            J.sp_off = (uint32_t*)((char*)k.sealed.p + k.off_spoff);
            J.sp_ent = (uint32_t*)((char*)k.sealed.p + k.off_spent);
            if (sz_dense) TF_TRY(hipMemsetAsync(J.dense, 0, sz_dense, stream));
        }
        TF_TRY(upload_jobs());
        const dim3 gt((unsigned)tiles, (unsigned)nb);
        seal_classify_kernel<<<ge, SEAL_BLOCK, 0, stream>>>(dj, jobs[0], did.as<int32_t>(), n_dense.as<uint32_t>(), bkt_D.as<uint32_t>(),
                                                           bkt_flags.as<uint32_t>(), h_n_dense);
        seal_tile_kernel<<<gt, SEAL_BLOCK, 0, stream>>>(dj, jobs[0]);
        seal_scan_kernel<<<gt, SEAL_BLOCK, 0, stream>>>(dj, jobs[0]);
        seal_scatter_kernel<<<ge, SEAL_BLOCK, 0, stream>>>(dj, jobs[0], did.as<int32_t>(), bkt_D.as<uint32_t>());
        TF_TRY(hipGetLastError());
        for (size_t j = 0; j < nb; ++j) {
            Bucket& k = buckets[ids[i0 + j]];
            k.state = 1;
            pool.put(&k.coo_pc);                                       // stream order: later users are enqueued after the scatter
            TF_TRY(set_bucket(ids[i0 + j]));
            seals += 1;
        }
    }
    if (d_jobs.p) { TF_TRY(hipStreamSynchronize(stream)); d_jobs.release(bytes_device); }
    return hipSuccess;
}

static RetireArgs take_pending(Tfidf& t) {
    RetireArgs r;
    r.n = 0;
    for (int i = 0; i < 4; ++i) { r.slot[i] = 0; r.coo_w[i] = nullptr; }
    while (r.n < 4 && !t.pending_retire.empty()) {
        const int64_t slot = t.pending_retire.back();
        t.pending_retire.pop_back();
        r.slot[r.n] = slot;
        r.coo_w[r.n] = t.buckets[(size_t)(slot / TF_R)].coo_w.as<uint32_t>();
        r.n += 1;
    }
    return r;
}

// apply every pending retirement now (stand-alone launch): needed before a bucket's memory is released
hipError_t Tfidf::flush_retire() {
    while (!pending_retire.empty()) {
        const int64_t slot = pending_retire.back();
        pending_retire.pop_back();
        retire_kernel<<<1, 256, 0, stream>>>((long long)slot, buckets[(size_t)(slot / TF_R)].coo_w.as<uint32_t>(), slot_begin.as<uint32_t>(),
                                             slot_cnt.as<uint32_t>(), nw.as<uint32_t>(), slot_ni.as<uint32_t>(), slot_sig.as<int32_t>());
        TF_TRY(hipGetLastError());
    }
    return hipSuccess;
}

static hipError_t run_frame_words(Tfidf& t, const int32_t* d_src, int n, bool ids_given, bool reg, int32_t sig_id, int64_t slot, int32_t ni,
                                  float N, const ResolveArgs* resolve, TailLaunch* defer, const WsRuns* new_ws = nullptr,
                                  const ShardAppendJob* shard_app = nullptr) {
    if (shard_app && (resolve || defer || shard_app->q <= 0 || shard_app->q > 8192)) return hipErrorInvalidValue;
    const int H = next_pow2(std::max(2 * n, 128));
    size_t shmem = ((size_t)H * 2 + H / 64 + 8) * 4;
    if (ids_given) TF_TRY(t.sync_id2ws());
    FwArgs a;
    a.src = d_src; a.n = n; a.xlate = ids_given ? t.d_id2ws.as<int32_t>() : nullptr; a.xlate_n = ids_given ? t.d_id2ws_n : 0;
    a.H = H; a.do_register = reg ? 1 : 0; a.want_q = 1;
    a.sig_id = sig_id; a.slot = (long long)slot; a.slot_local = (uint32_t)(slot % TF_R); a.ni = (uint32_t)ni; a.N = N;
    a.coo_w = nullptr; a.coo_pc = nullptr; a.ne_counter = nullptr;
    if (reg) {
        const int bi = (int)(slot / TF_R);
        a.coo_w = t.buckets[bi].coo_w.as<uint32_t>();
        a.coo_pc = t.buckets[bi].coo_pc.as<uint32_t>();
        a.ne_counter = t.bkt_ne.as<uint32_t>() + bi;
    }
    if (t.pending_retire.size() > 4) TF_TRY(t.flush_retire());
    const RetireArgs ret = take_pending(t);
    t.stamp += 1;
    if (t.stamp == 0) t.stamp = 1;
    a.stamp = t.stamp;
    a.nw = t.nw.as<uint32_t>(); a.did = t.did.as<int32_t>();
    a.slot_sig = t.slot_sig.as<int32_t>(); a.slot_ni = t.slot_ni.as<uint32_t>(); a.slot_begin = t.slot_begin.as<uint32_t>();
    a.slot_cnt = t.slot_cnt.as<uint32_t>();
    a.q_w = t.q_w.as<uint32_t>(); a.q_idf = t.q_idf.as<int32_t>(); a.q_did = t.q_did.as<int32_t>(); a.qd_did = t.qd_did.as<int32_t>();
    a.qd_idf = t.qd_idf.as<int32_t>(); a.q_meta = t.q_meta.as<uint32_t>(); a.idf_tab = t.idf_tab.as<uint2>();
    a.new_ws = new_ws ? *new_ws : WsRuns();
    a.wrow = t.wrow.as<uint32_t>();
    a.row_wslot = nullptr;
    if (defer && !resolve) {                                            // registration alone, launched inside a later filter launch
        defer->a = a; defer->ret = ret; defer->shmem = shmem;
        t.q_n_ub = n;
        return hipSuccess;
    }
    if (resolve) {
        a.src = resolve->out_wslot;
        const int mw = (resolve->q + 63) / 64 * 2;
        shmem = std::max(shmem, (size_t)(3 * mw + 2) * 4) + (size_t)n * 8;    // + the word slots handed over in LDS + the appender's list
        const int block = defer ? pipe_block_size() : FW_BLOCK;
        const int n_redo = (resolve->rp.enabled && resolve->fail_count) ? std::min((resolve->rp.n_rows + block - 1) / block, defer ? REDO_WGS_MAX : (1 << 30)) : 0;
        if (defer) {                                                    // launched later, inside the next frame's filter launch
            defer->r = *resolve; defer->a = a; defer->ret = ret; defer->n_redo = n_redo; defer->shmem = shmem;
            t.q_n_ub = n;
            return hipSuccess;
        }
        frame_tail_kernel<<<1 + n_redo, FW_BLOCK, shmem, t.stream>>>(*resolve, a, ret);
    } else if (shard_app) {
        frame_words_append_kernel<<<2, FW_BLOCK, std::max(shmem, (size_t)shard_app->q * 4), t.stream>>>(a, ret, *shard_app);
    } else {
        frame_words_kernel<<<1, FW_BLOCK, shmem, t.stream>>>(a, ret);
    }
    t.q_n_ub = n;
    return hipGetLastError();
}

hipError_t Tfidf::register_dev(int32_t sig_id, const int32_t* d_wslots, int n, int32_t ni, float N, const ResolveArgs* resolve, bool ids_given,
                               TailLaunch* defer, const WsRuns* new_ws, const ShardAppendJob* shard_app) {
    if (n > TF_MAX_WORDS) return hipErrorInvalidValue;
    const int64_t slot = n_slots;
    TF_TRY(ensure_slots(slot + 1));
    const int bi = (int)(slot / TF_R);
    if (bi >= (int)buckets.size()) {
        if (bi > 0 && buckets[bi - 1].state == 0) TF_TRY(seal_batch(std::vector<int>(1, bi - 1), false));
        TF_TRY(new_bucket());
    }
    Bucket& b = buckets[bi];
    TF_TRY(ensure_log(*this, bi, b.ub_entries + n));
    TF_TRY(run_frame_words(*this, d_wslots, n, ids_given, true, sig_id, slot, ni, N, resolve, defer, new_ws, shard_app));
    b.ub_entries += n;
    b.n_slots += 1;
    b.live += 1;
    postings_ub += n;
    n_slots += 1;
    live_sigs += 1;
    sig_slot[sig_id] = slot;
    return hipSuccess;
}

hipError_t Tfidf::query_dev(const int32_t* d_wslots, int n, float N, const ResolveArgs* resolve, bool ids_given, TailLaunch* defer,
                            const WsRuns* new_ws, const ShardAppendJob* shard_app) {
    if (n > TF_MAX_WORDS) return hipErrorInvalidValue;
    return run_frame_words(*this, d_wslots, n, ids_given, false, 0, 0, 0, N, resolve, defer, new_ws, shard_app);
}

// the decision loop of a frame as a workgroup of `block` threads inside a later filter launch: its redo helpers and its dynamic LDS
void resolve_launch_info(const ResolveArgs& r, int block, int* n_redo, size_t* shmem) {
    const int mw = (r.q + 63) / 64 * 2;
    *shmem = (size_t)(3 * mw + 4) * 4 + (size_t)r.q * 4;               // + the appender's list of word-creating descriptors
    *n_redo = (r.rp.enabled && r.fail_count) ? std::min((r.rp.n_rows + block - 1) / block, REDO_WGS_MAX) : 0;
}

hipError_t Tfidf::register_bulk(int n_sigs, const int32_t* sig_ids, const int64_t* offsets, const int32_t* ni, const int32_t* d_ids,
                                int64_t total_ids, int max_n) {
    if (n_sigs <= 0) return hipSuccess;
    (void)total_ids;
    TF_TRY(sync_id2ws());
    const int64_t slot0 = n_slots;
    TF_TRY(ensure_slots(slot0 + n_sigs));
    // buckets touched by the call: the open one is topped up, the others are new; every one that ends up full is sealed
    const int b_first = (int)(slot0 / TF_R), b_last = (int)((slot0 + n_sigs - 1) / TF_R);
    if (b_first > 0 && b_first >= (int)buckets.size() && buckets[b_first - 1].state == 0)
        TF_TRY(seal_batch(std::vector<int>(1, b_first - 1), false));
    while ((int)buckets.size() <= b_last) TF_TRY(new_bucket());
    std::vector<int64_t> add(b_last - b_first + 1, 0);
    for (int s = 0; s < n_sigs; ++s) add[(size_t)((slot0 + s) / TF_R - b_first)] += offsets[s + 1] - offsets[s];
    for (int b = b_first; b <= b_last; ++b) TF_TRY(ensure_log(*this, b, buckets[b].ub_entries + add[b - b_first]));
    // per-signature tables on the device (synchronous call: plain staging)
    DevBuf d_off, d_sig, d_ni;
    TF_TRY(d_off.reserve((size_t)(n_sigs + 1) * 8, 0, stream, bytes_device));
    TF_TRY(d_sig.reserve((size_t)n_sigs * 4, 0, stream, bytes_device));
    TF_TRY(hipMemcpyAsync(d_off.p, offsets, (size_t)(n_sigs + 1) * 8, hipMemcpyHostToDevice, stream));
    TF_TRY(hipMemcpyAsync(d_sig.p, sig_ids, (size_t)n_sigs * 4, hipMemcpyHostToDevice, stream));
    if (ni) {
        TF_TRY(d_ni.reserve((size_t)n_sigs * 4, 0, stream, bytes_device));
        TF_TRY(hipMemcpyAsync(d_ni.p, ni, (size_t)n_sigs * 4, hipMemcpyHostToDevice, stream));
    }
    const int H = next_pow2(std::max(2 * max_n, 64));
    const size_t shmem = ((size_t)H * 2 + H / 64 + 8) * 4;
    bulk_register_kernel<<<(unsigned)n_sigs, BR_BLOCK, shmem, stream>>>(d_ids, (const long long*)d_off.p, d_sig.as<int32_t>(),
                                                                         ni ? d_ni.as<int32_t>() : nullptr, (long long)slot0,
                                                                         d_id2ws.as<int32_t>(), d_id2ws_n, bkt_tab.as<BucketDev>(),
                                                                         bkt_ne.as<uint32_t>(), nw.as<uint32_t>(), slot_sig.as<int32_t>(),
                                                                         slot_ni.as<uint32_t>(), slot_begin.as<uint32_t>(), slot_cnt.as<uint32_t>());
    TF_TRY(hipGetLastError());
    std::vector<int> full;
    for (int s = 0; s < n_sigs; ++s) {
        const int64_t slot = slot0 + s;
        Bucket& b = buckets[(size_t)(slot / TF_R)];
        b.n_slots += 1; b.live += 1;
        sig_slot[sig_ids[s]] = slot;
    }
    for (int b = b_first; b <= b_last; ++b) {
        buckets[b].ub_entries += add[b - b_first];
        if (buckets[b].n_slots == TF_R && b < b_last) full.push_back(b);   // the last bucket stays open until the next one is started
    }
    postings_ub += offsets[n_sigs] - offsets[0];
    n_slots += n_sigs;
    live_sigs += n_sigs;
    TF_TRY(seal_batch(full, true));
    TF_TRY(hipStreamSynchronize(stream));
    d_off.release(bytes_device); d_sig.release(bytes_device); d_ni.release(bytes_device);
    return hipSuccess;
}

hipError_t Tfidf::score_args(float* d_likelihood, long long* lfix, int block, ScoreArgs* out, int* n_wgs) {
    *n_wgs = 0;
    if (n_slots == 0) return hipSuccess;
    TF_TRY(flush_retire());
    const bool has_open = !buckets.empty() && buckets.back().state == 0;
    ScoreArgs& A = *out;
    A.tab = bkt_tab.as<BucketDev>(); A.bkt_D = bkt_D.as<uint32_t>(); A.bkt_flags = bkt_flags.as<uint32_t>();
    A.n_closed = (int)buckets.size() - (has_open ? 1 : 0);
    A.n_open_slots = has_open ? buckets.back().n_slots : 0;
    A.wcap = std::max(q_n_ub, 1);
    A.q_w = q_w.as<uint32_t>(); A.q_idf = q_idf.as<int32_t>(); A.q_did = q_did.as<int32_t>(); A.qd_did = qd_did.as<int32_t>();
    A.qd_idf = qd_idf.as<int32_t>(); A.q_meta = q_meta.as<uint32_t>();
    A.slot_ni = slot_ni.as<uint32_t>(); A.slot_begin = slot_begin.as<uint32_t>(); A.slot_cnt = slot_cnt.as<uint32_t>();
    A.idf_tab = idf_tab.as<uint2>(); A.stamp = stamp;
    A.out_like = d_likelihood; A.out_fix = lfix;
    A.dir2 = dir2.as<uint32_t>(); A.dir2_stride = dir2_stride;
    A.n_closed_pad = (A.n_closed + 7) / 8 * 8;
    *n_wgs = A.n_closed_pad + (A.n_open_slots + block / 64 - 1) / (block / 64);
    return hipSuccess;
}

hipError_t Tfidf::launch_score(float* d_likelihood, long long* lfix) {
    const int scb = score_block == 256 || score_block == 1024 ? score_block : 512;
    ScoreArgs A;
    int grid = 0;
    TF_TRY(score_args(d_likelihood, lfix, scb, &A, &grid));
    if (grid == 0) return hipSuccess;
    if (prof_b) TF_TRY(hipEventRecord(prof_b, stream));
    if (scb == 256) score_kernel<256><<<grid, 256, 0, stream>>>(A);
    else if (scb == 512) score_kernel<512><<<grid, 512, 0, stream>>>(A);
    else score_kernel<1024><<<grid, 1024, 0, stream>>>(A);
    TF_TRY(hipGetLastError());
    if (prof_e) TF_TRY(hipEventRecord(prof_e, stream));
    prof_b = prof_e = nullptr;
    return hipSuccess;
}

// diagnostic (synchronises): the work of one scoring launch for the frame currently in q_*
hipError_t Tfidf::score_work(int64_t out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (n_slots == 0) return hipSuccess;
    TF_TRY(flush_retire());
    const bool has_open = !buckets.empty() && buckets.back().state == 0;
    ScoreArgs A;
    A.tab = bkt_tab.as<BucketDev>(); A.bkt_D = bkt_D.as<uint32_t>(); A.bkt_flags = bkt_flags.as<uint32_t>();
    A.n_closed = (int)buckets.size() - (has_open ? 1 : 0);
    A.n_open_slots = has_open ? buckets.back().n_slots : 0;
    A.wcap = std::max(q_n_ub, 1);
    A.q_w = q_w.as<uint32_t>(); A.q_idf = q_idf.as<int32_t>(); A.q_did = q_did.as<int32_t>(); A.qd_did = qd_did.as<int32_t>();
    A.qd_idf = qd_idf.as<int32_t>(); A.q_meta = q_meta.as<uint32_t>();
    A.slot_ni = slot_ni.as<uint32_t>(); A.slot_begin = slot_begin.as<uint32_t>(); A.slot_cnt = slot_cnt.as<uint32_t>();
    A.idf_tab = idf_tab.as<uint2>(); A.stamp = stamp; A.out_like = nullptr; A.out_fix = nullptr;
    DevBuf cnt;
    TF_TRY(cnt.reserve(64, 0, stream, bytes_device));
    TF_TRY(hipMemsetAsync(cnt.p, 0, 64, stream));
    score_work_kernel<<<A.n_closed + 1, 256, 0, stream>>>(A, nw.as<uint32_t>(), (unsigned long long*)cnt.p);
    TF_TRY(hipGetLastError());
    TF_TRY(hipMemcpyAsync(out, cnt.p, 64, hipMemcpyDeviceToHost, stream));
    TF_TRY(hipStreamSynchronize(stream));
    cnt.release(bytes_device);
    return hipSuccess;
}

hipError_t Tfidf::score(float* d_likelihood) { return launch_score(d_likelihood, nullptr); }
hipError_t Tfidf::score_fix(long long* lfix) { return launch_score(nullptr, lfix); }

hipError_t Tfidf::finalize(const long long* lfix_src, long long n, float* d_likelihood) {
    if (n <= 0) return hipSuccess;
    finalize_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(lfix_src, n, slot_ni.as<uint32_t>(), d_likelihood);
    return hipGetLastError();
}

hipError_t Tfidf::retire(int32_t sig_id) {
    auto it = sig_slot.find(sig_id);
    if (it == sig_slot.end()) return hipErrorInvalidValue;
    const int64_t slot = it->second;
    const int bi = (int)(slot / TF_R);
    Bucket& b = buckets[bi];
    // the device side (nw -= 1 for the signature's words, ni = 0) rides along with the next frame-words launch
    pending_retire.push_back(slot);
    sig_slot.erase(it);
    live_sigs -= 1;
    b.live -= 1;
    if (b.live == 0 && b.state == 1) {
        // every signature of the bucket is gone: its memory goes back to the pool (pending retirements read its log: apply them
        // first; everything already enqueued on the stream still sees the old contents, later users are ordered behind it)
        TF_TRY(flush_retire());
        postings_ub -= b.ub_entries;
        pool.put(&b.coo_w); pool.put(&b.coo_pc); pool.put(&b.sealed);
        b.W = 0; b.D_alloc = 0; b.ub_entries = 0; b.state = 2;
        TF_TRY(set_bucket(bi));
    }
    return hipSuccess;
}

}  // namespace lcd

// The interval set of recycled postings keys, as the engine keeps it (host code, no device needed: tests).  keys[0 .. n) with ok[i] != 0 are
// freed -- by_runs: consecutive keys as ONE interval operation (Tfidf::free_wslot_run, what harvest_released does), else key by key --
// then `take` keys are taken back; out receives the intervals as (start, length) pairs in ascending order.  Returns the number of pairs
// (or -1 if out is too small); *count = free keys as the set counts them.
extern "C" int lcd_debug_key_intervals(const int32_t* keys, const unsigned char* ok, int n, int by_runs, int take, int32_t* out, int cap, long long* count) {
    lcd::Tfidf t;
    int32_t run_start = 0, run_len = 0;
    for (int i = 0; i < n; ++i) {
        if (!ok[i]) { if (by_runs) { t.free_wslot_run(run_start, run_len); run_len = 0; } continue; }
        if (!by_runs) { t.free_wslot(keys[i]); continue; }
        if (run_len > 0 && keys[i] == run_start + run_len) { run_len += 1; continue; }
        t.free_wslot_run(run_start, run_len);
        run_start = keys[i]; run_len = 1;
    }
    if (by_runs) t.free_wslot_run(run_start, run_len);
    for (int i = 0; i < take; ++i) (void)t.take_wslot();
    if ((int)t.ws_free.size() > cap) return -1;
    int k = 0;
    for (const auto& kv : t.ws_free) { out[2 * k] = kv.first; out[2 * k + 1] = kv.second; k += 1; }
    if (count) *count = (long long)t.ws_free_count;
    return k;
}

#ifdef LCD_SCORE_TIMING
extern "C" int lcd_debug_score_timing(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_score_timing), (size_t)n_words * 8);
}
#endif
#ifdef LCD_TAIL_TIMING
extern "C" int lcd_debug_tail_timing(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_tail_timing), 64) != hipSuccess) return -2;
    return (int)hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(lcd::g_resolve_timing), 64);
}
#endif
