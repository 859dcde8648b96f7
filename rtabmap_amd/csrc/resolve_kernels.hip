// resolve_kernels.hip -- the per-descriptor decision loop of VWDictionary::addNewWords (reference VWDictionary.cpp:1089-1219)
// and the candidate merge of VWDictionary::findNN (:1457-1542), on the device, plus the small vocabulary maintenance
// kernels behind lcd_vocab_remove / lcd_vocab_rebuild (VWDictionary::update(), :571-690).
//
// addNewWords is sequential in the reference: descriptor i is also matched against the words created by descriptors
// j < i of the same call (Kp/NewWordsComparedTogether, :1140-1160).  That is a lower-triangular system
//     isNew[i] = f_i(isNew[0..i-1]),
// solved here by Jacobi sweeps over all descriptors in parallel until a sweep changes nothing.  Entry i is final
// once entries < i are, so after t sweeps the first t entries are exact and a sweep without change is the unique
// solution -- i.e. exactly the reference's sequential result (worst case q sweeps, in practice 1-3).
//
// A sweep is cheap because a same-frame new word j can only matter for descriptor i when it is strictly closer than
// i's second indexed neighbour: the self-distance kernel already reduced that test to one bit per (i, j)
// (cand_bits, knn2_kernels.hip).  A sweep is then "AND my bit row with the current new-word mask (LDS)" and a distance
// is fetched only for the surviving bits -- none at all for most descriptors of a mature vocabulary.
//
// One workgroup (1024 threads) handles the frame: the data is tiny and the work is latency-, not throughput-bound;
// the frame's heavy part is knn2_kernels.hip.
#include "lcd_kernels.h"
#include "resolve_body.cuh"
#include "shard_body.cuh"

namespace lcd {
namespace {

__global__ __launch_bounds__(RBLOCK) void resolve_kernel(int q, int flags, float nndr, int have_index,
                                                         const int32_t* __restrict__ knn_word, const float* __restrict__ knn_dist,
                                                         const float* __restrict__ selfdist, int ld,
                                                         const uint32_t* __restrict__ cand_bits, int bw,
                                                         int32_t* __restrict__ out_word, int32_t* __restrict__ out_n_new,
                                                         const int32_t* __restrict__ knn_row, const int32_t* __restrict__ row_wslot,
                                                         int32_t* __restrict__ out_wslot, WsRuns new_ws,
                                                         const uint2* __restrict__ cand_list, const int32_t* __restrict__ cand_cnt) {
    extern __shared__ uint32_t rs_dyn_smem[];
    if (q <= RBLOCK)
        resolve_body_fast<RBLOCK, 1>(rs_dyn_smem, nullptr, q, flags, nndr, have_index, knn_word, knn_dist, selfdist, ld, cand_bits, bw, out_word, out_n_new,
                                     knn_row, row_wslot, out_wslot, new_ws, cand_list, cand_cnt);
    else
        resolve_body<RBLOCK>(rs_dyn_smem, q, flags, nndr, have_index, knn_word, knn_dist, selfdist, ld, cand_bits, bw, out_word, out_n_new, knn_row,
                     row_wslot, out_wslot, new_ws);
}

// findNN (:1457-1542): indexed candidates are NOT cut at the first invalid one; not-indexed candidates are
__global__ void findnn_resolve_kernel(int q, int flags, float nndr, int have_index, const int32_t* __restrict__ knn_word,
                                      const float* __restrict__ knn_dist, int have_extra, const int32_t* __restrict__ extra_word,
                                      const float* __restrict__ extra_dist, int32_t* __restrict__ out_word) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
    Cand c0, c1; int n = 0;
    c0.d = 0.f; c0.id = 0; c1.d = 0.f; c1.id = 0;
    if (have_index) {
        for (int j = 0; j < 2; ++j) {
            const float d = knn_dist[2 * i + j]; const int id = knn_word[2 * i + j];
            if (d >= 0.0f && id != 0) cand_push(c0, c1, n, d, id);
        }
    }
    if (have_extra) {
        for (int j = 0; j < 2; ++j) {
            const float d = extra_dist[2 * i + j]; const int id = extra_word[2 * i + j];
            if (d >= 0.0f && id != 0) cand_push(c0, c1, n, d, id); else break;
        }
    }
    int w = 0;
    if (flags & LCD_Q_INCREMENTAL) {
        if (n >= 2 && !(c0.d > nndr * c1.d)) w = c0.id;
    } else if (n > 0) {
        w = c0.id;
    }
    out_word[i] = w;
}

// ------------------------------------------------------------------------------------------------ sharded vocabulary
// Word-ID-range sharding: every rank searched its own rows; the all-gathered per-rank candidates
// cand[rank][q][2] = {key = distance bits << 32 | local row, word id, postings key on the owning rank} are merged here.
// Global row order = (rank, local row): the lower rank, then the lower row, wins ties -- the single-GPU order when the
// shards are consecutive id ranges.  out_wslot[q*2] is the postings key if THIS rank owns the neighbour, else -1.
__global__ void shard_merge_kernel(const ShardCand* __restrict__ cand, int world, int rank, int q, int32_t* __restrict__ out_word,
                                   float* __restrict__ out_dist, int32_t* __restrict__ out_wslot, int by_word) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
    const ShardMerged m = shard_merge_one(cand, world, rank, q, i, by_word);          // shard_body.cuh
    for (int j = 0; j < 2; ++j) { out_word[2 * i + j] = m.word[j]; out_dist[2 * i + j] = m.dist[j]; out_wslot[2 * i + j] = m.wslot[j]; }
}
// local candidates of one rank in ShardCand form (dist as float already converted for Hamming by the merge kernel)
__global__ void shard_pack_kernel(ShardPackArgs p, int32_t* __restrict__ fail_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // the search's last consumer: its counters ([0] rejected queries, [1] arrivals of the redo, [3] redo done) are left clean for the next search
    // (the fused frame tail does the same for the unsharded handle: one reset launch less per frame)
    if (fail_count && i < 4 && i != 2) fail_count[i] = 0;
    if (i < p.q2) shard_pack_one(p, i);
}

// ------------------------------------------------------------------------------------------------ vocabulary upkeep
// dst row i = src row perm[i]; a row is row_bytes/4 dwords; consecutive lanes copy consecutive dwords (coalesced)
__global__ void gather_rows_kernel(const uint32_t* __restrict__ src, const int32_t* __restrict__ src_id,
                                   const int32_t* __restrict__ perm, int n, int row_dwords,
                                   uint32_t* __restrict__ dst, int32_t* __restrict__ dst_id) {
    const size_t total = (size_t)n * row_dwords;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / row_dwords), c = (int)(e % row_dwords);
        const int sr = perm[r];
        dst[e] = src[(size_t)sr * row_dwords + c];
        if (c == 0) dst_id[r] = src_id[sr];
    }
}
// Memory::cleanUnusedWords' selection (VWDictionary::getUnusedWords: words without a reference) over the device's own reference counts:
// the live rows whose word nobody references are listed (any order; the host sorts)
__global__ void unused_rows_kernel(const int32_t* __restrict__ row_id, const int32_t* __restrict__ row_wslot, const uint32_t* __restrict__ nw,
                                   int n_rows, int32_t* __restrict__ out_rows, int32_t* __restrict__ out_count, int cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows || row_id[r] == 0) return;
    const int32_t ws = row_wslot[r];
    if (ws >= 0 && nw[ws] != 0u) return;
    const int pos = atomicAdd(out_count, 1);
    if (pos < cap) out_rows[pos] = r;
}
// Memory::cleanUnusedWords (Memory.cpp:6899-6920) entirely on the device, enqueued behind the frames in flight: every live row whose word
// nobody references is tombstoned here (row_id = 0, |row|^2 = +inf so that no filter ranks it, its postings key released from the row)
// and logged for the host, which catches up the next time the handle is drained.  dev_cnt: the two alternating row counters of a handle
// whose frames append their words on the device (the larger one is the newest), NULL: n_rows is exact.  reg_cnt: the counter that holds
// the row count as of the newest REGISTERED frame -- the rows behind it belong to a frame in flight whose decision loop has run but whose
// registration (the only place its words get their first reference) has not: VWDictionary::addNewWords gives a new word its reference
// at once (VWDictionary.cpp:1185-1195), so Memory::cleanUnusedWords never sees such a word; those rows are not scanned.
__global__ void clean_unused_kernel(int32_t* __restrict__ row_id, const int32_t* __restrict__ row_wslot, const uint32_t* __restrict__ nw,
                                    uint32_t* __restrict__ wrow, float* __restrict__ aug, int n_rows, const int32_t* __restrict__ dev_cnt,
                                    const int32_t* __restrict__ reg_cnt, int32_t* __restrict__ rmlog, int cap) {
    int n = n_rows;
    if (dev_cnt) n = min(n_rows, max(dev_cnt[0], dev_cnt[1]));
    if (reg_cnt) n = min(n, reg_cnt[0]);
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        if (row_id[r] == 0) continue;
        const int32_t ws = row_wslot[r];
        if (ws >= 0 && nw[ws] != 0u) continue;
        row_id[r] = 0;
        if (aug) aug[2 * (size_t)r] = __int_as_float(0x7f800000);
        // the key stays out of circulation (0xFFFFFFFF reads as "a row's key" to the batched key check) until the host has caught up with
        // the log -- which it does with nothing in flight: a frame that matched this row before it was tombstoned may still register
        // references under the key, and a key recycled in between would be shared by two words
        if (ws >= 0) wrow[ws] = 0xFFFFFFFFu;
        const int pos = atomicAdd(&rmlog[0], 1);
        if (pos < cap) { rmlog[16 + 2 * pos] = r; rmlog[16 + 2 * pos + 1] = ws; }
    }
}
__global__ void tombstone_kernel(int32_t* __restrict__ row_id, const int32_t* __restrict__ rows, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) row_id[rows[i]] = 0;
}

}  // namespace

hipError_t launch_resolve(int q, int flags, float nndr, int have_index, const int32_t* knn_word, const float* knn_dist,
                          const float* selfdist, int ld, const uint32_t* cand_bits, int bw, int32_t* out_word, int32_t* out_n_new,
                          hipStream_t s, const int32_t* knn_row, const int32_t* row_wslot, int32_t* out_wslot, const WsRuns* new_ws,
                          const uint2* cand_list, const int32_t* cand_cnt) {
    if (q <= 0) return hipSuccess;
    if (q > 8 * RBLOCK) return hipErrorInvalidValue;
    const int mw = (q + 63) / 64 * 2;
    resolve_kernel<<<1, RBLOCK, (size_t)(3 * mw + 2) * 4, s>>>(q, flags, nndr, have_index, knn_word, knn_dist, selfdist, ld, cand_bits, bw,
                                                            out_word, out_n_new, knn_row, row_wslot, out_wslot, new_ws ? *new_ws : WsRuns(), cand_list,
                                                            cand_cnt);
    return hipGetLastError();
}

hipError_t launch_findnn_resolve(int q, int flags, float nndr, int have_index, const int32_t* knn_word,
                                 const float* knn_dist, int have_extra, const int32_t* extra_word,
                                 const float* extra_dist, int32_t* out_word, hipStream_t s) {
    if (q <= 0) return hipSuccess;
    findnn_resolve_kernel<<<(q + 255) / 256, 256, 0, s>>>(q, flags, nndr, have_index, knn_word, knn_dist, have_extra,
                                                          extra_word, extra_dist, out_word);
    return hipGetLastError();
}

hipError_t launch_shard_pack(const int32_t* knn_row, const int32_t* knn_word, const float* knn_dist, const int32_t* row_wslot, int q,
                             void* out_cand, hipStream_t s, int32_t* fail_count) {
    if (q <= 0) return hipSuccess;
    ShardPackArgs p;
    p.knn_row = knn_row; p.knn_word = knn_word; p.knn_dist = knn_dist; p.row_wslot = row_wslot; p.q2 = 2 * q; p.out = (ShardCand*)out_cand;
    shard_pack_kernel<<<(2 * q + 255) / 256, 256, 0, s>>>(p, fail_count);
    return hipGetLastError();
}
hipError_t launch_shard_merge(const void* all_cand, int world, int rank, int q, int32_t* out_word, float* out_dist, int32_t* out_wslot,
                              hipStream_t s, bool by_word) {
    if (q <= 0) return hipSuccess;
    shard_merge_kernel<<<(q + 255) / 256, 256, 0, s>>>((const ShardCand*)all_cand, world, rank, q, out_word, out_dist, out_wslot, by_word ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const void* src, const int32_t* src_id, const int32_t* perm, int n, int row_bytes,
                              void* dst, int32_t* dst_id, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int row_dwords = row_bytes / 4;
    const size_t total = (size_t)n * row_dwords;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    gather_rows_kernel<<<blocks, 256, 0, s>>>((const uint32_t*)src, src_id, perm, n, row_dwords, (uint32_t*)dst, dst_id);
    return hipGetLastError();
}

hipError_t launch_unused_rows(const int32_t* row_id, const int32_t* row_wslot, const uint32_t* nw, int n_rows, int32_t* out_rows, int32_t* out_count,
                              int cap, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    unused_rows_kernel<<<(n_rows + 255) / 256, 256, 0, s>>>(row_id, row_wslot, nw, n_rows, out_rows, out_count, cap);
    return hipGetLastError();
}
hipError_t launch_clean_unused(int32_t* row_id, const int32_t* row_wslot, const uint32_t* nw, uint32_t* wrow, float* aug, int n_rows,
                               const int32_t* dev_cnt, const int32_t* reg_cnt, int32_t* rmlog, int cap, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    const int blocks = (n_rows + 255) / 256 < 1024 ? (n_rows + 255) / 256 : 1024;
    clean_unused_kernel<<<blocks, 256, 0, s>>>(row_id, row_wslot, nw, wrow, aug, n_rows, dev_cnt, reg_cnt, rmlog, cap);
    return hipGetLastError();
}
hipError_t launch_tombstone(int32_t* row_id, const int32_t* rows, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    tombstone_kernel<<<(n + 255) / 256, 256, 0, s>>>(row_id, rows, n);
    return hipGetLastError();
}

}  // namespace lcd
