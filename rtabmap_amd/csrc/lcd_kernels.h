// lcd_kernels.h -- host-callable launchers of the hand-written gfx950 kernels (internal to liblcd_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcd {

// A candidate is one 64-bit key: (distance << 32) | row.  For LCD_F32 the distance half is the IEEE bit pattern of
// the (non-negative) squared L2 distance, for LCD_U8 it is the integer Hamming distance; both order like unsigned
// integers, and the row in the low half makes "lower row wins ties" (result_set.h:151-171) part of the comparison.
static const uint64_t KEY_NONE = ~0ull;

struct KnnPlan {
    int q;             // queries
    int qpad;          // q rounded up to 64
    int n_rows;        // vocabulary rows (incl. tombstones)
    int rows_per_block;
    int n_blocks;      // row blocks (grid.x)
};
KnnPlan knn_plan(int q, int n_rows, int dim_bytes);
size_t knn_partial_bytes(const KnnPlan& p);   // [n_blocks][2][qpad] keys

// 2-NN of `queries` [q x dim] against `vocab` [n_rows x dim] (row_id[r] == 0 -> tombstone, skipped).
// dtype: 0 = f32 (dim floats), 1 = u8 (dim bytes, dim % 4 == 0).  Writes per-block partial top-2 keys.
hipError_t launch_knn2_partial(int dtype, int dim, const void* vocab, const int32_t* row_id, const void* queries,
                               const KnnPlan& p, uint64_t* partial, hipStream_t s);
// Merge the partial keys: out_row[q*2] (row or -1), out_word[q*2] (row_id[row] or 0), out_dist[q*2] (float, -1 = none)
hipError_t launch_knn2_merge(int dtype, const KnnPlan& p, const uint64_t* partial, const int32_t* row_id,
                             int32_t* out_row, int32_t* out_word, float* out_dist, hipStream_t s);
// Hamming frames: the merge AND the same-frame distance matrix [q x ld] AND the candidate bit rows [q x bw] in one launch (one wave per query)
hipError_t launch_knn2_merge_selfdist_hamming(const KnnPlan& p, const uint64_t* partial, const int32_t* row_id, int32_t* out_row, int32_t* out_word,
                                              float* out_dist, const void* queries, int dim_bytes, float* selfdist, int ld, int have_index, uint32_t* bits,
                                              int bw, hipStream_t s);

// Optional by-product of the MFMA 2-NN: the candidate bit matrix of the addNewWords resolution (see launch_selfdist) from an
// already computed same-frame distance matrix.  bits == nullptr: not wanted.
struct CandBits {
    const float* selfdist = nullptr;   // [nq x ld], symmetric
    int ld = 0, nq = 0;
    uint32_t* bits = nullptr;          // [nq x bw], bw = ld / 32 (even)
    int bw = 0;
    int have_index = 0;
    // the same information in the form the decision loop wants: per descriptor i the number of set bits below i and, when there are
    // at most CAND_LIST, the (j, distance bits) pairs themselves -- one coalesced read instead of a bit-row scan and distance gathers
    uint2* list = nullptr;             // [nq x CAND_LIST] {j, distance bits}
    int32_t* cnt = nullptr;            // [nq] (> CAND_LIST: the list is incomplete, use the bit row)
};
// (round 6 tried sixteen entries: the decision loop then needs 128 more registers per thread and spills; the descriptors with long lists are the
// copies of a place's popular words, dozens per frame, which no practical list length covers -- they keep their bit row in registers instead)
constexpr int CAND_LIST = 4;
inline size_t cand_bits_bytes(int q, int bw) { return (size_t)q * (bw + 2 * CAND_LIST + 1) * 4; }   // bit rows | lists | counts in one buffer
inline void cand_bits_layout(CandBits& cb, uint32_t* base, int q, int bw) {
    cb.bits = base; cb.bw = bw;
    cb.list = reinterpret_cast<uint2*>(base + (size_t)q * bw);
    cb.cnt = reinterpret_cast<int32_t*>(base + (size_t)q * bw + (size_t)q * 2 * CAND_LIST);
}

// Postings keys reserved for the words a frame may create: the k-th new word of the frame gets the k-th key of the
// concatenated runs (recycled keys come back as a handful of intervals; the rest is one fresh interval).  n == 0: none.
// n < 0 (decision loop only): the keys are not known yet -- the loop leaves the code -(k + 2) for the frame's k-th new word and the
// registration, which runs one launch later on a pipelined handle, translates it with the runs it was given.
// Sharded vocabulary with balanced growth (own_block > 0): the k-th new word of the frame has id own_id0 + k and belongs to rank
// ((id - own_first) / own_block) % own_world -- every rank reserves the same keys (identical numbering), only the owner references them.
struct WsRuns {
    int32_t start[16];
    int32_t len[16];
    int32_t n = 0;
    int32_t own_id0 = 0, own_first = 0, own_block = 0, own_rank = 0, own_world = 1;
};
__host__ __device__ inline int32_t ws_runs_at(const WsRuns& r, int k) {
    if (r.own_block > 0) {
        const int32_t id = r.own_id0 + k;
        if (id < r.own_first || ((id - r.own_first) / r.own_block) % r.own_world != r.own_rank) return -1;
    }
    for (int i = 0; i < r.n; ++i) { if (k < r.len[i]) return r.start[i] + k; k -= r.len[i]; }
    return -1;
}

#if defined(__HIPCC__)
// Device call sites (decision loop, registration, row writers).  A variant that walks ALL runs in a uniform loop (start / len through scalar
// loads, only k per lane: no per-lane read of the argument segment) was measured and lost: launch A 12.7 -> 15.5 us on the driver's command -- the
// loop's sixteen dependent selects sit in front of every key in the decision loop's last phase (profiles/r06_ab_notes.txt, item 9).
__device__ __forceinline__ int32_t ws_runs_at_dev(const WsRuns& r, int k) { return ws_runs_at(r, k); }
#endif

// Arguments of the exact redo of rejected queries (rowpar_body.cuh).  enabled == 0: no redo wanted.
struct RowparArgs {
    int enabled = 0;
    const float* vocab = nullptr; const int32_t* row_id = nullptr; int n_rows = 0;
    const int32_t* n_rows_dev = nullptr;               // the row count on the device (<= n_rows, the host's upper bound) when rows are appended there
    const float* queries = nullptr; const int32_t* fail_list = nullptr;
    unsigned long long* partial = nullptr;             // knn_rowpar_partial_bytes()
    int32_t* out_row = nullptr; int32_t* out_word = nullptr; float* out_dist = nullptr;
    CandBits cb;
};

// ---- squared-L2 2-NN on the matrix cores (knn_mfma_kernels.hip): MFMA filter + exact re-rank + certificate.
// Queries whose result cannot be certified are appended to fail_list / fail_count and redone by launch_knn_rowpar.
struct MfmaPlan {
    int q, qpad, n_rows, tiles_per_block, n_blocks;
    int filter_units = -1;   // compute units the persistent bf16 filter plans for (-1: built-in; 0: one workgroup per strip always)
    int other_wgs = 0;       // long-running workgroups of other kinds in the same launch (each holds a compute unit like a filter workgroup)
    int one_strip = 0;       // 1: one workgroup per strip whatever the number of strips (lcd_set_option "strip_tiles")
    int f16 = 0;             // 1: the operand tables hold IEEE half and the filter multiplies ONE product per fp32 product (LCD_KNN_F16)
    int n_shadow = 0;        // > 0: the launch also computes the shadow scores (PipeKnn::sh_bf, sh_x): one workgroup per 32-row tile of the frame before
};
constexpr uint32_t SHADOW_ROW_BASE = 1u << 30;   // candidate keys of the re-rank name descriptor j of the frame before as row SHADOW_ROW_BASE + j
bool knn_mfma_supported(int dtype, int dim);
void knn_set_compute_units(int cus);           // the device's compute units: what the filter launch plans fill (256 unless told otherwise)
bool knn_bf16_persistent(const MfmaPlan& p);   // the bf16 filter launch of this plan uses the persistent kernels (..._kernel_p)
MfmaPlan knn_mfma_plan(int q, int n_rows);
MfmaPlan knn_bf16_plan_pipelined(int q, int n_rows, int n_tile_wgs, int filter_units);   // launch A of a pipelined frame (knn_mfma_kernels.hip)
size_t knn_mfma_partial_bytes(const MfmaPlan& p);
// |row|^2 for rows [first, first + n) (+inf for tombstones), running maximum in norm_max_bits
hipError_t launch_row_norms(const void* vocab, const int32_t* row_id, int first, int n, int dim, float* norm, uint32_t* norm_max_bits,
                            hipStream_t s);
hipError_t launch_norm_tombstone(float* norm, const int32_t* rows, int n, hipStream_t s);
// rows [first, first + n) beyond the vocabulary: +inf norm entries ({+inf, 1}; entry first + n is NOT written) and a zero bf16 split
hipError_t launch_vocab_tail(float* norm, void* bf, long long first, long long n, hipStream_t s);
// exact redo of the queries in fail_list (fail_count[0] of them; fail_count[1] is scratch) with one lane per vocabulary row
size_t knn_rowpar_partial_bytes(int n_rows, int q);
// A sharded search's candidate record (what the ranks all-gather) and the job that packs a rank's 2 * q local candidates into records:
// it rides at the end of the exact redo's launch (the search's last), which then also leaves the search's counters zeroed.
struct ShardCand { unsigned long long key; int32_t word; int32_t wslot; };
struct ShardPackArgs {
    const int32_t* knn_row = nullptr; const int32_t* knn_word = nullptr; const float* knn_dist = nullptr; const int32_t* row_wslot = nullptr;
    int q2 = 0; ShardCand* out = nullptr;
};
hipError_t launch_knn_rowpar(int dim, const void* vocab, const int32_t* row_id, int n_rows, const void* queries, const int32_t* fail_list,
                             int32_t* fail_count, void* partial, int32_t* out_row, int32_t* out_word, float* out_dist, hipStream_t s,
                             const CandBits* cb = nullptr, const ShardPackArgs* pack = nullptr);
// ... and the job that merges the all-gathered records [world][q][2] into the frame's global 2-NN: it rides at the head of the same-frame distance
// launch (every workgroup merges its 64 queries itself -- the thresholds of their candidate bits; the first row block's workgroups write the result)
struct ShardMergeJob {
    const ShardCand* cand = nullptr; int world = 0, rank = 0, by_word = 0;
    int32_t* out_word = nullptr; float* out_dist = nullptr; int32_t* out_wslot = nullptr;
};
hipError_t launch_knn_mfma(int dim, const void* vocab, const float* row_norm, const uint32_t* norm_max_bits, const int32_t* row_id,
                           const void* queries, const MfmaPlan& p, void* partial, int32_t* out_row, int32_t* out_word, float* out_dist,
                           int32_t* fail_list, int32_t* fail_count, hipStream_t s, hipEvent_t ev_begin = nullptr,
                           hipEvent_t ev_end = nullptr,    // optional events bracketing the filter kernel alone
                           bool reset_count = true,        // false: fail_count[0..1] are already zero
                           const CandBits* cb = nullptr);  // also emit the candidate bits (cb->selfdist must be complete on `s`)
// q x q distances of a block against itself; out[i*ld + j], ld >= q.  When `bits` is given ([q][bw] words, bw >= ceil(q/32))
// also the candidate bit matrix of the addNewWords resolution: bit j of row i = dist(j, i) < (distance of i's second
// indexed neighbour, +inf if it has none) -- see knn2_kernels.hip.
// `merge` (sharded frames, selfdist_can_merge() kernels only): knn_word / knn_dist are not read but WRITTEN, from the all-gathered records.
hipError_t launch_selfdist(int dtype, int dim, const void* queries, int q, float* out, int ld, hipStream_t s, int have_index = 0,
                           const int32_t* knn_word = nullptr, const float* knn_dist = nullptr, uint32_t* bits = nullptr, int bw = 0,
                           const ShardMergeJob* merge = nullptr);
bool selfdist_can_merge(int dtype, int dim);
// A sharded frame whose search left the same-frame distance matrix behind (it rode in the filter's launch, cb.selfdist): the merge of the
// all-gathered records and the candidate bit rows, one wave per query -- all that is left of the same-frame distance launch behind the all-gather.
hipError_t launch_shard_merge_bits(const ShardMergeJob& mj, const CandBits& cb, hipStream_t s);

// The addNewWords decision loop (VWDictionary.cpp:1089-1219) for a whole frame, on the device.
//   knn_word/knn_dist [q*2] indexed candidates (word 0 / dist < 0 = none); have_index = vocabulary had >= 2 live rows
//   selfdist [q x ld] and cand_bits [q x bw] from launch_selfdist (may be NULL when !(flags & NEW_WORDS_COMPARED))
//   out_word[q]: > 0 existing word, < 0: -(k+1) for the k-th new word, 0: no entry (fixed dictionary, no candidate)
//   out_n_new[1]
hipError_t launch_resolve(int q, int flags, float nndr, int have_index, const int32_t* knn_word, const float* knn_dist,
                          const float* selfdist, int ld, const uint32_t* cand_bits, int bw, int32_t* out_word, int32_t* out_n_new,
                          hipStream_t s, const int32_t* knn_row = nullptr, const int32_t* row_wslot = nullptr,
                          int32_t* out_wslot = nullptr,    // out_wslot[q]: postings key of the chosen word (-1: none)
                          const WsRuns* new_ws = nullptr,  // postings keys of the frame's new words (NULL: they get none)
                          const uint2* cand_list = nullptr, const int32_t* cand_cnt = nullptr);   // CandBits::list / cnt when available
// findNN merge (VWDictionary.cpp:1457-1542): indexed candidates + candidates among the not-indexed words + NNDR.
hipError_t launch_findnn_resolve(int q, int flags, float nndr, int have_index, const int32_t* knn_word,
                                 const float* knn_dist, int have_extra, const int32_t* extra_word,
                                 const float* extra_dist, int32_t* out_word, hipStream_t s);

// Sharded vocabulary (one rank per word-id range): local candidates -> 16-byte records {u64 key, i32 word, i32 wslot}, and the
// merge of the all-gathered records [world][q][2] (ties: lower rank, then lower local row); out_wslot is -1 for foreign words.
hipError_t launch_shard_pack(const int32_t* knn_row, const int32_t* knn_word, const float* knn_dist, const int32_t* row_wslot, int q,
                             void* out_cand, hipStream_t s, int32_t* fail_count = nullptr);
// by_word: ties go to the lower WORD ID instead of (rank, local row) -- the single-GPU row order when every rank's rows ascend by id and
// the ranks' id sets interleave (block-cyclic ownership of the words frames create)
hipError_t launch_shard_merge(const void* all_cand, int world, int rank, int q, int32_t* out_word, float* out_dist, int32_t* out_wslot,
                              hipStream_t s, bool by_word = false);

// ---- the same filter on the bf16 matrix pipe (three bf16 products per f32 product, fp32 accumulate): needs the hi/lo bf16
// split of the vocabulary (256 bytes per row) kept by launch_vocab_bf16 next to the rows.
MfmaPlan knn_bf16_plan(int q, int n_rows, int other_wgs = 0 /* long-running workgroups sharing the launch */);
int knn_selfdist_wgs(int q);   // distance-matrix workgroups of a q-descriptor frame
size_t knn_bf16_partial_bytes(const MfmaPlan& p);
hipError_t launch_vocab_bf16(const void* vocab, int first, int n, int dim, void* bf, hipStream_t s, int f16 = 0);
hipError_t launch_knn_bf16(int dim, const void* vocab, const void* vocab_bf, const float* row_norm, const uint32_t* norm_max_bits,
                           const int32_t* row_id, const void* queries, const MfmaPlan& p, void* partial, int32_t* out_row, int32_t* out_word,
                           float* out_dist, int32_t* fail_list, int32_t* fail_count, hipStream_t s, hipEvent_t ev_begin = nullptr,
                           hipEvent_t ev_end = nullptr, bool reset_count = true, const CandBits* cb = nullptr,
                           bool with_selfdist = false,     // true: extra workgroups of the filter launch fill cb->selfdist (queries x queries)
                           hipStream_t s_rerank = nullptr, hipEvent_t ev_bridge = nullptr);   // re-rank on its own stream behind ev_bridge

// Row gather used by lcd_vocab_rebuild: dst[i] = src[perm[i]] (rows of row_bytes bytes, multiple of 4), ids likewise.
hipError_t launch_gather_rows(const void* src, const int32_t* src_id, const int32_t* perm, int n, int row_bytes,
                              void* dst, int32_t* dst_id, hipStream_t s);
// the live rows whose word has no reference (nw[row_wslot[r]] == 0): out_rows[0 .. min(*out_count, cap)), any order
hipError_t launch_unused_rows(const int32_t* row_id, const int32_t* row_wslot, const uint32_t* nw, int n_rows, int32_t* out_rows, int32_t* out_count,
                              int cap, hipStream_t s);
// cleanUnusedWords on the device (resolve_kernels.hip, clean_unused_kernel): rmlog[0] counts, rmlog[16 ..] lists {row, postings key} of the rows tombstoned (cap pairs)
hipError_t launch_clean_unused(int32_t* row_id, const int32_t* row_wslot, const uint32_t* nw, uint32_t* wrow, float* aug, int n_rows,
                               const int32_t* dev_cnt, const int32_t* reg_cnt, int32_t* rmlog, int cap, hipStream_t s);
// row_id[rows[i]] = 0
hipError_t launch_tombstone(int32_t* row_id, const int32_t* rows, int n, hipStream_t s);

}  // namespace lcd
