// tfidf.h -- device-resident inverted index + TF-IDF likelihood (internal to liblcd_hip.so).
//
// Replaces VisualWord::_references (reference VisualWord.h:62, std::map<sigId,count> per word), Memory::getNi
// (Memory.cpp:4955) and the scoring loop of Memory::computeLikelihood (Memory.cpp:2215-2291).
//
// Data layout in HBM ("blocked inverted index", second version):
//   * a signature gets a SLOT (dense, arrival order); slots are grouped in BUCKETS of TF_R = 256 consecutive slots;
//   * a word gets a WSLOT (dense, recycled once the word is removed and the device has confirmed that nothing references
//     it any more); nw[wslot] = number of live signatures referencing the word;
//   * every bucket keeps the arrival-order log of its postings (coo_w[e] = wslot, coo_pc[e] = slot_local << 22 | count;
//     the entries of one signature are contiguous: slot_begin / slot_cnt), which is also the forward index used to retire a
//     signature.  The bucket that is still filling is scored from this log, one wavefront per signature;
//   * when a bucket is full it is SEALED on the device (no host round trip) into two parts:
//       DENSE ROWS  -- words that occur in many signatures (>= TF_DENSE_T of the 256 of some bucket) get a global dense id;
//                      a sealed bucket holds one 256-byte row per dense id known when it was sealed: row[d][slot] = count
//                      (saturating at 255, the excess goes to the sparse part).  1 byte per signature instead of 4 bytes per
//                      posting, no directory lookup, no decode, no atomics: a wavefront reads a row with one coalesced load
//                      and every lane owns four signatures.  With a heavy-tailed (Zipf) vocabulary these rows carry > 90 %
//                      of the postings a frame touches;
//       SPARSE PART -- every other posting, 4 bytes each, grouped by word, found through a compact directory whose size
//                      is bounded by the postings, not by buckets x words: one {presence bits, rank} pair per 32 wslots
//                      and one offset per word PRESENT in the bucket.
// Arithmetic: idf(w) = log10f(N / nw) as the reference computes it, then rounded ONCE to Q5.26 fixed point; a signature's
// score is the exact 64-bit integer sum of count x idf over the frame's words, converted to float and divided by ni once.
// Integer accumulation is order-free: the result does not depend on workgroup scheduling, on how a count is split between the
// dense and the sparse part, or on how the words are sharded over GPUs (an int64 all-reduce of the partial sums gives the
// single-GPU bits).  It differs from the reference's float accumulation (sum of (count * idf) / ni in ascending word order) by
// rounding only: ~1e-7 relative (bound 1e-4 in tests/test_gpu_likelihood.py, abs floor 1e-7).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "bayes.h"
#include "devbuf.h"
#include "lcd_kernels.h"

namespace lcd {

constexpr int TF_R = 256;                  // slots per bucket
constexpr int TF_CNT_BITS = 22;            // posting = slot_local << 22 | count
constexpr uint32_t TF_CNT_MASK = (1u << TF_CNT_BITS) - 1;
constexpr int TF_MAX_WORDS = 8192;         // words of one signature / one query frame handled by the 1-workgroup kernels
constexpr int TF_IDF_SHIFT = 26;           // idf in Q5.26 (|idf| < 32: N up to 1e32)
constexpr int TF_DENSE_T = 32;             // a word present in >= 32 of a bucket's 256 signatures becomes dense
constexpr int TF_DENSE_MAX = 4096;         // dense ids (1 MB of rows per bucket at most)
// WORD-MAJOR DIRECTORY of the sparse part (dir2): one 32-byte record per (block of 32 wslots, bucket), laid out
// dir2[block][bucket] -- the records of consecutive buckets share a 128-byte line, and consecutive buckets are scored by
// workgroups of the same XCD, so a probe costs a quarter of a line instead of two whole ones:
//   d[0] = offset of the block's first sparse posting in the bucket's sp_ent, d[1] = number of present words before the block
//   (bit 31: some count of the block does not fit its field -> look the word up through dirb / sp_off instead),
//   d[2..7] = the 32 words' posting counts, 5 bits each, six per dword.
// start and length of a word's postings follow from ONE record: no second, dependent lookup of an offset.
constexpr int TF_DIR2_DWORDS = 8;
constexpr uint32_t TF_DIR2_SAT = 1u << 31;
__host__ __device__ inline uint32_t dir2_sum6(uint32_t x) {           // sum of the six 5-bit fields of a dword
    const uint32_t t = (x & 0x01F07C1Fu) + ((x >> 5) & 0x01F07C1Fu);
    return (t & 0x3FFu) + ((t >> 10) & 0x3FFu) + (t >> 20);
}

// one bucket as the kernels see it
struct BucketDev {
    const uint32_t* coo_w;     // [cap] arrival-order log: wslot (kept after sealing: forward index for retirement)
    const uint32_t* coo_pc;    // [cap] arrival-order log: slot_local << 22 | count (released after sealing)
    const uint8_t* dense;      // sealed: [D_alloc][256] counts
    const uint2* dirb;         // sealed: [(W + 31) / 32] {presence bits, number of present words before the block}
    const uint32_t* sp_off;    // sealed: [present + 1] first sparse posting of each present word
    const uint32_t* sp_ent;    // sealed: sparse postings grouped by word
    uint32_t W;                // wslots covered by dirb
    uint32_t D_alloc;          // dense rows allocated
    uint32_t state;            // 0 = open, 1 = sealed, 2 = dead (every signature retired, memory released)
    uint32_t pad;
};

// one sealing job (kernel argument, by value: no staging buffer to keep alive)
struct SealJob {
    int bucket;
    const uint32_t* coo_w; const uint32_t* coo_pc; const uint32_t* ne;   // log and its length (device counter)
    uint8_t* dense; uint32_t D_alloc;
    uint2* dirb; uint32_t* sp_off; uint32_t* sp_ent;
    uint32_t* dir2; uint32_t dir2_stride;   // the word-major directory (all buckets) and its buckets-per-block stride
    uint32_t* cntw;            // [W] scratch, zeroed
    uint32_t* tile_sums;       // [2 * tiles] scratch
    uint32_t W;
    uint32_t ent_cap;          // capacity of sp_ent (host upper bound of the log length)
};

struct Bucket {
    int state = 0;            // 0 open, 1 sealed, 2 dead
    int n_slots = 0;          // slots handed out in this bucket
    int live = 0;             // live signatures among them
    int64_t ub_entries = 0;   // host upper bound of log entries (device appends without telling the host)
    DevBuf coo_w, coo_pc, sealed;
    size_t off_dirb = 0, off_spoff = 0, off_spent = 0;   // byte offsets inside `sealed` (dense rows at 0)
    uint32_t W = 0, D_alloc = 0;
};

// VWDictionary::update()'s append branch (VWDictionary.cpp:571-609) on the device, run by the decision loop's workgroup right after the
// loop: the descriptors that created words become vocabulary rows (row, word id = first_id + k, postings key, |row|^2, bf16 split) in
// descriptor order behind the rows that exist.  The number of rows lives on the device: cnt_in is read, cnt_out = cnt_in + new words
// written (two alternating counters: the filter of the next frame, which runs in the same launch, keeps reading cnt_in).
struct AppendArgs {
    int enabled = 0;
    const float* descriptors = nullptr;        // [q x dim floats] of the frame (dim == 64) or [q x row_bytes] bytes
    int row_dwords = 0;                        // dwords per stored row
    int is_f32_64 = 0;                         // rows are 64 floats: also the augmentation entries and the bf16 split
    int f16 = 0;                               // ... the split table holds IEEE half (LCD_KNN_F16) instead of bf16
    uint32_t* vocab = nullptr; int32_t* row_id = nullptr; int32_t* row_wslot = nullptr;
    uint32_t* wrow = nullptr;                  // Tfidf::wrow: the appended rows claim their postings keys
    float* row_norm = nullptr; uint32_t* norm_max_bits = nullptr; uint32_t* vocab_bf = nullptr;
    const int32_t* cnt_in = nullptr; int32_t* cnt_out = nullptr;
    int32_t* log_slot = nullptr;               // receives the number of rows appended (host reconciliation)
    int32_t first_id = 0;                      // > 0: the k-th new word gets the id first_id + k.  <= 0 (LCD_NEW_WORD_IDS_AUTO): the id FOLLOWS THE ROW, id = row - first_id
                                               // (-first_id = next word id - rows when the chain of appending frames started: every new word is one row and one id, so the
                                               // device numbers exactly as ++_lastWordId does, VWDictionary.cpp:1188, without the host knowing how many words a frame made)
    int32_t* first_out = nullptr;              // receives the id of the frame's first new word (may be NULL)
    long long capacity = 0;                    // rows the buffers hold: appends beyond are dropped (cannot happen: the host reserves q per frame)
    int lds_bytes = 0;                         // dynamic LDS of the workgroup that appends (launch A of a pipelined frame): what the decision loop's
                                               // own tables leave of it stages the new rows (0: no staging)
    // defer_rows (pipelined frames): the decision loop's workgroup only PUBLISHES which descriptors became words (list_out[k] = index of the
    // k-th new word's descriptor) and the new row count; the rows themselves -- copy, |row|^2, operand split, ids, keys -- are written by a
    // few workgroups of launch B of the same pair (append_rows_body), off the single-workgroup chain that bounds launch A.  The re-rank of
    // the next frame, which runs in that same launch B, reads its pending rows straight from the descriptors through the same list.
    int defer_rows = 0;
    uint32_t* list_out = nullptr;
    unsigned long long* host_mirror = nullptr; // pinned: (tag << 32 | rows) after this append, read by the host WITHOUT synchronising to
    uint32_t tag = 0;                          // bound the row count it plans the next launches for
    uint32_t* mask_out = nullptr;              // receives the final new-word mask (mw words) and its word prefix sums (mw + 1), mw = ceil(q / 64) * 2: what the
                                               // re-rank of the next frame needs to tell which shadow rows are words (zeros when nothing is appended)
    int mirror_later = 0;                      // 1 (PipeOpts::mirror_from_b): the decision loop leaves the mirror alone -- a workgroup of launch B of the same
                                               // pair stores it (a write to pinned HOST memory at the end of launch A's longest chain, waited for at s_endpgm)
};
// one rank's share of update()'s append on a sharded vocabulary (shard_append_body, tfidf.hip): codes = the replicated decision loop's output;
// own_block > 0: block-cyclic owners from own_first on, else the last rank owns every new word
struct ShardAppendJob {
    AppendArgs ap; WsRuns new_ws; const int32_t* codes = nullptr; int q = 0, rank = 0, world = 1; int32_t own_first = 0, own_block = 0;
};

// the row-writing half of a deferred append (launch B): the appender's arguments + the postings keys of the frame's new words
struct AppendRowsArgs { AppendArgs ap; WsRuns new_ws; int n_wgs = 0; };
constexpr int APPEND_ROW_WGS = 8;          // workgroups of launch B that write the rows a frame appended

// arguments of the addNewWords decision loop (resolve_body.cuh) when it is fused into the frame-words launch
struct ResolveArgs {
    int q, flags; float nndr; int have_index;
    const int32_t* knn_word; const float* knn_dist; const float* selfdist; int ld; const uint32_t* cand_bits; int bw;
    int32_t* out_word; int32_t* out_n_new; const int32_t* knn_row; const int32_t* row_wslot; int32_t* out_wslot;
    const uint2* cand_list; const int32_t* cand_cnt;   // CandBits::list / cnt (NULL: only the bit rows exist)
    int straight = 0;          // 1: the decision loop's first round trip as ONE straight line of unconditional requests (resolve_body_fast; PipeOpts::decision_straight)
    int slots_are_rows = 0;    // 1 (with row_wslot == NULL): out_wslot gets the vocabulary ROW of a matched word (>= 0; FwArgs::row_wslot translates it one
                           // launch later) and -(key + 2) for a word the frame creates (its key comes from new_ws here and is no row)
    WsRuns new_ws;         // postings keys of the frame's new words (n == 0: new words get no postings)
    int32_t* fail_count;   // reset for the next frame's certificate (saves a memset launch); may be NULL
    RowparArgs rp;         // rp.enabled: the exact redo of rejected queries runs as extra workgroups of the tail launch
    AppendArgs ap;         // ap.enabled: the frame's new words become vocabulary rows behind the decision loop
};

// kernel argument blocks of the registration / scoring launches (frame_tail_body.cuh, score_body.cuh)
struct FwArgs {
    const int32_t* src; int n;                    // word slots of the frame (or word ids when xlate != NULL); < 0 / <= 0 = no word
    const int32_t* xlate; long long xlate_n;      // word id -> wslot table (device copy of Tfidf::id2ws)
    int H; int do_register; int want_q;
    int32_t sig_id; long long slot; uint32_t slot_local; uint32_t ni; float N; uint32_t stamp;
    uint32_t* nw; const int32_t* did;
    uint32_t* coo_w; uint32_t* coo_pc; uint32_t* ne_counter;
    int32_t* slot_sig; uint32_t* slot_ni; uint32_t* slot_begin; uint32_t* slot_cnt;
    uint32_t* q_w; int32_t* q_idf; int32_t* q_did; int32_t* qd_did; int32_t* qd_idf; uint32_t* q_meta; uint2* idf_tab;
    WsRuns new_ws;                                // src entries <= -2 are codes -(k + 2) of the frame's k-th new word (WsRuns, n < 0)
    const int32_t* row_wslot;                     // NULL: src holds postings keys.  Otherwise its entries >= 0 are vocabulary ROWS (row_wslot[row] is the key) and its
                                                  // entries <= -2 are keys themselves, -(key + 2): the words the frame created (ResolveArgs::slots_are_rows)
                                                  // (the decision loop of a pipelined frame: the gather moved from its chain to the registration's, lcd_set_option "slots_from_rows")
    const uint32_t* wrow;                         // Tfidf::wrow (NULL: not wanted): a registered word whose key reads 0xFFFFFFFF is a word an enqueued
                                                  // cleanUnusedWords tombstoned while this frame was in flight -- counted in q_meta[8] (lcd_stats.clean_divergent_refs)
};
// signatures whose retirement was requested since the last frame (Memory::disableWordsRef -> removeAllWordRef): their
// words lose one reference each and the slot is marked dead (ni = 0).  Up to 4 ride along with the next frame-words launch.
struct RetireArgs { long long slot[4]; const uint32_t* coo_w[4]; int n; };
struct ScoreArgs {
    const BucketDev* tab; const uint32_t* bkt_D; const uint32_t* bkt_flags;
    const uint32_t* dir2; uint32_t dir2_stride;
    int n_closed;                           // buckets [0, n_closed) are sealed or dead; bucket n_closed (if any) is the open one
    int n_closed_pad;                       // workgroups launched for them: a multiple of 8, workgroup g scores bucket (g % 8) * pad / 8 + g / 8
    int n_open_slots; int wcap;
    const uint32_t* q_w; const int32_t* q_idf; const int32_t* q_did; const int32_t* qd_did; const int32_t* qd_idf; const uint32_t* q_meta;
    const uint32_t* slot_ni; const uint32_t* slot_begin; const uint32_t* slot_cnt;
    const uint2* idf_tab; uint32_t stamp;
    float* out_like; long long* out_fix;    // exactly one is non-NULL
};
// a frame tail ready to be launched (stand-alone, or inside the filter launches of the following frames).  On a pipelined handle
// the tail is split over two workgroups of two consecutive launches: the decision loop (r, n_redo, shmem_resolve) and, one launch
// later, retirement + registration (a, ret, shmem): each is a chain of dependent round trips, and together they outlasted the filter.
struct TailLaunch {
    ResolveArgs r; FwArgs a; RetireArgs ret;
    int n_redo = 0;            // extra workgroups for the exact redo of rejected queries
    size_t shmem = 0;          // dynamic LDS of the tail workgroup (whole tail, or the registration alone)
    size_t shmem_resolve = 0;  // dynamic LDS of the decision loop alone
};
// the 2-NN stage of a pipelined frame: everything the fused launches need (pointers into one of the two scratch sets)
struct PipeKnn {
    MfmaPlan plan;
    const void* vocab; const void* vocab_bf; const float* row_norm; const uint32_t* norm_max_bits; const int32_t* row_id; const void* queries;
    void* partial; int32_t* out_row; int32_t* out_word; float* out_dist; int32_t* fail_list; int32_t* fail_count;
    CandBits cb;               // cb.selfdist != NULL: the filter launch also fills the same-frame distance matrix
    const void* qsplit = nullptr;    // the frame's queries, pre-split into bf16 MFMA operands by the previous launch (QSplitArgs), and their
    const float* qnorm = nullptr;    // norms: what the one-strip filter of a pipelined launch reads instead of the descriptors
    const int32_t* n_lo = nullptr;   // device row counts (NULL: the host's plan.n_rows is exact): the filter sees rows [0, n_lo[0]), the re-rank
    const int32_t* n_hi = nullptr;   // also scans [n_lo[0], n_hi[0]) exactly -- the words the previous frame appended meanwhile (AppendArgs)
    // shadow rows (round 6): the descriptors of the frame whose decision loop rides in the same launch A, as operand-table rows written by that
    // frame's query pre-split; the filter ranks them in sh_blocks extra strips, the re-rank takes the ones that became words (ShadowArgs)
    const void* sh_bf = nullptr; const float* sh_norm = nullptr; int sh_rows = 0;   // sh_rows: that frame's padded descriptor count
    const uint32_t* sh_mask = nullptr; int sh_q = 0;   // that frame's final new-word mask + prefix sums (AppendArgs::mask_out) and its descriptor count
    float* sh_x = nullptr; int sh_ld = 0;              // [q x sh_ld] the filter's scores of this frame's queries against those rows (launch A writes, launch B reads)
    float* cross = nullptr;          // [q x cross_ld] distances of this frame's queries to the cross_ncols descriptors at cross_cols (the frame
    int cross_ld = 0;                // before it): written by extra tiles of launch A, read by the re-rank of launch B for its pending rows (which are
    const void* cross_cols = nullptr; int cross_ncols = 0;   // descriptors of that frame); NULL: the re-rank stages the pending rows and computes them
};
// the new frame's queries -> MFMA operand order in global memory (knn_mfma_kernels.hip, qsplit_body): a few workgroups of launch A
struct QSplitArgs { const float* queries; int nq, qpad; uint4* qsplit; float* qnorm; int n_wgs; int f16 = 0; /* operands as IEEE half */
                    // round 6 ("shadow rows"): the same descriptors once more as ROWS of an operand table (256 B each, the layout of vocab_bf) with their
                    // augmentation entries ({|d|^2, 1}; {+inf, 1} for the padding rows and the sentinel at qpad) -- what the NEXT frame's filter multiplies
                    // to rank the words this frame is about to create; NULL: not wanted
                    uint32_t* shadow_bf = nullptr; float* shadow_norm = nullptr;
                    uint32_t* norm_max_bits = nullptr;   // the vocabulary's running maximum of |row|^2 (the re-rank's error bound is made from it): shadow rows count from now on
};
size_t knn_qsplit_bytes(int q);
int pipe_block_size();      // workgroup size of launch A (the filter's)
int pipe_b_block_size();    // workgroup size of launch B (re-rank + scoring)
// exact-redo helper workgroups of a fused frame launch (they leave at once when nothing was rejected; rowpar_body walks the rows in chunks)
#ifndef LCD_REDO_WGS_MAX      // (timing experiments: tools/build_variant.py <name> -DLCD_REDO_WGS_MAX=<n>)
#define LCD_REDO_WGS_MAX 32
#endif
constexpr int REDO_WGS_MAX = LCD_REDO_WGS_MAX;
void resolve_launch_info(const ResolveArgs& r, int block, int* n_redo, size_t* shmem);
// filter of the newest frame + the decision loop of one earlier frame (resolve: r / n_redo / shmem_resolve of that TailLaunch) + the
// registration of a still earlier one (reg: a / ret / shmem); either may be NULL
// k: the frame whose filter runs (NULL: none); qs: the frame whose queries are pre-split for the NEXT launch's filter (NULL: none)
// the handle's options that shape the two fused launches (per handle: lcd_set_option writes them into lcd_engine, nothing is process-wide)
struct PipeOpts {
    int f16 = 0;                     // the handle's filter multiplies fp16 operands: also picks the kernel variant of launches WITHOUT a filter
                                     // (pipeline fill / drain), so that they run the code the steady state keeps hot
    int cross_frames = 0;            // "cross_frame_tiles"
    int append_from_rerank = 1;      // "append_from_rerank" (0: the eight row-writer workgroups of round 4, for A/B runs)
    int append_split_buckets = -1;   // "append_split_buckets" (< 0: built-in)
    int filter_delay = 0;            // "filter_delay": s_sleep units (64 clocks) a filter workgroup waits in front of its first request
    // round 6, measured by kernel trace on the driver's command (profiles/r06_ab_notes.txt): all three on by default
    int shadow_rows = 1;             // "shadow_rows" (1: while the stream creates >= 16 words per frame; 2: always; 0: never): launch A also scores the frame against the descriptors of the frame before (whose new words are not rows
                                     // yet), the re-rank keeps the words' scores under its threshold: no workgroup stages or scans the new rows (launch B 19.6 -> 15.4 us)
    int mirror_from_b = 1;           // "mirror_from_b": the pinned row-count mirror of an appending frame is stored by launch B instead of by the decision loop (launch A -0.3 us)
    int slots_from_rows = 1;         // "slots_from_rows" (1: while the stream creates >= 16 words per frame; 2: always; 0: never): the decision loop leaves the ROW of the word a
                                     // descriptor matched; the registration (one launch later, in the round trip that fetches the retired signature's words anyway) looks the
                                     // postings key up.  0: the decision loop gathers the keys itself
    int decision_straight = 1;       // "decision_straight" (1: while the stream creates >= 16 words per frame; 2: always; 0: never): the decision loop requests everything its first round trip
                                     // reads unconditionally, in one straight line, the helpers' counter and the appender's row count with it (launch A 13.6 -> 12.6 us while frames
                                     // create words; once they only revisit the launch is 0.4-0.5 us LONGER that way, r06_ab_notes.txt 10)
    int row_writer_wgs = 16;         // "row_writer_wgs": > 0 = that many extra workgroups of launch B's re-rank role write the appended rows (launch B -0.9 us
                                     // without the shadow scores; with them nobody else could); 0: the re-rank workgroups write them at the end of their own chains
};
hipError_t launch_frame_a(const PipeKnn* k, const QSplitArgs* qs, const TailLaunch* resolve, const TailLaunch* reg, hipStream_t s,
                          hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr, const PipeOpts& opt = PipeOpts());
// app: the rows the decision loop of launch A of this pair published (NULL: none); they are also the pending rows of k's re-rank
hipError_t launch_frame_b(const PipeKnn* k, const ScoreArgs* score, int score_wgs, hipStream_t s, hipEvent_t ev_begin = nullptr,
                          hipEvent_t ev_end = nullptr, const AppendRowsArgs* app = nullptr, const PipeOpts& opt = PipeOpts());

// recycled allocations of bucket-sized device buffers (a bucket is born and dies every 256 frames in steady state:
// hipMalloc / hipFree there would synchronise the device)
struct BufPool {
    std::vector<DevBuf> free_list;
    hipError_t get(size_t bytes, DevBuf* out, int64_t* total);
    void put(DevBuf* b);
    void destroy(int64_t* total);
};

struct Tfidf {
    hipStream_t stream = nullptr;
    int64_t* bytes_device = nullptr;
    // per slot
    DevBuf slot_sig, slot_ni, slot_begin, slot_cnt;
    // per wslot
    DevBuf nw, did;                      // references, dense id (-1: none)
    DevBuf wrow;                         // vocabulary row that carries the key + 1 (0: none): a key held by a live row is never recycled, whatever
                                         // its reference count (rows appended on the device get their key there, the host learns of it later)
    DevBuf idf_tab;                      // {stamp, idf Q5.26} of the words of the current frame (valid iff stamp matches)
    uint32_t stamp = 0;
    // word id -> wslot: host vector (ids are small consecutive integers in the reference, ++_lastWordId) mirrored on the device
    std::vector<int32_t> id2ws;          // -1 = none
    DevBuf d_id2ws;
    int64_t d_id2ws_n = 0;               // entries valid on the device
    std::vector<int32_t> id2ws_dirty;    // ids whose device entry is out of date
    std::map<int32_t, int32_t> ws_free;  // recycled wslots (confirmed unreferenced by the device) as intervals: start -> length
    int64_t ws_free_count = 0;
    // wslots on their way back: a kernel checks nw == 0 for each and reports through pinned memory.  ids[i] != 0: the wslot was
    // reserved for new word ids[i] of a frame; if it turns out to be referenced, that word exists and keeps the wslot.
    struct PinBlock { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; };   // pinned memory + event, recycled (one batch per frame)
    std::vector<PinBlock> pin_free;
    struct ReleaseBatch { std::vector<int32_t> ws, ids; PinBlock blk; const uint8_t* ok = nullptr; bool recheck = false; };
    std::vector<ReleaseBatch> releasing;
    std::vector<int32_t> held_ws, held_ids;   // keys of superseded reservations waiting for a batched check
    std::vector<int32_t> ghost_ws;            // keys of removed words that a batch found still referenced: asked about again every 8th batch
    uint32_t flushes = 0;
    hipError_t flush_held();
    struct Reservation { int32_t first_id = 0, n = 0; WsRuns runs; } resv;   // wslots reserved for the new words of the last frame
    // per bucket
    DevBuf bkt_tab, bkt_ne, bkt_D, bkt_flags;
    std::vector<Bucket> buckets;
    int score_block = 512;               // threads per scoring workgroup (256 / 512 / 1024; lcd_set_option "score_block")
    int q_n_ub = 0;                      // word count of the last frame handed to frame_words (upper bound of its unique words)
    BufPool pool;
    DevBuf dir2;                         // word-major directory of the sparse parts: [dir2_blocks][dir2_stride] records of 32 bytes
    uint32_t dir2_blocks = 0, dir2_stride = 0, dir2_hint_blocks = 0, dir2_hint_stride = 0;
    hipError_t ensure_dir2(uint32_t blocks, uint32_t buckets_needed);
    DevBuf n_dense;                      // [0] number of dense ids handed out (device counter)
    uint32_t* h_n_dense = nullptr;       // pinned host mirror written by the sealing kernels (read without synchronising: stale is fine)
    DevBuf seal_cntw, seal_tiles;        // sealing scratch
    // per frame
    DevBuf q_w, q_idf, q_did, qd_did, qd_idf, q_meta;   // the frame's unique words: wslot, idf, dense id; its dense words; [0] = unique, [1] = dense
    DevBuf d_stage;                      // staged word ids of host-side calls
    DevBuf d_pairs;                      // (id, wslot) pairs on their way into d_id2ws
    PinBuf h_stage;
    // host maps
    std::unordered_map<int32_t, int64_t> sig_slot;    // live signature id -> slot
    int32_t n_wslots = 0;
    int64_t n_slots = 0, live_sigs = 0;
    int64_t postings_ub = 0;
    int64_t seals = 0;
    std::string err;

    hipError_t init(hipStream_t s, int64_t* bytes, int64_t sig_capacity, int64_t vocab_capacity);
    void destroy();
    // wslot of a word id (assigned on first sight when `create`); -1 if unknown and !create
    hipError_t wslot_of(int32_t word_id, bool create, int32_t* out);
    // bring the device copy of id2ws up to date
    hipError_t sync_id2ws();
    // the word left the dictionary (VWDictionary::removeWords): its wslot is recycled once the device confirms nw == 0
    hipError_t release_words(const int32_t* word_ids, int n);
    // recheck: keys without a word id that turn out to be still referenced are checked again with a later batch
    hipError_t release_wslots(const std::vector<int32_t>& ws, const std::vector<int32_t>* ids = nullptr, bool recheck = false);
    // the vocabulary rows' claim on their keys (wrow): rows [first_row, first_row + n) carry d_ws[0 .. n); rows d_rows[0 .. n) are gone;
    // the vocabulary was cleared
    hipError_t rows_take_keys(const int32_t* d_ws, int n, int64_t first_row);
    hipError_t rows_drop_keys(const int32_t* d_row_wslot, const int32_t* d_rows, int n);
    hipError_t rows_clear();
    // the device-side cleanUnusedWords keeps the keys of the rows it tombstones out of circulation (wrow = 0xFFFFFFFF) until the host has
    // caught up with its log of {row, key} pairs -- with nothing in flight: then they are released like any removed word's key
    hipError_t rows_unlog_keys(const int32_t* d_pairs, int n);
    // the device tombstoned the row of word `word_id` (key `ws`): if that is the word's permanent key it goes to the batched check
    void forget_word(int32_t word_id, int32_t ws);
    void free_wslot(int32_t w);          // into the interval set
    void free_wslot_run(int32_t start, int32_t len);   // a run of consecutive keys, one operation
    int32_t take_wslot();                // one recycled wslot, or -1
    void harvest_released(bool wait);
    // reserve n wslots for the new words first_id, first_id + 1, ... of the coming frame (recycled intervals first)
    // may_flush = false: the batched check of superseded reservations is not launched here (a pipelined handle launches it with
    // flush_held_if_due() once the registration that may still use those keys is enqueued)
    hipError_t reserve_new_words(int32_t first_id, int n, WsRuns* runs, bool may_flush = true);
    void adopt_key(int32_t word_id, int32_t ws);   // a word numbered on the device: id and key read from its row at reconciliation
    hipError_t flush_held_if_due() { return held_ws.size() >= 16384 ? flush_held() : hipSuccess; }
    // register one signature whose word slots are already on the device (d_wslots[n]; < 0 = no word); if N > 0 the
    // frame's unique words / idf are left in q_* for a following score()
    // defer != NULL (needs resolve): do not launch the frame tail, leave its launch arguments there -- sized for a workgroup of
    // pipe_block_size() threads -- for the filter launch of the next frame to carry (knn_mfma_kernels.hip, frame_a_kernel)
    // defer without resolve: the registration alone is left there (its word slots were written by a decision loop launched earlier;
    // new_ws translates that loop's new-word codes, see WsRuns)
    // shard_app (sharded frames; neither resolve nor defer): a second workgroup of the registration's launch appends this rank's share of the
    // frame's new words to its shard (ShardAppendJob) -- the two single-workgroup chains side by side instead of one launch behind the other
    hipError_t register_dev(int32_t sig_id, const int32_t* d_wslots, int n, int32_t ni, float N, const ResolveArgs* resolve = nullptr,
                            bool ids_given = false /* d_wslots holds word ids, translated on the device */, TailLaunch* defer = nullptr,
                            const WsRuns* new_ws = nullptr, const ShardAppendJob* shard_app = nullptr);
    // prepare q_* from word slots on the device without registering anything
    hipError_t query_dev(const int32_t* d_wslots, int n, float N, const ResolveArgs* resolve = nullptr, bool ids_given = false,
                         TailLaunch* defer = nullptr, const WsRuns* new_ws = nullptr, const ShardAppendJob* shard_app = nullptr);
    // Memory::loadDataFromDb replay: many signatures in O(1) launches; d_ids = word ids on the device, offsets[n_sigs + 1]
    hipError_t register_bulk(int n_sigs, const int32_t* sig_ids, const int64_t* offsets, const int32_t* ni, const int32_t* d_ids,
                             int64_t total_ids, int max_n);
    hipError_t flush_retire();
    std::vector<int64_t> pending_retire;   // slots whose device-side retirement rides along with the next frame-words launch
    // score q_* against every live signature: dense float likelihood over slots [0, n_slots)
    hipEvent_t prof_b = nullptr, prof_e = nullptr;   // one-shot: bracket the next scoring launch (lcd_profile_*)
    hipError_t score(float* d_likelihood);
    hipError_t score_work(int64_t out[8]);   // diagnostic, synchronises: what one scoring launch reads for the frame in q_*
    // sharded path: exact integer partial sums per slot (every slot written), and fixed point -> float after the all-reduce
    hipError_t score_fix(long long* lfix);
    hipError_t finalize(const long long* lfix_src, long long n, float* d_likelihood);
    hipError_t retire(int32_t sig_id);
    hipError_t seal_batch(const std::vector<int>& bucket_ids, bool bulk);
    hipError_t ensure_slots(int64_t n);
    hipError_t ensure_wslots(int32_t n);
    hipError_t ensure_buckets(int n);
    hipError_t set_bucket(int b);          // upload one bucket descriptor (tiny kernel: the data travels as kernel arguments)
    hipError_t new_bucket();
    hipError_t launch_score(float* d_likelihood, long long* lfix);
    // the arguments of a scoring launch with workgroups of `block` threads, and the number of workgroups (pending retirements are
    // applied first); for the fused launch of a pipelined frame
    hipError_t score_args(float* d_likelihood, long long* lfix, int block, ScoreArgs* out, int* n_wgs);
};

// gather of the dense likelihood: out[k] = slots[k] >= 0 ? dense[slots[k]] : 0
hipError_t launch_gather_f32(const float* dense, const int64_t* slots, int n, float* out, hipStream_t s);
// one rank's share of update()'s append on a sharded vocabulary as a launch of its own (tfidf.hip, shard_append_kernel): codes = the replicated
// decision loop's output; own_block > 0: block-cyclic owners from own_first on, else the last rank owns every new word.  lcd_shard_frame_dev no
// longer calls it -- the append rides in the registration's launch (ShardAppendJob) -- it stays as the stand-alone form of the same body
hipError_t launch_shard_append(const AppendArgs& ap, const WsRuns& new_ws, const int32_t* codes, int q, int rank, int world, int32_t own_first,
                               int32_t own_block, hipStream_t s);


}  // namespace lcd
