// engine.h -- internal state of an lcd_engine handle (liblcd_hip.so).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <deque>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lcd.h"
#include "devbuf.h"
#include "lcd_kernels.h"
#include "bayes.h"
#include "tfidf.h"


struct lcd_engine {
    int device = 0;
    int dtype = 0;
    int dim = 0;            // columns as given by the caller
    int row_bytes = 0;      // bytes per stored row (u8 rows are zero-padded to a multiple of 4)
    int kdim = 0;           // `dim` as the kernels see it (floats, or padded bytes)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int64_t bytes_device = 0;

    // ---- vocabulary: rows [n_rows x row_bytes], row_id[r] = word id (0 = tombstone), row_wslot[r] = postings key
    lcd::DevBuf vocab, row_id, row_wslot;
    lcd::DevBuf vocab_alt, row_id_alt, row_wslot_alt;   // rebuild target (swapped in)
    int64_t n_rows = 0, n_live = 0;
    // host mirror of the row order (the tie-break contract): key = word id the row was appended with, live = not tombstoned.
    // While the keys are ascending (the usual case: word ids only grow) a word's row is found by binary search; only a
    // vocabulary with out-of-order appends (re-activated old words) needs the id -> row map, built lazily.
    std::vector<int32_t> h_row_key;
    std::vector<char> h_row_live;
    bool rows_sorted = true;
    std::unordered_map<int32_t, int32_t> word_row;      // only valid when !rows_sorted && word_row_valid
    bool word_row_valid = false;
    lcd::DevBuf row_norm_alt;
    lcd::DevBuf vocab_bf;                               // hi/lo bf16 split of the rows (256 B per row) for the bf16x3 filter
    int find_row(int32_t word_id);

    // ---- per-call scratch
    lcd::DevBuf d_queries, d_partial, d_knn_row, d_knn_word, d_knn_wslot, d_knn_dist, d_selfdist, d_out_word, d_out_wslot,
        d_n_new, d_tmp_i32, d_extra_rows, d_extra_id, d_extra_word, d_extra_dist, d_extra_row, d_like, d_slots, d_bits, row_norm, norm_max, d_partial2, d_partial3, d_fail_list, d_fail_count;
    // a sharded search (lcd_shard_knn2_dev) leaves the frame's same-frame distance matrix here when its filter launch can carry it; the frame call
    // behind the all-gather (lcd_shard_frame_dev, same descriptors) then only merges and derives the bit rows
    lcd::DevBuf d_shard_selfdist;
    const void* shard_sd_desc = nullptr; int shard_sd_q = 0;
    bool fail_count_clean = false;                      // d_fail_count[0..1] known to be zero (the fused frame tail resets them)
    bool bf_family() const { return knn_mode == 2 || knn_mode == 3; }   // the bf16x3 / fp16 filters share kernels, tables and the pipelined frame
    int f16() const { return knn_mode == 3 ? 1 : 0; }
    int knn_mode = 2;                                   // f32 dim 64: 2 = bf16x3 MFMA filter + exact re-rank (default), 3 = fp16 one-product filter, 1 = f32 MFMA filter
                                                        // + exact re-rank, 0 = exact VALU scan only (lcd_config.knn_mode)
    // ---- pipelined frames (lcd_config.pipeline): three frames are in flight.  The call for frame t launches
    //        A = filter of frame t  +  decision loop of frame t - 1  +  retirement / registration of frame t - 2
    //        B = re-rank of frame t  +  scoring of frame t - 2            (then the decision stage of frame t - 2, if asked for)
    // so every single-workgroup latency chain hides behind the matrix-core filter.  The scratch a frame's stages hand to each other
    // lives in a ring indexed by the frame's sequence number; what a frame still owes (`stage`) and the calls made behind it
    // (retirements, neighbour lists, event records) wait in `inflight` until a later lcd_frame_dev carries them or any other call
    // on the handle completes them stand-alone (drain()).
    int pipeline = 0;
    hipStream_t kst = nullptr;                          // the stream the 2-NN stage is enqueued on (== stream)
    struct FrameScratch {
        lcd::DevBuf d_knn_row, d_knn_word, d_knn_dist, d_selfdist, d_bits, d_partial2, d_partial3, d_fail_list, d_fail_count, d_out_wslot;
        lcd::DevBuf d_qsplit, d_qnorm;                  // the frame's queries pre-split into bf16 matrix-core operands, their norms
        lcd::DevBuf d_applist;                          // deferred append: which descriptors of the frame became words (AppendArgs::list_out)
        lcd::DevBuf d_cross;                            // the frame's distances to the descriptors of the frame before it (PipeKnn::cross)
        lcd::DevBuf d_shadow_bf, d_shadow_norm, d_newmask;   // shadow rows: the frame's descriptors as operand-table rows + augmentation entries (written by its
                                                        // query pre-split), its final new-word mask + prefix sums (written by its decision loop)
        bool fail_count_clean = false;
    };
    static constexpr int PIPE_SETS = 4;                 // a frame's set is in use for four calls (pre-split .. registration)
    FrameScratch ring[PIPE_SETS];
    uint64_t frame_seq = 0;
    const void* last_fail_count = nullptr;              // certificate counters of the latest pipelined frame (lcd_get_stats)
    struct DeferredLink { std::vector<int32_t> triples, restart; };
    struct InFlight {
        lcd_frame_args a; lcd::ResolveArgs r; int set = 0;
        uint64_t vseq = 0; bool chained = false;        // the frame takes part in the device row-count chain (vcnt_active at its call)
        bool has_shadow = false;                        // its query pre-split also wrote its shadow rows (FrameScratch::d_shadow_bf)
        bool slots_are_rows = false;                    // its decision loop left vocabulary ROWS in r.out_wslot (PipeOpts::slots_from_rows): the registration looks the keys up
        lcd::WsRuns runs; bool reserved = false;        // postings keys of its new words (reserved when its decision loop is prepared)
        int stage = 0;                                  // what is owed next: 0 filter + re-rank, 1 the decision loop, 2 registration + scoring
        std::vector<int32_t> retire_after;              // lcd_sig_remove calls made while this was the newest frame
        std::vector<void*> events_after;                // lcd_record_event calls ...
        std::vector<DeferredLink> links_after;          // lcd_bayes_set_neighbors calls ...
        int cleans_after = 0;                           // lcd_vocab_remove_unused_async calls ...
    };
    std::deque<InFlight> inflight;                      // oldest first
    // ---- VWDictionary::update()'s append branch on the device (lcd_frame_args.append_new_words): the decision loop's workgroup turns the
    // frame's new words into vocabulary rows, so the row count lives on the device (d_vcnt: two alternating counters + a log of rows
    // appended per frame).  The host plans launches for an upper bound (the pinned mirror the appender writes, + q per younger frame)
    // and catches up with the exact rows (h_row_key ...) the next time the handle is drained (reconcile()).
    static constexpr int VLOG = 4096;
    lcd::DevBuf d_vcnt;                                 // int32: [0], [1] row counters, [16 .. 16 + VLOG) rows appended by frame seq % VLOG
    int64_t vocab_capacity_cfg = 0;                     // lcd_config.vocab_capacity: every per-row buffer is sized for it
    unsigned long long* h_vmirror = nullptr;            // pinned: (seq + 1) << 32 | rows after that frame's append
    struct DevAppend { uint64_t seq; int32_t first_id; int32_t q; bool enabled;
                       // sharded append (lcd_shard_frame_dev): the log holds the frame's TOTAL of new words, this rank owns the ids the rule gives it
                       int32_t own_world = 0, own_rank = 0, own_first = 0, own_block = 0; };
    // LCD_NEW_WORD_IDS_AUTO: the words frames create are numbered on the device, id = row + id_delta (AppendArgs::first_id <= 0; DevAppend::first_id = -id_delta).
    // next_word_id: one past the highest word id this handle has seen (rows appended by any call; "next_word_id" sets it: VWDictionary::_lastWordId + 1);
    // id_delta is fixed while appends are unreconciled (every new word is one row and one id), auto_window says the unreconciled appenders are numbered that way
    int32_t next_word_id = 1, id_delta = 1; bool auto_window = false;
    std::deque<DevAppend> unreconciled;                 // frames whose appends the host mirror has not caught up with
    uint64_t vseq = 0;                                  // sequence number of the next frame in the chain: it reads counter vseq & 1, writes the other
    bool vcnt_active = false;                           // the counters hold the row count (set when the first appending frame arrives)
    bool tail_dirty = true;                             // the host wrote (or reallocated) behind the rows since the tail was last filled
    int64_t tail_filled_rows = 0;                       // rows [n_rows, tail_filled_rows) carry +inf norms and a zero bf16 split
    // ---- Memory::cleanUnusedWords on the device without completing the frames in flight (lcd_vocab_remove_unused_async): the rows a
    // clean_unused_kernel tombstoned are logged on the device; the host's row mirror and the postings keys of the removed words catch up
    // with the log the next time the handle is drained (reconcile()).
    lcd::DevBuf d_rmlog;                                // int32: [0] rows logged, [16 ..] the rows
    int64_t rm_seen = 0;                                // log entries the host mirror has caught up with
    bool rm_pending = false;                            // a clean was enqueued since the last reconciliation
    int frames_since_reconcile = 0;                     // pipelined frames submitted with rm_pending set
    int enqueue_clean(const int32_t* reg_cnt = nullptr);  // flush the pending retirements, launch the kernel (nothing is synchronised); reg_cnt:
                                                        // device row count as of the newest registered frame (rows behind it are not scanned)
    bool clean_armed = false;                           // a clean waits for the next fused launch pair, whose registration applies the
                                                        // retirements asked for before it (they ride there: no launches of their own)
    int reconcile();
    void mirror_push_row(int32_t id, int64_t row);     // a row the device appended enters h_row_key / h_row_live / word_row
    int64_t rows_ub() const;
    // The rows the FILTER of chain frame `fseq` will most likely see (the count its launch reads is the one written a launch earlier: the
    // words of the frames up to fseq - 2): what the newest finished appender reported + an estimate per appending frame between that one
    // and fseq - 2, from the growth the reports have shown.  Only the launch PLAN is made for it -- correctness does not rest on it: the
    // filter masks rows beyond the device's count, and the re-rank scans exactly everything from min(plan, device count) on.
    int64_t rows_plan(uint64_t fseq);
    uint32_t est_tag = 0; int64_t est_cnt = 0; double est_new = 0.0;   // last report seen, decaying maximum of new rows per appending frame
    // sharded vocabulary, balanced growth (lcd_set_option "shard_growth_first" / "shard_growth_block"): the words frames create (ids >=
    // shard_first) belong to rank ((id - shard_first) / shard_block) % world; 0 = they belong to the last rank
    int32_t shard_first = 0, shard_block = 0;
    int shard_append = 0;                               // lcd_set_option("shard_append"): lcd_shard_frame_dev appends the new words this rank owns on the device
    int filter_units = -1;                              // lcd_set_option("filter_units")
    int strip_tiles = 0;                                // lcd_set_option("strip_tiles"): tiles per filter workgroup of a pipelined frame (0: planner)
    lcd::PipeOpts popt;                                 // options of the fused launches ("cross_frame_tiles", "append_from_rerank", "append_split_buckets", "filter_delay"); f16 follows knn_mode
    int sync_all();                                     // stream drained
    int drain(bool rows = true);                        // complete the owed index stage (stand-alone launches); rows: the host's row mirror
                                                        // catches up with the rows appended / removed on the device (reconcile(): synchronises, two small reads)
    const char* prof2_kernel = "score_kernel";
    lcd::PinBuf h_in, h_out, h_out2;
    lcd::PinBuf h_frame_in, h_frame_out;                // lcd_frame_host: descriptors in, word ids + likelihood out (one synchronisation per call)
    lcd::DevBuf d_frame_desc, d_frame_words, d_frame_like;
    lcd::Bayes bayes;                                   // Bayes filter over the signature slots (bayes.h)
    lcd::DevBuf d_adj_scratch;                          // adjusted likelihood when the caller wants the posterior but not that vector
    lcd::DevBuf d_hyp_scratch;                          // hypothesis record when the caller only wants the adjusted vector

    // ---- inverted index / TF-IDF
    lcd::Tfidf tfidf;

    // ---- event bracketing of the dominant kernel (lcd_profile_*)
    std::vector<hipEvent_t> prof_ev, prof2_ev;          // 2-NN scan kernel / fused likelihood kernel
    int prof_n = 0, prof_cap = 0, prof2_n = 0;
    int prof_skip = 0;                                  // "profile_skip": pipelined launches lcd_profile_begin lets pass before it samples
    bool prof_likelihood = true;                        // lcd_set_option("profile_likelihood"): also bracket launch B of a pipelined frame
    const char* prof_kernel = "";

    // ---- statistics
    int64_t knn_launches = 0, likelihood_launches = 0, rebuilds = 0, frame_calls = 0, frame_host_ns = 0;
    // where the host time of a pipelined lcd_frame_dev goes (lcd_debug_host_profile): ns accumulated per section
    //   0 checks + throttle + capacity   1 ring reservations   2 reserve_frame_words + decision-loop arguments   3 registration + scoring arguments
    //   4 filter plan (build_knn)   5 launch A   6 launch B   7 the rest of pipeline_launch (flush_held, finish_frame_ops)   8 calls
    int64_t host_prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // ---- roctx ranges (lcd_set_option "roctx"): resolved from libroctx64.so at run time, NULL while off
    int (*roctx_push)(const char*) = nullptr;
    int (*roctx_pop)() = nullptr;
    struct Range {   // a range for the lifetime of a scope
        lcd_engine* h;
        Range(lcd_engine* e, const char* name) : h(e && e->roctx_push ? e : nullptr) { if (h) h->roctx_push(name); }
        ~Range() { if (h) h->roctx_pop(); }
    };

    int fail(int code, const std::string& msg) { err = msg; return code; }
    int hip_fail(hipError_t e, const char* what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return LCD_ERR_HIP;
    }
};
