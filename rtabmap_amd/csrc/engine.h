// engine.h -- internal state of an lcd_engine handle (liblcd_hip.so).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/lcd.h"
#include "devbuf.h"
#include "lcd_kernels.h"
#include "tfidf.h"


// lcd_config.pipeline == 2: the index stage of a frame (registration + scoring launches and all the host bookkeeping of the
// inverted index) is enqueued by this thread while the caller's thread already enqueues the 2-NN stage of the next frame -- the
// HIP launch cost of a frame (~45 us on one thread) is what bounds a fully device-resident loop.  Jobs run in posting order, so
// the index sees exactly the call sequence; every API entry other than lcd_frame_dev / lcd_sig_remove drains the queue first.
struct IndexWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<std::function<int(std::string*)>> jobs;
    bool stop = false;
    uint64_t posted = 0, done = 0;
    int err_code = 0;               // first failure of an asynchronous job since it was last reported
    std::string err_msg;
    int device = 0;
    void start(int dev);
    uint64_t post(std::function<int(std::string*)> f);
    void wait(uint64_t n);          // until job number n (1-based posting order) has run
    void drain() { wait(posted); }
    void shutdown();
};

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
    bool fail_count_clean = false;                      // d_fail_count[0..1] known to be zero (the fused frame tail resets them)
    int knn_mode = 2;                                   // f32 dim 64: 2 = bf16x3 MFMA filter + exact re-rank (default), 1 = f32 MFMA filter
                                                        // + exact re-rank, 0 = exact VALU scan only (lcd_config.knn_mode)
    // ---- pipelined frames (lcd_config.pipeline): the 2-NN stage of a frame runs on `kstream`, its registration / scoring on
    // `stream`; the scratch the two stages share exists twice (the set in use above and `alt`), swapped every frame
    hipStream_t kstream = nullptr;                      // NULL: not pipelined
    hipStream_t kst = nullptr;                          // the stream the 2-NN stage is being enqueued on right now
    hipStream_t rstream = nullptr;                      // pipelined: the re-rank of frame t runs here, next to the filter of frame t + 1
    hipStream_t rst = nullptr;                          // the stream the re-rank is being enqueued on right now (NULL: same as kst)
    hipEvent_t ev_filter[2] = {nullptr, nullptr};       // filter of the frame that uses set i finished (recorded on kstream)
    struct AltScratch {
        lcd::DevBuf d_knn_row, d_knn_word, d_knn_dist, d_selfdist, d_bits, d_partial2, d_partial3, d_fail_list, d_fail_count, d_out_wslot;
        bool fail_count_clean = false;
    } alt;
    int ks_idx = 0;                                     // which of the two sets is the current one
    int ks_q[2] = {0, 0};                               // queries each set has been sized for
    hipEvent_t ev_knn[2] = {nullptr, nullptr};          // 2-NN stage of the frame that uses set i finished (recorded on kstream)
    hipEvent_t ev_tail[2] = {nullptr, nullptr};         // the frame tail that read set i finished (recorded on stream)
    bool k_busy = false;                                // work may be in flight on kstream
    int sync_all();                                     // both streams drained
    IndexWorker* worker = nullptr;                      // pipeline == 2
    uint64_t set_job[2] = {0, 0};                       // the index job that last used scratch set i
    int drain();                                        // run every queued index job; reports a failure one of them had
    lcd::PinBuf h_in, h_out, h_out2;
    lcd::DevBuf d_hyp_scratch;                          // hypothesis record when the caller only wants the adjusted vector

    // ---- inverted index / TF-IDF
    lcd::Tfidf tfidf;

    // ---- event bracketing of the dominant kernel (lcd_profile_*)
    std::vector<hipEvent_t> prof_ev, prof2_ev;          // 2-NN scan kernel / fused likelihood kernel
    int prof_n = 0, prof_cap = 0, prof2_n = 0;
    const char* prof_kernel = "";

    // ---- statistics
    int64_t knn_launches = 0, likelihood_launches = 0, rebuilds = 0, frame_calls = 0, frame_host_ns = 0;

    int fail(int code, const std::string& msg) { err = msg; return code; }
    int hip_fail(hipError_t e, const char* what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return LCD_ERR_HIP;
    }
};
