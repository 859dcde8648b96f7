// engine.hip -- implementation of the C-ABI declared in include/lcd.h (liblcd_hip.so).
//
// Host-side orchestration only: device memory, staging, stream ordering and the bookkeeping that keeps the
// vocabulary row order (the distance tie-break of the reference) and the signature/word slot maps.  All arithmetic
// of the hot path runs in the gfx950 kernels (knn2_kernels.hip, resolve_kernels.hip, tfidf.hip).  There is no CPU
// fallback: every entry point either runs on the device or returns an error status.
#include "engine.h"
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <thread>
#include <unordered_map>

using namespace lcd;

static_assert(sizeof(lcd::HypothesisOut) == sizeof(lcd_hypothesis), "lcd_hypothesis is the kernel's output record");

// row of a live word, -1 if absent
int lcd_engine::find_row(int32_t word_id) {
    if (rows_sorted) {
        auto it = std::lower_bound(h_row_key.begin(), h_row_key.begin() + n_rows, word_id);
        if (it == h_row_key.begin() + n_rows || *it != word_id) return -1;
        const int r = (int)(it - h_row_key.begin());
        return h_row_live[r] ? r : -1;
    }
    if (!word_row_valid) {
        word_row.clear();
        word_row.reserve((size_t)n_rows * 2);
        for (int64_t r = 0; r < n_rows; ++r) if (h_row_live[r]) word_row[h_row_key[r]] = (int32_t)r;
        word_row_valid = true;
    }
    auto it = word_row.find(word_id);
    return it == word_row.end() ? -1 : it->second;
}

#define LCD_CHECK_HANDLE(h) do { if (!(h)) return LCD_ERR_INVALID; } while (0)
#define LCD_HIP(h, x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return (h)->hip_fail(e__, #x); } while (0)
// every entry selects the device; every entry except lcd_frame_dev / lcd_sig_remove / lcd_record_event first completes the index
// stage a pipelined handle still owes for its last frame
#define LCD_DEV_NODRAIN(h) LCD_HIP(h, hipSetDevice((h)->device))
#define LCD_DEV(h) do { LCD_DEV_NODRAIN(h); int rc__ = (h)->drain(); if (rc__) return rc__; } while (0)

// The sharded stages between a frame that appended on the device and the next one need no exact row mirror: the search plans for rows_ub() (the
// rows behind the device's count carry +inf norms, a zero operand split and row id 0: no scan ranks them), the index calls do not look at rows
// at all.  Round 6: they complete what is owed WITHOUT reconciling (drain(false)) -- the per-frame synchronisation of the sharded path -- as long
// as the append log has room and the caller is within 8 frames of the device (the bound grows by q per unreported frame).
static int drain_keep_rows_lazy(lcd_engine* h) {
    { int rc = h->drain(false); if (rc) return rc; }
    const bool lazy = h->vcnt_active && h->shard_append && !h->rm_pending && h->unreconciled.size() < (size_t)lcd_engine::VLOG / 2;
    if (!lazy) return h->reconcile();
    if (h->h_vmirror && h->unreconciled.size() > 8) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 0;; ++spins) {
            const uint32_t tag = (uint32_t)(*(volatile const unsigned long long*)h->h_vmirror >> 32);
            if ((uint32_t)h->vseq - tag <= 8u) break;
            if (spins > 4096) std::this_thread::yield();
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                if (hipStreamSynchronize(h->stream) != hipSuccess) return h->fail(LCD_ERR_HIP, "hipStreamSynchronize");
                break;
            }
        }
    }
    return LCD_OK;
}

int lcd_engine::sync_all() {
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize(stream)");
    return LCD_OK;
}
#define LCD_JOIN_K(h) do { } while (0)

namespace {

inline hipError_t dreserve(lcd_engine* h, DevBuf& b, size_t bytes, size_t keep = 0) {
    return b.reserve(bytes, keep, h->stream, &h->bytes_device);
}

// One scratch buffer of a pipelined frame: in the frame's own set of the ring -- and, while nothing is in flight, in every other set as
// well, so that a steady stream of frames does not meet a hipMalloc (hundreds of microseconds) each time a set sees its first frame.
// The other sets are NOT touched while frames are in flight: those frames' launch arguments hold pointers into them.
inline hipError_t ring_reserve(lcd_engine* h, int own_set, DevBuf lcd_engine::FrameScratch::*member, size_t bytes) {
    hipError_t e = dreserve(h, h->ring[own_set].*member, bytes);
    if (e != hipSuccess || !h->inflight.empty()) return e;
    for (lcd_engine::FrameScratch& sc : h->ring) {
        e = dreserve(h, sc.*member, bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ... for a buffer whose size follows the vocabulary: `need` bytes now; when that takes a (re)allocation, `want` >= need bytes are asked for, so that the set does
// not outgrow the buffer again a few frames later (a reallocation with frames in flight waits for the stream, and every set of the ring pays its own)
inline hipError_t ring_reserve_grow(lcd_engine* h, int own_set, DevBuf lcd_engine::FrameScratch::*member, size_t need, size_t want) {
    if ((h->ring[own_set].*member).cap >= need) return hipSuccess;
    return ring_reserve(h, own_set, member, std::max(need, want));
}

// copy `rows` host rows (h->dim columns) into a device buffer laid out with h->row_bytes per row (u8 rows zero-padded)
int upload_rows(lcd_engine* h, const void* rows, int n, DevBuf& dst) {
    const size_t bytes = (size_t)n * h->row_bytes;
    LCD_HIP(h, dreserve(h, dst, std::max<size_t>(bytes, 4)));
    if (n == 0) return LCD_OK;
    LCD_HIP(h, h->h_in.reserve(bytes));
    const size_t src_row = (size_t)h->dim * (h->dtype == LCD_F32 ? 4 : 1);
    if (src_row == (size_t)h->row_bytes) {
        std::memcpy(h->h_in.p, rows, bytes);
    } else {
        std::memset(h->h_in.p, 0, bytes);
        for (int i = 0; i < n; ++i) std::memcpy((char*)h->h_in.p + (size_t)i * h->row_bytes, (const char*)rows + (size_t)i * src_row, src_row);
    }
    LCD_HIP(h, hipMemcpyAsync(dst.p, h->h_in.p, bytes, hipMemcpyHostToDevice, h->stream));
    // the staging buffer is reused by the next call: the copy must have left it
    LCD_HIP(h, hipStreamSynchronize(h->stream));
    return LCD_OK;
}

// the exact redo of rejected queries is left to the caller's next launch (the fused frame tail): describe it
static void fill_redo(lcd_engine* h, RowparArgs* a, const void* vocab, const int32_t* row_id, int n_rows, const void* d_queries, int32_t* o_row,
                      int32_t* o_word, float* o_dist, const CandBits* cb) {
    a->enabled = 1; a->vocab = (const float*)vocab; a->row_id = row_id; a->n_rows = n_rows; a->queries = (const float*)d_queries;
    a->fail_list = h->d_fail_list.as<int32_t>(); a->partial = (unsigned long long*)h->d_partial3.p;
    a->out_row = o_row; a->out_word = o_word; a->out_dist = o_dist;
    if (cb) a->cb = *cb;
}

// 2-NN of q device-resident queries against a row matrix -> o_{row,word,dist}[q*2].  `main_vocab` selects the resident
// vocabulary (which has row norms and may use the MFMA filter); other matrices (findNN's not-indexed words) use the exact scan.
int run_knn2_raw(lcd_engine* h, const void* d_queries, int q, const void* vocab, const int32_t* row_id, int64_t n_rows, bool main_vocab,
                 int32_t* o_row, int32_t* o_word, float* o_dist, const CandBits* cb = nullptr, RowparArgs* defer_redo = nullptr,
                 const ShardPackArgs* pack = nullptr /* a sharded search: the candidate records ride in the redo's launch ... */,
                 bool* packed = nullptr /* ... when the search has one (matrix-core filter): told here */) {
    if (packed) *packed = false;
    if (q == 0) return LCD_OK;
    const bool mfma = main_vocab && h->knn_mode != 0 && knn_mfma_supported(h->dtype, h->kdim) && n_rows >= 256;
    const KnnPlan p = knn_plan(q, (int)n_rows, h->row_bytes);
    if (mfma && h->bf_family()) {
        MfmaPlan mp = knn_bf16_plan(q, (int)n_rows, cb != nullptr ? knn_selfdist_wgs(q) : 0);
        mp.filter_units = h->filter_units;
        mp.f16 = h->f16();
        LCD_HIP(h, dreserve(h, h->d_partial2, knn_bf16_partial_bytes(mp)));
        LCD_HIP(h, dreserve(h, h->d_fail_list, (size_t)q * 4));
        const bool prof = h->prof_cap > 0 && h->prof_n < h->prof_cap;
        LCD_HIP(h, launch_knn_bf16(h->kdim, vocab, h->vocab_bf.p, h->row_norm.as<float>(), h->norm_max.as<uint32_t>(), row_id, d_queries, mp,
                                   h->d_partial2.p, o_row, o_word, o_dist, h->d_fail_list.as<int32_t>(), h->d_fail_count.as<int32_t>(),
                                   h->kst, prof ? h->prof_ev[2 * h->prof_n] : nullptr, prof ? h->prof_ev[2 * h->prof_n + 1] : nullptr,
                                   !h->fail_count_clean, cb, cb != nullptr));
        h->fail_count_clean = false;
        if (prof) { h->prof_n += 1; h->prof_kernel = h->f16() ? (knn_bf16_persistent(mp) ? "knn_bf16_filter_kernel_p (fp16 operands)" : "knn_bf16_filter_kernel (fp16 operands)")
                                                             : (knn_bf16_persistent(mp) ? "knn_bf16_filter_kernel_p" : "knn_bf16_filter_kernel"); }
        LCD_HIP(h, dreserve(h, h->d_partial3, knn_rowpar_partial_bytes((int)n_rows, q)));
        if (defer_redo) fill_redo(h, defer_redo, vocab, row_id, (int)n_rows, d_queries, o_row, o_word, o_dist, cb);
        else {
            LCD_HIP(h, launch_knn_rowpar(h->kdim, vocab, row_id, (int)n_rows, d_queries, h->d_fail_list.as<int32_t>(),
                                         h->d_fail_count.as<int32_t>(), h->d_partial3.p, o_row, o_word, o_dist, h->kst, cb, pack));
            if (pack && packed) *packed = true;
        }
    } else if (mfma) {
        const MfmaPlan mp = knn_mfma_plan(q, (int)n_rows);
        LCD_HIP(h, dreserve(h, h->d_partial2, knn_mfma_partial_bytes(mp)));
        LCD_HIP(h, dreserve(h, h->d_fail_list, (size_t)q * 4));
        const bool prof = h->prof_cap > 0 && h->prof_n < h->prof_cap;
        if (cb) LCD_HIP(h, launch_selfdist(h->dtype, h->kdim, d_queries, q, const_cast<float*>(cb->selfdist), cb->ld, h->kst));
        LCD_HIP(h, launch_knn_mfma(h->kdim, vocab, h->row_norm.as<float>(), h->norm_max.as<uint32_t>(), row_id, d_queries, mp, h->d_partial2.p,
                                   o_row, o_word, o_dist, h->d_fail_list.as<int32_t>(), h->d_fail_count.as<int32_t>(), h->kst,
                                   prof ? h->prof_ev[2 * h->prof_n] : nullptr, prof ? h->prof_ev[2 * h->prof_n + 1] : nullptr,
                                   !h->fail_count_clean, cb));
        h->fail_count_clean = false;
        if (prof) { h->prof_n += 1; h->prof_kernel = "knn_mfma_filter_kernel"; }
        // the queries the certificate rejected are redone exactly by the row-parallel kernel (usually none: it leaves at once)
        LCD_HIP(h, dreserve(h, h->d_partial3, knn_rowpar_partial_bytes((int)n_rows, q)));
        if (defer_redo) fill_redo(h, defer_redo, vocab, row_id, (int)n_rows, d_queries, o_row, o_word, o_dist, cb);
        else {
            LCD_HIP(h, launch_knn_rowpar(h->kdim, vocab, row_id, (int)n_rows, d_queries, h->d_fail_list.as<int32_t>(),
                                         h->d_fail_count.as<int32_t>(), h->d_partial3.p, o_row, o_word, o_dist, h->kst, cb, pack));
            if (pack && packed) *packed = true;
        }
    } else {
        LCD_HIP(h, dreserve(h, h->d_partial, knn_partial_bytes(p)));
        const bool prof = main_vocab && h->prof_cap > 0 && h->prof_n < h->prof_cap;
        if (prof) LCD_HIP(h, hipEventRecord(h->prof_ev[2 * h->prof_n], h->kst));
        LCD_HIP(h, launch_knn2_partial(h->dtype, h->kdim, vocab, row_id, d_queries, p, h->d_partial.as<uint64_t>(), h->kst));
        if (prof) { LCD_HIP(h, hipEventRecord(h->prof_ev[2 * h->prof_n + 1], h->kst)); h->prof_n += 1; h->prof_kernel = h->dtype == LCD_F32 ? "knn2_l2_kernel" : "knn2_hamming_kernel"; }
        LCD_HIP(h, launch_knn2_merge(h->dtype, p, h->d_partial.as<uint64_t>(), row_id, o_row, o_word, o_dist, h->kst));
    }
    h->knn_launches += 1;
    if (mfma) h->last_fail_count = h->d_fail_count.p;
    return LCD_OK;
}

int run_knn2(lcd_engine* h, const void* d_queries, int q, const void* vocab, const int32_t* row_id, const int32_t* row_wslot,
             int64_t n_rows, DevBuf& o_row, DevBuf& o_word, DevBuf& o_dist) {
    (void)row_wslot;
    LCD_HIP(h, dreserve(h, o_row, (size_t)std::max(q, 1) * 2 * 4));
    LCD_HIP(h, dreserve(h, o_word, (size_t)std::max(q, 1) * 2 * 4));
    LCD_HIP(h, dreserve(h, o_dist, (size_t)std::max(q, 1) * 2 * 4));
    return run_knn2_raw(h, d_queries, q, vocab, row_id, n_rows, vocab == h->vocab.p, o_row.as<int32_t>(), o_word.as<int32_t>(),
                        o_dist.as<float>());
}

int download(lcd_engine* h, void* dst, const void* d_src, size_t bytes, PinBuf& pin) {
    if (!bytes) return LCD_OK;
    LCD_HIP(h, pin.reserve(bytes));
    LCD_HIP(h, hipMemcpyAsync(pin.p, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
    LCD_HIP(h, hipStreamSynchronize(h->stream));
    std::memcpy(dst, pin.p, bytes);
    return LCD_OK;
}

// ---- VWDictionary::update()'s append branch on the device (see engine.h)
int64_t vocab_cap_rows(const lcd_engine* h) {
    int64_t c = (int64_t)(h->vocab.cap / (size_t)h->row_bytes);
    c = std::min<int64_t>(c, (int64_t)(h->row_id.cap / 4));
    c = std::min<int64_t>(c, (int64_t)(h->row_wslot.cap / 4));
    if (h->dtype == LCD_F32) c = std::min<int64_t>(c, (int64_t)(h->row_norm.cap / 8) - 1);
    if (knn_mfma_supported(h->dtype, h->kdim)) c = std::min<int64_t>(c, (int64_t)(h->vocab_bf.cap / 256));
    return std::max<int64_t>(c, 0);
}

// the row buffers hold `rows` rows; what lies behind the rows in use carries +inf norms and a zero bf16 split
int ensure_append_capacity(lcd_engine* h, int64_t rows) {
    const int64_t keep = h->rows_ub();
    if (rows > vocab_cap_rows(h)) {
        LCD_HIP(h, dreserve(h, h->vocab, (size_t)rows * h->row_bytes, (size_t)keep * h->row_bytes));
        LCD_HIP(h, dreserve(h, h->row_id, (size_t)rows * 4, (size_t)keep * 4));
        LCD_HIP(h, dreserve(h, h->row_wslot, (size_t)rows * 4, (size_t)keep * 4));
        if (h->dtype == LCD_F32) LCD_HIP(h, dreserve(h, h->row_norm, ((size_t)rows + 1) * 8, ((size_t)keep + 1) * 8));
        if (knn_mfma_supported(h->dtype, h->kdim)) LCD_HIP(h, dreserve(h, h->vocab_bf, (size_t)rows * 256, (size_t)keep * 256));
        h->tail_filled_rows = std::min(h->tail_filled_rows, keep);
    }
    const int64_t cap = vocab_cap_rows(h);
    const int64_t first = std::max(h->tail_filled_rows, keep);
    if (first < cap) {
        if (knn_mfma_supported(h->dtype, h->kdim)) LCD_HIP(h, launch_vocab_tail(h->row_norm.as<float>(), h->vocab_bf.p, first, cap - first, h->stream));
        // row id 0 behind the rows: a scan planned for an upper bound of the row count skips what does not exist yet like a tombstone
        LCD_HIP(h, hipMemsetAsync(h->row_id.as<int32_t>() + first, 0, (size_t)(cap - first) * 4, h->stream));
        h->tail_filled_rows = cap;
    }
    return LCD_OK;
}

// the first appending frame since the host last changed the vocabulary: the device counters take over the row count
int activate_dev_rows(lcd_engine* h) {
    if (h->vcnt_active) return LCD_OK;
    LCD_HIP(h, dreserve(h, h->d_vcnt, (size_t)(16 + lcd_engine::VLOG) * 4));
    if (!h->h_vmirror) {
        LCD_HIP(h, hipHostMalloc((void**)&h->h_vmirror, 64, hipHostMallocDefault));
        *h->h_vmirror = 0ull;
    }
    LCD_HIP(h, hipMemsetD32Async((hipDeviceptr_t)h->d_vcnt.p, (int)h->n_rows, 2, h->stream));
    if (h->tail_dirty) h->tail_filled_rows = 0;                      // host-side appends / rebuilds wrote behind the rows (or reallocated)
    h->tail_dirty = false;
    h->vcnt_active = true;
    return LCD_OK;
}

// the vocabulary buffers may have been reallocated since a frame's arguments were stored (device-side appends grow them)
void refresh_vocab_ptrs(lcd_engine* h, ResolveArgs* r) {
    r->row_wslot = h->row_wslot.as<int32_t>();
    if (r->rp.enabled) { r->rp.vocab = (const float*)h->vocab.p; r->rp.row_id = h->row_id.as<int32_t>(); }
}

// the append (or, for a frame that appends nothing, the hand-over of the row count) that rides with the decision loop of chain frame `vseq`
void fill_append(lcd_engine* h, const lcd_frame_args& a, uint64_t vseq, bool enabled, ResolveArgs* r, uint32_t* list_out = nullptr) {
    AppendArgs& ap = r->ap;
    ap = AppendArgs();
    ap.enabled = enabled ? 1 : 0;
    // pipelined frames of 64-float rows: the decision loop publishes the list, workgroups of launch B write the rows (append_rows_body)
    if (list_out && knn_mfma_supported(h->dtype, h->kdim)) { ap.defer_rows = 1; ap.list_out = list_out; }
    ap.descriptors = (const float*)a.d_descriptors; ap.row_dwords = h->row_bytes / 4; ap.is_f32_64 = knn_mfma_supported(h->dtype, h->kdim) ? 1 : 0;
    ap.vocab = h->vocab.as<uint32_t>(); ap.row_id = h->row_id.as<int32_t>(); ap.row_wslot = h->row_wslot.as<int32_t>();
    ap.row_norm = h->row_norm.as<float>(); ap.norm_max_bits = h->norm_max.as<uint32_t>(); ap.vocab_bf = h->vocab_bf.as<uint32_t>();
    ap.wrow = h->tfidf.wrow.as<uint32_t>(); ap.f16 = h->f16();
    ap.cnt_in = h->d_vcnt.as<int32_t>() + (vseq & 1); ap.cnt_out = h->d_vcnt.as<int32_t>() + ((vseq + 1) & 1);
    ap.log_slot = h->d_vcnt.as<int32_t>() + 16 + (vseq % lcd_engine::VLOG);
    ap.first_id = a.first_new_word_id == LCD_NEW_WORD_IDS_AUTO ? -h->id_delta : a.first_new_word_id; ap.capacity = vocab_cap_rows(h);
    ap.first_out = (int32_t*)a.d_first_new_word_id;
    ap.host_mirror = h->h_vmirror; ap.tag = (uint32_t)(vseq + 1);
}

}  // namespace

// rows the vocabulary can have by now: exact when nothing was appended on the device since the last reconciliation, else the count the
// newest finished appender reported (pinned memory, read without synchronising) + q per younger appending frame
int64_t lcd_engine::rows_ub() const {
    if (unreconciled.empty()) return n_rows;
    uint32_t tag = 0; int64_t cnt = 0;
    if (h_vmirror) { const unsigned long long v = *(volatile const unsigned long long*)h_vmirror; tag = (uint32_t)(v >> 32); cnt = (int64_t)(uint32_t)v; }
    int64_t extra = 0;
    for (auto it = unreconciled.rbegin(); it != unreconciled.rend(); ++it) {
        if (tag != 0 && (uint32_t)(it->seq + 1) == tag) return cnt + extra;
        if (it->enabled) extra += it->q;
    }
    return n_rows + extra;
}

int64_t lcd_engine::rows_plan(uint64_t fseq) {
    if (unreconciled.empty() || !h_vmirror) return n_rows;
    const unsigned long long v = *(volatile const unsigned long long*)h_vmirror;
    const uint32_t tag = (uint32_t)(v >> 32);
    const int64_t cnt = (int64_t)(uint32_t)v;
    if (tag == 0) return rows_ub();                                  // nothing reported yet
    if (est_tag != 0 && tag != est_tag) {                             // the reports moved on: rows per frame since the last look
        const double per = (double)(cnt - est_cnt) / (double)(uint32_t)(tag - est_tag);
        est_new = std::max(est_new * 0.9, per);
    }
    est_tag = tag; est_cnt = cnt;
    const int64_t ub = rows_ub();
    int64_t frames = 0; bool found = false;
    for (auto it = unreconciled.rbegin(); it != unreconciled.rend(); ++it) {
        if ((uint32_t)(it->seq + 1) == tag) { found = true; break; }
        if (it->enabled && it->seq + 2 <= fseq) frames += 1;          // an appender the filter's count includes, not reported yet
    }
    if (!found) return ub;
    const int64_t est = cnt + (int64_t)std::ceil((double)frames * (est_new * 1.25 + 8.0));
    return std::min(std::max(est, cnt), ub);
}

// one row the device appended enters the host's row mirror (the caller adds to n_rows / n_live)
void lcd_engine::mirror_push_row(int32_t id, int64_t row) {
    if (rows_sorted && !h_row_key.empty() && id <= h_row_key.back()) rows_sorted = false;
    if (word_row_valid) word_row[id] = (int32_t)row;
    h_row_key.push_back(id);
    if (id >= next_word_id) next_word_id = id + 1;
    h_row_live.push_back(1);
}

// the host's row mirror catches up with the device (synchronises)
int lcd_engine::reconcile() {
    if (unreconciled.empty() && !rm_pending) return LCD_OK;
    { int rc = sync_all(); if (rc) return rc; }
    std::vector<int32_t> log((size_t)VLOG);
    if (!unreconciled.empty()) {   // the log is a ring: the entries of the frames to catch up with form at most two stretches of it
        const size_t first = (size_t)(unreconciled.front().seq % VLOG), n = unreconciled.size();
        const size_t n1 = std::min(n, (size_t)VLOG - first);
        hipError_t e = hipMemcpy(log.data() + first, d_vcnt.as<int32_t>() + 16 + first, n1 * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && n > n1) e = hipMemcpy(log.data(), d_vcnt.as<int32_t>() + 16, std::min(n - n1, (size_t)VLOG) * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpy(append log)");
    }
    std::vector<std::pair<int64_t, int> > auto_rows;
    for (const DevAppend& a : unreconciled) {
        if (!a.enabled) continue;
        const int n = log[(size_t)(a.seq % VLOG)];
        est_new = std::max(est_new * 0.9, (double)n);                 // (the estimate the launch plans and the shadow-score switch use: rows_plan() only sees it move while frames are in flight)
        int taken = 0;
        const int64_t rows0 = n_rows;
        for (int k = 0; k < n; ++k) {
            const int32_t id = a.first_id > 0 ? a.first_id + k : (int32_t)(n_rows + taken) - a.first_id;   // (<= 0: the id follows the row)
            if (a.own_world > 0) {                                    // a sharded append: n is the frame's total, this rank wrote the ids it owns
                const bool mine = a.own_block > 0 ? (id >= a.own_first && ((id - a.own_first) / a.own_block) % a.own_world == a.own_rank)
                                                  : a.own_rank == a.own_world - 1;
                if (!mine) continue;
            }
            mirror_push_row(id, n_rows + taken);
            taken += 1;
        }
        n_rows += taken;
        n_live += taken;
        if (a.first_id <= 0 && taken > 0) auto_rows.push_back(std::pair<int64_t, int>(rows0, taken));
    }
    if (!auto_rows.empty()) {
        // words numbered on the device: the host learns their postings keys from the rows themselves, in ONE copy (with ids the caller supplies
        // the reservation check pairs them; here nobody knew the ids when the keys were reserved)
        const int64_t r0 = auto_rows.front().first;
        std::vector<int32_t> keys((size_t)(n_rows - r0));
        hipError_t e = hipMemcpy(keys.data(), row_wslot.as<int32_t>() + r0, keys.size() * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpy(row keys)");
        for (const std::pair<int64_t, int>& ar : auto_rows)
            for (int k = 0; k < ar.second; ++k) tfidf.adopt_key(h_row_key[(size_t)(ar.first + k)], keys[(size_t)(ar.first + k - r0)]);
    }
    unreconciled.clear();
    auto_window = false;
    if (rm_pending) {
        // rows tombstoned by the device-side cleanUnusedWords since the last reconciliation: the words are gone (removeWords,
        // VWDictionary.cpp:1595-1607), their postings keys go to the batched check that recycles them once nothing references them
        int32_t cnt = 0;
        hipError_t e = hipMemcpy(&cnt, d_rmlog.p, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpy(removal log)");
        const int64_t cap = ((int64_t)(d_rmlog.cap / 4) - 16) / 2;
        const int64_t n = std::min<int64_t>(cnt, cap);
        if (n > rm_seen) {
            std::vector<int32_t> ent((size_t)(n - rm_seen) * 2);
            e = hipMemcpy(ent.data(), d_rmlog.as<int32_t>() + 16 + 2 * rm_seen, ent.size() * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return hip_fail(e, "hipMemcpy(removal log)");
            // every batched key check enqueued so far has finished (the stream is drained): with their verdicts in, a key is either the
            // word's permanent one -- released here -- or still on its way through the reservation checks, which will find it free
            tfidf.harvest_released(true);
            e = tfidf.rows_unlog_keys(d_rmlog.as<int32_t>() + 16 + 2 * rm_seen, (int)(n - rm_seen));   // nothing is in flight: the keys may circulate again
            if (e != hipSuccess) return hip_fail(e, "wrow_unlog_kernel");
            for (size_t i = 0; i < ent.size(); i += 2) {
                const int32_t r = ent[i];
                if (r < 0 || r >= n_rows || !h_row_live[(size_t)r]) continue;
                h_row_live[(size_t)r] = 0;
                n_live -= 1;
                const int32_t id = h_row_key[(size_t)r];
                if (word_row_valid) word_row.erase(id);
                tfidf.forget_word(id, ent[i + 1]);
            }
            rm_seen = n;
        }
        rm_pending = false;
    }
    frames_since_reconcile = 0;
    return LCD_OK;
}

int lcd_engine::enqueue_clean(const int32_t* reg_cnt) {
    const int64_t rows = rows_ub();
    if (rows <= 0) return LCD_OK;
    hipError_t e = tfidf.flush_retire();                 // retirements ride with the next registration otherwise: the counts would be stale
    if (e != hipSuccess) return hip_fail(e, "flush_retire");
    const size_t need = ((size_t)std::max<int64_t>(rows, (int64_t)(vocab.cap / (size_t)row_bytes)) * 2 + 16) * 4;   // a row is logged at most once
    if (need > d_rmlog.cap) {
        const bool fresh = d_rmlog.p == nullptr;
        e = d_rmlog.reserve(need, d_rmlog.cap, stream, &bytes_device);
        if (e == hipSuccess && fresh) e = hipMemsetAsync(d_rmlog.p, 0, 64, stream);
        if (e != hipSuccess) return hip_fail(e, "removal log");
    }
    e = launch_clean_unused(row_id.as<int32_t>(), row_wslot.as<int32_t>(), tfidf.nw.as<uint32_t>(), tfidf.wrow.as<uint32_t>(),
                            dtype == LCD_F32 ? row_norm.as<float>() : nullptr, (int)rows, vcnt_active ? d_vcnt.as<int32_t>() : nullptr,
                            vcnt_active ? reg_cnt : nullptr, d_rmlog.as<int32_t>(), (int)((d_rmlog.cap / 4 - 16) / 2), stream);
    if (e != hipSuccess) return hip_fail(e, "clean_unused_kernel");
    rm_pending = true;
    return LCD_OK;
}

// No exception crosses the C-ABI (lcd.h): the bookkeeping of every entry point uses std:: containers, whose allocations may throw
static int lcd_catch(const lcd_engine* h, int code, const char* what) noexcept {
    if (h) { try { const_cast<lcd_engine*>(h)->err = what; } catch (...) { } }
    return code;
}
#define LCD_TRY try {
#define LCD_CATCH(h) } catch (const std::bad_alloc&) { return lcd_catch(h, LCD_ERR_NOMEM, "out of host memory"); } \
    catch (const std::exception& e__) { return lcd_catch(h, LCD_ERR_STATE, e__.what()); } \
    catch (...) { return lcd_catch(h, LCD_ERR_STATE, "unexpected exception"); }

extern "C" {

int lcd_abi_version(void) { return LCD_ABI_VERSION; }

int lcd_create(const lcd_config* cfg, lcd_engine** out) {
    LCD_TRY
    if (!cfg || !out) return LCD_ERR_INVALID;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(lcd_config)) return LCD_ERR_INVALID;
    if (cfg->dim <= 0 || cfg->dim > 4096 || (cfg->dtype != LCD_F32 && cfg->dtype != LCD_U8)) return LCD_ERR_INVALID;
    if (cfg->knn_mode < LCD_KNN_DEFAULT || cfg->knn_mode > LCD_KNN_F16) return LCD_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return LCD_ERR_HIP;
    if (hipSetDevice(cfg->device) != hipSuccess) return LCD_ERR_HIP;
    {   // the filter's launch plans fill THIS device's compute units (256 on MI355X; fewer on a partitioned part)
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) knn_set_compute_units(cus);
    }
    lcd_engine* h = new (std::nothrow) lcd_engine();
    if (!h) return LCD_ERR_NOMEM;
    h->device = cfg->device;
    h->dtype = cfg->dtype;
    h->dim = cfg->dim;
    if (cfg->dtype == LCD_F32) { h->row_bytes = cfg->dim * 4; h->kdim = cfg->dim; }
    else { h->row_bytes = (cfg->dim + 3) / 4 * 4; h->kdim = h->row_bytes; }
    if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; h->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return LCD_ERR_HIP; }
        h->own_stream = true;
    }
    const int64_t vcap = cfg->vocab_capacity > 0 ? cfg->vocab_capacity : 4096;
    h->vocab_capacity_cfg = vcap;
    hipError_t e = h->vocab.reserve((size_t)vcap * h->row_bytes, 0, h->stream, &h->bytes_device);
    if (e == hipSuccess) e = h->row_id.reserve((size_t)vcap * 4, 0, h->stream, &h->bytes_device);
    if (e == hipSuccess) e = h->row_wslot.reserve((size_t)vcap * 4, 0, h->stream, &h->bytes_device);
    if (e == hipSuccess) e = h->d_n_new.reserve(64, 0, h->stream, &h->bytes_device);
    if (e == hipSuccess) e = h->norm_max.reserve(64, 0, h->stream, &h->bytes_device);
    if (e == hipSuccess) e = hipMemsetAsync(h->norm_max.p, 0, 64, h->stream);
    if (e == hipSuccess) e = h->row_norm.reserve(((size_t)vcap + 1) * 8, 0, h->stream, &h->bytes_device);
    if (e == hipSuccess) e = h->d_fail_count.reserve(64, 0, h->stream, &h->bytes_device);   // [0] rejected, [1] arrivals, [2] max err / eps
    if (e == hipSuccess) e = h->d_hyp_scratch.reserve(64, 0, h->stream, &h->bytes_device);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_fail_count.p, 0, 64, h->stream);
    h->knn_mode = cfg->knn_mode == LCD_KNN_EXACT_VALU ? 0 : cfg->knn_mode == LCD_KNN_F32_MFMA ? 1 : cfg->knn_mode == LCD_KNN_F16 ? 3 : 2;
    h->kst = h->stream;
    if (cfg->pipeline < 0 || cfg->pipeline > 1) { delete h; return LCD_ERR_INVALID; }
    h->pipeline = cfg->pipeline;
    if (cfg->pipeline) {
        for (lcd_engine::FrameScratch& sc : h->ring) {
            if (e == hipSuccess) e = sc.d_fail_count.reserve(64, 0, h->stream, &h->bytes_device);
            if (e == hipSuccess) e = hipMemsetAsync(sc.d_fail_count.p, 0, 64, h->stream);
            sc.fail_count_clean = true;
        }
    }
    if (e == hipSuccess) e = h->tfidf.init(h->stream, &h->bytes_device, cfg->sig_capacity, cfg->vocab_capacity);
    h->bayes.init(h->stream, &h->bytes_device);
    if (e != hipSuccess) { lcd_destroy(h); return LCD_ERR_HIP; }
    *out = h;
    return LCD_OK;
    LCD_CATCH((const lcd_engine*)nullptr)
}

void lcd_destroy(lcd_engine* h) {
    try {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)h->drain();
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->tfidf.destroy();
    h->bayes.destroy();
    for (lcd_engine::FrameScratch& sc : h->ring) {
        DevBuf* all[] = {&sc.d_knn_row, &sc.d_knn_word, &sc.d_knn_dist, &sc.d_selfdist, &sc.d_bits, &sc.d_partial2, &sc.d_partial3, &sc.d_fail_list,
                         &sc.d_fail_count, &sc.d_out_wslot, &sc.d_qsplit, &sc.d_qnorm, &sc.d_applist, &sc.d_cross};
        for (DevBuf* d : all) d->release(&h->bytes_device);
    }
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->prof2_ev) (void)hipEventDestroy(e);
    DevBuf* all[] = {&h->vocab, &h->row_id, &h->row_wslot, &h->vocab_alt, &h->row_id_alt, &h->row_wslot_alt, &h->d_queries,
                     &h->d_partial, &h->d_knn_row, &h->d_knn_word, &h->d_knn_wslot, &h->d_knn_dist, &h->d_selfdist, &h->d_out_word,
                     &h->d_out_wslot, &h->d_n_new, &h->d_tmp_i32, &h->d_extra_rows, &h->d_extra_id, &h->d_extra_word,
                     &h->d_extra_dist, &h->d_extra_row, &h->d_like, &h->d_slots, &h->d_bits, &h->row_norm, &h->norm_max, &h->d_partial2,
                     &h->d_fail_list, &h->d_fail_count, &h->d_partial3, &h->row_norm_alt, &h->vocab_bf, &h->d_hyp_scratch, &h->d_adj_scratch,
                     &h->d_shard_selfdist};
    for (DevBuf* d : all) d->release(&h->bytes_device);
    h->d_vcnt.release(&h->bytes_device);
    h->d_rmlog.release(&h->bytes_device);
    if (h->h_vmirror) (void)hipHostFree(h->h_vmirror);
    h->h_in.release(); h->h_out.release(); h->h_out2.release();
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    } catch (...) { }
}

const char* lcd_last_error(const lcd_engine* h) { return h ? h->err.c_str() : "null handle"; }

int lcd_synchronize(lcd_engine* h) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV_NODRAIN(h);
    // what the frames in flight owe is completed and awaited; the host's row mirror is NOT brought up to date here (nothing a caller can
    // observe after this call needs it: every call that does completes the reconciliation itself) -- two device reads less per call
    { int rc = h->drain(false); if (rc) return rc; }
    return h->sync_all();
    LCD_CATCH(h)
}

void* lcd_stream(lcd_engine* h) { return h ? (void*)h->stream : nullptr; }

int lcd_pipeline_depth(const lcd_engine* h) { return (h && h->pipeline) ? 3 : 0; }

int lcd_record_event(lcd_engine* h, void* event) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    if (!event) return h->fail(LCD_ERR_INVALID, "lcd_record_event: null event");
    LCD_DEV_NODRAIN(h);
    if (!h->inflight.empty()) { h->inflight.back().events_after.push_back(event); return LCD_OK; }   // recorded behind the stages still owed
    LCD_HIP(h, hipEventRecord((hipEvent_t)event, h->stream));
    return LCD_OK;
    LCD_CATCH(h)
}

// ------------------------------------------------------------------------------------------------ vocabulary
int lcd_vocab_clear(lcd_engine* h) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    { int rc = h->sync_all(); if (rc) return rc; }
    h->n_rows = 0; h->n_live = 0;
    h->vcnt_active = false; h->tail_dirty = true;
    LCD_HIP(h, h->tfidf.rows_clear());
    if (h->d_rmlog.p) LCD_HIP(h, hipMemsetAsync(h->d_rmlog.p, 0, 4, h->stream));
    h->rm_seen = 0;
    h->h_row_key.clear();
    h->h_row_live.clear();
    h->rows_sorted = true;
    h->word_row.clear();
    h->word_row_valid = false;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_vocab_append(lcd_engine* h, const void* rows, int n, const int32_t* word_ids) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (n < 0 || (n > 0 && (!rows || !word_ids))) return h->fail(LCD_ERR_INVALID, "lcd_vocab_append: null input");
    if (n == 0) return LCD_OK;
    { int rc = h->sync_all(); if (rc) return rc; }                   // the 2-NN stage of a pipelined frame may still read the vocabulary
    for (int i = 0; i < n; ++i) {
        if (word_ids[i] <= 0) return h->fail(LCD_ERR_INVALID, "lcd_vocab_append: word ids must be > 0");
        if (h->find_row(word_ids[i]) >= 0) return h->fail(LCD_ERR_STATE, "lcd_vocab_append: word already in the vocabulary");
    }
    {   // the same id twice in one call would create two live rows for one word
        std::vector<int32_t> sorted(word_ids, word_ids + n);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
            return h->fail(LCD_ERR_INVALID, "lcd_vocab_append: duplicate word id in the call");
    }
    const int64_t total = h->n_rows + n;
    if (total > 0x7FFFFFF0ll) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_vocab_append: more than 2^31 rows");
    LCD_HIP(h, dreserve(h, h->vocab, (size_t)total * h->row_bytes, (size_t)h->n_rows * h->row_bytes));
    LCD_HIP(h, dreserve(h, h->row_id, (size_t)total * 4, (size_t)h->n_rows * 4));
    LCD_HIP(h, dreserve(h, h->row_wslot, (size_t)total * 4, (size_t)h->n_rows * 4));
    // stage rows | ids | wslots in one pinned block
    const size_t rb = (size_t)n * h->row_bytes;
    LCD_HIP(h, h->h_in.reserve(rb + (size_t)n * 8));
    char* st = (char*)h->h_in.p;
    const size_t src_row = (size_t)h->dim * (h->dtype == LCD_F32 ? 4 : 1);
    if (src_row == (size_t)h->row_bytes) std::memcpy(st, rows, rb);
    else {
        std::memset(st, 0, rb);
        for (int i = 0; i < n; ++i) std::memcpy(st + (size_t)i * h->row_bytes, (const char*)rows + (size_t)i * src_row, src_row);
    }
    int32_t* ids = (int32_t*)(st + rb);
    int32_t* ws = ids + n;
    for (int i = 0; i < n; ++i) {
        ids[i] = word_ids[i];
        LCD_HIP(h, h->tfidf.wslot_of(word_ids[i], true, &ws[i]));
    }
    LCD_HIP(h, hipMemcpyAsync((char*)h->vocab.p + (size_t)h->n_rows * h->row_bytes, st, rb, hipMemcpyHostToDevice, h->stream));
    LCD_HIP(h, hipMemcpyAsync(h->row_id.as<int32_t>() + h->n_rows, ids, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    LCD_HIP(h, hipMemcpyAsync(h->row_wslot.as<int32_t>() + h->n_rows, ws, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    LCD_HIP(h, h->tfidf.rows_take_keys(h->row_wslot.as<int32_t>() + h->n_rows, n, h->n_rows));
    if (h->dtype == LCD_F32) {   // |row|^2 for the MFMA filter
        LCD_HIP(h, dreserve(h, h->row_norm, ((size_t)total + 1) * 8, (size_t)h->n_rows * 8));
        LCD_HIP(h, launch_row_norms(h->vocab.p, h->row_id.as<int32_t>(), (int)h->n_rows, n, h->kdim, h->row_norm.as<float>(),
                                    h->norm_max.as<uint32_t>(), h->stream));
        if (knn_mfma_supported(h->dtype, h->kdim)) {   // hi/lo bf16 split of the new rows
            // (for the capacity the caller configured, like the other row buffers: a stream of appending frames must not meet a reallocation -- 32 MB copied
            // behind a synchronisation -- where this table alone was sized for the rows it was given: 125 000 rows fill a 32 MiB allocation to 131 072)
            LCD_HIP(h, dreserve(h, h->vocab_bf, (size_t)std::max<int64_t>(total, h->vocab_capacity_cfg) * 256, (size_t)h->n_rows * 256));
            LCD_HIP(h, launch_vocab_bf16(h->vocab.p, (int)h->n_rows, n, h->kdim, h->vocab_bf.p, h->stream, h->f16()));
        }
    }
    LCD_HIP(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; ++i) {
        if (h->rows_sorted && !h->h_row_key.empty() && word_ids[i] <= h->h_row_key.back()) h->rows_sorted = false;   // out-of-order id
        if (h->word_row_valid) h->word_row[word_ids[i]] = (int32_t)(h->n_rows + i);
        h->h_row_key.push_back(word_ids[i]);
        if (word_ids[i] >= h->next_word_id) h->next_word_id = word_ids[i] + 1;
        h->h_row_live.push_back(1);
    }
    h->n_rows = total;
    h->n_live += n;
    h->vcnt_active = false; h->tail_dirty = true;                    // the device row counters (appends by frames) start over from this count
    return LCD_OK;
    LCD_CATCH(h)
}

static int vocab_remove_ids(lcd_engine* h, const int32_t* word_ids, int n);

int lcd_vocab_remove(lcd_engine* h, const int32_t* word_ids, int n) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (n < 0 || (n > 0 && !word_ids)) return h->fail(LCD_ERR_INVALID, "lcd_vocab_remove: null input");
    if (n == 0) return LCD_OK;
    return vocab_remove_ids(h, word_ids, n);
    LCD_CATCH(h)
}

int lcd_vocab_remove_unused(lcd_engine* h, int32_t* out_word_ids, int capacity, int32_t* out_n) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (capacity < 0 || (capacity > 0 && !out_word_ids)) return h->fail(LCD_ERR_INVALID, "lcd_vocab_remove_unused: bad output buffer");
    if (out_n) *out_n = 0;
    if (h->n_rows == 0) return LCD_OK;
    LCD_HIP(h, h->tfidf.flush_retire());                             // retirements ride with the next frame otherwise: nw would be stale
    LCD_HIP(h, dreserve(h, h->d_tmp_i32, ((size_t)h->n_rows + 16) * 4));
    int32_t* d_cnt = h->d_tmp_i32.as<int32_t>();
    LCD_HIP(h, hipMemsetAsync(d_cnt, 0, 4, h->stream));
    LCD_HIP(h, launch_unused_rows(h->row_id.as<int32_t>(), h->row_wslot.as<int32_t>(), h->tfidf.nw.as<uint32_t>(), (int)h->n_rows, d_cnt + 16, d_cnt,
                                  (int)h->n_rows, h->stream));
    int32_t n = 0;
    { int rc = download(h, &n, d_cnt, 4, h->h_out2); if (rc) return rc; }
    if (n <= 0) return LCD_OK;
    std::vector<int32_t> rows((size_t)n);
    { int rc = download(h, rows.data(), d_cnt + 16, (size_t)n * 4, h->h_out); if (rc) return rc; }
    std::sort(rows.begin(), rows.end());
    std::vector<int32_t> ids((size_t)n);
    for (int i = 0; i < n; ++i) ids[(size_t)i] = h->h_row_key[(size_t)rows[(size_t)i]];
    { int rc = vocab_remove_ids(h, ids.data(), n); if (rc) return rc; }
    if (out_n) *out_n = n;
    for (int i = 0; i < n && i < capacity; ++i) out_word_ids[i] = ids[(size_t)i];
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_vocab_remove_unused_async(lcd_engine* h) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV_NODRAIN(h);
    // a pipelined handle still owes stages of its latest frames: the clean takes its place behind the newest of them (its registration
    // and the retirements asked for since), like lcd_sig_remove -- nothing is completed, nothing is synchronised
    if (!h->inflight.empty()) { h->inflight.back().cleans_after += 1; return LCD_OK; }
    return h->enqueue_clean();
    LCD_CATCH(h)
}

static int vocab_remove_ids(lcd_engine* h, const int32_t* word_ids, int n) {
    { int rc = h->sync_all(); if (rc) return rc; }
    std::vector<int32_t> rows;
    rows.reserve(n);
    for (int i = 0; i < n; ++i) {
        const int r = h->find_row(word_ids[i]);
        if (r >= 0) { rows.push_back(r); continue; }
        // not a row: a word that was created by a frame (_notIndexedWords) and dies before update() indexed it only gives its
        // postings key back (removeWords erases it from _notIndexedWords, :1602); anything else is an error
        int32_t ws = -1;
        LCD_HIP(h, h->tfidf.wslot_of(word_ids[i], false, &ws));
        if (ws < 0) return h->fail(LCD_ERR_STATE, "lcd_vocab_remove: unknown word");
    }
    {   // the same word twice would be counted out of n_live twice
        std::vector<int32_t> sorted(rows);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
            return h->fail(LCD_ERR_INVALID, "lcd_vocab_remove: duplicate word id in the call");
    }
    const int nr = (int)rows.size();
    if (nr) {
        LCD_HIP(h, dreserve(h, h->d_tmp_i32, (size_t)nr * 4));
        LCD_HIP(h, h->h_in.reserve((size_t)nr * 4));
        std::memcpy(h->h_in.p, rows.data(), (size_t)nr * 4);
        LCD_HIP(h, hipMemcpyAsync(h->d_tmp_i32.p, h->h_in.p, (size_t)nr * 4, hipMemcpyHostToDevice, h->stream));
        LCD_HIP(h, h->tfidf.rows_drop_keys(h->row_wslot.as<int32_t>(), h->d_tmp_i32.as<int32_t>(), nr));
        LCD_HIP(h, launch_tombstone(h->row_id.as<int32_t>(), h->d_tmp_i32.as<int32_t>(), nr, h->stream));
        if (h->dtype == LCD_F32) LCD_HIP(h, launch_norm_tombstone(h->row_norm.as<float>(), h->d_tmp_i32.as<int32_t>(), nr, h->stream));
        LCD_HIP(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < nr; ++i) h->h_row_live[rows[i]] = 0;
        h->vcnt_active = false;                                      // the counters restart: the next frame's filter sees every row (tombstones carry
                                                                     // +inf norms), its re-rank has no pending rows -- one of them might be gone now
        if (h->word_row_valid) for (int i = 0; i < n; ++i) h->word_row.erase(word_ids[i]);
        h->n_live -= nr;
    }
    // removeWords: the words are gone; their postings keys come back once the device has found them unreferenced
    LCD_HIP(h, h->tfidf.release_words(word_ids, n));
    return LCD_OK;
}

int lcd_vocab_rebuild(lcd_engine* h) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    { int rc = h->sync_all(); if (rc) return rc; }
    // permutation: live rows ordered by ascending word id (VWDictionary.cpp:636-660 walks std::map<int,VisualWord*>).
    // Word ids only grow in normal operation, so the live rows are already ascending and this is a stable compaction;
    // the sort is only needed after out-of-order appends (re-activated old words).
    std::vector<int32_t> perm;
    perm.reserve((size_t)h->n_live);
    for (int64_t r = 0; r < h->n_rows; ++r) if (h->h_row_live[r]) perm.push_back((int32_t)r);
    if (!h->rows_sorted)
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return h->h_row_key[a] < h->h_row_key[b]; });
    const int n = (int)perm.size();
    if (n == (int)h->n_rows && h->rows_sorted) return LCD_OK;       // nothing to drop, nothing to reorder
    LCD_HIP(h, dreserve(h, h->vocab_alt, std::max<size_t>(h->vocab.cap, 4)));
    LCD_HIP(h, dreserve(h, h->row_id_alt, std::max<size_t>(h->row_id.cap, 4)));
    LCD_HIP(h, dreserve(h, h->row_wslot_alt, std::max<size_t>(h->row_wslot.cap, 4)));
    if (h->dtype == LCD_F32) LCD_HIP(h, dreserve(h, h->row_norm_alt, std::max<size_t>(h->row_norm.cap, 16)));
    if (n) {
        LCD_HIP(h, dreserve(h, h->d_tmp_i32, (size_t)n * 4));
        LCD_HIP(h, h->h_in.reserve((size_t)n * 4 + 16));
        std::memcpy(h->h_in.p, perm.data(), (size_t)n * 4);
        LCD_HIP(h, hipMemcpyAsync(h->d_tmp_i32.p, h->h_in.p, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
        LCD_HIP(h, launch_gather_rows(h->vocab.p, h->row_id.as<int32_t>(), h->d_tmp_i32.as<int32_t>(), n, h->row_bytes,
                                      h->vocab_alt.p, h->row_id_alt.as<int32_t>(), h->stream));
        LCD_HIP(h, launch_gather_rows(h->row_wslot.p, h->row_id.as<int32_t>(), h->d_tmp_i32.as<int32_t>(), n, 4,
                                      h->row_wslot_alt.p, h->row_id_alt.as<int32_t>(), h->stream));
        if (h->dtype == LCD_F32) {
            // the augmentation table ({|row|^2, 1} per row) moves with the rows; the sentinel entry follows the last row
            LCD_HIP(h, launch_gather_rows(h->row_norm.p, h->row_id.as<int32_t>(), h->d_tmp_i32.as<int32_t>(), n, 8,
                                          h->row_norm_alt.p, h->row_id_alt.as<int32_t>(), h->stream));
            float* sentinel = (float*)((char*)h->h_in.p + (size_t)n * 4);
            const uint32_t inf_bits = 0x7f800000u;
            std::memcpy(&sentinel[0], &inf_bits, 4);
            sentinel[1] = 1.0f;
            LCD_HIP(h, hipMemcpyAsync(h->row_norm_alt.as<float>() + 2 * (size_t)n, sentinel, 8, hipMemcpyHostToDevice, h->stream));
        }
        LCD_HIP(h, hipStreamSynchronize(h->stream));                 // staging buffer reuse
    }
    std::swap(h->vocab, h->vocab_alt);
    std::swap(h->row_id, h->row_id_alt);
    std::swap(h->row_wslot, h->row_wslot_alt);
    if (h->dtype == LCD_F32) std::swap(h->row_norm, h->row_norm_alt);
    LCD_HIP(h, h->tfidf.rows_take_keys(h->row_wslot.as<int32_t>(), n, 0));   // the rows moved (the keys of dropped rows were released with them)
    if (h->d_rmlog.p) LCD_HIP(h, hipMemsetAsync(h->d_rmlog.p, 0, 4, h->stream));   // (reconciled by the drain above: the log starts over)
    h->rm_seen = 0;
    if (n && knn_mfma_supported(h->dtype, h->kdim)) {   // the split is recomputed from the compacted rows
        LCD_HIP(h, dreserve(h, h->vocab_bf, (size_t)n * 256));
        LCD_HIP(h, launch_vocab_bf16(h->vocab.p, 0, n, h->kdim, h->vocab_bf.p, h->stream, h->f16()));
    }
    std::vector<int32_t> keys(n);
    for (int i = 0; i < n; ++i) keys[i] = h->h_row_key[perm[i]];
    h->h_row_key.swap(keys);
    h->h_row_live.assign((size_t)n, 1);
    h->rows_sorted = true;
    h->word_row.clear();
    h->word_row_valid = false;
    h->n_rows = n;
    h->n_live = n;
    h->vcnt_active = false; h->tail_dirty = true;
    h->rebuilds += 1;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_vocab_count(const lcd_engine* h, int64_t* rows, int64_t* live) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    if (hipSetDevice(h->device) != hipSuccess) return LCD_ERR_HIP;
    { int rc = const_cast<lcd_engine*>(h)->drain(); if (rc) return rc; }   // rows appended on the device: the owed stages run, the mirror catches up
    if (rows) *rows = h->n_rows;
    if (live) *live = h->n_live;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_vocab_read(lcd_engine* h, int64_t first, int n, void* out_rows, int32_t* out_word_ids) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (first < 0 || n < 0 || first + n > h->n_rows) return h->fail(LCD_ERR_INVALID, "lcd_vocab_read: range");
    if (n == 0) return LCD_OK;
    if (out_rows) {
        const size_t bytes = (size_t)n * h->row_bytes;
        LCD_HIP(h, h->h_out.reserve(bytes));
        LCD_HIP(h, hipMemcpyAsync(h->h_out.p, (const char*)h->vocab.p + (size_t)first * h->row_bytes, bytes, hipMemcpyDeviceToHost, h->stream));
        LCD_HIP(h, hipStreamSynchronize(h->stream));
        const size_t dst_row = (size_t)h->dim * (h->dtype == LCD_F32 ? 4 : 1);
        for (int i = 0; i < n; ++i) std::memcpy((char*)out_rows + (size_t)i * dst_row, (const char*)h->h_out.p + (size_t)i * h->row_bytes, dst_row);
    }
    if (out_word_ids) { int rc = download(h, out_word_ids, h->row_id.as<int32_t>() + first, (size_t)n * 4, h->h_out2); if (rc) return rc; }
    return LCD_OK;
    LCD_CATCH(h)
}

// ------------------------------------------------------------------------------------------------ 2-NN
int lcd_knn2(lcd_engine* h, const void* queries, int q, int32_t* out_word_ids, float* out_dist) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    LCD_JOIN_K(h);
    if (q < 0 || (q > 0 && (!queries || !out_word_ids || !out_dist))) return h->fail(LCD_ERR_INVALID, "lcd_knn2: null input");
    if (q == 0) return LCD_OK;
    int rc = upload_rows(h, queries, q, h->d_queries);
    if (rc) return rc;
    rc = run_knn2(h, h->d_queries.p, q, h->vocab.p, h->row_id.as<int32_t>(), h->row_wslot.as<int32_t>(), h->n_rows, h->d_knn_row,
                  h->d_knn_word, h->d_knn_dist);
    if (rc) return rc;
    rc = download(h, out_word_ids, h->d_knn_word.p, (size_t)q * 8, h->h_out);
    if (rc) return rc;
    return download(h, out_dist, h->d_knn_dist.p, (size_t)q * 8, h->h_out);
    LCD_CATCH(h)
}

int lcd_selfdist(lcd_engine* h, const void* queries, int q, float* out_qxq) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    LCD_JOIN_K(h);
    if (q < 0 || (q > 0 && (!queries || !out_qxq))) return h->fail(LCD_ERR_INVALID, "lcd_selfdist: null input");
    if (q == 0) return LCD_OK;
    int rc = upload_rows(h, queries, q, h->d_queries);
    if (rc) return rc;
    LCD_HIP(h, dreserve(h, h->d_selfdist, (size_t)q * q * 4));
    LCD_HIP(h, launch_selfdist(h->dtype, h->kdim, h->d_queries.p, q, h->d_selfdist.as<float>(), q, h->stream));
    return download(h, out_qxq, h->d_selfdist.p, (size_t)q * q * 4, h->h_out);
    LCD_CATCH(h)
}

// device part of addNewWords up to (not including) the decision loop: 2-NN, same-frame distances + candidate bits.
// Fills the decision loop's arguments.
static int prepare_resolve(lcd_engine* h, const void* d_desc, int q, int flags, float nndr, int32_t* d_out_word, int32_t* d_out_wslot,
                           ResolveArgs* r, bool defer_redo = false /* the caller's next launch is the fused frame tail */,
                           int64_t rows_now = -1 /* rows to scan when the host's count lags the device's (an upper bound: the rows behind the
                                                    device's count carry row id 0 and are skipped like tombstones) */) {
    r->rp = RowparArgs{};
    if (rows_now < 0) rows_now = h->n_rows;
    const int have_index = h->n_live >= 2 ? 1 : 0;                  // VWDictionary.cpp:1015
    const bool incremental = (flags & LCD_Q_INCREMENTAL) != 0;
    const bool together = incremental && (flags & LCD_Q_NEW_WORDS_COMPARED);
    int ld = (q + 63) / 64 * 64;
    const int bw = ld / 32;
    if (together) {
        LCD_HIP(h, dreserve(h, h->d_selfdist, (size_t)q * ld * 4));
        LCD_HIP(h, dreserve(h, h->d_bits, cand_bits_bytes(q, bw)));
    }
    // With the MFMA filter the same-frame distance matrix does not wait for the 2-NN: extra workgroups of the filter launch
    // compute it, and the re-rank workgroup of a query -- the first to know the query's second neighbour -- derives the query's
    // candidate bits from its (symmetric) row: two launches fewer per frame.
    const int64_t knn_rows = have_index ? rows_now : 0;
    const bool side = together && h->knn_mode != 0 && knn_mfma_supported(h->dtype, h->kdim) && knn_rows >= 256 && q > 0;
    int rc;
    if (side) {
        CandBits cb;
        cb.selfdist = h->d_selfdist.as<float>(); cb.ld = ld; cb.nq = q; cb.have_index = have_index;
        cand_bits_layout(cb, h->d_bits.as<uint32_t>(), q, bw);
        LCD_HIP(h, dreserve(h, h->d_knn_row, (size_t)q * 2 * 4));
        LCD_HIP(h, dreserve(h, h->d_knn_word, (size_t)q * 2 * 4));
        LCD_HIP(h, dreserve(h, h->d_knn_dist, (size_t)q * 2 * 4));
        rc = run_knn2_raw(h, d_desc, q, h->vocab.p, h->row_id.as<int32_t>(), knn_rows, true, h->d_knn_row.as<int32_t>(),
                          h->d_knn_word.as<int32_t>(), h->d_knn_dist.as<float>(), &cb, defer_redo ? &r->rp : nullptr);
        if (rc) return rc;
    } else if (together && h->dtype == LCD_U8 && q > 0) {
        // Hamming frames (config 3): the scan, then ONE launch for the merge of its partial keys, the same-frame distance matrix and the candidate
        // bit rows (round 6: they were two dependent launches of ~5 us each behind the scan)
        LCD_HIP(h, dreserve(h, h->d_knn_row, (size_t)q * 2 * 4));
        LCD_HIP(h, dreserve(h, h->d_knn_word, (size_t)q * 2 * 4));
        LCD_HIP(h, dreserve(h, h->d_knn_dist, (size_t)q * 2 * 4));
        const KnnPlan p = knn_plan(q, (int)knn_rows, h->row_bytes);
        LCD_HIP(h, dreserve(h, h->d_partial, knn_partial_bytes(p)));
        const bool prof = h->prof_cap > 0 && h->prof_n < h->prof_cap;
        if (prof) LCD_HIP(h, hipEventRecord(h->prof_ev[2 * h->prof_n], h->kst));
        LCD_HIP(h, launch_knn2_partial(h->dtype, h->kdim, h->vocab.p, h->row_id.as<int32_t>(), d_desc, p, h->d_partial.as<uint64_t>(), h->kst));
        if (prof) { LCD_HIP(h, hipEventRecord(h->prof_ev[2 * h->prof_n + 1], h->kst)); h->prof_n += 1; h->prof_kernel = "knn2_hamming_kernel"; }
        LCD_HIP(h, launch_knn2_merge_selfdist_hamming(p, h->d_partial.as<uint64_t>(), h->row_id.as<int32_t>(), h->d_knn_row.as<int32_t>(),
                                                      h->d_knn_word.as<int32_t>(), h->d_knn_dist.as<float>(), d_desc, h->kdim, h->d_selfdist.as<float>(), ld,
                                                      have_index, h->d_bits.as<uint32_t>(), bw, h->kst));
        h->knn_launches += 1;
    } else {
        rc = run_knn2(h, d_desc, q, h->vocab.p, h->row_id.as<int32_t>(), h->row_wslot.as<int32_t>(), knn_rows, h->d_knn_row, h->d_knn_word,
                      h->d_knn_dist);
        if (rc) return rc;
        if (together)
            LCD_HIP(h, launch_selfdist(h->dtype, h->kdim, d_desc, q, h->d_selfdist.as<float>(), ld, h->kst, have_index,
                                       h->d_knn_word.as<int32_t>(), h->d_knn_dist.as<float>(), h->d_bits.as<uint32_t>(), bw));
    }
    r->q = q;
    r->flags = (incremental ? LCD_Q_INCREMENTAL : 0) | (together ? LCD_Q_NEW_WORDS_COMPARED : 0);
    r->nndr = nndr;
    r->have_index = have_index;
    r->knn_word = h->d_knn_word.as<int32_t>();
    r->knn_dist = h->d_knn_dist.as<float>();
    r->selfdist = together ? h->d_selfdist.as<float>() : nullptr;
    r->ld = ld;
    r->cand_bits = together ? h->d_bits.as<uint32_t>() : nullptr;
    r->bw = bw;
    r->out_word = d_out_word;
    r->out_n_new = h->d_n_new.as<int32_t>();
    r->knn_row = h->d_knn_row.as<int32_t>();
    r->row_wslot = h->row_wslot.as<int32_t>();
    r->out_wslot = d_out_wslot;
    r->new_ws = WsRuns();
    r->cand_list = nullptr; r->cand_cnt = nullptr;
    if (side) {                                                      // the re-rank also left the compact candidate lists
        CandBits lay;
        cand_bits_layout(lay, h->d_bits.as<uint32_t>(), q, bw);
        r->cand_list = lay.list; r->cand_cnt = lay.cnt;
    }
    r->fail_count = nullptr;
    return LCD_OK;
}

// device part of addNewWords: d_queries already holds q descriptors.  Leaves d_out_word[q], d_n_new[1].
static int quantize_dev(lcd_engine* h, const void* d_desc, int q, int flags, float nndr, int32_t* d_out_word, int32_t* d_out_wslot = nullptr) {
    ResolveArgs r;
    int rc = prepare_resolve(h, d_desc, q, flags, nndr, d_out_word, d_out_wslot, &r);
    if (rc) return rc;
    LCD_HIP(h, launch_resolve(r.q, r.flags, r.nndr, r.have_index, r.knn_word, r.knn_dist, r.selfdist, r.ld, r.cand_bits, r.bw, r.out_word,
                              r.out_n_new, h->stream, r.knn_row, r.row_wslot, r.out_wslot, nullptr, r.cand_list, r.cand_cnt));
    return LCD_OK;
}

int lcd_quantize(lcd_engine* h, const void* descriptors, int q, int flags, float nndr_ratio, int32_t* out_word_ids, int32_t* out_n_new) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    lcd_engine::Range range__(h, "lcd_quantize");
    LCD_DEV(h);
    LCD_JOIN_K(h);
    if (q < 0 || (q > 0 && (!descriptors || !out_word_ids))) return h->fail(LCD_ERR_INVALID, "lcd_quantize: null input");
    if (out_n_new) *out_n_new = 0;
    if (q == 0) return LCD_OK;
    if (q > 8192) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_quantize: more than 8192 descriptors per call");
    int rc = upload_rows(h, descriptors, q, h->d_queries);
    if (rc) return rc;
    LCD_HIP(h, dreserve(h, h->d_out_word, (size_t)q * 4));
    rc = quantize_dev(h, h->d_queries.p, q, flags, nndr_ratio, h->d_out_word.as<int32_t>());
    if (rc) return rc;
    rc = download(h, out_word_ids, h->d_out_word.p, (size_t)q * 4, h->h_out);
    if (rc) return rc;
    if (out_n_new) return download(h, out_n_new, h->d_n_new.p, 4, h->h_out2);
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_find_nn(lcd_engine* h, const void* queries, int q, const void* extra_rows, const int32_t* extra_word_ids, int n_extra, int flags,
                float nndr_ratio, int32_t* out_word_ids) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    LCD_JOIN_K(h);
    if (q < 0 || n_extra < 0 || (q > 0 && (!queries || !out_word_ids)) || (n_extra > 0 && (!extra_rows || !extra_word_ids)))
        return h->fail(LCD_ERR_INVALID, "lcd_find_nn: null input");
    if (q == 0) return LCD_OK;
    int rc = upload_rows(h, queries, q, h->d_queries);
    if (rc) return rc;
    const int have_index = h->n_live >= 2 ? 1 : 0;                  // VWDictionary.cpp:1347
    rc = run_knn2(h, h->d_queries.p, q, h->vocab.p, h->row_id.as<int32_t>(), h->row_wslot.as<int32_t>(), have_index ? h->n_rows : 0,
                  h->d_knn_row, h->d_knn_word, h->d_knn_dist);
    if (rc) return rc;
    if (n_extra > 0) {   // the not-yet-indexed words: a second, temporary vocabulary (VWDictionary.cpp:1416-1451)
        // upload_rows stages through h_in: upload extra rows after the queries are on the device
        rc = upload_rows(h, extra_rows, n_extra, h->d_extra_rows);
        if (rc) return rc;
        LCD_HIP(h, dreserve(h, h->d_extra_id, (size_t)n_extra * 4));
        LCD_HIP(h, h->h_in.reserve((size_t)n_extra * 4));
        std::memcpy(h->h_in.p, extra_word_ids, (size_t)n_extra * 4);
        LCD_HIP(h, hipMemcpyAsync(h->d_extra_id.p, h->h_in.p, (size_t)n_extra * 4, hipMemcpyHostToDevice, h->stream));
        LCD_HIP(h, hipStreamSynchronize(h->stream));
        // the merge kernel of the indexed search has consumed d_partial (stream order), so it can be reused
        rc = run_knn2(h, h->d_queries.p, q, h->d_extra_rows.p, h->d_extra_id.as<int32_t>(), nullptr, n_extra, h->d_extra_row,
                      h->d_extra_word, h->d_extra_dist);
        if (rc) return rc;
    }
    LCD_HIP(h, dreserve(h, h->d_out_word, (size_t)q * 4));
    LCD_HIP(h, launch_findnn_resolve(q, flags, nndr_ratio, have_index, h->d_knn_word.as<int32_t>(), h->d_knn_dist.as<float>(),
                                     n_extra > 0 ? 1 : 0, h->d_extra_word.as<int32_t>(), h->d_extra_dist.as<float>(),
                                     h->d_out_word.as<int32_t>(), h->stream));
    return download(h, out_word_ids, h->d_out_word.p, (size_t)q * 4, h->h_out);
    LCD_CATCH(h)
}

// ------------------------------------------------------------------------------------------------ inverted index
// word ids of a host-side call -> device (t.d_stage); the id -> postings-key translation happens in the kernels through the
// device copy of the table.  create: unknown ids get a postings key now (addWordRef on a word the index has not seen yet).
static int stage_word_ids(lcd_engine* h, const int32_t* word_ids, int64_t n, bool create) {
    Tfidf& t = h->tfidf;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t id = word_ids[i];
        if (id <= 0) continue;
        if ((size_t)id < t.id2ws.size() && t.id2ws[id] >= 0) continue;          // the common case: one vector read
        int32_t ws;
        hipError_t e = t.wslot_of(id, create, &ws);
        if (e == hipErrorInvalidValue) return h->fail(LCD_ERR_UNSUPPORTED, "word ids must be below 2^28");
        if (e != hipSuccess) return h->hip_fail(e, "wslot_of");
    }
    LCD_HIP(h, dreserve(h, t.d_stage, (size_t)std::max<int64_t>(n, 1) * 4));
    if (n) LCD_HIP(h, hipMemcpyAsync(t.d_stage.p, word_ids, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));   // caller-owned source:
    return LCD_OK;                                                                                               // every caller synchronises
}

int lcd_sig_add(lcd_engine* h, int32_t sig_id, const int32_t* word_ids, int n, int32_t ni) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (sig_id == 0 || n < 0 || (n > 0 && !word_ids) || ni < 0) return h->fail(LCD_ERR_INVALID, "lcd_sig_add: bad argument");
    if (n > TF_MAX_WORDS) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_sig_add: more than 8192 words in one signature");
    if (h->tfidf.sig_slot.count(sig_id)) return h->fail(LCD_ERR_STATE, "lcd_sig_add: signature already registered");
    int rc = stage_word_ids(h, word_ids, n, true);
    if (rc) return rc;
    LCD_HIP(h, h->tfidf.register_dev(sig_id, h->tfidf.d_stage.as<int32_t>(), n, ni, 0.0f, nullptr, true));
    LCD_HIP(h, hipStreamSynchronize(h->stream));   // the caller's buffer was the copy source
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_sig_add_bulk(lcd_engine* h, int n_sigs, const int32_t* sig_ids, const int64_t* sig_offsets, const int32_t* word_ids, const int32_t* ni) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (n_sigs < 0 || (n_sigs > 0 && (!sig_ids || !sig_offsets || !word_ids))) return h->fail(LCD_ERR_INVALID, "lcd_sig_add_bulk: null input");
    if (n_sigs == 0) return LCD_OK;
    Tfidf& t = h->tfidf;
    const int64_t total = sig_offsets[n_sigs] - sig_offsets[0];
    int max_n = 0;
    {
        std::vector<int32_t> sorted(sig_ids, sig_ids + n_sigs);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return h->fail(LCD_ERR_STATE, "lcd_sig_add_bulk: duplicate signature id");
    }
    for (int s = 0; s < n_sigs; ++s) {
        if (sig_ids[s] == 0 || t.sig_slot.count(sig_ids[s])) return h->fail(LCD_ERR_STATE, "lcd_sig_add_bulk: bad or duplicate signature id");
        const int64_t a = sig_offsets[s], b = sig_offsets[s + 1];
        if (b < a || b - a > TF_MAX_WORDS) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_sig_add_bulk: bad offsets / more than 8192 words");
        if (ni && ni[s] < 0) return h->fail(LCD_ERR_INVALID, "lcd_sig_add_bulk: negative ni");
        max_n = std::max(max_n, (int)(b - a));
    }
    // all word ids in one copy, one registration launch for every signature, a fixed number of launches per batch of sealed buckets
    int rc = stage_word_ids(h, word_ids + sig_offsets[0], total, true);
    if (rc) return rc;
    LCD_HIP(h, t.register_bulk(n_sigs, sig_ids, sig_offsets, ni, t.d_stage.as<int32_t>(), total, max_n));   // synchronises
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_sig_remove(lcd_engine* h, int32_t sig_id) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV_NODRAIN(h);
    if (!h->inflight.empty()) {
        // a pipelined handle still owes stages of its latest frames: the retirement takes its place behind the newest of them
        bool known = h->tfidf.sig_slot.count(sig_id) != 0, queued = false;
        for (const lcd_engine::InFlight& f : h->inflight) {
            if (f.a.sig_id != 0 && f.a.sig_id == sig_id) known = true;
            if (std::find(f.retire_after.begin(), f.retire_after.end(), sig_id) != f.retire_after.end()) queued = true;
        }
        if (!known || queued) return h->fail(LCD_ERR_STATE, "lcd_sig_remove: unknown signature");
        h->inflight.back().retire_after.push_back(sig_id);
        return LCD_OK;
    }
    if (!h->tfidf.sig_slot.count(sig_id)) return h->fail(LCD_ERR_STATE, "lcd_sig_remove: unknown signature");
    LCD_HIP(h, h->tfidf.retire(sig_id));
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_sig_count(const lcd_engine* h, int64_t* live_signatures, int64_t* postings) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    { int rc = const_cast<lcd_engine*>(h)->drain(); if (rc) return rc; }
    if (live_signatures) *live_signatures = h->tfidf.live_sigs;
    if (postings) *postings = h->tfidf.postings_ub;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_word_nrefs(lcd_engine* h, int32_t word_id, int32_t* out_nw) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (!out_nw) return h->fail(LCD_ERR_INVALID, "lcd_word_nrefs: null output");
    int32_t ws = -1;
    LCD_HIP(h, h->tfidf.wslot_of(word_id, false, &ws));
    if (ws < 0) { *out_nw = 0; return LCD_OK; }
    LCD_HIP(h, h->tfidf.flush_retire());             // retirements ride with the next frame otherwise: nw would be stale
    return download(h, out_nw, h->tfidf.nw.as<uint32_t>() + ws, 4, h->h_out2);
    LCD_CATCH(h)
}

int lcd_likelihood(lcd_engine* h, const int32_t* query_word_ids, int nq, const int32_t* sig_ids, int n_ids, float N, float* out) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    lcd_engine::Range range__(h, "lcd_likelihood");
    LCD_DEV(h);
    if (nq < 0 || n_ids < 0 || (nq > 0 && !query_word_ids) || (n_ids > 0 && (!sig_ids || !out)))
        return h->fail(LCD_ERR_INVALID, "lcd_likelihood: null input");
    if (n_ids == 0) return LCD_OK;                                   // reference: empty ids -> empty map (Memory.cpp:2227)
    if (nq > TF_MAX_WORDS) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_likelihood: more than 8192 query words");
    Tfidf& t = h->tfidf;
    if (t.n_slots == 0 || !(N > 0.0f)) { std::memset(out, 0, (size_t)n_ids * 4); return LCD_OK; }
    int rc = stage_word_ids(h, query_word_ids, nq, false);
    if (rc) return rc;
    LCD_HIP(h, t.query_dev(t.d_stage.as<int32_t>(), nq, N, nullptr, true));
    LCD_HIP(h, dreserve(h, h->d_like, (size_t)(t.n_slots + n_ids) * 4));
    LCD_HIP(h, t.score(h->d_like.as<float>()));
    h->likelihood_launches += 1;
    // gather the requested signatures
    LCD_HIP(h, h->h_in.reserve((size_t)n_ids * 8));
    int64_t* slots = h->h_in.as<int64_t>();
    for (int i = 0; i < n_ids; ++i) { auto it = t.sig_slot.find(sig_ids[i]); slots[i] = it == t.sig_slot.end() ? -1 : it->second; }
    LCD_HIP(h, dreserve(h, h->d_slots, (size_t)n_ids * 8));
    LCD_HIP(h, hipMemcpyAsync(h->d_slots.p, slots, (size_t)n_ids * 8, hipMemcpyHostToDevice, h->stream));
    float* d_out = h->d_like.as<float>() + t.n_slots;
    LCD_HIP(h, launch_gather_f32(h->d_like.as<float>(), h->d_slots.as<int64_t>(), n_ids, d_out, h->stream));
    return download(h, out, d_out, (size_t)n_ids * 4, h->h_out);
    LCD_CATCH(h)
}

// Rtabmap::adjustLikelihood on a device vector whose entry 0 is the virtual place, in place: the decision stage's two passes
// (bayes.hip) with every entry taking part
static int adjust_vector(lcd_engine* h, float* d_L, int n, float ratio) {
    DecideArgs d;
    d.like = d_L + 1; d.ratio = ratio; d.adj_out = d_L;
    LCD_HIP(h, h->bayes.decide(d, nullptr, (int64_t)n - 1, (int64_t)n - 1));
    return LCD_OK;
}

int lcd_adjust_likelihood(lcd_engine* h, float* likelihood, int n, float virtual_place_ratio) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (n < 0 || (n > 0 && !likelihood)) return h->fail(LCD_ERR_INVALID, "lcd_adjust_likelihood: null input");
    if (n == 0) return LCD_OK;
    LCD_HIP(h, dreserve(h, h->d_like, (size_t)n * 4));
    LCD_HIP(h, h->h_in.reserve((size_t)n * 4));
    std::memcpy(h->h_in.p, likelihood, (size_t)n * 4);
    LCD_HIP(h, hipMemcpyAsync(h->d_like.p, h->h_in.p, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    { int rc = adjust_vector(h, h->d_like.as<float>(), n, virtual_place_ratio); if (rc) return rc; }
    return download(h, likelihood, h->d_like.p, (size_t)n * 4, h->h_out);
    LCD_CATCH(h)
}

int lcd_adjust_likelihood_dev(lcd_engine* h, float* d_likelihood, int n, float virtual_place_ratio) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (n < 0 || (n > 0 && !d_likelihood)) return h->fail(LCD_ERR_INVALID, "lcd_adjust_likelihood_dev: null input");
    if (n == 0) return LCD_OK;
    return adjust_vector(h, d_likelihood, n, virtual_place_ratio);
    LCD_CATCH(h)
}

// ------------------------------------------------------------------------------------------------ device-resident frame
namespace {
struct FrameHostTimer {   // host time spent inside lcd_frame_dev (lcd_stats.frame_host_ns)
    lcd_engine* h; std::chrono::steady_clock::time_point t0;
    explicit FrameHostTimer(lcd_engine* e) : h(e), t0(std::chrono::steady_clock::now()) {}
    ~FrameHostTimer() { h->frame_host_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); h->frame_calls += 1; }
};
}  // namespace

// the index stage of a frame, launched on its own: registration (or query preparation), scoring, hypothesis
// Rtabmap::adjustLikelihood + the best candidate (Rtabmap.cpp:2121-2158), then the Bayes filter's update and its highest
// hypothesis (Rtabmap.cpp:2133-2158), without any vector leaving the device.  The frame's likelihood is already enqueued.
static int hypothesis_stage(lcd_engine* h, const lcd_frame_args& a) {
    Tfidf& t = h->tfidf;
    const bool bayes = a.d_posterior || a.d_bayes;
    if (!(a.d_hypothesis || a.d_adjusted || bayes)) return LCD_OK;
    const long long n_cons = (long long)t.n_slots - std::max(a.exclude_recent, 0);
    DecideArgs d;
    d.like = a.d_likelihood; d.ratio = a.virtual_place_ratio; d.adj_out = a.d_adjusted; d.hyp = (HypothesisOut*)a.d_hypothesis;
    d.bayes = bayes; d.d_posterior = a.d_posterior; d.d_bayes = (BayesOut*)a.d_bayes;
    LCD_HIP(h, h->bayes.decide(d, t.slot_sig.as<int32_t>(), t.n_slots, n_cons));
    return LCD_OK;
}

// likelihood + decision stage of a frame whose registration (or query preparation) has just been enqueued stand-alone
static int frame_score_s(lcd_engine* h, const lcd_frame_args& a) {
    Tfidf& t = h->tfidf;
    if (!a.d_likelihood) return LCD_OK;
    if (h->prof_cap > 0 && h->prof2_n < h->prof_cap) {
        t.prof_b = h->prof2_ev[2 * h->prof2_n]; t.prof_e = h->prof2_ev[2 * h->prof2_n + 1];
        h->prof2_n += 1;
        h->prof2_kernel = "score_kernel";
    }
    LCD_HIP(h, t.score(a.d_likelihood));
    if (t.prof_b) { t.prof_b = t.prof_e = nullptr; h->prof2_n -= 1; }     // the launch that would have been bracketed did not happen
    h->likelihood_launches += 1;
    return hypothesis_stage(h, a);
}

// (any descriptor type: rows that are not 64 floats are copied without the matrix-core filter's tables -- such handles are never pipelined)
static bool frame_appends(const lcd_engine* h, const lcd_frame_args& a) {
    return a.append_new_words != 0 && (a.first_new_word_id > 0 || a.first_new_word_id == LCD_NEW_WORD_IDS_AUTO) && (a.flags & LCD_Q_INCREMENTAL) != 0 &&
           h->row_bytes == h->dim * (h->dtype == LCD_F32 ? 4 : 1);     // (rows are stored as they arrive: no padding to add on the device)
}

// Who numbers the words this frame creates.  LCD_NEW_WORD_IDS_AUTO: the device, id = row + id_delta -- exact as long as every unreconciled appender is
// numbered that way (one new word = one row = one id), so a change of mode completes what is owed first; id_delta is set while the host's row
// mirror is current.
static int id_window(lcd_engine* h, const lcd_frame_args& a) {
    const bool is_auto = a.first_new_word_id == LCD_NEW_WORD_IDS_AUTO;
    if (a.first_new_word_id < 0 && !is_auto) return h->fail(LCD_ERR_INVALID, "lcd_frame_dev: first_new_word_id");
    if (is_auto && !frame_appends(h, a))
        return h->fail(LCD_ERR_INVALID, "lcd_frame_dev: LCD_NEW_WORD_IDS_AUTO needs append_new_words on an incremental dictionary of unpadded rows");
    if (!frame_appends(h, a)) return LCD_OK;
    if (!h->unreconciled.empty() && h->auto_window != is_auto) { int rc = h->drain(); if (rc) return rc; }
    if (h->unreconciled.empty()) {
        h->auto_window = is_auto;
        if (is_auto) {
            const int64_t d = (int64_t)h->next_word_id - h->n_rows;
            if (d < 1 || d >= (1ll << 28)) return h->fail(LCD_ERR_STATE, "lcd_frame_dev: next_word_id lies below the ids the vocabulary holds (lcd_set_option \"next_word_id\")");
            h->id_delta = (int32_t)d;
        }
    }
    return LCD_OK;
}

static int reserve_frame_words(lcd_engine* h, const lcd_frame_args& a, WsRuns* runs, bool may_flush = true) {
    *runs = WsRuns();
    // postings keys for the words this frame may create (VisualWord(id, descriptor, signatureId) references the signature; a word that
    // becomes a vocabulary row on the device needs its key there as well)
    if ((a.sig_id != 0 || frame_appends(h, a)) && (a.first_new_word_id > 0 || (a.first_new_word_id == LCD_NEW_WORD_IDS_AUTO && frame_appends(h, a))) && (a.flags & LCD_Q_INCREMENTAL)) {
        hipError_t e = h->tfidf.reserve_new_words(a.first_new_word_id, a.q, runs, may_flush);
        if (e == hipErrorInvalidValue) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_frame_dev: word ids must be below 2^28");
        LCD_HIP(h, e);
    }
    return LCD_OK;
}

// the whole index stage of a frame, launched on its own: decision loop + registration (one workgroup), scoring, decision stage
// chained / vseq: the frame's place in the device row-count chain (a frame that takes part appends its new words, or hands the count on)
static int frame_stage_s(lcd_engine* h, const lcd_frame_args& a, ResolveArgs r, bool chained = false, uint64_t vseq = 0) {
    Tfidf& t = h->tfidf;
    const int q = a.q;
    if (a.sig_id != 0 && t.sig_slot.count(a.sig_id)) return h->fail(LCD_ERR_STATE, "lcd_frame_dev: signature already registered");
    const int64_t slots_after = t.n_slots + (a.sig_id != 0 ? 1 : 0);
    if (a.d_likelihood && a.likelihood_capacity < slots_after) return h->fail(LCD_ERR_INVALID, "lcd_frame_dev: likelihood buffer too small");
    { int rc = reserve_frame_words(h, a, &r.new_ws); if (rc) return rc; }
    if (chained) fill_append(h, a, vseq, frame_appends(h, a), &r); else r.ap = AppendArgs();
    if (a.sig_id != 0) LCD_HIP(h, t.register_dev(a.sig_id, r.out_wslot, q, q, a.N, &r));
    else LCD_HIP(h, t.query_dev(r.out_wslot, q, a.N, &r));
    return frame_score_s(h, a);
}

// the same for a frame whose decision loop has already run (it left the word slots in r.out_wslot, new words as codes)
static int frame_stage_reg_s(lcd_engine* h, const lcd_frame_args& a, const ResolveArgs& r) {
    Tfidf& t = h->tfidf;
    if (a.sig_id != 0 && t.sig_slot.count(a.sig_id)) return h->fail(LCD_ERR_STATE, "lcd_frame_dev: signature already registered");
    if (a.sig_id != 0) LCD_HIP(h, t.register_dev(a.sig_id, r.out_wslot, a.q, a.q, a.N));
    else LCD_HIP(h, t.query_dev(r.out_wslot, a.q, a.N));
    return frame_score_s(h, a);
}

// the calls made while `f` was the newest frame of a pipelined handle, in call order, once every stage of `f` is enqueued
static int finish_frame_ops(lcd_engine* h, lcd_engine::InFlight& f) {
    for (int32_t sig : f.retire_after) LCD_HIP(h, h->tfidf.retire(sig));
    f.retire_after.clear();
    for (const lcd_engine::DeferredLink& dl : f.links_after) {
        LCD_HIP(h, h->bayes.ensure(std::max<int64_t>(h->tfidf.n_slots, 1)));
        const hipError_t e = h->bayes.link(dl.triples, dl.restart);
        if (e == hipErrorInvalidValue) { f.links_after.clear(); return h->fail(LCD_ERR_UNSUPPORTED, "lcd_bayes_set_neighbors: a neighbour list longer than 8192 entries"); }
        LCD_HIP(h, e);
    }
    f.links_after.clear();
    if (f.cleans_after > 0) { f.cleans_after = 0; h->clean_armed = true; }   // runs behind the next launch pair (pipeline_launch) or the drain
    for (void* ev : f.events_after) LCD_HIP(h, hipEventRecord((hipEvent_t)ev, h->stream));
    f.events_after.clear();
    return LCD_OK;
}

static int pipeline_launch(lcd_engine* h, const QSplitArgs* qs);
struct HostLap {   // section timer of the pipelined frame's host path (host_prof)
    lcd_engine* h; std::chrono::steady_clock::time_point t;
    explicit HostLap(lcd_engine* e) : h(e), t(std::chrono::steady_clock::now()) {}
    void lap(int i) { const auto n = std::chrono::steady_clock::now(); h->host_prof[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count(); t = n; }
};

int lcd_engine::drain(bool rows) {
    int rc_all = LCD_OK;
    Range range__(inflight.empty() ? nullptr : this, "lcd:drain");
    while (!inflight.empty()) {                                      // three fused launch pairs complete what is owed, oldest first
        const size_t before = inflight.size();
        const int stage_front = inflight.front().stage;
        const int rc = pipeline_launch(this, nullptr);
        if (rc && !rc_all) rc_all = rc;
        if (rc && inflight.size() == before && inflight.front().stage == stage_front) {   // no progress: drop the frame instead of spinning
            InFlight f = std::move(inflight.front());
            inflight.pop_front();
            (void)finish_frame_ops(this, f);
        }
    }
    if (clean_armed) { clean_armed = false; const int rc = enqueue_clean(); if (rc && !rc_all) rc_all = rc; }
    const int rc3 = rows ? reconcile() : LCD_OK;                     // rows appended on the device: the host mirror catches up
    return rc_all ? rc_all : rc3;
}

// The 2-NN stage of an in-flight frame, planned when its filter is about to be launched (the row count may have grown since the frame was
// submitted): scratch of the frame's ring set, launch plan, and the arguments its decision loop will need one launch later.
// f_sh: the frame whose decision loop rides in the same launch A and whose shadow rows this filter ranks (NULL: none)
static int build_knn(lcd_engine* h, lcd_engine::InFlight& f, PipeKnn* kp, const lcd_engine::InFlight* f_sh) {
    PipeKnn& k = *kp;
    const lcd_frame_args& a = f.a;
    const int q = a.q;
    lcd_engine::FrameScratch& sc = h->ring[f.set];
    const bool incremental = (a.flags & LCD_Q_INCREMENTAL) != 0;
    const bool together = incremental && (a.flags & LCD_Q_NEW_WORDS_COMPARED);
    const int ld = (q + 63) / 64 * 64, bw = ld / 32;
    const int64_t rows_bound = f.chained ? h->rows_ub() : h->n_rows;   // a true upper bound: the exact redo and the buffers are sized for it
    const int64_t plan_rows = f.chained ? h->rows_plan(f.vseq) : h->n_rows;
    if (plan_rows > 0x7FFFFFF0ll) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_frame_dev: more than 2^31 rows");
    // The distance tiles get compute units of their own (a tile that shares one with a strip takes twice as long, and so does the
    // strip); the two tail workgroups do not: a filter workgroup holds 66 KB of LDS, so two of the launch's workgroups can share a
    // compute unit, and one strip less per workgroup is worth more than the two shared units (49 000 words x 500 descriptors:
    // 219 seven-tile strips + 36 tiles + 2 = 257 workgroups, frame 31.6 us; 192 eight-tile strips 32.1; 256 six-tile strips 34.5).
    k.plan = knn_bf16_plan_pipelined(q, (int)plan_rows, together ? knn_selfdist_wgs(q) : 0, h->filter_units);
    k.plan.f16 = h->f16();
    if (h->strip_tiles > 0 && plan_rows > 0) {                       // timing experiments: a fixed strip length, one workgroup per strip
        const int n_tiles = (int)((plan_rows + 31) / 32);
        k.plan.tiles_per_block = h->strip_tiles; k.plan.n_blocks = (n_tiles + h->strip_tiles - 1) / h->strip_tiles; k.plan.one_strip = 1;
        k.plan.f16 = h->f16();
    }
    if (f_sh && !knn_bf16_persistent(k.plan) && plan_rows < (int64_t)SHADOW_ROW_BASE) {
        const lcd_engine::FrameScratch& ss = h->ring[f_sh->set];
        k.plan.n_shadow = 1;
        k.sh_bf = ss.d_shadow_bf.p; k.sh_norm = ss.d_shadow_norm.as<float>(); k.sh_rows = (f_sh->a.q + 63) / 64 * 64;
        k.sh_mask = ss.d_newmask.as<uint32_t>(); k.sh_q = f_sh->a.q;
        k.sh_ld = k.sh_rows;
        LCD_HIP(h, ring_reserve(h, f.set, &lcd_engine::FrameScratch::d_cross, (size_t)q * k.sh_ld * 4));   // (the buffer the cross-frame tiles use: never both)
        k.sh_x = sc.d_cross.as<float>();
    }
    {   // the candidate records: sized for this plan AND for the one the upper bound would get (a growing vocabulary crosses the planner's
        // thresholds: a reallocation drains the stream)
        // ... and, when that takes an allocation, for a vocabulary half as large again: a set that outgrows its buffers while frames are in flight reallocates
        // behind a synchronisation of the stream, and so does each of the other sets when its turn comes -- four stalls of ~0.24 ms in a row where
        // 125 000 rows happen to fill their allocation (profiles/dead_ends_r06.txt 15)
        size_t bytes = knn_bf16_partial_bytes(k.plan), want = 0;
        if (f.chained) {
            bytes = std::max(bytes, knn_bf16_partial_bytes(knn_bf16_plan_pipelined(q, (int)(rows_bound + 8 * (int64_t)q), together ? knn_selfdist_wgs(q) : 0, h->filter_units)));
            want = knn_bf16_partial_bytes(knn_bf16_plan_pipelined(q, (int)std::min<int64_t>(rows_bound + rows_bound / 2 + 65536, 0x7FFFFF00ll), together ? knn_selfdist_wgs(q) : 0, h->filter_units));
        }
        LCD_HIP(h, ring_reserve_grow(h, f.set, &lcd_engine::FrameScratch::d_partial2, bytes, want));
    }
    {
        const int64_t rows3 = rows_bound + (f.chained ? 8 * (int64_t)q : 0);
        LCD_HIP(h, ring_reserve_grow(h, f.set, &lcd_engine::FrameScratch::d_partial3, knn_rowpar_partial_bytes((int)rows3, q),
                                     f.chained ? knn_rowpar_partial_bytes((int)std::min<int64_t>(rows3 + rows3 / 2 + 65536, 0x7FFFFF00ll), q) : 0));
    }
    k.vocab = h->vocab.p; k.vocab_bf = h->vocab_bf.p; k.row_norm = h->row_norm.as<float>(); k.norm_max_bits = h->norm_max.as<uint32_t>();
    k.row_id = h->row_id.as<int32_t>(); k.queries = a.d_descriptors; k.partial = sc.d_partial2.p;
    k.qsplit = sc.d_qsplit.p; k.qnorm = sc.d_qnorm.as<float>();
    k.out_row = sc.d_knn_row.as<int32_t>(); k.out_word = sc.d_knn_word.as<int32_t>(); k.out_dist = sc.d_knn_dist.as<float>();
    k.fail_list = sc.d_fail_list.as<int32_t>(); k.fail_count = sc.d_fail_count.as<int32_t>();
    k.n_lo = nullptr; k.n_hi = nullptr;
    if (f.chained) { k.n_lo = h->d_vcnt.as<int32_t>() + ((f.vseq + 1) & 1); k.n_hi = h->d_vcnt.as<int32_t>() + (f.vseq & 1); }
    k.cb = CandBits();
    if (together) { k.cb.selfdist = sc.d_selfdist.as<float>(); k.cb.ld = ld; k.cb.nq = q; k.cb.have_index = 1; cand_bits_layout(k.cb, sc.d_bits.as<uint32_t>(), q, bw); }
    if (!sc.fail_count_clean) LCD_HIP(h, hipMemsetAsync(sc.d_fail_count.p, 0, 8, h->stream));
    sc.fail_count_clean = true;                                      // the frame's decision loop (a later launch A) resets the counters
    h->last_fail_count = sc.d_fail_count.p;
    // ---- the decision loop's arguments (launched one call later), the redo of rejected queries riding with it
    ResolveArgs& r = f.r;
    r = ResolveArgs();
    r.q = q; r.flags = (incremental ? LCD_Q_INCREMENTAL : 0) | (together ? LCD_Q_NEW_WORDS_COMPARED : 0); r.nndr = a.nndr_ratio; r.have_index = 1;
    r.knn_word = k.out_word; r.knn_dist = k.out_dist; r.selfdist = together ? sc.d_selfdist.as<float>() : nullptr; r.ld = ld;
    r.cand_bits = together ? sc.d_bits.as<uint32_t>() : nullptr; r.bw = bw; r.out_word = a.d_word_ids; r.out_n_new = h->d_n_new.as<int32_t>();
    r.cand_list = together ? k.cb.list : nullptr; r.cand_cnt = together ? k.cb.cnt : nullptr;
    r.knn_row = k.out_row; r.row_wslot = h->row_wslot.as<int32_t>(); r.out_wslot = sc.d_out_wslot.as<int32_t>(); r.new_ws = WsRuns();
    r.fail_count = sc.d_fail_count.as<int32_t>();
    RowparArgs& rp = r.rp;
    rp.enabled = 1; rp.vocab = (const float*)h->vocab.p; rp.row_id = h->row_id.as<int32_t>(); rp.n_rows = (int)rows_bound;
    rp.n_rows_dev = k.n_hi;                                          // the rows that exist when the redo runs: after the previous frame's append
    rp.queries = (const float*)a.d_descriptors; rp.fail_list = sc.d_fail_list.as<int32_t>(); rp.partial = (unsigned long long*)sc.d_partial3.p;
    rp.out_row = k.out_row; rp.out_word = k.out_word; rp.out_dist = k.out_dist;
    if (together) rp.cb = k.cb;
    return LCD_OK;
}

// One pair of fused launches of a pipelined handle.  With frame t the newest:
//   A = queries of frame t pre-split into matrix-core operands  +  filter (+ same-frame distance tiles) of frame t-1
//       + decision loop of frame t-2 (+ its redo helpers, + the append of its new words)  +  retirement / registration of frame t-3
//   B = re-rank of frame t-1  +  scoring of frame t-3   (then the decision stage of frame t-3 and the calls queued behind it)
// qs == NULL: nothing new -- drain() advances what is in flight with the same fused launches.
static int pipeline_launch(lcd_engine* h, const QSplitArgs* qs) {
    Tfidf& t = h->tfidf;
    lcd_engine::InFlight* f_reg = nullptr; lcd_engine::InFlight* f_res = nullptr; lcd_engine::InFlight* f_knn = nullptr;
    for (lcd_engine::InFlight& f : h->inflight) {
        if (f.stage == 2 && !f_reg) f_reg = &f;
        else if (f.stage == 1 && !f_res) f_res = &f;
        else if (f.stage == 0 && !f_knn) f_knn = &f;
    }
    TailLaunch tl_reg, tl_res; ScoreArgs sa; int score_wgs = 0;
    bool reg_like = false;
    HostLap lap(h);
    if (f_res) {
        // FIRST, before any launch argument is built: the reservation may move the word-indexed tables (they double when the keys run
        // out -- every ~3 000 frames at 150 new words per frame), and the registration / scoring arguments below hold pointers into
        // them.  The postings keys of the words that frame may create are reserved now (the batched check of older reservations waits until
        // launch A is enqueued: the registration that rides in it may still use some of those keys)
        { int rc = reserve_frame_words(h, f_res->a, &f_res->runs, false); if (rc) return rc; }
        f_res->reserved = true;
        tl_res.r = f_res->r;
        tl_res.r.new_ws = f_res->runs;
        refresh_vocab_ptrs(h, &tl_res.r);
        // (rows instead of postings keys in out_wslot: NULL is the decision loop's "knn_row already holds the word slot"; the registration translates)
        // (built-in: only while the stream creates words, like the shadow scores -- the gather leaves the decision loop's chain for the registration's, and once frames
        // revisit, the decision loop is short and the registration is what ends launch A: 13.8 -> 14.5 us in the revisit phase with the rows always on, r06_ab_notes.txt 10)
        f_res->slots_are_rows = h->popt.slots_from_rows && (h->popt.slots_from_rows >= 2 || h->est_new >= 16.0) && tl_res.r.row_wslot && tl_res.r.knn_row && tl_res.r.q <= 1024;
        if (f_res->slots_are_rows) { tl_res.r.row_wslot = nullptr; tl_res.r.slots_are_rows = 1; }
        tl_res.r.straight = (h->popt.decision_straight >= 2 || (h->popt.decision_straight == 1 && h->est_new >= 16.0)) ? 1 : 0;
        if (f_res->chained) fill_append(h, f_res->a, f_res->vseq, frame_appends(h, f_res->a), &tl_res.r, h->ring[f_res->set].d_applist.as<uint32_t>());
        // the pinned row-count mirror is a store to HOST memory, waited for at the end of the decision loop's chain: with "mirror_from_b" a
        // workgroup of launch B of this pair (which writes the frame's rows anyway) stores it instead
        if (h->popt.mirror_from_b && tl_res.r.ap.enabled && tl_res.r.ap.defer_rows) tl_res.r.ap.mirror_later = 1;
        resolve_launch_info(tl_res.r, pipe_block_size(), &tl_res.n_redo, &tl_res.shmem_resolve);
    }
    lap.lap(2);
    if (f_reg) {
        const lcd_frame_args& pa = f_reg->a;
        if (pa.sig_id != 0) LCD_HIP(h, t.register_dev(pa.sig_id, f_reg->r.out_wslot, pa.q, pa.q, pa.N, nullptr, false, &tl_reg));
        else LCD_HIP(h, t.query_dev(f_reg->r.out_wslot, pa.q, pa.N, nullptr, false, &tl_reg));
        if (f_reg->slots_are_rows) tl_reg.a.row_wslot = h->row_wslot.as<int32_t>();
        if (pa.d_likelihood) {
            LCD_HIP(h, t.score_args(pa.d_likelihood, nullptr, pipe_b_block_size(), &sa, &score_wgs));
            reg_like = true;
            h->likelihood_launches += 1;
        }
    }
    lap.lap(3);
    PipeKnn k;
    // shadow rows: f_res's new words are not rows when f_knn's filter runs (its decision loop rides in the same launch) -- the filter ranks f_res's
    // descriptors from the operand rows its query pre-split left, the re-rank keeps the ones the mask f_res's decision loop publishes names
    const bool sh_ok = f_knn && f_res && f_res->has_shadow && f_res->chained && f_knn->chained && tl_res.r.ap.enabled && tl_res.r.ap.defer_rows && tl_res.r.ap.is_f32_64 &&
                       h->popt.shadow_rows && !h->popt.cross_frames && h->popt.append_from_rerank;
    if (f_knn) { int rc = build_knn(h, *f_knn, &k, sh_ok ? f_res : nullptr); if (rc) return rc; h->knn_launches += 1; }
    if (f_res && f_res->has_shadow && tl_res.r.ap.enabled && tl_res.r.ap.defer_rows) tl_res.r.ap.mask_out = h->ring[f_res->set].d_newmask.as<uint32_t>();
    if (f_knn && f_res && tl_res.r.ap.enabled && tl_res.r.ap.defer_rows && tl_res.r.ap.is_f32_64 && h->popt.cross_frames) {
        // The rows f_res appends (its decision loop rides in this launch A) are descriptors of f_res, and f_knn's re-rank (this launch B)
        // must scan them exactly: extra distance tiles of launch A compute f_knn x f_res in the reference's arithmetic, the re-rank reads
        // its pending rows' distances there instead of staging the rows (the buffer was sized when f_knn was submitted: no reallocation here)
        const int ncols = f_res->a.q, ldx = (ncols + 63) / 64 * 64;
        lcd::DevBuf& xb = h->ring[f_knn->set].d_cross;
        if (ncols > 0 && f_knn->a.q > 0 && xb.cap >= (size_t)f_knn->a.q * ldx * 4) {
            k.cross = xb.as<float>(); k.cross_ld = ldx; k.cross_cols = tl_res.r.ap.descriptors; k.cross_ncols = ncols;
        }
    }
    lap.lap(4);
    bool prof = f_knn && h->prof_cap > 0 && h->prof_n < h->prof_cap;
    if (prof && h->prof_skip > 0) { h->prof_skip -= 1; prof = false; }     // ("profile_skip": not the first launches behind an idle queue)
    h->popt.f16 = h->f16();
    if (h->roctx_push) h->roctx_push("lcd:launch_A");
    const hipError_t ea__ = launch_frame_a(f_knn ? &k : nullptr, qs, f_res ? &tl_res : nullptr, f_reg ? &tl_reg : nullptr, h->stream,
                              prof ? h->prof_ev[2 * h->prof_n] : nullptr, prof ? h->prof_ev[2 * h->prof_n + 1] : nullptr, h->popt);
    if (h->roctx_pop) h->roctx_pop();
    LCD_HIP(h, ea__);
    if (prof) {
        h->prof_n += 1;
        if (h->f16())
            h->prof_kernel = knn_bf16_persistent(k.plan) ? "frame_a_kernel_p (persistent fp16 filter of frame t-1 + query pre-split of t + decision loop of t-2 + registration of t-3)"
                                                         : "frame_a_kernel (fp16 filter of frame t-1 + query pre-split of t + decision loop of t-2 + registration of t-3)";
        else
            h->prof_kernel = knn_bf16_persistent(k.plan) ? "frame_a_kernel_p (persistent bf16 filter of frame t-1 + query pre-split of t + decision loop of t-2 + registration of t-3)"
                                                         : "frame_a_kernel (bf16 filter of frame t-1 + query pre-split of t + decision loop of t-2 + registration of t-3)";
    }
    lap.lap(5);
    const bool prof2 = f_knn && reg_like && h->prof_likelihood && h->prof_cap > 0 && h->prof2_n < h->prof_cap;
    AppendRowsArgs app;
    if (f_res && tl_res.r.ap.enabled && tl_res.r.ap.defer_rows) { app.ap = tl_res.r.ap; app.new_ws = tl_res.r.new_ws; }
    if (h->roctx_push) h->roctx_push("lcd:launch_B");
    const hipError_t eb__ = launch_frame_b(f_knn ? &k : nullptr, reg_like ? &sa : nullptr, score_wgs, h->stream, prof2 ? h->prof2_ev[2 * h->prof2_n] : nullptr,
                                           prof2 ? h->prof2_ev[2 * h->prof2_n + 1] : nullptr, app.ap.enabled ? &app : nullptr, h->popt);
    if (h->roctx_pop) h->roctx_pop();
    LCD_HIP(h, eb__);
    lap.lap(6);
    LCD_HIP(h, t.flush_held_if_due());                               // (behind launch B: the rows it writes claim their postings keys there)
    if (prof2) { h->prof2_n += 1; h->prof2_kernel = "frame_b_kernel (re-rank of frame t-1 + scoring of frame t-3)"; }
    if (h->clean_armed && f_reg) {
        // cleanUnusedWords asked for behind an earlier frame: the retirements made in front of it rode with the registration of this
        // launch A, the reference counts are what Memory::preUpdate would see -- one kernel, between this launch B and the next launch A
        // f_res's decision loop ran in this launch A and its new words are rows since this launch B, but they get their first reference
        // with its registration, in the NEXT launch A: the clean stops at the count f_res started from (the counter it read, untouched
        // until the next decision loop writes it) -- addNewWords references a word as it creates it, cleanUnusedWords never sees one
        h->clean_armed = false;
        const int32_t* reg_cnt = f_res && f_res->chained ? h->d_vcnt.as<int32_t>() + (f_res->vseq & 1) : nullptr;
        int rc = h->enqueue_clean(reg_cnt); if (rc) return rc;        // (flushes what more than four retirements per frame left over)
    }
    if (f_res) f_res->stage = 2;
    if (f_knn) f_knn->stage = 1;
    if (f_reg) {                                                     // that frame is complete: its decision stage and the calls queued behind it
        lcd_engine::InFlight done = std::move(*f_reg);
        h->inflight.pop_front();                                     // (f_reg is the oldest entry: stages advance in order)
        if (done.a.d_likelihood) { int rc = hypothesis_stage(h, done.a); if (rc) return rc; }
        int rc = finish_frame_ops(h, done);
        if (rc) return rc;
    }
    lap.lap(7);
    return LCD_OK;
}

// Pipelined handle, matrix-core 2-NN (knn_mfma_kernels.hip, frame_a_kernel / frame_b_kernel): four frames are in flight.  The call for
// frame t pre-splits its queries and carries one stage of each of the three frames before it (pipeline_launch); what the frames still
// owe afterwards waits in h->inflight.
static int frame_pipelined(lcd_engine* h, const lcd_frame_args* a) {
    Tfidf& t = h->tfidf;
    const int q = a->q;
    HostLap lap0(h);
    h->host_prof[8] += 1;
    // validate against the index as it will be once the owed stages have run
    int64_t owed_slots = 0;
    for (const lcd_engine::InFlight& f : h->inflight) {
        if (f.a.sig_id == 0) continue;
        if (f.a.sig_id == a->sig_id) return h->fail(LCD_ERR_STATE, "lcd_frame_dev: signature already registered");
        owed_slots += 1;
    }
    if (a->sig_id != 0 && t.sig_slot.count(a->sig_id)) return h->fail(LCD_ERR_STATE, "lcd_frame_dev: signature already registered");
    const int64_t slots_after = t.n_slots + owed_slots + (a->sig_id != 0 ? 1 : 0);
    if (a->d_likelihood && a->likelihood_capacity < slots_after) return h->fail(LCD_ERR_INVALID, "lcd_frame_dev: likelihood buffer too small");
    if (h->unreconciled.size() >= (size_t)lcd_engine::VLOG / 2) { int rc = h->drain(); if (rc) return rc; }   // the append log is a ring
    // rows tombstoned by enqueued cleans keep their postings keys out of circulation until the host has caught up with the log: a stream
    // that never completes anything does so every 512 frames (three fused launch pairs, ~0.2 us per frame)
    if (h->rm_pending && ++h->frames_since_reconcile >= 512) { int rc = h->drain(); if (rc) return rc; }
    { int rc = id_window(h, *a); if (rc) return rc; }
    // rows appended on the device: the counters take over the row count, the buffers keep room for the words of the frames in flight
    const bool app = frame_appends(h, *a);
    if (app) { int rc = activate_dev_rows(h); if (rc) return rc; }
    const bool chained = h->vcnt_active;
    if (chained && h->h_vmirror && h->unreconciled.size() > 8) {
        // The launches are planned for an upper bound of the row count: what the newest FINISHED appender reported + q per younger
        // frame.  A caller that enqueues frames much faster than the device runs them would inflate that bound without limit (the
        // filter would scan mostly empty rows): such a caller waits here until the device is at most 8 frames behind.
        // (the wait spins for the few microseconds a frame takes, then yields; a stream that makes no progress for a long time --
        // a caller-provided one may legitimately sit behind an event -- is waited for with hipStreamSynchronize instead of failing)
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 0;; ++spins) {
            const uint32_t tag = (uint32_t)(*(volatile const unsigned long long*)h->h_vmirror >> 32);
            if ((uint32_t)h->vseq - tag <= 8u) break;
            if (spins > 4096) std::this_thread::yield();
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                LCD_HIP(h, hipStreamSynchronize(h->stream));
                break;
            }
        }
    }
    if (chained) { int rc = ensure_append_capacity(h, h->rows_ub() + 3 * (int64_t)q); if (rc) return rc; }
    const uint64_t vseq = h->vseq;
    const int set = (int)(h->frame_seq % lcd_engine::PIPE_SETS);
    lcd_engine::FrameScratch& sc = h->ring[set];
    const bool incremental = (a->flags & LCD_Q_INCREMENTAL) != 0;
    const bool together = incremental && (a->flags & LCD_Q_NEW_WORDS_COMPARED);
    const int ld = (q + 63) / 64 * 64, bw = ld / 32;
    lap0.lap(0);
    // ---- the frame's scratch set (what does not depend on the launch plan; the partial keys are sized when the filter is planned)
    LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_qsplit, knn_qsplit_bytes(q)));
    LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_qnorm, (size_t)ld * 4));
    LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_fail_list, (size_t)q * 4));
    LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_knn_row, (size_t)q * 2 * 4));
    LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_knn_word, (size_t)q * 2 * 4));
    LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_knn_dist, (size_t)q * 2 * 4));
    LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_out_wslot, (size_t)q * 4));
    LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_applist, (size_t)std::max(q, 512) * 4));   // (the re-rank reads 512 entries unconditionally)
    if (together) {
        LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_selfdist, (size_t)q * ld * 4));
        LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_bits, cand_bits_bytes(q, bw)));
    }
    if (chained && !h->inflight.empty() && h->dtype == LCD_F32 && h->kdim == 64 && (h->popt.cross_frames || h->popt.shadow_rows))   // (pipeline_launch: this frame x the frame before it)
        LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_cross, (size_t)q * ((h->inflight.back().a.q + 63) / 64 * 64) * 4));
    // shadow rows: a frame that appends on the device leaves its descriptors as operand-table rows too, for the filter of the frame behind it
    // (built-in: only while the stream creates words -- the scores cost launch A ~1 us (16 more workgroups, the pre-split's extra stores) and buy launch B
    // ~3.5 us per frame whose predecessor appended ~150 rows, nothing when it appended none; est_new is the decaying maximum of rows per appending frame
    // that the launch plans already keep.  "shadow_rows" = 2: always)
    const bool with_shadow = chained && app && h->popt.shadow_rows && h->dtype == LCD_F32 && h->kdim == 64 && q <= 4096 &&
                             (h->popt.shadow_rows >= 2 || h->est_new >= 16.0);
    if (with_shadow) {
        LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_shadow_bf, (size_t)ld * 256));
        LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_shadow_norm, (size_t)(ld + 1) * 8));
        LCD_HIP(h, ring_reserve(h, set, &lcd_engine::FrameScratch::d_newmask, (size_t)(2 * (ld / 32) + 4) * 4));
    }
    QSplitArgs qs;
    qs.queries = (const float*)a->d_descriptors; qs.nq = q; qs.qpad = ld; qs.qsplit = (uint4*)sc.d_qsplit.p; qs.qnorm = sc.d_qnorm.as<float>(); qs.n_wgs = 0; qs.f16 = h->f16();
    if (with_shadow) { qs.shadow_bf = sc.d_shadow_bf.as<uint32_t>(); qs.shadow_norm = sc.d_shadow_norm.as<float>(); qs.norm_max_bits = h->norm_max.as<uint32_t>(); }
    lap0.lap(1);
    // ---- what the frames in flight owe rides with this frame's launches
    { int rc = pipeline_launch(h, &qs); if (rc) return rc; }
    // ---- this frame's filter, re-rank, decision loop, registration and scoring are owed from here on
    lcd_engine::InFlight nf;
    nf.a = *a; nf.set = set; nf.stage = 0; nf.vseq = vseq; nf.chained = chained; nf.has_shadow = with_shadow;
    if (chained) { h->unreconciled.push_back(lcd_engine::DevAppend{vseq, a->first_new_word_id == LCD_NEW_WORD_IDS_AUTO ? -h->id_delta : a->first_new_word_id, q, app}); h->vseq += 1; }
    h->inflight.push_back(std::move(nf));
    h->frame_seq += 1;
    return LCD_OK;
}

static int frame_dev_body(lcd_engine* h, const lcd_frame_args* a);

int lcd_frame_dev(lcd_engine* h, const lcd_frame_args* a) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    FrameHostTimer timer__(h);
    lcd_engine::Range range__(h, "lcd_frame_dev");
    LCD_DEV_NODRAIN(h);
    return frame_dev_body(h, a);
    LCD_CATCH(h)
}

// where the host time of the pipelined lcd_frame_dev calls went so far (engine.h: host_prof): out9[0..7] ns per section, out9[8] calls.  Not part of lcd.h.
int lcd_debug_host_profile(const lcd_engine* h, int64_t* out9) {
    if (!h || !out9) return LCD_ERR_INVALID;
    for (int i = 0; i < 9; ++i) out9[i] = h->host_prof[i];
    return LCD_OK;
}

int lcd_slot_count(const lcd_engine* h, int64_t* n_slots) {
    if (!h || !n_slots) return LCD_ERR_INVALID;
    int64_t owed = 0;
    for (const lcd_engine::InFlight& f : h->inflight) if (f.a.sig_id != 0) owed += 1;
    *n_slots = h->tfidf.n_slots + owed;
    return LCD_OK;
}

int lcd_frame_host(lcd_engine* h, const lcd_frame_host_args* a) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    lcd_engine::Range range__(h, "lcd_frame_host");
    LCD_DEV(h);                                                      // completes what a pipelined handle owes
    if (!a || a->struct_size != (int32_t)sizeof(lcd_frame_host_args)) return h->fail(LCD_ERR_INVALID, "lcd_frame_host: bad argument block");
    const int q = a->q;
    if (q <= 0 || q > 8192 || !a->descriptors || !a->word_ids) return h->fail(LCD_ERR_INVALID, "lcd_frame_host: bad argument");
    const size_t src_row = (size_t)h->dim * (h->dtype == LCD_F32 ? 4 : 1);
    if (src_row != (size_t)h->row_bytes) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_frame_host: rows of this size are padded on the device (use lcd_quantize)");
    const int64_t slots_after = h->tfidf.n_slots + (a->sig_id != 0 ? 1 : 0);
    if (a->likelihood && a->likelihood_capacity < slots_after) return h->fail(LCD_ERR_INVALID, "lcd_frame_host: likelihood buffer too small");
    // descriptors: host -> pinned staging -> device, on the engine's stream (the one synchronisation at the end frees the staging)
    const size_t dbytes = (size_t)q * h->row_bytes;
    LCD_HIP(h, h->h_frame_in.reserve(dbytes));
    std::memcpy(h->h_frame_in.p, a->descriptors, dbytes);
    LCD_HIP(h, dreserve(h, h->d_frame_desc, std::max<size_t>(dbytes, 16)));
    LCD_HIP(h, dreserve(h, h->d_frame_words, (size_t)q * 4));
    if (a->likelihood) LCD_HIP(h, dreserve(h, h->d_frame_like, (size_t)std::max<int64_t>(slots_after, 1) * 4));
    LCD_HIP(h, hipMemcpyAsync(h->d_frame_desc.p, h->h_frame_in.p, dbytes, hipMemcpyHostToDevice, h->stream));
    // from here on a copy out of / into the pinned staging may be in flight: a failure synchronises before it returns (the next call --
    // the mirror falls back to the call-by-call path on the same engine straight away -- reuses h_frame_in / h_frame_out)
    auto bail = [&](int rc) { (void)hipStreamSynchronize(h->stream); return rc; };
#define LCD_HIP_B(h, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return bail((h)->hip_fail(e__, #call)); } while (0)
    lcd_frame_args fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.struct_size = (int32_t)sizeof(fa); fa.q = q; fa.d_descriptors = h->d_frame_desc.p; fa.flags = a->flags; fa.nndr_ratio = a->nndr_ratio;
    fa.sig_id = a->sig_id; fa.first_new_word_id = a->first_new_word_id; fa.N = a->N; fa.append_new_words = a->append_new_words;
    fa.d_word_ids = h->d_frame_words.as<int32_t>();
    if (a->likelihood) { fa.d_likelihood = h->d_frame_like.as<float>(); fa.likelihood_capacity = (int64_t)(h->d_frame_like.cap / 4); }
    { int rc = frame_dev_body(h, &fa); if (rc) return bail(rc); }
    { int rc = h->drain(false); if (rc) return bail(rc); }            // a pipelined handle: the frame's stages stand-alone (the row mirror is not needed here)
    const size_t wbytes = (size_t)q * 4, lbytes = a->likelihood ? (size_t)slots_after * 4 : 0;
    LCD_HIP_B(h, h->h_frame_out.reserve(wbytes + lbytes + 16));
    LCD_HIP_B(h, hipMemcpyAsync(h->h_frame_out.p, h->d_frame_words.p, wbytes, hipMemcpyDeviceToHost, h->stream));
    if (lbytes) LCD_HIP_B(h, hipMemcpyAsync((char*)h->h_frame_out.p + wbytes, h->d_frame_like.p, lbytes, hipMemcpyDeviceToHost, h->stream));
    LCD_HIP(h, hipStreamSynchronize(h->stream));
#undef LCD_HIP_B
    std::memcpy(a->word_ids, h->h_frame_out.p, wbytes);
    if (lbytes) std::memcpy(a->likelihood, (const char*)h->h_frame_out.p + wbytes, lbytes);
    if (a->n_slots) *a->n_slots = slots_after;
    // The word ids are here and the stream is idle: the rows this frame appended on the device are known without asking the device's log
    // (the k-th new word carries the code -(k + 1)), so the host's row mirror catches up now -- the next call finds nothing to reconcile
    // (a synchronisation and two small blocking copies less per frame).  Only when this frame is the one unreconciled appender.
    if (h->unreconciled.size() == 1 && h->unreconciled.front().enabled && h->unreconciled.front().own_world == 0 && !h->rm_pending &&
        h->unreconciled.front().first_id == a->first_new_word_id && h->h_vmirror) {
        int n_new = 0;
        for (int i = 0; i < q; ++i) n_new = std::max(n_new, -a->word_ids[i]);
        const unsigned long long v = *(volatile const unsigned long long*)h->h_vmirror;   // what the appender reported: (tag << 32) | rows
        if ((uint32_t)(v >> 32) == (uint32_t)(h->unreconciled.front().seq + 1) && (int64_t)(uint32_t)v == h->n_rows + n_new) {
            for (int k = 0; k < n_new; ++k) h->mirror_push_row(a->first_new_word_id + k, h->n_rows + k);
            h->n_rows += n_new;
            h->n_live += n_new;
            h->unreconciled.clear();
            h->frames_since_reconcile = 0;                               // (what reconcile() leaves: nothing is owed to the mirror)
        }
    }
    return LCD_OK;
    LCD_CATCH(h)
}

static int frame_dev_body(lcd_engine* h, const lcd_frame_args* a) {
    if (!a || a->struct_size != (int32_t)sizeof(lcd_frame_args)) return h->fail(LCD_ERR_INVALID, "lcd_frame_dev: bad argument block");
    const int q = a->q;
    if (q <= 0 || q > 8192 || !a->d_descriptors || !a->d_word_ids) return h->fail(LCD_ERR_INVALID, "lcd_frame_dev: bad argument");
    if (((uintptr_t)a->d_descriptors & 15u) != 0) return h->fail(LCD_ERR_INVALID, "lcd_frame_dev: d_descriptors must be 16-byte aligned");
    if ((a->d_hypothesis || a->d_adjusted || a->d_posterior || a->d_bayes) && !a->d_likelihood)
        return h->fail(LCD_ERR_INVALID, "lcd_frame_dev: the hypothesis needs d_likelihood");
    if ((a->d_posterior || a->d_bayes) && !h->bayes.configured) return h->fail(LCD_ERR_STATE, "lcd_frame_dev: lcd_bayes_configure first");
    if (h->pipeline && q <= 4096 && h->bf_family() && knn_mfma_supported(h->dtype, h->kdim) && h->n_live >= 2 && h->n_rows >= 256)
        return frame_pipelined(h, a);
    { int rc = id_window(h, *a); if (rc) return rc; }
    const bool app = frame_appends(h, *a);
    // A stream of appending frames on a plain handle with the exact scan (ORB: config 3) does not wait for the device between frames:
    // the host's row mirror lags (as on a pipelined handle), the scan is planned for an upper bound of the row count.  Anything else
    // completes what is owed and brings the mirror up to date first.
    const bool lazy = app && h->inflight.empty() && h->vcnt_active && h->n_live >= 2 && !(h->knn_mode != 0 && knn_mfma_supported(h->dtype, h->kdim)) &&
                      h->unreconciled.size() < (size_t)lcd_engine::VLOG / 2 && !(h->rm_pending && h->frames_since_reconcile >= 512);
    if (!lazy) { int rc = h->drain(); if (rc) return rc; }         // (also brings the host's row mirror up to date)
    else {
        if (h->rm_pending) h->frames_since_reconcile += 1;
        if (h->h_vmirror && h->unreconciled.size() > 8) {           // the bound grows by q per unreported frame: stay within 8 frames of the device
            const auto t0 = std::chrono::steady_clock::now();
            for (int spins = 0;; ++spins) {
                const uint32_t tag = (uint32_t)(*(volatile const unsigned long long*)h->h_vmirror >> 32);
                if ((uint32_t)h->vseq - tag <= 8u) break;
                if (spins > 4096) std::this_thread::yield();
                if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { LCD_HIP(h, hipStreamSynchronize(h->stream)); break; }
            }
        }
    }
    LCD_HIP(h, dreserve(h, h->d_out_wslot, (size_t)q * 4));
    if (app) {
        { int rc = activate_dev_rows(h); if (rc) return rc; }
        { int rc = ensure_append_capacity(h, h->rows_ub() + 2 * (int64_t)q); if (rc) return rc; }
    }
    // 2-NN + same-frame distances, then ONE single-workgroup launch: decision loop -> pending retirements -> registration / idf
    ResolveArgs r;
    int rc = prepare_resolve(h, a->d_descriptors, q, a->flags, a->nndr_ratio, a->d_word_ids, h->d_out_wslot.as<int32_t>(), &r, true,
                             lazy ? h->rows_ub() : -1);
    if (rc) return rc;
    if (h->d_fail_count.p) { r.fail_count = h->d_fail_count.as<int32_t>(); h->fail_count_clean = true; }   // the tail resets the counters
    const uint64_t vseq = h->vseq;
    if (app) { h->unreconciled.push_back(lcd_engine::DevAppend{vseq, a->first_new_word_id == LCD_NEW_WORD_IDS_AUTO ? -h->id_delta : a->first_new_word_id, q, true}); h->vseq += 1; }
    return frame_stage_s(h, *a, r, app, vseq);
}

int lcd_knn2_dev(lcd_engine* h, const void* d_queries, int q, int32_t* d_word_ids, float* d_dist) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    LCD_JOIN_K(h);
    if (q <= 0 || !d_queries || !d_word_ids || !d_dist) return h->fail(LCD_ERR_INVALID, "lcd_knn2_dev: bad argument");
    if (((uintptr_t)d_queries & 15u) != 0) return h->fail(LCD_ERR_INVALID, "lcd_knn2_dev: d_queries must be 16-byte aligned");
    LCD_HIP(h, dreserve(h, h->d_knn_row, (size_t)q * 2 * 4));
    return run_knn2_raw(h, d_queries, q, h->vocab.p, h->row_id.as<int32_t>(), h->n_rows, true, h->d_knn_row.as<int32_t>(), d_word_ids, d_dist);
    LCD_CATCH(h)
}

// ---------------------------------------------------------------------------------------------------------------- Bayes filter
int lcd_bayes_configure(lcd_engine* h, const double* prediction_lc, int n_values, float virtual_place_prior) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (!prediction_lc) return h->fail(LCD_ERR_INVALID, "lcd_bayes_configure: bad argument");
    if (h->bayes.configure(prediction_lc, n_values, virtual_place_prior) != hipSuccess)
        return h->fail(LCD_ERR_INVALID, "lcd_bayes_configure: 2..32 values in [0, 1] and a prior in [0, 1] expected");   // the reference logs UERROR (:83, :103)
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_bayes_reset(lcd_engine* h) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    LCD_HIP(h, h->bayes.reset());
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_bayes_set_neighbors(lcd_engine* h, int n_sigs, const int32_t* sig_ids, const int64_t* offsets, const int32_t* nbr_sig_ids,
                            const int32_t* nbr_margins) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
#ifdef LCD_DEBUG_TIMING
    static double dbg[4]; static int dbg_n;
    const auto d0 = std::chrono::steady_clock::now();
#endif
    LCD_DEV_NODRAIN(h);                                       // a pipelined handle queues the lists behind the index stage it still owes
#ifdef LCD_DEBUG_TIMING
    const auto d1 = std::chrono::steady_clock::now();
    struct Rep { std::chrono::steady_clock::time_point a, b; double* d; int* n; ~Rep() { auto c = std::chrono::steady_clock::now();
        d[0] += std::chrono::duration<double, std::micro>(b - a).count(); d[1] += std::chrono::duration<double, std::micro>(c - b).count();
        if (++*n % 100 == 0) { fprintf(stderr, "[set_neighbors] setdevice %.2f us, rest %.2f us (avg of 100)\n", d[0] / 100, d[1] / 100); d[0] = d[1] = 0; } } } rep__{d0, d1, dbg, &dbg_n};
#endif
    if (n_sigs < 0 || (n_sigs > 0 && (!sig_ids || !offsets))) return h->fail(LCD_ERR_INVALID, "lcd_bayes_set_neighbors: bad argument");
    if (!h->bayes.configured) return h->fail(LCD_ERR_STATE, "lcd_bayes_set_neighbors: lcd_bayes_configure first");
    if (n_sigs == 0) return LCD_OK;
    if (offsets[n_sigs] > offsets[0] && (!nbr_sig_ids || !nbr_margins)) return h->fail(LCD_ERR_INVALID, "lcd_bayes_set_neighbors: bad argument");
    Tfidf& t = h->tfidf;
    if (t.n_slots >= (1ll << BAYES_SLOT_BITS)) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_bayes_set_neighbors: at most 2^27 signature slots");
    const int max_margin = h->bayes.prm.n_lc - 2;
    std::vector<int32_t> triples, restart;                    // restart: the slots of the listed signatures (their lists start over)
    std::unordered_map<uint64_t, int32_t> seen;
    seen.reserve((size_t)(offsets[n_sigs] - offsets[0]) * 2 + 16);
    // the signatures of frames whose registration is still owed have no slots yet: they will get the next ones, in frame order
    auto gone = [&](int32_t id) {
        for (const lcd_engine::InFlight& f : h->inflight)
            if (std::find(f.retire_after.begin(), f.retire_after.end(), id) != f.retire_after.end()) return true;
        return false;
    };
    auto slot_of = [&](int32_t id) -> int64_t {
        int64_t k = 0;
        for (const lcd_engine::InFlight& f : h->inflight) {
            if (f.a.sig_id == 0) continue;
            if (f.a.sig_id == id) return gone(id) ? -1 : t.n_slots + k;
            k += 1;
        }
        auto it = t.sig_slot.find(id);
        if (it == t.sig_slot.end() || gone(id)) return -1;
        return it->second;
    };
    for (int i = 0; i < n_sigs; ++i) {
        const int64_t a = slot_of(sig_ids[i]);
        if (offsets[i + 1] < offsets[i]) return h->fail(LCD_ERR_INVALID, "lcd_bayes_set_neighbors: offsets must not decrease");
        // a signature the engine does not hold -- one without a single word never got references, hence no slot (a featureless
        // frame, Rtabmap.cpp:2234 "bad signature") -- has no likelihood and no posterior: its list is skipped, the reference's filter
        // carries such signatures along with probability 0 as well
        if (a < 0) continue;
        restart.push_back((int32_t)a);
        for (int64_t e = offsets[i]; e < offsets[i + 1]; ++e) {
            const int32_t m = nbr_margins[e];
            if (m < 0 || m > max_margin) return h->fail(LCD_ERR_INVALID, "lcd_bayes_set_neighbors: margin outside the prediction's levels");   // UASSERT :263
            if (nbr_sig_ids[e] < 0) continue;                 // "if(iter->first>=0)" (:254)
            const int64_t b = slot_of(nbr_sig_ids[e]);
            if (b < 0) continue;                              // not in memory: it can not be in a likelihood
            const uint64_t key = ((uint64_t)std::min(a, b) << 32) | (uint64_t)std::max(a, b);
            auto st = seen.find(key);
            if (st != seen.end()) { triples[(size_t)st->second * 3 + 2] = m; continue; }   // listed from both ends: the later margin stays
            seen.emplace(key, (int32_t)(triples.size() / 3));
            triples.push_back((int32_t)std::min(a, b)); triples.push_back((int32_t)std::max(a, b)); triples.push_back(m);
        }
    }
    if (!h->inflight.empty()) { h->inflight.back().links_after.push_back(lcd_engine::DeferredLink{std::move(triples), std::move(restart)}); return LCD_OK; }
    LCD_HIP(h, h->bayes.ensure(std::max<int64_t>(t.n_slots, 1)));
    const hipError_t le = h->bayes.link(triples, restart);
    if (le == hipErrorInvalidValue) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_bayes_set_neighbors: a neighbour list longer than 8192 entries");
    LCD_HIP(h, le);
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_bayes_update_dev(lcd_engine* h, const float* d_adjusted, int exclude_recent, float* d_posterior, lcd_bayes_result* d_result) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (!d_adjusted) return h->fail(LCD_ERR_INVALID, "lcd_bayes_update_dev: bad argument");
    if (!h->bayes.configured) return h->fail(LCD_ERR_STATE, "lcd_bayes_update_dev: lcd_bayes_configure first");
    Tfidf& t = h->tfidf;
    LCD_HIP(h, t.flush_retire());                             // slot_sig must show the retirements asked for so far
    const long long n_cons = (long long)t.n_slots - std::max(exclude_recent, 0);
    DecideArgs d;
    d.adj_in = d_adjusted; d.bayes = true; d.d_posterior = d_posterior; d.d_bayes = (BayesOut*)d_result;
    LCD_HIP(h, h->bayes.decide(d, t.slot_sig.as<int32_t>(), t.n_slots, n_cons));
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_bayes_update(lcd_engine* h, const int32_t* sig_ids, const float* adjusted, int n, lcd_bayes_result* result) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (n < 1 || !sig_ids || !adjusted) return h->fail(LCD_ERR_INVALID, "lcd_bayes_update: bad argument");
    if (!h->bayes.configured) return h->fail(LCD_ERR_STATE, "lcd_bayes_update: lcd_bayes_configure first");
    if (sig_ids[0] != -1) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_bayes_update: the likelihood must start with the virtual place (id -1)");
    Tfidf& t = h->tfidf;
    LCD_HIP(h, t.flush_retire());
    // scatter the map into slot order; its keys must be the registered signatures without the most recently registered ones
    const size_t bytes = ((size_t)t.n_slots + 1) * 4;
    LCD_HIP(h, h->h_in.reserve(bytes));
    float* adj = (float*)h->h_in.p;
    std::memset(adj, 0, bytes);
    adj[0] = adjusted[0];
    int64_t n_cons = 0, n_reg = 0;
    for (int i = 1; i < n; ++i) {
        if (sig_ids[i] <= sig_ids[i - 1]) return h->fail(LCD_ERR_INVALID, "lcd_bayes_update: ids must ascend (std::map order)");
        auto it = t.sig_slot.find(sig_ids[i]);
        if (it == t.sig_slot.end()) continue;                    // a signature without words holds no slot: no likelihood, posterior 0 (see set_neighbors)
        adj[(size_t)it->second + 1] = adjusted[i];
        n_cons = std::max<int64_t>(n_cons, it->second + 1);
        n_reg += 1;
    }
    int64_t live_below = 0;
    if (n_cons == t.n_slots) live_below = t.live_sigs;             // the usual case without a short-term memory: no walk over the table
    else for (const auto& kv : t.sig_slot) live_below += kv.second < n_cons ? 1 : 0;
    if (live_below != n_reg)
        return h->fail(LCD_ERR_UNSUPPORTED, "lcd_bayes_update: the likelihood must hold every registered signature up to its newest one "
                                            "(the working memory without the short-term memory)");
    LCD_HIP(h, dreserve(h, h->d_adj_scratch, bytes));
    LCD_HIP(h, hipMemcpyAsync(h->d_adj_scratch.p, adj, bytes, hipMemcpyHostToDevice, h->stream));
    DecideArgs d;
    d.adj_in = h->d_adj_scratch.as<float>(); d.bayes = true; d.d_bayes = (BayesOut*)h->d_hyp_scratch.p;
    LCD_HIP(h, h->bayes.decide(d, t.slot_sig.as<int32_t>(), t.n_slots, n_cons));
    lcd_bayes_result r;
    { int rc = download(h, &r, h->d_hyp_scratch.p, sizeof(r), h->h_out); if (rc) return rc; }   // (synchronises: h_in is free again)
    if (result) *result = r;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_bayes_posterior(lcd_engine* h, const int32_t* sig_ids, int n, float* out) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (n < 0 || (n > 0 && (!sig_ids || !out))) return h->fail(LCD_ERR_INVALID, "lcd_bayes_posterior: bad argument");
    if (n == 0) return LCD_OK;
    Tfidf& t = h->tfidf;
    std::vector<float> all;
    std::vector<uint8_t> in;
    const int64_t have = std::min<int64_t>(t.n_slots, h->bayes.cap);
    LCD_HIP(h, h->bayes.read_posterior(have, &all, &in));
    for (int i = 0; i < n; ++i) {
        float v = 0.0f;
        if (sig_ids[i] == -1) v = in[0] ? all[0] : 0.0f;
        else {
            auto it = t.sig_slot.find(sig_ids[i]);
            if (it != t.sig_slot.end() && it->second < have && in[(size_t)it->second + 1]) v = all[(size_t)it->second + 1];
        }
        out[i] = v;
    }
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_shard_knn2_dev(lcd_engine* h, const void* d_descriptors, int q, lcd_shard_cand* d_cand) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV_NODRAIN(h);
    { int rc = drain_keep_rows_lazy(h); if (rc) return rc; }
    LCD_JOIN_K(h);
    if (q <= 0 || !d_descriptors || !d_cand) return h->fail(LCD_ERR_INVALID, "lcd_shard_knn2_dev: bad argument");
    if (((uintptr_t)d_descriptors & 15u) != 0) return h->fail(LCD_ERR_INVALID, "lcd_shard_knn2_dev: d_descriptors must be 16-byte aligned");
    const int64_t rows_scan = h->rows_ub();                           // == n_rows unless this rank appended on the device since the mirror last caught up
    if (rows_scan >= (1 << 26)) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_shard_knn2_dev: a shard holds at most 2^26 - 1 rows (merge key: 26-bit row, 6-bit rank)");
    LCD_HIP(h, dreserve(h, h->d_knn_row, (size_t)q * 2 * 4));
    LCD_HIP(h, dreserve(h, h->d_knn_word, (size_t)q * 2 * 4));
    LCD_HIP(h, dreserve(h, h->d_knn_dist, (size_t)q * 2 * 4));
    // the candidate records are the work of the search's last launch: the exact redo's when the search has one (the matrix-core filter; one launch
    // less per frame and rank), a launch of their own behind the exact scan otherwise.  Either way the search's counters are left zeroed.
    ShardPackArgs pk;
    pk.knn_row = h->d_knn_row.as<int32_t>(); pk.knn_word = h->d_knn_word.as<int32_t>(); pk.knn_dist = h->d_knn_dist.as<float>();
    pk.row_wslot = h->row_wslot.as<int32_t>(); pk.q2 = 2 * q; pk.out = reinterpret_cast<ShardCand*>(d_cand);
    bool packed = false;
    // the same-frame distance matrix needs nothing but the descriptors: it rides in the filter's launch (extra workgroups, as in the single-GPU
    // frame) instead of waiting behind the all-gather.  No bit rows yet (bits == nullptr): their thresholds are the MERGED second neighbours'.
    CandBits sd;
    h->shard_sd_desc = nullptr;
    bool with_sd = h->knn_mode != 0 && h->bf_family() && knn_mfma_supported(h->dtype, h->kdim) && rows_scan >= 256;
    if (with_sd) {
        // ... unless the compute units its tiles take turn a one-strip-per-workgroup filter into the persistent one (between ~56 000 and ~65 000
        // rows at 500 descriptors: the filter then costs 4 us more, what the ride saves behind the all-gather; r06_call55)
        MfmaPlan p0 = knn_bf16_plan(q, (int)rows_scan, 0), p1 = knn_bf16_plan(q, (int)rows_scan, knn_selfdist_wgs(q));
        p0.filter_units = p1.filter_units = h->filter_units;
        with_sd = knn_bf16_persistent(p0) == knn_bf16_persistent(p1);
    }
    if (with_sd) {
        const int ld = (q + 63) / 64 * 64;
        LCD_HIP(h, dreserve(h, h->d_shard_selfdist, (size_t)q * ld * 4));
        sd.selfdist = h->d_shard_selfdist.as<float>(); sd.ld = ld; sd.nq = q;
    }
    int rc = run_knn2_raw(h, d_descriptors, q, h->vocab.p, h->row_id.as<int32_t>(), rows_scan, true, h->d_knn_row.as<int32_t>(),
                          h->d_knn_word.as<int32_t>(), h->d_knn_dist.as<float>(), with_sd ? &sd : nullptr, nullptr, &pk, &packed);
    if (rc) return rc;
    if (with_sd) { h->shard_sd_desc = d_descriptors; h->shard_sd_q = q; }
    if (!packed) LCD_HIP(h, launch_shard_pack(pk.knn_row, pk.knn_word, pk.knn_dist, pk.row_wslot, q, d_cand, h->stream, h->d_fail_count.as<int32_t>()));
    h->fail_count_clean = true;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_shard_frame_dev(lcd_engine* h, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id,
                        int32_t first_new_word_id, float N, int rank, int world, const lcd_shard_cand* d_all_cand, int64_t total_live_rows,
                        int32_t* d_word_ids, int64_t* d_lfix, int64_t lfix_capacity) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV_NODRAIN(h);
    { int rc = drain_keep_rows_lazy(h); if (rc) return rc; }
    LCD_JOIN_K(h);
    const bool matrix_left = h->shard_sd_desc != nullptr && h->shard_sd_desc == d_descriptors && h->shard_sd_q == q;   // by this frame's search
    h->shard_sd_desc = nullptr;                                       // (one frame call per search: whatever happens below, it is used up)
    if (q <= 0 || q > 8192 || !d_descriptors || !d_all_cand || !d_word_ids || world < 1 || world > 64 || rank < 0 || rank >= world)
        return h->fail(LCD_ERR_INVALID, "lcd_shard_frame_dev: bad argument");
    Tfidf& t = h->tfidf;
    if (sig_id != 0 && t.sig_slot.count(sig_id)) return h->fail(LCD_ERR_STATE, "lcd_shard_frame_dev: signature already registered");
    const int64_t slots_after = t.n_slots + (sig_id != 0 ? 1 : 0);
    if (d_lfix && lfix_capacity < slots_after) return h->fail(LCD_ERR_INVALID, "lcd_shard_frame_dev: lfix buffer too small");
    LCD_HIP(h, dreserve(h, h->d_knn_row, (size_t)q * 2 * 4));
    LCD_HIP(h, dreserve(h, h->d_knn_word, (size_t)q * 2 * 4));
    LCD_HIP(h, dreserve(h, h->d_knn_dist, (size_t)q * 2 * 4));
    // global 2-NN from the gathered per-rank candidates; d_knn_row holds the postings keys of the neighbours this rank owns
    // ONE flag decides both the owner of a new word and the order of equal distances in the merge (block-cyclic owners need ties by word
    // id; ties by (rank, row) need every new word on the last rank): a block without a first id is no growth policy, and a frame whose
    // new ids would lie in front of the policy's origin has no owner rule at all -- refused rather than guessed
    const bool cyclic = h->shard_block > 0 && h->shard_first > 0;
    // the frame's new words become rows of this rank's shard on the device (shard_append): they need postings keys and an owner whether or
    // not the frame is registered as a signature (a query-only frame: the single-GPU path reserves keys whenever frame_appends() holds)
    const bool dev_append = h->shard_append && (flags & LCD_Q_INCREMENTAL) && first_new_word_id > 0 && h->row_bytes == h->dim * (h->dtype == LCD_F32 ? 4 : 1);
    if (cyclic && (sig_id != 0 || dev_append) && first_new_word_id > 0 && (flags & LCD_Q_INCREMENTAL) && first_new_word_id < h->shard_first)
        return h->fail(LCD_ERR_INVALID, "lcd_shard_frame_dev: first_new_word_id lies in front of shard_growth_first");
    const int have_index = total_live_rows >= 2 ? 1 : 0;
    const bool incremental = (flags & LCD_Q_INCREMENTAL) != 0;
    const bool together = incremental && (flags & LCD_Q_NEW_WORDS_COMPARED);
    const int ld = (q + 63) / 64 * 64, bw = ld / 32;
    // the merge rides at the head of the same-frame distance launch when the frame has one that can carry it (one launch less per frame and rank)
    const bool have_matrix = together && matrix_left;
    const bool merge_in_selfdist = together && !have_matrix && selfdist_can_merge(h->dtype, h->kdim);
    if (have_matrix) {
        LCD_HIP(h, dreserve(h, h->d_bits, (size_t)q * bw * 4));
        ShardMergeJob mj;
        mj.cand = reinterpret_cast<const ShardCand*>(d_all_cand); mj.world = world; mj.rank = rank; mj.by_word = cyclic ? 1 : 0;
        mj.out_word = h->d_knn_word.as<int32_t>(); mj.out_dist = h->d_knn_dist.as<float>(); mj.out_wslot = h->d_knn_row.as<int32_t>();
        CandBits cb;
        cb.selfdist = h->d_shard_selfdist.as<float>(); cb.ld = ld; cb.nq = q; cb.bits = h->d_bits.as<uint32_t>(); cb.bw = bw; cb.have_index = have_index;
        LCD_HIP(h, launch_shard_merge_bits(mj, cb, h->stream));
    } else if (!merge_in_selfdist)
        LCD_HIP(h, launch_shard_merge(d_all_cand, world, rank, q, h->d_knn_word.as<int32_t>(), h->d_knn_dist.as<float>(),
                                      h->d_knn_row.as<int32_t>(), h->stream, cyclic));
    if (together && !have_matrix) {
        LCD_HIP(h, dreserve(h, h->d_selfdist, (size_t)q * ld * 4));
        LCD_HIP(h, dreserve(h, h->d_bits, (size_t)q * bw * 4));
        ShardMergeJob mj;
        mj.cand = reinterpret_cast<const ShardCand*>(d_all_cand); mj.world = world; mj.rank = rank; mj.by_word = cyclic ? 1 : 0;
        mj.out_word = h->d_knn_word.as<int32_t>(); mj.out_dist = h->d_knn_dist.as<float>(); mj.out_wslot = h->d_knn_row.as<int32_t>();
        LCD_HIP(h, launch_selfdist(h->dtype, h->kdim, d_descriptors, q, h->d_selfdist.as<float>(), ld, h->stream, have_index,
                                   h->d_knn_word.as<int32_t>(), h->d_knn_dist.as<float>(), h->d_bits.as<uint32_t>(), bw,
                                   merge_in_selfdist ? &mj : nullptr));
    }
    LCD_HIP(h, dreserve(h, h->d_out_wslot, (size_t)q * 4));
    // new words: every rank reserves the same keys (identical call sequence => identical numbering); only the LAST rank, which
    // will hold their rows, references them
    WsRuns new_ws;
    if ((sig_id != 0 || dev_append) && first_new_word_id > 0 && incremental) {
        hipError_t e = t.reserve_new_words(first_new_word_id, q, &new_ws);
        if (e == hipErrorInvalidValue) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_shard_frame_dev: word ids must be below 2^28");
        if (e != hipSuccess) return h->hip_fail(e, "reserve_new_words");
        if (cyclic) {                                      // block-cyclic ownership of the frame's new words (balanced growth)
            new_ws.own_id0 = first_new_word_id; new_ws.own_first = h->shard_first; new_ws.own_block = h->shard_block;
            new_ws.own_rank = rank; new_ws.own_world = world;
        } else if (rank != world - 1) new_ws.n = 0;
    }
    const int rflags = (incremental ? LCD_Q_INCREMENTAL : 0) | (together ? LCD_Q_NEW_WORDS_COMPARED : 0);
    LCD_HIP(h, launch_resolve(q, rflags, nndr_ratio, have_index, h->d_knn_word.as<int32_t>(), h->d_knn_dist.as<float>(),
                              together ? (have_matrix ? h->d_shard_selfdist.as<float>() : h->d_selfdist.as<float>()) : nullptr, ld,
                              together ? h->d_bits.as<uint32_t>() : nullptr, bw,
                              d_word_ids, h->d_n_new.as<int32_t>(), h->stream, h->d_knn_row.as<int32_t>(), nullptr,
                              h->d_out_wslot.as<int32_t>(), &new_ws));
    ShardAppendJob app;
    if (dev_append) {
        // VWDictionary::update()'s append branch, this rank's share, on the device: the words the frame created that this rank owns become
        // rows of its shard before the next frame is searched (lcd_shard_knn2_dev catches the host's row mirror up: one synchronisation,
        // no lcd_vocab_append, nothing read back by the caller)
        { int rc = activate_dev_rows(h); if (rc) return rc; }
        { int rc = ensure_append_capacity(h, h->rows_ub() + (int64_t)q); if (rc) return rc; }
        lcd_frame_args fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.d_descriptors = d_descriptors; fa.first_new_word_id = first_new_word_id; fa.q = q;
        ResolveArgs ra;
        const uint64_t vseq = h->vseq;
        fill_append(h, fa, vseq, true, &ra);
        // ... as a second workgroup of the registration's launch (the two chains need nothing of each other: one launch less per frame and rank)
        app.ap = ra.ap; app.new_ws = new_ws;               // (n == 0 on a rank that owns nothing in last-rank mode: it appends nothing either)
        app.codes = d_word_ids; app.q = q; app.rank = rank; app.world = world;
        app.own_first = cyclic ? h->shard_first : 0; app.own_block = cyclic ? h->shard_block : 0;
    }
    if (sig_id != 0) LCD_HIP(h, t.register_dev(sig_id, h->d_out_wslot.as<int32_t>(), q, q, N, nullptr, false, nullptr, nullptr, dev_append ? &app : nullptr));
    else LCD_HIP(h, t.query_dev(h->d_out_wslot.as<int32_t>(), q, N, nullptr, false, nullptr, nullptr, dev_append ? &app : nullptr));
    if (dev_append) {                                      // (enqueued: the host's record of it)
        lcd_engine::DevAppend da{h->vseq, first_new_word_id, q, true};
        da.own_world = world; da.own_rank = rank; da.own_first = app.own_first; da.own_block = app.own_block;
        h->unreconciled.push_back(da);
        h->vseq += 1;
    }
    if (d_lfix) {
        LCD_HIP(h, t.score_fix((long long*)d_lfix));       // every slot written: no zero-fill needed
        h->likelihood_launches += 1;
    }
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_finalize_dev(lcd_engine* h, int64_t* d_lfix, int64_t n, float* d_likelihood) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV_NODRAIN(h);
    { int rc = drain_keep_rows_lazy(h); if (rc) return rc; }
    if (n < 0 || (n > 0 && (!d_lfix || !d_likelihood))) return h->fail(LCD_ERR_INVALID, "lcd_finalize_dev: bad argument");
    if (n > h->tfidf.n_slots) return h->fail(LCD_ERR_INVALID, "lcd_finalize_dev: more entries than signature slots");
    LCD_HIP(h, h->tfidf.finalize((const long long*)d_lfix, (long long)n, d_likelihood));
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_slots_dev(lcd_engine* h, const int32_t** d_slot_sig, int64_t* n_slots) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    { int rc = drain_keep_rows_lazy(h); if (rc) return rc; }           // (the slot table does not depend on the row mirror)
    if (d_slot_sig) *d_slot_sig = h->tfidf.slot_sig.as<int32_t>();
    if (n_slots) *n_slots = h->tfidf.n_slots;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_profile_begin(lcd_engine* h, int max_samples) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (max_samples <= 0 || max_samples > (1 << 20)) return h->fail(LCD_ERR_INVALID, "lcd_profile_begin: bad sample count");
    while ((int)h->prof_ev.size() < 2 * max_samples) {
        hipEvent_t e;
        LCD_HIP(h, hipEventCreate(&e));
        h->prof_ev.push_back(e);
        LCD_HIP(h, hipEventCreate(&e));
        h->prof_ev.push_back(e);
        LCD_HIP(h, hipEventCreate(&e));
        h->prof2_ev.push_back(e);
        LCD_HIP(h, hipEventCreate(&e));
        h->prof2_ev.push_back(e);
    }
    h->prof_n = 0;
    h->prof2_n = 0;
    h->prof_cap = max_samples;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_profile_read(lcd_engine* h, float* avg_ms, int* n_samples, const char** kernel_name) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    { int rc = h->sync_all(); if (rc) return rc; }
    double sum = 0.0;
    for (int i = 0; i < h->prof_n; ++i) {
        float ms = 0.0f;
        LCD_HIP(h, hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
        sum += ms;
    }
    if (avg_ms) *avg_ms = h->prof_n ? (float)(sum / h->prof_n) : 0.0f;
    if (n_samples) *n_samples = h->prof_n;
    if (kernel_name) *kernel_name = h->prof_kernel;
    h->prof_cap = 0;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_profile_read_likelihood(lcd_engine* h, float* avg_ms, int* n_samples, const char** kernel_name) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    { int rc = h->sync_all(); if (rc) return rc; }
    double sum = 0.0;
    for (int i = 0; i < h->prof2_n; ++i) {
        float ms = 0.0f;
        LCD_HIP(h, hipEventElapsedTime(&ms, h->prof2_ev[2 * i], h->prof2_ev[2 * i + 1]));
        sum += ms;
    }
    if (avg_ms) *avg_ms = h->prof2_n ? (float)(sum / h->prof2_n) : 0.0f;
    if (n_samples) *n_samples = h->prof2_n;
    if (kernel_name) *kernel_name = h->prof2_kernel;
    h->prof_cap = 0;
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_set_option(lcd_engine* h, const char* key, int64_t value) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    { int rc = h->drain(); if (rc) return rc; }
    if (!key) return h->fail(LCD_ERR_INVALID, "lcd_set_option: null key");
    if (!std::strcmp(key, "score_block") && (value == 256 || value == 512 || value == 1024)) { h->tfidf.score_block = (int)value; return LCD_OK; }
    // compute units the bf16 filter's persistent launch plans for (vocabularies of more 256-word strips than that): -1 built-in, 0 off
    if (!std::strcmp(key, "filter_units") && value >= -1 && value <= 4096) { h->filter_units = (int)value; return LCD_OK; }
    if (!std::strcmp(key, "strip_tiles") && value >= 0 && value <= 8) { h->strip_tiles = (int)value; return LCD_OK; }
    if (!std::strcmp(key, "shard_growth_first") && value >= 0 && value < (1ll << 28)) { h->shard_first = (int32_t)value; return LCD_OK; }
    if (!std::strcmp(key, "shard_growth_block") && value >= 0 && value <= (1 << 20)) { h->shard_block = (int32_t)value; return LCD_OK; }
    if (!std::strcmp(key, "shard_append") && (value == 0 || value == 1)) { h->shard_append = (int)value; return LCD_OK; }
    // 0: lcd_profile_begin brackets only the 2-NN launch of a pipelined frame (an event pair costs the stream ~10 us)
    if (!std::strcmp(key, "profile_likelihood") && (value == 0 || value == 1)) { h->prof_likelihood = value != 0; return LCD_OK; }
    // (per handle, like every option) sealed buckets from which the rows of a deferred append are written by a launch of their own; -1: built-in
    if (!std::strcmp(key, "append_split_buckets") && value >= -1 && value <= (1 << 24)) { h->popt.append_split_buckets = (int)value; return LCD_OK; }
    if (!std::strcmp(key, "cross_frame_tiles") && value >= -1 && value <= 1) { h->popt.cross_frames = value > 0 ? 1 : 0; return LCD_OK; }
    if (!std::strcmp(key, "append_from_rerank") && value >= -1 && value <= 1) { h->popt.append_from_rerank = value != 0 ? 1 : 0; return LCD_OK; }
    if (!std::strcmp(key, "roctx") && (value == 0 || value == 1)) {
        if (!value) { h->roctx_push = nullptr; h->roctx_pop = nullptr; return LCD_OK; }
        static void* lib = nullptr;                                  // (stays loaded: ranges of other handles may be open)
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return h->fail(LCD_ERR_UNSUPPORTED, "lcd_set_option(roctx): libroctx64.so not found");
        h->roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        h->roctx_pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!h->roctx_push || !h->roctx_pop) { h->roctx_push = nullptr; h->roctx_pop = nullptr; return h->fail(LCD_ERR_UNSUPPORTED, "lcd_set_option(roctx): roctxRangePushA / roctxRangePop not exported"); }
        return LCD_OK;
    }
    // timing experiments: the filter workgroups of launch A wait value x 64 clocks in front of their first request (the single-workgroup
    // chains of the launch then get their first round trip ahead of the strips' opening burst)
    if (!std::strcmp(key, "filter_delay") && value >= 0 && value <= 127) { h->popt.filter_delay = (int)value; return LCD_OK; }
    if (!std::strcmp(key, "shadow_rows") && value >= -1 && value <= 2) { h->popt.shadow_rows = value < 0 ? 1 : (int)value; return LCD_OK; }   // (-1: built-in = 1)
    if (!std::strcmp(key, "mirror_from_b") && value >= -1 && value <= 1) { h->popt.mirror_from_b = value != 0 ? 1 : 0; return LCD_OK; }
    if (!std::strcmp(key, "next_word_id") && value >= 1 && value < (1ll << 28)) { h->next_word_id = std::max(h->next_word_id, (int32_t)value); return LCD_OK; }   // (the drain above has brought the row mirror up to date)
    if (!std::strcmp(key, "profile_skip") && value >= 0 && value <= (1 << 20)) { h->prof_skip = (int)value; return LCD_OK; }
    if (!std::strcmp(key, "decision_straight") && value >= -1 && value <= 2) { h->popt.decision_straight = value >= 0 ? (int)value : PipeOpts().decision_straight; return LCD_OK; }
    if (!std::strcmp(key, "slots_from_rows") && value >= -1 && value <= 2) { h->popt.slots_from_rows = value >= 0 ? (int)value : PipeOpts().slots_from_rows; return LCD_OK; }
    if (!std::strcmp(key, "row_writer_wgs") && value >= -1 && value <= 256) { h->popt.row_writer_wgs = value >= 0 ? (int)value : PipeOpts().row_writer_wgs; return LCD_OK; }
    return h->fail(LCD_ERR_INVALID, "lcd_set_option: unknown key or value");
    LCD_CATCH(h)
}

int lcd_trace_push(lcd_engine* h, const char* name) {
    if (!h || !name) return LCD_ERR_INVALID;
    if (h->roctx_push) h->roctx_push(name);
    return LCD_OK;
}
int lcd_trace_pop(lcd_engine* h) {
    if (!h) return LCD_ERR_INVALID;
    if (h->roctx_pop) h->roctx_pop();
    return LCD_OK;
}

int lcd_profile_score_work(lcd_engine* h, int64_t* out8) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    LCD_DEV(h);
    if (!out8) return h->fail(LCD_ERR_INVALID, "lcd_profile_score_work: null output");
    { int rc = h->sync_all(); if (rc) return rc; }
    LCD_HIP(h, h->tfidf.score_work(out8));
    return LCD_OK;
    LCD_CATCH(h)
}

int lcd_get_stats(lcd_engine* h, lcd_stats* out) {
    LCD_TRY
    LCD_CHECK_HANDLE(h);
    if (!out) return LCD_ERR_INVALID;
    LCD_DEV(h);
    { int rc = h->sync_all(); if (rc) return rc; }
    out->knn_last_fallback_queries = 0;
    out->knn_max_err_ratio = 0.0;
    const void* fc = h->last_fail_count ? h->last_fail_count : h->d_fail_count.p;
    if (fc) {
        int32_t n[3] = {0, 0, 0};
        int rc = download(h, n, fc, 12, h->h_out2);
        if (rc) return rc;
        out->knn_last_fallback_queries = n[0];
        float r;
        std::memcpy(&r, &n[2], 4);
        out->knn_max_err_ratio = r;
    }
    out->vocab_rows = h->n_rows; out->vocab_live = h->n_live;
    out->signatures = h->tfidf.live_sigs; out->postings = h->tfidf.postings_ub;
    out->knn_launches = h->knn_launches; out->likelihood_launches = h->likelihood_launches; out->rebuilds = h->rebuilds;
    h->tfidf.harvest_released(false);
    out->frame_calls = h->frame_calls; out->frame_host_ns = h->frame_host_ns;
    out->buckets_sealed = h->tfidf.seals;
    out->word_slots = (int64_t)h->tfidf.n_wslots - h->tfidf.ws_free_count;
    out->dense_words = h->tfidf.h_n_dense ? (int64_t)*(volatile uint32_t*)h->tfidf.h_n_dense : 0;
    out->bytes_device = h->bytes_device;
    out->clean_divergent_refs = 0;
    if (h->tfidf.q_meta.p) {
        uint32_t n = 0;
        int rc = download(h, &n, h->tfidf.q_meta.as<uint32_t>() + 8, 4, h->h_out2);
        if (rc) return rc;
        out->clean_divergent_refs = (int64_t)n;
    }
    return LCD_OK;
    LCD_CATCH(h)
}

}  // extern "C"
