// ShardedLcd.cpp -- liblcd_shard.so: the sharded frame of include/lcd_shard.h as C++ host code over the C-ABI of liblcd_hip.so and RCCL.
// (rtabmap_amd/sharded.py drives the same engine entries through torch.distributed for bench.py --gpus N; this is the path a C++ caller --
// the reference is C++ -- links.)
#include "../../include/lcd_shard.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <new>
#include <string>
#include <vector>

struct lcd_shard_comm {
    lcd_engine* eng = nullptr;
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;                                    // exchanges on the engine's stream (the all-gather, the in-order all-reduce)
    ncclComm_t comm2 = nullptr;                                   // the deferred all-reduce's own communicator (operations on ONE communicator are
                                                                  // serialised in issue order: the next frame's all-gather would wait for it)
    lcd_shard_transport tr{};                                     // tr.all_gather != NULL: the caller's exchanges instead of RCCL
    hipStream_t stream = nullptr;                                 // the engine's
    hipStream_t stream2 = nullptr;                                // the deferred all-reduce runs here, under the next frame's search
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    lcd_shard_cand* d_cand = nullptr; size_t cand_cap = 0;        // this rank's q x 2 records
    lcd_shard_cand* d_all = nullptr; size_t all_cap = 0;          // world x q x 2 records, rank-major
    long long* d_lfix[2] = {nullptr, nullptr}; size_t lfix_cap[2] = {0, 0};   // integer partial likelihood per signature slot (two: one may be in flight)
    int parity = 0;
    struct Owed { bool any = false; int buf = 0; int64_t n = 0; float* d_like = nullptr; } owed;
    std::vector<int32_t> retire_q;                                // lcd_shard_sig_remove calls made while a likelihood is owed
    int32_t grow_first = 0, grow_block = 0;
    std::string err;
    int fail(int code, const std::string& m) { err = m; return code; }
};

namespace {
template <typename T>
int grow(lcd_shard_comm* c, T*& p, size_t& cap, size_t need) {
    if (need <= cap) return LCD_OK;
    size_t n = cap ? cap : 1024;
    while (n < need) n *= 2;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(LCD_ERR_HIP, "hipSetDevice");   // the buffers live on the ENGINE's device
    if (p) {
        if (hipStreamSynchronize(c->stream) != hipSuccess || (c->stream2 && hipStreamSynchronize(c->stream2) != hipSuccess))
            return c->fail(LCD_ERR_HIP, "hipStreamSynchronize");
        (void)hipFree(p); p = nullptr; cap = 0;
    }
    if (hipMalloc((void**)&p, n * sizeof(T)) != hipSuccess) return c->fail(LCD_ERR_NOMEM, "hipMalloc(exchange buffer)");
    cap = n;
    return LCD_OK;
}

int all_gather(lcd_shard_comm* c, const void* send, void* recv, size_t bytes_per_rank) {
    if (c->world == 1) {
        if (hipMemcpyAsync(recv, send, bytes_per_rank, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return c->fail(LCD_ERR_HIP, "hipMemcpyAsync(candidates)");
        return LCD_OK;
    }
    if (c->tr.all_gather) return c->tr.all_gather(c->tr.user, send, recv, bytes_per_rank, c->stream) == 0 ? LCD_OK : c->fail(LCD_ERR_HIP, "transport all_gather");
    // 8 KB per rank at 500 descriptors -- latency-bound, one RCCL call on the engine's stream
    if (ncclAllGather(send, recv, bytes_per_rank / 8, ncclInt64, c->comm, c->stream) != ncclSuccess) return c->fail(LCD_ERR_HIP, "ncclAllGather(candidates)");
    return LCD_OK;
}
// 0.8 MB at 100k signatures, 8 MB at 1M (int64: order-free, bit-reproducible)
int all_reduce(lcd_shard_comm* c, long long* buf, size_t count, bool second_stream) {
    if (c->world == 1 || count == 0) return LCD_OK;
    hipStream_t s = second_stream ? c->stream2 : c->stream;
    if (c->tr.all_reduce_sum_i64) return c->tr.all_reduce_sum_i64(c->tr.user, buf, count, s) == 0 ? LCD_OK : c->fail(LCD_ERR_HIP, "transport all_reduce");
    ncclComm_t cm = (second_stream && c->comm2) ? c->comm2 : c->comm;
    if (ncclAllReduce(buf, buf, count, ncclInt64, ncclSum, cm, s) != ncclSuccess) return c->fail(LCD_ERR_HIP, "ncclAllReduce(likelihood)");
    return LCD_OK;
}

// the owed likelihood becomes final on the engine's stream (behind its all-reduce), then the queued retirements are applied
int complete_owed(lcd_shard_comm* c) {
    if (c->owed.any) {
        c->owed.any = false;
        if (c->world > 1 && hipStreamWaitEvent(c->stream, c->ev_done, 0) != hipSuccess) return c->fail(LCD_ERR_HIP, "hipStreamWaitEvent(all-reduce)");
        if (lcd_finalize_dev(c->eng, (int64_t*)c->d_lfix[c->owed.buf], c->owed.n, c->owed.d_like) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
    }
    for (int32_t s : c->retire_q)
        if (lcd_sig_remove(c->eng, s) != LCD_OK) { c->retire_q.clear(); return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng)); }
    c->retire_q.clear();
    return LCD_OK;
}

int create_common(lcd_engine* engine, int rank, int world, lcd_shard_comm** out, lcd_shard_comm** made) {
    if (!engine || !out || world < 1 || world > 64 || rank < 0 || rank >= world) return LCD_ERR_INVALID;
    *out = nullptr;
    lcd_shard_comm* c = new (std::nothrow) lcd_shard_comm();
    if (!c) return LCD_ERR_NOMEM;
    c->eng = engine; c->rank = rank; c->world = world;
    c->stream = (hipStream_t)lcd_stream(engine);
    // the exchange buffers must live on the ENGINE's device, which need not be the calling thread's current one: ask the engine's stream
    hipDevice_t dev = 0;
    if (c->stream && hipStreamGetDevice(c->stream, &dev) == hipSuccess) c->device = (int)dev;
    else if (hipGetDevice(&c->device) != hipSuccess) { delete c; return LCD_ERR_HIP; }
    // ... and so must the second stream and the events made next (finish_create): they are created on the calling thread's CURRENT device
    if (hipSetDevice(c->device) != hipSuccess) { delete c; return LCD_ERR_HIP; }
    *made = c;
    return LCD_OK;
}

int frame_impl(lcd_shard_comm* c, bool deferred, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id, int32_t first_new_word_id,
               float N, int64_t total_live_rows, int32_t* d_word_ids, float* d_likelihood, int64_t likelihood_capacity) {
    if (q <= 0 || !d_descriptors || !d_word_ids) return c->fail(LCD_ERR_INVALID, "lcd_shard_frame: bad argument");
    // one thread may drive several engines: the copies, events and collectives below are issued for THIS engine's device
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(LCD_ERR_HIP, "hipSetDevice");
    const size_t n_rec = (size_t)q * 2;
    { int rc = grow(c, c->d_cand, c->cand_cap, n_rec); if (rc) return rc; }
    { int rc = grow(c, c->d_all, c->all_cap, n_rec * (size_t)c->world); if (rc) return rc; }
    // (1) local exact 2-NN of this rank's rows -> 16-byte records {key = distance bits << 32 | local row, word id, postings key}.  The
    //     vocabulary does not depend on the likelihood still owed: its all-reduce (second stream) runs under this search.
    if (lcd_shard_knn2_dev(c->eng, d_descriptors, q, c->d_cand) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
    // (2) all-gather of the records
    { int rc = all_gather(c, c->d_cand, c->d_all, n_rec * sizeof(lcd_shard_cand)); if (rc) return rc; }
    // (2b) the previous frame's likelihood, before this frame's registration and scoring touch the index
    { int rc = complete_owed(c); if (rc) return rc; }
    // (3) merge + same-frame resolution (replicated), registration and integer scoring of the words this rank owns
    const int32_t* slot_sig = nullptr;
    int64_t n_slots = 0;
    if (lcd_slots_dev(c->eng, &slot_sig, &n_slots) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
    const int64_t after = n_slots + (sig_id != 0 ? 1 : 0);
    if (d_likelihood && likelihood_capacity < after) return c->fail(LCD_ERR_INVALID, "lcd_shard_frame: likelihood buffer too small");
    const int b = c->parity;
    if (d_likelihood) { int rc = grow(c, c->d_lfix[b], c->lfix_cap[b], (size_t)after + 1); if (rc) return rc; }
    if (lcd_shard_frame_dev(c->eng, d_descriptors, q, flags, nndr_ratio, sig_id, first_new_word_id, N, c->rank, c->world, c->d_all, total_live_rows,
                            d_word_ids, d_likelihood ? (int64_t*)c->d_lfix[b] : nullptr, (int64_t)c->lfix_cap[b]) != LCD_OK)
        return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
    if (!d_likelihood) return LCD_OK;
    c->parity ^= 1;
    if (!deferred) {
        // (4) all-reduce of the partial likelihood, (5) fixed point -> float, / ni -- in order on the engine's stream
        { int rc = all_reduce(c, c->d_lfix[b], (size_t)after, false); if (rc) return rc; }
        if (lcd_finalize_dev(c->eng, (int64_t*)c->d_lfix[b], after, d_likelihood) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
        return LCD_OK;
    }
    // (4') the all-reduce on the second stream, behind this frame's scoring; (5) waits until the next call (or lcd_shard_flush)
    if (c->world > 1) {
        if (hipEventRecord(c->ev_ready, c->stream) != hipSuccess || hipStreamWaitEvent(c->stream2, c->ev_ready, 0) != hipSuccess)
            return c->fail(LCD_ERR_HIP, "hipEventRecord / hipStreamWaitEvent");
        { int rc = all_reduce(c, c->d_lfix[b], (size_t)after, true); if (rc) return rc; }
        if (hipEventRecord(c->ev_done, c->stream2) != hipSuccess) return c->fail(LCD_ERR_HIP, "hipEventRecord(all-reduce)");
    }
    c->owed.any = true; c->owed.buf = b; c->owed.n = after; c->owed.d_like = d_likelihood;
    return LCD_OK;
}

int finish_create(lcd_shard_comm* c) {
    if (c->world == 1) return LCD_OK;
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) return LCD_ERR_HIP;
    if (hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess)
        return LCD_ERR_HIP;
    return LCD_OK;
}
}  // namespace

#define SHARD_TRY try {
#define SHARD_CATCH(c) } catch (const std::bad_alloc&) { return (c)->fail(LCD_ERR_NOMEM, "out of host memory"); } \
    catch (...) { return (c)->fail(LCD_ERR_STATE, "unexpected exception"); }

extern "C" {

int lcd_shard_unique_id(unsigned char out128[128]) {
    if (!out128) return LCD_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return LCD_ERR_HIP;
    for (int i = 0; i < 128; ++i) out128[i] = (unsigned char)id.internal[i];
    return LCD_OK;
}

int lcd_shard_comm_create(lcd_engine* engine, int rank, int world, const unsigned char id128[128], lcd_shard_comm** out) {
    if (world > 1 && !id128) return LCD_ERR_INVALID;
    try {
        lcd_shard_comm* c = nullptr;
        { int rc = create_common(engine, rank, world, out, &c); if (rc) return rc; }
        if (world > 1) {
            ncclUniqueId id;
            for (int i = 0; i < 128; ++i) id.internal[i] = (char)id128[i];
            if (ncclCommInitRank(&c->comm, world, id, rank) != ncclSuccess) { delete c; return LCD_ERR_HIP; }
            // a second communicator over the same ranks for the deferred all-reduce (without it the driver still works: the next frame's
            // all-gather then queues behind the all-reduce on the one communicator)
            if (ncclCommSplit(c->comm, 0, rank, &c->comm2, nullptr) != ncclSuccess) c->comm2 = nullptr;
        }
        if (finish_create(c) != LCD_OK) { lcd_shard_comm_destroy(c); return LCD_ERR_HIP; }
        *out = c;
        return LCD_OK;
    } catch (...) { return LCD_ERR_NOMEM; }
}

int lcd_shard_comm_create_transport(lcd_engine* engine, int rank, int world, const lcd_shard_transport* transport, lcd_shard_comm** out) {
    if (!transport || transport->struct_size != (int32_t)sizeof(lcd_shard_transport) || !transport->all_gather || !transport->all_reduce_sum_i64)
        return LCD_ERR_INVALID;
    try {
        lcd_shard_comm* c = nullptr;
        { int rc = create_common(engine, rank, world, out, &c); if (rc) return rc; }
        c->tr = *transport;
        if (finish_create(c) != LCD_OK) { lcd_shard_comm_destroy(c); return LCD_ERR_HIP; }
        *out = c;
        return LCD_OK;
    } catch (...) { return LCD_ERR_NOMEM; }
}

void lcd_shard_comm_destroy(lcd_shard_comm* c) {
    if (!c) return;
    try {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
        if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
        if (c->ev_done) (void)hipEventDestroy(c->ev_done);
        if (c->comm2) (void)ncclCommDestroy(c->comm2);
        if (c->comm) (void)ncclCommDestroy(c->comm);
        if (c->d_cand) (void)hipFree(c->d_cand);
        if (c->d_all) (void)hipFree(c->d_all);
        for (int i = 0; i < 2; ++i) if (c->d_lfix[i]) (void)hipFree(c->d_lfix[i]);
        delete c;
    } catch (...) { }
}

const char* lcd_shard_last_error(const lcd_shard_comm* c) { return c ? c->err.c_str() : "null communicator"; }

int lcd_shard_set_growth(lcd_shard_comm* c, int32_t first_incremental_id, int32_t block) {
    if (!c) return LCD_ERR_INVALID;
    SHARD_TRY
    if (block < 0 || (block > 0 && first_incremental_id <= 0)) return c->fail(LCD_ERR_INVALID, "lcd_shard_set_growth: bad argument");
    if (lcd_set_option(c->eng, "shard_growth_first", block > 0 ? first_incremental_id : 0) != LCD_OK ||
        lcd_set_option(c->eng, "shard_growth_block", block) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
    c->grow_first = block > 0 ? first_incremental_id : 0; c->grow_block = block;
    return LCD_OK;
    SHARD_CATCH(c)
}

int lcd_shard_set_append(lcd_shard_comm* c, int on) {
    if (!c) return LCD_ERR_INVALID;
    SHARD_TRY
    if (lcd_set_option(c->eng, "shard_append", on ? 1 : 0) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
    return LCD_OK;
    SHARD_CATCH(c)
}

int lcd_shard_owner_of(const lcd_shard_comm* c, int32_t word_id) {
    if (!c) return -1;
    if (c->grow_block > 0) return word_id >= c->grow_first ? ((word_id - c->grow_first) / c->grow_block) % c->world : -1;
    return c->world - 1;
}

int lcd_shard_frame(lcd_shard_comm* c, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id, int32_t first_new_word_id,
                    float N, int64_t total_live_rows, int32_t* d_word_ids, float* d_likelihood, int64_t likelihood_capacity) {
    if (!c) return LCD_ERR_INVALID;
    SHARD_TRY
    return frame_impl(c, false, d_descriptors, q, flags, nndr_ratio, sig_id, first_new_word_id, N, total_live_rows, d_word_ids, d_likelihood, likelihood_capacity);
    SHARD_CATCH(c)
}

int lcd_shard_frame_deferred(lcd_shard_comm* c, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id,
                             int32_t first_new_word_id, float N, int64_t total_live_rows, int32_t* d_word_ids, float* d_likelihood,
                             int64_t likelihood_capacity) {
    if (!c) return LCD_ERR_INVALID;
    SHARD_TRY
    return frame_impl(c, true, d_descriptors, q, flags, nndr_ratio, sig_id, first_new_word_id, N, total_live_rows, d_word_ids, d_likelihood, likelihood_capacity);
    SHARD_CATCH(c)
}

int lcd_shard_flush(lcd_shard_comm* c) {
    if (!c) return LCD_ERR_INVALID;
    SHARD_TRY
    return complete_owed(c);
    SHARD_CATCH(c)
}

int lcd_shard_sig_remove(lcd_shard_comm* c, int32_t sig_id) {
    if (!c) return LCD_ERR_INVALID;
    SHARD_TRY
    if (c->owed.any) { c->retire_q.push_back(sig_id); return LCD_OK; }   // the owed likelihood is finalised against the memory as its frame left it
    if (lcd_sig_remove(c->eng, sig_id) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
    return LCD_OK;
    SHARD_CATCH(c)
}

}  // extern "C"
