// ShardedLcd.cpp -- liblcd_shard.so: the sharded frame of include/lcd_shard.h as C++ host code over the C-ABI of liblcd_hip.so and RCCL.
// (rtabmap_amd/sharded.py drives the same engine entries through torch.distributed for the Python tests and bench.py; this is the path
// a C++ caller -- the reference is C++ -- links.)
#include "../../include/lcd_shard.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <new>
#include <string>

struct lcd_shard_comm {
    lcd_engine* eng = nullptr;
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    lcd_shard_cand* d_cand = nullptr; size_t cand_cap = 0;        // this rank's q x 2 records
    lcd_shard_cand* d_all = nullptr; size_t all_cap = 0;          // world x q x 2 records, rank-major
    long long* d_lfix = nullptr; size_t lfix_cap = 0;             // integer partial likelihood per signature slot
    std::string err;
    int fail(int code, const std::string& m) { err = m; return code; }
};

namespace {
template <typename T>
int grow(lcd_shard_comm* c, T*& p, size_t& cap, size_t need) {
    if (need <= cap) return LCD_OK;
    size_t n = cap ? cap : 1024;
    while (n < need) n *= 2;
    if (p) { if (hipStreamSynchronize(c->stream) != hipSuccess) return c->fail(LCD_ERR_HIP, "hipStreamSynchronize"); (void)hipFree(p); p = nullptr; cap = 0; }
    if (hipMalloc((void**)&p, n * sizeof(T)) != hipSuccess) return c->fail(LCD_ERR_NOMEM, "hipMalloc(exchange buffer)");
    cap = n;
    return LCD_OK;
}
}  // namespace

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
    if (!engine || !out || world < 1 || world > 64 || rank < 0 || rank >= world || (world > 1 && !id128)) return LCD_ERR_INVALID;
    *out = nullptr;
    try {
        lcd_shard_comm* c = new (std::nothrow) lcd_shard_comm();
        if (!c) return LCD_ERR_NOMEM;
        c->eng = engine; c->rank = rank; c->world = world;
        c->stream = (hipStream_t)lcd_stream(engine);
        if (hipGetDevice(&c->device) != hipSuccess) { delete c; return LCD_ERR_HIP; }
        if (world > 1) {
            ncclUniqueId id;
            for (int i = 0; i < 128; ++i) id.internal[i] = (char)id128[i];
            if (ncclCommInitRank(&c->comm, world, id, rank) != ncclSuccess) { delete c; return LCD_ERR_HIP; }
        }
        *out = c;
        return LCD_OK;
    } catch (...) { return LCD_ERR_NOMEM; }
}

void lcd_shard_comm_destroy(lcd_shard_comm* c) {
    if (!c) return;
    try {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->comm) (void)ncclCommDestroy(c->comm);
        if (c->d_cand) (void)hipFree(c->d_cand);
        if (c->d_all) (void)hipFree(c->d_all);
        if (c->d_lfix) (void)hipFree(c->d_lfix);
        delete c;
    } catch (...) { }
}

const char* lcd_shard_last_error(const lcd_shard_comm* c) { return c ? c->err.c_str() : "null communicator"; }

int lcd_shard_frame(lcd_shard_comm* c, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id, int32_t first_new_word_id,
                    float N, int64_t total_live_rows, int32_t* d_word_ids, float* d_likelihood, int64_t likelihood_capacity) {
    if (!c) return LCD_ERR_INVALID;
    try {
        if (q <= 0 || !d_descriptors || !d_word_ids) return c->fail(LCD_ERR_INVALID, "lcd_shard_frame: bad argument");
        const size_t n_rec = (size_t)q * 2;
        { int rc = grow(c, c->d_cand, c->cand_cap, n_rec); if (rc) return rc; }
        { int rc = grow(c, c->d_all, c->all_cap, n_rec * (size_t)c->world); if (rc) return rc; }
        // (1) local exact 2-NN of this rank's rows -> 16-byte records {key = distance bits << 32 | local row, word id, postings key}
        if (lcd_shard_knn2_dev(c->eng, d_descriptors, q, c->d_cand) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
        // (2) all-gather: 8 KB per rank at 500 descriptors -- latency-bound, one RCCL call on the engine's stream
        if (c->world == 1) {
            if (hipMemcpyAsync(c->d_all, c->d_cand, n_rec * sizeof(lcd_shard_cand), hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
                return c->fail(LCD_ERR_HIP, "hipMemcpyAsync(candidates)");
        } else if (ncclAllGather(c->d_cand, c->d_all, n_rec * 2, ncclInt64, c->comm, c->stream) != ncclSuccess) {
            return c->fail(LCD_ERR_HIP, "ncclAllGather(candidates)");
        }
        // (3) merge + same-frame resolution (replicated), registration and integer scoring of the words this rank owns
        const int32_t* slot_sig = nullptr;
        int64_t n_slots = 0;
        if (lcd_slots_dev(c->eng, &slot_sig, &n_slots) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
        const int64_t after = n_slots + (sig_id != 0 ? 1 : 0);
        if (d_likelihood && likelihood_capacity < after) return c->fail(LCD_ERR_INVALID, "lcd_shard_frame: likelihood buffer too small");
        if (d_likelihood) { int rc = grow(c, c->d_lfix, c->lfix_cap, (size_t)after + 1); if (rc) return rc; }
        if (lcd_shard_frame_dev(c->eng, d_descriptors, q, flags, nndr_ratio, sig_id, first_new_word_id, N, c->rank, c->world, c->d_all, total_live_rows,
                                d_word_ids, d_likelihood ? (int64_t*)c->d_lfix : nullptr, (int64_t)c->lfix_cap) != LCD_OK)
            return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
        if (!d_likelihood) return LCD_OK;
        // (4) all-reduce of the partial likelihood: 0.8 MB at 100k signatures, 8 MB at 1M (int64: order-free, bit-reproducible)
        if (c->world > 1 && ncclAllReduce(c->d_lfix, c->d_lfix, (size_t)after, ncclInt64, ncclSum, c->comm, c->stream) != ncclSuccess)
            return c->fail(LCD_ERR_HIP, "ncclAllReduce(likelihood)");
        // (5) fixed point -> float, / ni
        if (lcd_finalize_dev(c->eng, (int64_t*)c->d_lfix, after, d_likelihood) != LCD_OK) return c->fail(LCD_ERR_STATE, lcd_last_error(c->eng));
        return LCD_OK;
    } catch (const std::bad_alloc&) { return c->fail(LCD_ERR_NOMEM, "out of host memory"); }
    catch (...) { return c->fail(LCD_ERR_STATE, "unexpected exception"); }
}

}  // extern "C"
