// tfidf.h -- device-resident inverted index + TF-IDF likelihood (internal to liblcd_hip.so).
//
// Replaces VisualWord::_references (reference VisualWord.h:62, std::map<sigId,count> per word), Memory::getNi
// (Memory.cpp:4955) and the scoring loop of Memory::computeLikelihood (Memory.cpp:2215-2291).
//
// Data layout in HBM ("blocked inverted index"):
//   * a signature gets a SLOT (dense, arrival order); slots are grouped in BUCKETS of TF_R = 256 consecutive slots;
//   * a word gets a WSLOT (dense); nw[wslot] = number of live signatures referencing the word;
//   * every bucket keeps the arrival-order log of its postings (coo_w[e] = wslot, coo_pc[e] = slot_local << 22 | count),
//     which is also the forward index used to retire a signature;
//   * when a bucket is full it is SEALED: its postings are regrouped by word (counting sort on the device) into
//     ent[] (4 B per posting: slot_local << 22 | count) with a directory dir[wslot] -> first posting, so that
//     "the postings of word w that fall in bucket b" is one contiguous segment found with two loads;
//   * ni[slot] (0 = retired) is read once per workgroup into LDS.
// Scoring a frame = for every sealed bucket, one workgroup (x G word groups) walks the segments of the frame's words
// flattened into one load-balanced index space (heavy-tailed posting lists cannot starve a wave), accumulates into an
// LDS array of TF_R fixed-point (Q15.48, int64) sums with ds_add_u64 and flushes each slot once.  The open bucket is
// scanned in arrival order against the frame's sorted word list.  Integer accumulation makes the result independent
// of the order of the adds (bit-reproducible run to run and across any sharding of the words over GPUs); each term
// is computed in fp32 exactly as the reference does, (nwi * log10(N/nw)) / ni.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "devbuf.h"
#include "lcd_kernels.h"

namespace lcd {

constexpr int TF_R = 256;                  // slots per bucket
constexpr int TF_CNT_BITS = 22;            // posting = slot_local << 22 | count
constexpr uint32_t TF_CNT_MASK = (1u << TF_CNT_BITS) - 1;
constexpr int TF_MAX_WORDS = 8192;         // words of one signature / one query frame handled by the 1-workgroup kernels
constexpr int TF_FIX_SHIFT = 48;           // Q15.48 fixed point

// one bucket as the kernels see it
struct BucketDev {
    uint32_t* coo_w;       // [cap] arrival-order log: wslot
    uint32_t* coo_pc;      // [cap] arrival-order log: slot_local << 22 | count
    const uint32_t* dir;   // sealed: [W + 1] first posting of each wslot
    const uint32_t* ent;   // sealed: [n_e] postings grouped by wslot
    uint32_t W;            // wslots covered by dir
    uint32_t sealed;
    uint32_t n_e_sealed;   // postings in ent
    uint32_t pad;
};

struct Bucket {
    bool sealed = false;
    int n_slots = 0;          // slots handed out in this bucket
    int live = 0;             // live signatures among them
    int64_t ub_entries = 0;   // host upper bound of log entries (device appends without telling the host)
    DevBuf coo_w, coo_pc, dir, ent;
    uint32_t W = 0;
    uint32_t n_e_sealed = 0;
};

// arguments of the addNewWords decision loop (resolve_body.cuh) when it is fused into the frame-words launch
struct ResolveArgs {
    int q, flags; float nndr; int have_index;
    const int32_t* knn_word; const float* knn_dist; const float* selfdist; int ld; const uint32_t* cand_bits; int bw;
    int32_t* out_word; int32_t* out_n_new; const int32_t* knn_row; const int32_t* row_wslot; int32_t* out_wslot;
    int32_t* fail_count;   // reset for the next frame's certificate (saves a memset launch); may be NULL
    RowparArgs rp;         // rp.enabled: the exact redo of rejected queries runs as extra workgroups of the tail launch
};

struct Tfidf {
    hipStream_t stream = nullptr;
    int64_t* bytes_device = nullptr;
    // per slot
    DevBuf slot_sig, slot_ni, slot_begin, slot_cnt;
    // per wslot
    DevBuf nw;
    DevBuf idf_tab;                      // {stamp, idf bits} of the words of the current frame (valid iff stamp matches)
    uint32_t stamp = 0;
    // per bucket
    DevBuf bkt_tab, bkt_ne, bkt_list, bkt_list_all, open_done;
    std::vector<Bucket> buckets;
    std::vector<BucketDev> h_bkt;
    bool bkt_dirty = true;
    int n_list = 0;                      // sealed buckets with live signatures (entries of bkt_list)
    int n_list_all = 0;                  // all sealed buckets, retired ones included (entries of bkt_list_all)
    int q_n_ub = 0;                      // word count of the last frame handed to frame_words (upper bound of its unique words)
    // per frame
    DevBuf lfix;                         // int64 accumulator per slot
    DevBuf q_w, q_cnt, q_idf, q_meta;    // the frame's unique words (sorted wslots), counts, idf, [0] = unique count
    DevBuf tmp_cursor;                   // sealing scratch
    DevBuf d_stage;                      // staged word slots of host-side calls
    PinBuf h_stage;
    // host maps
    std::unordered_map<int32_t, int64_t> sig_slot;    // live signature id -> slot
    std::unordered_map<int32_t, int32_t> word_wslot;  // word id -> wslot (never recycled in this version)
    int32_t n_wslots = 0;
    int64_t n_slots = 0, live_sigs = 0;
    int64_t postings_ub = 0;
    std::string err;

    hipError_t init(hipStream_t s, int64_t* bytes, int64_t sig_capacity);
    void destroy();
    // wslot of a word id (assigned on first sight); grows nw[]
    hipError_t wslot_of(int32_t word_id, int32_t* out);
    // register one signature whose word slots are already on the device (d_wslots[n]; < 0 = no word); if N > 0 the
    // frame's unique words / idf are left in q_* for a following score()
    hipError_t register_dev(int32_t sig_id, const int32_t* d_wslots, int n, int32_t ni, float N, const ResolveArgs* resolve = nullptr);
    // prepare q_* from word slots on the device without registering anything
    hipError_t query_dev(const int32_t* d_wslots, int n, float N, const ResolveArgs* resolve = nullptr);
    hipError_t flush_retire();
    std::vector<int64_t> pending_retire;   // slots whose device-side retirement rides along with the next frame-words launch
    // score q_* against every live signature: dense float likelihood over slots [0, n_slots)
    hipEvent_t prof_b = nullptr, prof_e = nullptr;   // one-shot: bracket the next fused scoring launch (lcd_profile_*)
    hipError_t score(float* d_likelihood);
    // the two halves of score(): integer partial sums into a ZEROED caller buffer, and fixed point -> float (re-zeroes the source)
    hipError_t score_partial(unsigned long long* lfix_target);
    hipError_t finalize(long long* lfix_src, long long n, float* d_likelihood);
    hipError_t retire(int32_t sig_id);
    hipError_t seal(int b);
    hipError_t ensure_slots(int64_t n);
    hipError_t upload_buckets();
};

// gather of the dense likelihood: out[k] = slots[k] >= 0 ? dense[slots[k]] : 0
hipError_t launch_gather_f32(const float* dense, const int64_t* slots, int n, float* out, hipStream_t s);

// Rtabmap::adjustLikelihood on a device vector (entry 0 = virtual place), in place
hipError_t launch_adjust_likelihood(float* d_L, int n, float ratio, hipStream_t s);

}  // namespace lcd
