/* lcd.h -- C-ABI of the MI355X-native loop-closure detection engine (liblcd_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of introlab/rtabmap (reference v0.23.8, paths below are relative to
 * /root/reference): descriptor -> visual-word quantisation (VWDictionary) + TF-IDF likelihood
 * (Memory::computeLikelihood).  Every entry point is `extern "C"`, takes plain pointers and sizes, returns an int
 * status (LCD_OK == 0) and never throws or aborts across the boundary (the reference's own convention is
 * "UERROR + return empty", VWDictionary.cpp:920-930, 948-957).  Pointers are caller-owned; host buffers are copied
 * before the call returns.  A handle is single-owner and not re-entrant (VWDictionary has no locks either; its
 * calls come from the Rtabmap thread and, for update(), from PreUpdateThread joined before use, Memory.cpp:5284,5926);
 * calls may come from different threads at different times -- every entry selects the engine's device itself.
 *
 * Numeric contract (identical to the reference): distances are SQUARED L2 for float descriptors (rtflann L2 functor,
 * dist.h:150-177, and cv::NORM_L2SQR) accumulated in the reference's own order -> bit-exact; Hamming distances as
 * float (VWDictionary.cpp:1078-1083) over EVERY byte of the descriptor (cv::NORM_HAMMING, the metric of the brute-force
 * strategies this engine stands in for; rtflann::Hamming, used by the FLANN strategies, ignores the size % 8 trailing
 * bytes, dist.h:555-579 -- the two only differ for sizes that are not a multiple of 8); on equal distance the lower vocabulary row wins (result_set.h:151-171); word
 * ids >= 1, 0 == none (ID_INVALID, VWDictionary.cpp:59-60); signature ids are any non-zero int (virtual place -1).
 *
 * There is NO CPU fallback inside this library: without a gfx950 device lcd_create() fails with LCD_ERR_HIP.
 */
#ifndef LCD_H_
#define LCD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCD_ABI_VERSION 1

typedef struct lcd_engine lcd_engine;

enum lcd_status {
    LCD_OK = 0,
    LCD_ERR_INVALID = 1,      /* bad argument (size/type mismatch: the reference logs UERROR and returns empty) */
    LCD_ERR_HIP = 2,          /* HIP runtime / launch failure, or no gfx950 device */
    LCD_ERR_NOMEM = 3,
    LCD_ERR_STATE = 4,        /* call not valid in the current state (e.g. unknown word / signature) */
    LCD_ERR_UNSUPPORTED = 5
};

enum lcd_dtype {
    LCD_F32 = 0,              /* CV_32F rows, squared-L2 metric (SURF/SIFT...) */
    LCD_U8 = 1                /* CV_8U rows, Hamming metric (ORB/BRIEF...)      */
};

/* flags of lcd_quantize / lcd_find_nn (VWDictionary parameters, Parameters.h:243-266) */
enum lcd_quantize_flags {
    LCD_Q_INCREMENTAL = 1,              /* Kp/IncrementalDictionary: NNDR decides between "existing word" and "new word" */
    LCD_Q_NEW_WORDS_COMPARED = 2        /* Kp/NewWordsComparedTogether: also match words created earlier in the same call */
};

typedef struct lcd_config {
    int32_t struct_size;       /* sizeof(lcd_config), for ABI evolution */
    int32_t device;            /* HIP device ordinal */
    int32_t dtype;             /* lcd_dtype */
    int32_t dim;               /* columns: floats (LCD_F32) or bytes (LCD_U8) per descriptor */
    int64_t vocab_capacity;    /* initial row capacity (grows on demand) */
    int64_t sig_capacity;      /* initial signature-slot capacity (grows on demand) */
    int32_t max_queries;       /* initial per-call query capacity (Kp/MaxFeatures; grows on demand) */
    int32_t reserved0;
    void*   stream;            /* optional hipStream_t to enqueue on; NULL = engine-owned stream */
} lcd_config;

/* ---------------------------------------------------------------------------------------------------------------
 * life cycle.  Replaces `new VWDictionary(parameters)` (Memory.cpp:144) for the device-side state. */
int  lcd_abi_version(void);
int  lcd_create(const lcd_config* cfg, lcd_engine** out);
void lcd_destroy(lcd_engine* h);
/* text of the last error on this handle ("" if none); valid until the next call on the handle */
const char* lcd_last_error(const lcd_engine* h);
/* block until all work enqueued by this handle has finished */
int  lcd_synchronize(lcd_engine* h);

/* ---------------------------------------------------------------------------------------------------------------
 * vocabulary == VWDictionary::_dataTree + _mapIndexId (VWDictionary.h:146-149), maintained by update() :475-701.
 * Rows live in HBM; the row ORDER is part of the contract because it is the distance tie-break. */

/* VWDictionary::clear() :843-873 / the reset at the top of the rebuild branch :612-615 */
int lcd_vocab_clear(lcd_engine* h);
/* brute-force append branch :571-609: rows appended in the given order; word_ids[i] > 0, not already present */
int lcd_vocab_append(lcd_engine* h, const void* rows, int n, const int32_t* word_ids);
/* _removedIndexedWords (removeWords :1595-1607): rows are tombstoned at once (never returned by a search) */
int lcd_vocab_remove(lcd_engine* h, const int32_t* word_ids, int n);
/* full-rebuild branch :610-690: drop tombstones and reorder the live rows by ascending word id, on the device */
int lcd_vocab_rebuild(lcd_engine* h);
/* rows = rows in the matrix incl. tombstones, live = searchable rows */
int lcd_vocab_count(const lcd_engine* h, int64_t* rows, int64_t* live);
/* read back rows [first, first+n) and their word ids (0 = tombstone); either output may be NULL */
int lcd_vocab_read(lcd_engine* h, int64_t first, int n, void* out_rows, int32_t* out_word_ids);

/* ---------------------------------------------------------------------------------------------------------------
 * exact 2-NN == FlannIndex::knnSearch(k=2) (FlannIndex.cpp:701, linear index) == cv::BFMatcher::knnMatch(k=2)
 * (VWDictionary.cpp:1027-1028) == the cv::cuda brute-force matcher (:1053-1066, which re-uploads the vocabulary per
 * call; here it stays resident).  out_word_ids/out_dist are [q*2]; a missing neighbour is (0, -1.0f). */
int lcd_knn2(lcd_engine* h, const void* queries, int q, int32_t* out_word_ids, float* out_dist);

/* q x q distance matrix of a descriptor block against itself (same metric/arithmetics as lcd_knn2).
 * The reference computes these distances one cv::BFMatcher call per descriptor (VWDictionary.cpp:1140-1160). */
int lcd_selfdist(lcd_engine* h, const void* queries, int q, float* out_qxq);

/* VWDictionary::addNewWords search + decision loop (:1015-1219) without the bookkeeping:
 *   for each descriptor i (in order): candidates = indexed 2-NN (only if the vocabulary has >= 2 live rows, :1015)
 *   + [LCD_Q_NEW_WORDS_COMPARED] exact 2-NN among the descriptors j < i that became new words in this call (:1140);
 *   [LCD_Q_INCREMENTAL] new word iff fewer than 2 candidates or d1 > nndr_ratio * d2 (:1162-1183), else word = nearest;
 *   fixed dictionary: nearest word, or "no entry" when there is no candidate (:1211-1218).
 * out_word_ids[i] > 0  : existing word id (caller does addWordRef)
 * out_word_ids[i] < 0  : the (-out-1)-th new word of this call (caller assigns ++_lastWordId in that order, :1185)
 * out_word_ids[i] == 0 : fixed dictionary and no candidate (the reference emits no list entry)
 * out_n_new (may be NULL) receives the number of new words. */
int lcd_quantize(lcd_engine* h, const void* descriptors, int q, int flags, float nndr_ratio,
                 int32_t* out_word_ids, int32_t* out_n_new);

/* VWDictionary::findNN(cv::Mat) (:1273-1552): indexed 2-NN + exact 2-NN (1-NN if one row) over the caller's
 * not-yet-indexed words (`extra_rows` x dim with ids `extra_word_ids`, ascending id like _notIndexedWords; may be
 * NULL/0) + NNDR (LCD_Q_INCREMENTAL) -> out_word_ids[i] = matched word id or 0.  Read-only. */
int lcd_find_nn(lcd_engine* h, const void* queries, int q, const void* extra_rows, const int32_t* extra_word_ids,
                int n_extra, int flags, float nndr_ratio, int32_t* out_word_ids);

/* ---------------------------------------------------------------------------------------------------------------
 * inverted index == VisualWord::_references of every word (VisualWord.h:62) + Memory::getNi (Memory.cpp:4955).
 * The engine is signature-granular: a signature's word list is registered once and retired once. */

/* one signature's references: word_ids[n] in keypoint order, duplicates = occurrences (== n x addWordRef :880 for
 * ids > 0; ids <= 0 are features without a word: they only count in ni).  ni = Signature::getWords().size()
 * (Memory.cpp:4961), normally n.  The signature must not be registered already. */
int lcd_sig_add(lcd_engine* h, int32_t sig_id, const int32_t* word_ids, int n, int32_t ni);
/* Memory::disableWordsRef (:6877-6897) == removeAllWordRef(word, sig) for every word of the signature */
int lcd_sig_remove(lcd_engine* h, int32_t sig_id);
/* bulk registration (Memory::loadDataFromDb replay, Memory.cpp:447-480): sig_offsets[n_sigs+1] index word_ids */
int lcd_sig_add_bulk(lcd_engine* h, int n_sigs, const int32_t* sig_ids, const int64_t* sig_offsets,
                     const int32_t* word_ids, const int32_t* ni);
int lcd_sig_count(const lcd_engine* h, int64_t* live_signatures, int64_t* postings);
/* nw = VisualWord::getReferences().size() of a word (0 if unknown) */
int lcd_word_nrefs(lcd_engine* h, int32_t word_id, int32_t* out_nw);

/* Memory::computeLikelihood(signature, ids), TF-IDF branch (Memory.cpp:2215-2291):
 *   out[k] = sum over unique word ids w > 0 of the query of  (nwi(w, sig_ids[k]) * log10(N / nw(w))) / ni(sig_ids[k])
 * query_word_ids[nq]: the query signature's words (any order, duplicates allowed, ids <= 0 ignored);
 * sig_ids[n_ids]: the signatures to score (unknown / retired ids and the virtual place score 0);
 * N = (float)Memory::getSignatures().size() as the caller counts it (:2248).  out[n_ids] pairs with sig_ids. */
int lcd_likelihood(lcd_engine* h, const int32_t* query_word_ids, int nq, const int32_t* sig_ids, int n_ids,
                   float N, float* out);

/* Rtabmap::adjustLikelihood (Rtabmap.cpp:5691-5760) on a likelihood vector whose entry 0 is the virtual place;
 * in/out on the host, reduction on the device.  ("next" row f1 of the scope table) */
int lcd_adjust_likelihood(lcd_engine* h, float* likelihood, int n, float virtual_place_ratio);
/* the same in place on a DEVICE vector, enqueued on the engine stream and not synchronised: with the likelihood of lcd_frame_dev
 * written at d_likelihood + 1 and the virtual place's value at d_likelihood[0], the adjusted vector never leaves the device */
int lcd_adjust_likelihood_dev(lcd_engine* h, float* d_likelihood, int n, float virtual_place_ratio);

/* ---------------------------------------------------------------------------------------------------------------
 * device-resident frame path (no host round trip; what bench.py times).  All pointers are DEVICE pointers valid on
 * the engine's device; work is enqueued on the engine stream and NOT synchronised.
 *
 * lcd_frame_dev == Memory::update's quantisation (addNewWords :913) -> register the frame's references ->
 * Memory::computeLikelihood against every live signature:
 *   d_descriptors [q x dim]; out d_word_ids[q] as lcd_quantize; new words are NOT added to the vocabulary here
 *   (that is VWDictionary::update() of the next frame); if sig_id != 0 the frame is registered as signature sig_id
 *   (new words excluded: they reference only this signature and cannot score any other);
 *   d_likelihood[n_slots] receives the dense likelihood over signature slots (see lcd_slots_dev), the frame's own
 *   slot included.  N is the caller's signature count. */
int lcd_frame_dev(lcd_engine* h, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id,
                  float N, int32_t* d_word_ids, float* d_likelihood, int64_t likelihood_capacity);
/* lcd_knn2 with device-resident queries and outputs (d_word_ids[q*2], d_dist[q*2]); enqueued, not synchronised */
int lcd_knn2_dev(lcd_engine* h, const void* d_queries, int q, int32_t* d_word_ids, float* d_dist);
/* ---- vocabulary sharded by word-ID range over several engines/GPUs (one handle per rank).
 * Every rank holds a consecutive id range of the vocabulary and the references of those words; every rank registers
 * every signature (same order => same slots).  Per frame: (1) lcd_shard_knn2_dev on each rank, (2) all-gather of the
 * 16-byte candidate records, (3) lcd_shard_frame_dev on each rank (merge + same-frame resolution, replicated; registers the
 * frame with the words THIS rank owns; integer partial likelihood into d_lfix), (4) all-reduce(sum, int64) of d_lfix --
 * order-free, so the result equals the single-GPU one bit for bit -- (5) lcd_finalize_dev. */
typedef struct lcd_shard_cand { uint64_t key; int32_t word; int32_t wslot; } lcd_shard_cand;
int lcd_shard_knn2_dev(lcd_engine* h, const void* d_descriptors, int q, lcd_shard_cand* d_cand /* [q*2] */);
int lcd_shard_frame_dev(lcd_engine* h, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id, float N,
                        int rank, int world, const lcd_shard_cand* d_all_cand /* [world*q*2], rank-major */,
                        int64_t total_live_rows, int32_t* d_word_ids, int64_t* d_lfix, int64_t lfix_capacity);
int lcd_finalize_dev(lcd_engine* h, int64_t* d_lfix, int64_t n, float* d_likelihood);
/* slot table: d_slot_sig[slot] = signature id (0 = retired slot), n_slots = number of slots in use */
int lcd_slots_dev(lcd_engine* h, const int32_t** d_slot_sig, int64_t* n_slots);
/* the engine's hipStream_t (so a caller can record events around enqueued work) */
void* lcd_stream(lcd_engine* h);

/* ---------------------------------------------------------------------------------------------------------------
 * kernel timing: while enabled, every launch of the dominant kernel of a frame (the 2-NN scan: MFMA filter, or the VALU
 * scan when the filter does not apply) is bracketed by a pair of HIP events on the engine stream.  lcd_profile_read
 * synchronises, returns the average duration in milliseconds and the number of samples, and disables profiling. */
int lcd_profile_begin(lcd_engine* h, int max_samples);
int lcd_profile_read(lcd_engine* h, float* avg_ms, int* n_samples, const char** kernel_name);
/* the same for the fused likelihood kernel of lcd_frame_dev (both series are recorded while profiling is enabled) */
int lcd_profile_read_likelihood(lcd_engine* h, float* avg_ms, int* n_samples, const char** kernel_name);

/* ---------------------------------------------------------------------------------------------------------------
 * statistics (names follow Statistics.h:178,202,209-212 where one exists) */
typedef struct lcd_stats {
    int64_t vocab_rows, vocab_live;        /* Keypoint/Dictionary_size */
    int64_t signatures, postings;
    int64_t knn_launches, likelihood_launches, rebuilds;
    int64_t bytes_device;                  /* HBM held by the handle */
    int64_t knn_last_fallback_queries;     /* queries of the LAST 2-NN call that the MFMA certificate sent to the exact scan */
    double knn_max_err_ratio;              /* largest |filter score - exact distance| / eps seen by the re-rank so far (must stay < 1) */
} lcd_stats;
/* synchronises the engine stream (the fallback counter lives on the device) */
int lcd_get_stats(lcd_engine* h, lcd_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* LCD_H_ */
