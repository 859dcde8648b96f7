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

#define LCD_ABI_VERSION 7

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

/* how the squared-L2 2-NN of 64-float descriptors is computed.  Every mode returns the SAME bits (the reference's distances and
 * tie-break): the matrix-core modes only rank candidates, an exact re-rank in the reference's arithmetic plus a completeness
 * certificate (exact redo when it fails) produces the result.  Other descriptor types always use the exact scan. */
enum lcd_knn_mode {
    LCD_KNN_DEFAULT = 0,       /* = LCD_KNN_BF16X3 where it applies */
    LCD_KNN_EXACT_VALU = 1,    /* exact vector-ALU scan only */
    LCD_KNN_F32_MFMA = 2,      /* fp32 matrix-core filter (v_mfma_f32_32x32x2_f32) + exact re-rank */
    LCD_KNN_BF16X3 = 3,        /* bf16 matrix-core filter, three bf16 products per fp32 product + exact re-rank */
    LCD_KNN_F16 = 4            /* fp16 matrix-core filter, ONE product per fp32 product (operands rounded to IEEE half: a third of the
                                  matrix work, an error bound of ~2^-10 (|q|^2 + |v|^2) instead of ~2^-14) + exact re-rank.  Made for
                                  unit-scale descriptors (SURF/SIFT are L2-normalised); queries whose certificate the wider bound
                                  cannot give -- and descriptors beyond half's range -- go to the exact scan, so the results stay the
                                  same bits; a vocabulary of near-duplicate words makes that the common case and this mode the slower one */
};

typedef struct lcd_config {
    int32_t struct_size;       /* sizeof(lcd_config), for ABI evolution */
    int32_t device;            /* HIP device ordinal */
    int32_t dtype;             /* lcd_dtype */
    int32_t dim;               /* columns: floats (LCD_F32) or bytes (LCD_U8) per descriptor */
    int64_t vocab_capacity;    /* initial row capacity (grows on demand) */
    int64_t sig_capacity;      /* initial signature-slot capacity (grows on demand) */
    int32_t max_queries;       /* initial per-call query capacity (Kp/MaxFeatures; grows on demand) */
    int32_t knn_mode;          /* lcd_knn_mode, per handle */
    void*   stream;            /* optional hipStream_t to enqueue on; NULL = engine-owned stream */
    int32_t pipeline;          /* 1: consecutive lcd_frame_dev calls are software-pipelined (matrix-core 2-NN handles), lcd_pipeline_depth() = 3
                                  frames deep: the call for frame t only converts its descriptors into matrix-core operands; its launches
                                  carry the distance filter + re-rank of frame t - 1, the decision loop of frame t - 2 and the registration +
                                  scoring of frame t - 3, whose single-workgroup latency chains hide behind the filter.  Consequence for the
                                  caller: the outputs of a frame (d_word_ids, d_likelihood, d_bayes, ...) are written -- and its
                                  descriptors read -- by work that the NEXT THREE lcd_frame_dev calls enqueue (or any other call on the
                                  handle, which completes the owed stages first; lcd_synchronize to wait for them): keep
                                  lcd_pipeline_depth() + 1 sets of buffers and rotate.  lcd_sig_remove, lcd_record_event and
                                  lcd_bayes_set_neighbors are queued behind the owed stages of the frame they follow, so they keep their
                                  place in the call order.  Results are identical with and without. */
    int32_t reserved1;
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
/* 0 for a plain handle; for a pipelined one (lcd_config.pipeline) the number of later lcd_frame_dev calls that still enqueue work of a
 * frame: its outputs are complete (enqueued) once that many further frames have been submitted, or after any other call */
int  lcd_pipeline_depth(const lcd_engine* h);

/* ---------------------------------------------------------------------------------------------------------------
 * vocabulary == VWDictionary::_dataTree + _mapIndexId (VWDictionary.h:146-149), maintained by update() :475-701.
 * Rows live in HBM; the row ORDER is part of the contract because it is the distance tie-break. */

/* VWDictionary::clear() :843-873 / the reset at the top of the rebuild branch :612-615 */
int lcd_vocab_clear(lcd_engine* h);
/* brute-force append branch :571-609: rows appended in the given order; word_ids[i] > 0, not already present */
int lcd_vocab_append(lcd_engine* h, const void* rows, int n, const int32_t* word_ids);
/* VWDictionary::removeWords (:1595-1607, _removedIndexedWords): rows are tombstoned at once (never returned by a search) and
 * the words cease to exist.  As in the reference, only words without references are removed (its callers pass getUnusedWords(),
 * Memory.cpp:2867,6906); the postings key of a removed word is recycled once the device has confirmed that. */
int lcd_vocab_remove(lcd_engine* h, const int32_t* word_ids, int n);
/* Memory::cleanUnusedWords (Memory.cpp:6899-6920: removeWords(getUnusedWords()), run by preUpdate before every frame of an incremental
 * dictionary) from the DEVICE's reference counts: every vocabulary row whose word no signature references is removed as by
 * lcd_vocab_remove.  For callers that keep no host copy of the references (device-resident frame streams).  out_word_ids (may be NULL
 * with capacity 0) receives up to `capacity` of the removed ids in ascending row order, *out_n their number.  Synchronises. */
int lcd_vocab_remove_unused(lcd_engine* h, int32_t* out_word_ids, int capacity, int32_t* out_n);
/* The same cleanUnusedWords WITHOUT completing or synchronising anything: one kernel, enqueued behind the work the handle has taken on so
 * far, tombstones every row whose word no signature references (row id 0, |row|^2 = +inf: no search finds it any more) and logs it on the
 * device; the host's mirror of the rows and the postings keys of the removed words catch up the next time the handle is drained (any call
 * that completes the owed stages: lcd_vocab_count, lcd_vocab_rebuild, lcd_synchronize ...).  On a pipelined handle with frames in flight
 * the clean takes its place behind the newest frame -- its registration and the lcd_sig_remove calls made since, like those calls
 * themselves -- so it is what Memory::preUpdate (Memory.cpp:1004-1010) runs in front of the NEXT frame's update(); the frames already in
 * flight behind it (up to lcd_pipeline_depth()) took their snapshot of the vocabulary earlier: a word they still matched keeps the
 * references of their signatures (its key is recycled once those are gone) but is never matched again.  A caller that needs the
 * reference's order exactly calls lcd_vocab_remove_unused (or drains) instead. */
int lcd_vocab_remove_unused_async(lcd_engine* h);
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
/* bulk registration (Memory::loadDataFromDb replay, Memory.cpp:447-480): sig_offsets[n_sigs+1] index word_ids.  One
 * registration launch for the whole call and a fixed number of launches per 64 full buckets (16 384 signatures) sealed. */
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
 * the engine's device; work is enqueued and NOT synchronised (lcd_synchronize, or synchronise the engine stream).
 *
 * lcd_frame_dev == Memory::update's quantisation (addNewWords :913) -> the frame's references (addWordRef :880 for
 * existing words, the VisualWord constructor's addRef :1185 for new ones) -> Memory::computeLikelihood (Memory.cpp:2177)
 * against every live signature -> optionally Rtabmap::adjustLikelihood (Rtabmap.cpp:5691) and the best candidate.
 * New words are NOT added to the vocabulary here (that is VWDictionary::update() of the next frame: lcd_vocab_append). */
typedef struct lcd_hypothesis {
    int32_t sig_id;            /* signature with the highest likelihood among the considered ones (0: none is positive) */
    int32_t slot;              /* its slot (-1: none) */
    float likelihood;          /* its raw likelihood */
    float adjusted;            /* its value after adjustLikelihood (1.0 when it is not above mean + stddev) */
    float virtual_place;       /* adjustLikelihood's value for the virtual place (entry -1 of the reference's map) */
    float mean, stddev;        /* over the positive likelihoods considered (uMean / sqrt(uVariance), UMath.h:419,512) */
    int32_t n_positive;
} lcd_hypothesis;

#define LCD_NEW_WORD_IDS_AUTO (-1)     /* lcd_frame_args.first_new_word_id: the device numbers the frame's new words (see there) */
typedef struct lcd_frame_args {
    int32_t struct_size;               /* sizeof(lcd_frame_args) */
    int32_t q;                         /* descriptors in the frame (1..8192) */
    const void* d_descriptors;         /* [q x dim], 16-byte aligned (rows are read as 16-byte vectors) */
    int32_t flags;                     /* lcd_quantize_flags */
    float nndr_ratio;
    int32_t sig_id;                    /* != 0: register the frame as this signature (it must not exist yet) */
    int32_t first_new_word_id;         /* the id the caller gives the frame's first new word (VWDictionary::_lastWordId + 1); the
                                          k-th new word (descriptor order, the -(k+1) codes of d_word_ids) is first_new_word_id + k,
                                          and consecutive frames must number their new words consecutively, as ++_lastWordId does
                                          (:1185).  The frame's signature then references its new words as well, so that a later
                                          frame matching one of them -- after lcd_vocab_append -- scores this signature.
                                          0: new words get no references (fixed dictionary / caller never indexes them).
                                          LCD_NEW_WORD_IDS_AUTO (needs append_new_words): the DEVICE numbers the words, exactly as ++_lastWordId does, for a
                                          caller that does not read back how many words a frame created before it submits the next one: the id of a new
                                          word is its vocabulary row + (next word id - rows) as they stood when the run of appending frames began -- every
                                          new word is one row and one id.  The run starts from lcd_set_option(h, "next_word_id", _lastWordId + 1), or from one
                                          past the highest id the handle has seen; the frame's first id is written to d_first_new_word_id.  (Not in the
                                          sharded entry points.  ABI v7.) */
    float N;                           /* Memory::getSignatures().size() as the caller counts it (Memory.cpp:2248) */
    int32_t exclude_recent;            /* hypothesis only: the newest `exclude_recent` slots (short-term memory + this frame,
                                          Rtabmap.cpp:2050-2117 compares against the working memory only) are not considered */
    int32_t* d_word_ids;               /* out [q], as lcd_quantize */
    float* d_likelihood;               /* out, may be NULL: dense likelihood over signature slots [n_slots] (lcd_slots_dev), the
                                          frame's own slot included */
    int64_t likelihood_capacity;       /* floats available at d_likelihood */
    lcd_hypothesis* d_hypothesis;      /* out, may be NULL (needs d_likelihood): 32 bytes instead of the whole vector to the host */
    float* d_adjusted;                 /* out, may be NULL: [n_slots + 1], entry 0 = virtual place, entry 1 + slot = adjusted value
                                          (0 for slots that are retired or not considered) */
    float virtual_place_ratio;         /* Rtabmap/VirtualPlaceLikelihoodRatio (0 = default branch) */
    int32_t append_new_words;          /* 1 (needs first_new_word_id > 0, LCD_Q_INCREMENTAL, 64-float rows): the words this frame creates become
                                          vocabulary rows ON THE DEVICE, in descriptor order behind the rows that exist, before the next frame is
                                          searched == VWDictionary::update()'s append branch (:571-609) run by Memory::preUpdate of the next
                                          frame (Memory.cpp:1004-1016) -- no lcd_vocab_append, no host round trip.  On a pipelined handle the next
                                          frame's filter has already taken its snapshot by then: its re-rank scans the appended rows exactly, so
                                          the result is the 2-NN over the updated vocabulary.  Removals (lcd_vocab_remove / lcd_vocab_rebuild,
                                          cleanUnusedWords) stay host calls that complete the owed stages first. */
    int32_t* d_first_new_word_id;      /* out, may be NULL (device memory; chained frames: append_new_words on a handle whose rows live on the device): the id of
                                          the frame's first new word -- first_new_word_id itself, or what LCD_NEW_WORD_IDS_AUTO resolved to (the -(k+1) codes of
                                          d_word_ids are this + k).  (ABI v7: the field was a reserved pointer, NULL.) */
    float* d_posterior;                /* out, may be NULL (needs d_likelihood and lcd_bayes_configure): the Bayes filter's posterior
                                          after this frame, [n_slots + 1], entry 0 = virtual place, 0 for slots that are retired or
                                          not considered (BayesFilter::computePosterior, BayesFilter.cpp:145-235) */
    struct lcd_bayes_result* d_bayes;  /* out, may be NULL: the highest loop-closure hypothesis (Rtabmap.cpp:2147-2158), 32 bytes */
} lcd_frame_args;
int lcd_frame_dev(lcd_engine* h, const lcd_frame_args* args);

/* The same frame for a caller whose descriptors live in HOST memory -- what corelib hands VWDictionary::addNewWords (a cv::Mat,
 * VWDictionary.cpp:913) and gets back from Memory::computeLikelihood (Memory.cpp:2177): ONE call, one synchronisation.  The descriptors
 * are copied to the device (engine-owned pinned staging), the frame runs as lcd_frame_dev describes (quantisation -> the signature's
 * references -> [append_new_words: the words it creates become vocabulary rows, VWDictionary::update()'s append branch] -> TF-IDF
 * likelihood against every registered signature), and the word ids and the dense likelihood over the signature slots come back.
 * Slots are handed out in registration order (lcd_sig_add, lcd_sig_add_bulk in array order, frames) and never reused; a retired
 * signature's slot scores 0.  What the frames in flight of a pipelined handle owe is completed first and this frame is completed
 * before the call returns (a host caller needs its answer: use a plain handle).  (ABI v5; the reference-side caller is
 * rtabmap_amd/host/VWDictionaryHip::addNewWordsAndScore.) */
typedef struct lcd_frame_host_args {
    int32_t struct_size;               /* sizeof(lcd_frame_host_args) */
    int32_t q;                         /* descriptors in the frame (1..8192) */
    const void* descriptors;           /* HOST [q x dim], row-major, as cv::Mat::data of a continuous matrix */
    int32_t flags;                     /* lcd_quantize_flags */
    float nndr_ratio;
    int32_t sig_id;                    /* != 0: register the frame as this signature */
    int32_t first_new_word_id;         /* as lcd_frame_args */
    float N;                           /* as lcd_frame_args */
    int32_t append_new_words;          /* as lcd_frame_args */
    int32_t* word_ids;                 /* HOST out [q], as lcd_quantize */
    float* likelihood;                 /* HOST out, may be NULL: [likelihood_capacity], entry = signature slot */
    int64_t likelihood_capacity;       /* floats available at likelihood (>= slots after this frame, else LCD_ERR_INVALID) */
    int64_t* n_slots;                  /* HOST out, may be NULL: slots in use after this frame = entries written */
} lcd_frame_host_args;
int lcd_frame_host(lcd_engine* h, const lcd_frame_host_args* args);
/* slots in use (host bookkeeping, nothing is synchronised or completed): what lcd_frame_host's likelihood_capacity must cover is this + 1 */
int lcd_slot_count(const lcd_engine* h, int64_t* n_slots);

/* ---------------------------------------------------------------------------------------------------------------
 * Bayes filter over the signatures of the working memory ("next" row f2 of the scope table).
 * == BayesFilter (corelib/src/BayesFilter.cpp): the posterior lives on the device, one float per signature slot; the
 * reference's dense m x m prediction matrix (:313) is never formed -- each column's non-zeros are evaluated from the
 * signature's graph-neighbour list.  The considered signatures are the live slots below n_slots - exclude_recent, as for
 * the hypothesis of lcd_frame_dev (the reference passes the working memory's likelihood, Rtabmap.cpp:2050-2133). */
typedef struct lcd_bayes_result {
    int32_t sig_id;            /* _highestHypothesis.first: the considered signature with the highest posterior (0: none > 0) */
    int32_t slot;              /* its slot (-1: none) */
    float posterior;           /* its posterior */
    float value;               /* _highestHypothesis.second = 1 - posterior of the virtual place (Rtabmap.cpp:2157) */
    float virtual_place;       /* posterior of the virtual place */
    int32_t n_considered;      /* signatures that took part (the reference's likelihood.size() - 1) */
    float sum;                 /* normalisation constant of this update (BayesFilter.cpp:221) */
    int32_t reserved;
} lcd_bayes_result;
/* BayesFilter::setPredictionLC (:77-122) + Bayes/VirtualPlacePriorThr: prediction_lc = {virtual place, loop closure, neighbour
 * level 1, level 2, ...} as getPredictionLC() returns them (2..32 values in [0, 1]).  Needed before any update. */
int lcd_bayes_configure(lcd_engine* h, const double* prediction_lc, int n_values, float virtual_place_prior);
/* BayesFilter::reset (:138-143): forget the posterior and the neighbour lists */
int lcd_bayes_reset(lcd_engine* h);
/* The graph neighbourhood of n_sigs registered signatures: for signature sig_ids[i] the entries [offsets[i], offsets[i+1]) of
 * nbr_sig_ids / nbr_margins are what Memory::getNeighborsId(id, prediction_lc.size() - 1, 0, false, false, true, true) returned
 * (BayesFilter.cpp:328, :581) -- the signature itself with margin 0 included, margins in [0, n_values - 2].  The list REPLACES
 * what the engine held for that signature (the reference's _neighborsIndex entry of a new id is exactly this answer), and like the
 * reference's incremental update (:583-592) every entry is also entered into the neighbour's own list (an existing entry for the
 * same pair is replaced) -- so a signature's list is passed once, when it enters the working memory, and keeps growing through the
 * lists of the signatures that come after it; passing every list again (Bayes/FullPredictionUpdate = true, :328-352) rebuilds them
 * all.  Neighbours that are not registered are skipped (they are in the long-term memory and can not be in a likelihood).  Host
 * pointers. */
int lcd_bayes_set_neighbors(lcd_engine* h, int n_sigs, const int32_t* sig_ids, const int64_t* offsets, const int32_t* nbr_sig_ids,
                            const int32_t* nbr_margins);
/* One filter update from an adjusted likelihood that is already on the device: d_adjusted[n_slots + 1] laid out as lcd_frame_dev
 * writes it (entry 0 = virtual place, entry 1 + slot).  Enqueued, not synchronised.  d_posterior (may be NULL) like d_adjusted. */
int lcd_bayes_update_dev(lcd_engine* h, const float* d_adjusted, int exclude_recent, float* d_posterior, lcd_bayes_result* d_result);
/* BayesFilter::computePosterior(memory, likelihood) (:145-235) with the likelihood as the std::map hands it out: n parallel host
 * entries in ascending id order, the virtual place (id -1) first (Rtabmap.cpp:2111-2115 always compares against it; a likelihood
 * without it is LCD_ERR_UNSUPPORTED).  The other ids must be every registered signature up to the newest one named -- the working
 * memory without the short-term memory, the list Rtabmap.cpp:2046-2115 builds.  Synchronises; *result (may be NULL) receives the
 * highest hypothesis (Rtabmap.cpp:2147-2158); lcd_bayes_posterior reads the vector. */
int lcd_bayes_update(lcd_engine* h, const int32_t* sig_ids, const float* adjusted, int n, lcd_bayes_result* result);
/* the filter's current posterior for some signatures (host arrays; unknown / never considered signatures -> 0); sig id -1 = the
 * virtual place.  Synchronises. */
int lcd_bayes_posterior(lcd_engine* h, const int32_t* sig_ids, int n, float* out);
/* lcd_knn2 with device-resident queries and outputs (d_word_ids[q*2], d_dist[q*2]); enqueued, not synchronised */
int lcd_knn2_dev(lcd_engine* h, const void* d_queries, int q, int32_t* d_word_ids, float* d_dist);
/* ---- vocabulary sharded by word-ID range over several engines/GPUs (one handle per rank).
 * Every rank holds a consecutive id range of the vocabulary and the references of those words; every rank registers
 * every signature (same order => same slots).  Per frame: (1) lcd_shard_knn2_dev on each rank, (2) all-gather of the
 * 16-byte candidate records, (3) lcd_shard_frame_dev on each rank (merge + same-frame resolution, replicated; registers the
 * frame with the words THIS rank owns; integer partial likelihood into d_lfix), (4) all-reduce(sum, int64) of d_lfix --
 * order-free, so the result equals the single-GPU one bit for bit -- (5) lcd_finalize_dev.
 * (1) and (3) of a frame are one pair: the search also leaves the frame's same-frame distance matrix on the handle (it rides in the
 * filter's launch), and the frame call that follows it with the SAME d_descriptors pointer and q -- their content unchanged in between --
 * takes it from there; any other frame call computes the matrix itself. */
typedef struct lcd_shard_cand { uint64_t key; int32_t word; int32_t wslot; } lcd_shard_cand;
int lcd_shard_knn2_dev(lcd_engine* h, const void* d_descriptors, int q, lcd_shard_cand* d_cand /* [q*2] */);
int lcd_shard_frame_dev(lcd_engine* h, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id,
                        int32_t first_new_word_id /* as lcd_frame_args; new words belong to the LAST rank, or block-cyclically to all of
                                                     them: lcd_set_option "shard_growth_first" / "shard_growth_block" */, float N,
                        int rank, int world, const lcd_shard_cand* d_all_cand /* [world*q*2], rank-major */,
                        int64_t total_live_rows, int32_t* d_word_ids, int64_t* d_lfix, int64_t lfix_capacity);
int lcd_finalize_dev(lcd_engine* h, int64_t* d_lfix, int64_t n, float* d_likelihood);
/* slot table: d_slot_sig[slot] = signature id (0 = retired slot), n_slots = number of slots in use */
int lcd_slots_dev(lcd_engine* h, const int32_t** d_slot_sig, int64_t* n_slots);
/* record a caller-owned hipEvent_t on the engine stream BEHIND everything the calls made so far will enqueue there (a pipelined
 * handle enqueues the registration / scoring of its latest frame with the next call: recording on lcd_stream() directly would
 * land in front of it) */
int lcd_record_event(lcd_engine* h, void* event);
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

/* tuning knobs for experiments (results never depend on them; every one of them is per handle).  "filter_delay": the filter workgroups
 * of a pipelined frame's launch A wait value x 64 clocks in front of their first request (0 .. 127; timing experiments).  "roctx": 1 = roctx ranges (see lcd_trace_push below).  "shadow_rows": 1 (built-in: while the stream creates 16 words per frame or more; 2 = always) = a frame that appends its words on the device also leaves its descriptors as rows
 * of an operand table, and the matrix-core filter of the NEXT frame ranks them beside the vocabulary (its re-rank keeps the ones that became words)
 * instead of every re-rank workgroup staging the new rows and scanning them (0).  "mirror_from_b": 1 (built-in) = the pinned row-count mirror of an appending
 * frame is stored by a workgroup of launch B instead of at the end of the decision loop's chain in launch A.  "row_writer_wgs": the rows a
 * frame appends are written by that many extra workgroups of launch B's re-rank role (built-in 16; 0 = by the re-rank workgroups themselves; 0 .. 256).  "slots_from_rows": 1 (built-in: while the stream creates 16 words per frame or more; 2 = always) = the decision loop of a
 * pipelined frame hands the registration the vocabulary ROW of every matched word and the registration looks the postings key up (in the
 * round trip that fetches the retired signature's words); 0 = the decision loop gathers the keys itself.  "score_block": threads per workgroup of the scoring kernel
 * (256 / 512 / 1024).  "filter_units": compute units the bf16 filter plans its persistent workgroups for when the vocabulary has
 * more 256-word strips than that (-1 built-in, 0 never persistent).  "decision_straight": 1 (built-in: while the stream creates 16 words per frame or more; 2 = always; 0 = never) = the decision loop of a pipelined frame requests
 * everything its first round trip reads unconditionally, in one straight line (faster while frames create words, slower once they only revisit: DESIGN.md 4d).  "next_word_id": one past the highest word id handed out so far (VWDictionary::_lastWordId + 1): where LCD_NEW_WORD_IDS_AUTO continues (never lowered: the handle
 * keeps the maximum of this and the ids of the rows it has seen).  "profile_skip": the number of launches of a pipelined handle that lcd_profile_begin lets pass before it brackets one (0; the first launches behind an idle
 * queue are not the steady state).  "profile_likelihood": 0 = lcd_profile_begin brackets only the
 * 2-NN launch of a pipelined frame (every timed launch costs stream time).  "strip_tiles": 32-word tiles per filter workgroup of a
 * pipelined frame (1 .. 8; 0 = the built-in plan).  "append_split_buckets": sealed buckets of 256 signatures from which
 * the rows a frame appends are written by a kernel of their own behind launch B instead of by workgroups inside it (-1 = built-in, 1 024).
 * "append_from_rerank": 1 (built-in) = those rows are written by the re-rank workgroups of launch B, 0 = by eight row-writer
 * workgroups.  "cross_frame_tiles": 1 = launch A also computes a frame's distances to the frame before it and the re-rank reads
 * the distances of the rows that frame appended from there instead of staging the rows (0 / -1 = built-in: staged).
 * Unknown keys / values -> LCD_ERR_INVALID.
 * The two keys that DO change what a call means (sharded handles only, identical on every rank): "shard_growth_first" = F and
 * "shard_growth_block" = B > 0 make lcd_shard_frame_dev give the words frames create (ids >= F) to rank ((id - F) / B) % world instead of
 * the last rank, and merge the gathered candidates with ties going to the lower WORD ID -- the single-GPU row order as long as every rank
 * appends its words in ascending id ("block-cyclic so growth stays balanced").  "shard_append" = 1 (what
 * lcd_shard_set_append of include/lcd_shard.h sets): lcd_shard_frame_dev also turns the new words this rank owns into rows of its shard,
 * on the device, from the replicated decision -- VWDictionary::update()'s append, per rank. */
int lcd_set_option(lcd_engine* h, const char* key, int64_t value);

/* ---------------------------------------------------------------------------------------------------------------
 * tracing (tracing row; ABI v6).  lcd_set_option("roctx", 1) loads libroctx64.so at run time (no link dependency;
 * LCD_ERR_UNSUPPORTED when it is not installed) and from then on the engine brackets what it enqueues with roctx ranges that
 * `rocprofv3 --marker-trace` shows beside the kernels: "lcd_frame_dev", "lcd:launch_A", "lcd:launch_B", "lcd:drain", "lcd_frame_host",
 * "lcd_likelihood", "lcd_quantize".  lcd_trace_push / lcd_trace_pop put a caller's own range on the same track (the host mirror brackets its
 * CPU-side stages with the reference's names: "Memory::update", "VWDictionary::addNewWords", "Memory::computeLikelihood", ...).  Both are
 * no-ops (LCD_OK) while the option is off.  The reference has ULOGGER_DEBUG timings at these places (Memory.cpp:5931,6062; Rtabmap.cpp:4357). */
int lcd_trace_push(lcd_engine* h, const char* name);
int lcd_trace_pop(lcd_engine* h);

/* the work of ONE scoring launch for the words of the last frame (diagnostic, synchronises): out8[0] bytes of dense count rows
 * read, [1] sparse postings read (4 B each), [2] directory lookups, [3] lookups that found the word, [4] entries of the open
 * bucket's log (8 B each), [5] postings of the frame's words over all live signatures (the P of SURVEY.md 8d), [6] unique
 * words of the frame, [7] of them with dense rows */
int lcd_profile_score_work(lcd_engine* h, int64_t* out8);

/* ---------------------------------------------------------------------------------------------------------------
 * statistics (names follow Statistics.h:178,202,209-212 where one exists) */
typedef struct lcd_stats {
    int64_t vocab_rows, vocab_live;        /* Keypoint/Dictionary_size */
    int64_t signatures, postings;
    int64_t knn_launches, likelihood_launches, rebuilds;
    int64_t buckets_sealed;                /* 256-signature blocks of the inverted index regrouped on the device */
    int64_t word_slots;                    /* postings keys in use (recycled when words are removed) */
    int64_t dense_words;                   /* words whose postings are kept as dense count rows (last value the device reported) */
    int64_t frame_calls, frame_host_ns;    /* lcd_frame_dev calls and the host time spent inside them (enqueue cost) */
    int64_t bytes_device;                  /* HBM held by the handle */
    int64_t knn_last_fallback_queries;     /* queries of the LAST 2-NN call that the MFMA certificate sent to the exact scan */
    double knn_max_err_ratio;              /* largest |filter score - exact distance| / eps seen by the re-rank so far (must stay < 1) */
    int64_t clean_divergent_refs;          /* (ABI v6) references registered to a word that an ENQUEUED cleanUnusedWords (lcd_vocab_remove_unused_async) had tombstoned
                                            * while the referencing frame was in flight: the one documented departure of the device-resident mode from the
                                            * reference, whose clean runs behind that frame's addNewWords and keeps the word (Memory.cpp:6899-6920).  0 on streams
                                            * that drain before they clean */
} lcd_stats;
/* synchronises the engine stream (the fallback counter lives on the device) */
int lcd_get_stats(lcd_engine* h, lcd_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* LCD_H_ */
