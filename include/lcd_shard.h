/* lcd_shard.h -- C-ABI of the multi-GPU driver (liblcd_shard.so): the vocabulary sharded by word-ID range over the GPUs of one node,
 * one process (and one lcd_engine, include/lcd.h) per GPU, the two per-frame exchanges on RCCL over xGMI (SURVEY.md section 8e;
 * BASELINE.json north_star: "C++ host code ... with an RCCL all-reduce over xGMI of the per-node likelihood vector").
 *
 * The reference has no multi-GPU path: this is what its VWDictionary::addNewWords (VWDictionary.cpp:913-1229) + Memory::computeLikelihood
 * (Memory.cpp:2215-2291) become when rank r holds the rows of word ids in its range and the references of those words:
 *     lcd_shard_knn2_dev on every rank (local exact 2-NN)           -> ncclAllGather of q x 2 records of 16 bytes
 *     lcd_shard_frame_dev (merge, same-frame resolution -- replicated --, registration and integer scoring of the owned words)
 *                                                                    -> ncclAllReduce(sum, int64) of the partial likelihood
 *     lcd_finalize_dev (fixed point -> float, / ni)
 * Integer partial sums make the reduction order-free: the likelihood equals the single-GPU one bit for bit whatever algorithm RCCL picks.
 * Everything is enqueued on the engine's stream (the deferred all-reduce on a second stream of the driver's own); nothing is synchronised.
 * extern "C", plain pointers, status codes of lcd.h. */
#ifndef LCD_SHARD_H_
#define LCD_SHARD_H_

#include "lcd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LCD_SHARD_ABI_VERSION 2

typedef struct lcd_shard_comm lcd_shard_comm;

/* 128 opaque bytes (ncclUniqueId): produced by ONE rank, handed to every rank's lcd_shard_comm_create by whatever transport the caller has */
int lcd_shard_unique_id(unsigned char out128[128]);
/* rank `rank` of `world` (1..64) around an existing engine of that rank's GPU; world == 1 needs no id (may be NULL) and no RCCL call */
int lcd_shard_comm_create(lcd_engine* engine, int rank, int world, const unsigned char id128[128], lcd_shard_comm** out);
/* The same driver over exchanges the CALLER provides instead of RCCL (a fabric RCCL does not cover; include/lcd_p2p.h fills this struct with
 * its one-shot peer-to-peer kernels: lcd_p2p_transport; the 2-rank tests of this repo, whose box has one GPU, also stage them through the host).  Both callbacks work on DEVICE buffers of this rank and must be ordered like a RCCL call
 * on `stream`: behind the work already enqueued there, in front of what is enqueued after they return (completing before they return
 * is one way to do that).  Return 0 on success. */
typedef struct lcd_shard_transport {
    int32_t struct_size;       /* sizeof(lcd_shard_transport) */
    int32_t reserved;
    void* user;
    /* recv[r * bytes_per_rank ..) = rank r's send[0 .. bytes_per_rank) for every rank r (rank-major, this rank included) */
    int (*all_gather)(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream);
    /* buf[i] = sum over the ranks of buf[i], 64-bit integers, in place */
    int (*all_reduce_sum_i64)(void* user, void* d_buf, size_t count, void* stream);
} lcd_shard_transport;
int lcd_shard_comm_create_transport(lcd_engine* engine, int rank, int world, const lcd_shard_transport* transport, lcd_shard_comm** out);
void lcd_shard_comm_destroy(lcd_shard_comm* c);
const char* lcd_shard_last_error(const lcd_shard_comm* c);

/* Balanced growth ("block-cyclic so growth stays balanced"): the words frames create -- ids >= first_incremental_id --
 * belong to rank ((id - first_incremental_id) / block) % world instead of the last rank; ties between equally distant words then break
 * by word id (the single-GPU row order when every rank appends its words in ascending id).  Same values on every rank, before the first
 * frame.  block == 0: back to "the last rank owns them".  lcd_shard_owner_of tells the caller which rank's lcd_vocab_append receives the
 * row of a word VWDictionary::update() indexes (the initial vocabulary keeps the consecutive ranges it was loaded with: -1). */
int lcd_shard_set_growth(lcd_shard_comm* c, int32_t first_incremental_id, int32_t block);
int lcd_shard_owner_of(const lcd_shard_comm* c, int32_t word_id);
/* VWDictionary::update()'s append branch on the device, sharded (v3): with on = 1 every frame that creates words (LCD_Q_INCREMENTAL,
 * first_new_word_id > 0) leaves the words THIS rank owns as rows of its shard before the next frame is searched -- no
 * lcd_shard_owner_of / lcd_vocab_append round trip through the caller, no read-back; the host's row mirror catches up when the next
 * frame's search is planned (one stream synchronisation).  Same value on every rank.  What lcd_frame_args.append_new_words is to one GPU
 * (Memory.cpp:1004-1016: update() runs before every addNewWords). */
int lcd_shard_set_append(lcd_shard_comm* c, int on);

/* One frame through the sharded path (arguments as lcd_frame_args / lcd_shard_frame_dev; total_live_rows = live vocabulary rows over ALL
 * ranks, VWDictionary.cpp:1015).  d_word_ids[q] and d_likelihood[likelihood_capacity >= slots after the frame] are device buffers of this
 * rank; every rank receives the same word ids and the same likelihood.  Enqueued on the engine's stream. */
int lcd_shard_frame(lcd_shard_comm* c, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id, int32_t first_new_word_id,
                    float N, int64_t total_live_rows, int32_t* d_word_ids, float* d_likelihood, int64_t likelihood_capacity);
/* The same frame with its all-reduce left running on the driver's second stream, UNDER the nearest-neighbour search of the next frame
 * (hard part 9): d_likelihood of this call is written by the NEXT lcd_shard_frame_deferred / lcd_shard_frame call --
 * after that frame's search and all-gather, before it touches the index -- or by lcd_shard_flush; keep two likelihood buffers and
 * alternate.  d_word_ids is complete (enqueued) when the call returns.  Retirements made through lcd_shard_sig_remove while a likelihood
 * is owed wait until it is final, so every frame sees the memory the undeferred order gives: results are bit-identical. */
int lcd_shard_frame_deferred(lcd_shard_comm* c, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id,
                             int32_t first_new_word_id, float N, int64_t total_live_rows, int32_t* d_word_ids, float* d_likelihood,
                             int64_t likelihood_capacity);
/* finalise the likelihood a deferred frame still owes (enqueued on the engine's stream; no-op if none) and apply the queued retirements */
int lcd_shard_flush(lcd_shard_comm* c);
/* lcd_sig_remove on this rank's engine, queued behind an owed likelihood (every rank makes the same calls) */
int lcd_shard_sig_remove(lcd_shard_comm* c, int32_t sig_id);

#ifdef __cplusplus
}
#endif
#endif /* LCD_SHARD_H_ */
