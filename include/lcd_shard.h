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
 * Everything is enqueued on the engine's stream; nothing is synchronised.  extern "C", plain pointers, status codes of lcd.h. */
#ifndef LCD_SHARD_H_
#define LCD_SHARD_H_

#include "lcd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lcd_shard_comm lcd_shard_comm;

/* 128 opaque bytes (ncclUniqueId): produced by ONE rank, handed to every rank's lcd_shard_comm_create by whatever transport the caller has */
int lcd_shard_unique_id(unsigned char out128[128]);
/* rank `rank` of `world` (1..64) around an existing engine of that rank's GPU; world == 1 needs no id (may be NULL) and no RCCL call */
int lcd_shard_comm_create(lcd_engine* engine, int rank, int world, const unsigned char id128[128], lcd_shard_comm** out);
void lcd_shard_comm_destroy(lcd_shard_comm* c);
const char* lcd_shard_last_error(const lcd_shard_comm* c);
/* One frame through the sharded path (arguments as lcd_frame_args / lcd_shard_frame_dev; total_live_rows = live vocabulary rows over ALL
 * ranks, VWDictionary.cpp:1015).  d_word_ids[q] and d_likelihood[likelihood_capacity >= slots after the frame] are device buffers of this
 * rank; every rank receives the same word ids and the same likelihood.  Enqueued on the engine's stream. */
int lcd_shard_frame(lcd_shard_comm* c, const void* d_descriptors, int q, int flags, float nndr_ratio, int32_t sig_id, int32_t first_new_word_id,
                    float N, int64_t total_live_rows, int32_t* d_word_ids, float* d_likelihood, int64_t likelihood_capacity);

#ifdef __cplusplus
}
#endif
#endif /* LCD_SHARD_H_ */
