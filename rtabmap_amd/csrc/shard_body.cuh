// shard_body.cuh -- device code of the sharded search's candidate records (one rank per word-id range), shared by the
// stand-alone kernels (resolve_kernels.hip) and by the launches the two steps ride in: the records are packed at the end of the exact
// redo's launch (knn_mfma_kernels.hip) and merged at the head of the same-frame distance launch (knn2_kernels.hip).
#pragma once
#include "lcd_kernels.h"

namespace lcd {
namespace {

// local candidate i (of 2 * q) of one rank as a record
__device__ __forceinline__ void shard_pack_one(const ShardPackArgs& p, int i) {
    ShardCand c;
    const int row = p.knn_row[i];
    if (row < 0) { c.key = KEY_NONE; c.word = 0; c.wslot = -1; }
    else { c.key = ((unsigned long long)__float_as_uint(p.knn_dist[i]) << 32) | (uint32_t)row; c.word = p.knn_word[i]; c.wslot = p.row_wslot[row]; }
    p.out[i] = c;
}

// The global 2-NN of query i from the all-gathered records cand[world][q][2].  Global row order = (rank, local row): the lower rank, then the
// lower row, wins ties -- the single-GPU order when the shards are consecutive id ranges; by_word: (distance, word id) -- rows ascend by id on
// every rank, so this is the order one GPU holding all rows would see.  wslot[j] is the postings key if THIS rank owns neighbour j, else -1.
struct ShardMerged { int32_t word[2]; float dist[2]; int32_t wslot[2]; };
__device__ __forceinline__ ShardMerged shard_merge_one(const ShardCand* __restrict__ cand, int world, int rank, int q, int i, int by_word) {
    // composite (distance, rank, local row, slot index) compared lexicographically
    unsigned long long bk = KEY_NONE, sk = KEY_NONE;
    int bsrc = -1, ssrc = -1;
    for (int r = 0; r < world; ++r) {
        for (int j = 0; j < 2; ++j) {
            const ShardCand c = cand[((size_t)r * q + i) * 2 + j];
            if (c.key == KEY_NONE || c.word == 0) continue;
            const unsigned long long k = by_word ? ((c.key & 0xFFFFFFFF00000000ull) | (unsigned long long)(uint32_t)c.word)
                                                 : ((c.key & 0xFFFFFFFF00000000ull) | ((unsigned long long)r << 26) | (c.key & 0x3FFFFFFull));
            const int src = (r * q + i) * 2 + j;
            if (k < bk) { sk = bk; ssrc = bsrc; bk = k; bsrc = src; }
            else if (k < sk) { sk = k; ssrc = src; }
        }
    }
    ShardMerged m;
    const int srcs[2] = {bsrc, ssrc};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (srcs[j] < 0) { m.word[j] = 0; m.dist[j] = -1.0f; m.wslot[j] = -1; continue; }
        const ShardCand c = cand[srcs[j]];
        m.word[j] = c.word;
        m.dist[j] = __uint_as_float((uint32_t)(c.key >> 32));
        m.wslot[j] = (srcs[j] / (2 * q)) == rank ? c.wslot : -1;
    }
    return m;
}

}  // namespace
}  // namespace lcd
