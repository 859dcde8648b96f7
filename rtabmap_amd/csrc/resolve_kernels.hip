// resolve_kernels.hip -- the per-descriptor decision loop of VWDictionary::addNewWords (reference VWDictionary.cpp:1089-1219)
// and the candidate merge of VWDictionary::findNN (:1457-1542), on the device, plus the small vocabulary maintenance
// kernels behind lcd_vocab_remove / lcd_vocab_rebuild (VWDictionary::update(), :571-690).
//
// addNewWords is sequential in the reference: descriptor i is also matched against the words created by descriptors
// j < i of the same call (Kp/NewWordsComparedTogether, :1140-1160).  That is a lower-triangular system
//     isNew[i] = f_i(isNew[0..i-1]),
// solved here by Jacobi sweeps over all descriptors in parallel until a sweep changes nothing.  Entry i is final
// once entries < i are, so after t sweeps the first t entries are exact and a sweep without change is the unique
// solution -- i.e. exactly the reference's sequential result (worst case q sweeps, in practice 2-4).
//
// One workgroup (1024 threads) handles the frame: the data is tiny (q x q distances from L2) and the work is
// latency-, not throughput-bound; the frame's heavy part is knn2_kernels.hip.
#include "lcd_kernels.h"

namespace lcd {
namespace {

constexpr int RBLOCK = 1024;
constexpr int LCD_Q_INCREMENTAL = 1;
constexpr int LCD_Q_NEW_WORDS_COMPARED = 2;

struct Cand { float d; int id; };   // id > 0: word id, id < 0: -(j+1) = the new word created by descriptor j

// std::multimap<float,int> insertion (equal keys keep insertion order, VWDictionary.cpp:1091) restricted to what is
// read afterwards: the two smallest entries.
__device__ __forceinline__ void cand_push(Cand& c0, Cand& c1, int& n, float d, int id) {
    if (n == 0) { c0.d = d; c0.id = id; }
    else if (d < c0.d) { c1 = c0; c0.d = d; c0.id = id; }
    else if (n == 1 || d < c1.d) { c1.d = d; c1.id = id; }
    ++n;
}

// candidates of descriptor i given the current guess of which earlier descriptors are new words
__device__ __forceinline__ void gather_candidates(int i, int flags, int have_index, const int32_t* __restrict__ knn_word,
                                                  const float* __restrict__ knn_dist, const float* __restrict__ selfdist,
                                                  int ld, const unsigned char* __restrict__ is_new, int jmax,
                                                  Cand& c0, Cand& c1, int& n) {
    n = 0;
    c0.d = 0.f; c0.id = 0; c1.d = 0.f; c1.id = 0;
    if (have_index) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {                // :1092-1137, stop at the first invalid neighbour
            const float d = knn_dist[2 * i + j];
            const int id = knn_word[2 * i + j];
            if (d >= 0.0f && id != 0) cand_push(c0, c1, n, d, id); else break;
        }
    }
    if (flags & LCD_Q_NEW_WORDS_COMPARED) {
        // exact 2-NN (1-NN when only one exists) among the new words created before i, lowest j on ties (:1140-1160)
        uint64_t b = KEY_NONE, s = KEY_NONE;
        for (int j = 0; j < jmax; ++j) {             // jmax is wave-uniform (>= i for every lane), is_new[j] uniform
            if (!is_new[j]) continue;
            if (j < i) {
                const uint64_t k = ((uint64_t)__float_as_uint(selfdist[(size_t)j * ld + i]) << 32) | (uint32_t)j;
                const uint64_t hi = b > k ? b : k;
                b = b < k ? b : k;
                s = s < hi ? s : hi;
            }
        }
        if (b != KEY_NONE) cand_push(c0, c1, n, __uint_as_float((uint32_t)(b >> 32)), -((int)(uint32_t)b + 1));
        if (s != KEY_NONE) cand_push(c0, c1, n, __uint_as_float((uint32_t)(s >> 32)), -((int)(uint32_t)s + 1));
    }
}

__global__ __launch_bounds__(RBLOCK) void resolve_kernel(int q, int flags, float nndr, int have_index,
                                                         const int32_t* __restrict__ knn_word, const float* __restrict__ knn_dist,
                                                         const float* __restrict__ selfdist, int ld,
                                                         int32_t* __restrict__ out_word, int32_t* __restrict__ out_n_new) {
    extern __shared__ unsigned char smem[];          // is_new[qpad] (current guess) | is_new[qpad] (next guess)
    const int qpad = (q + 15) / 16 * 16;
    unsigned char* is_new = smem;
    unsigned char* is_next = smem + qpad;
    __shared__ int s_changed;
    __shared__ int s_scan[RBLOCK];
    const int tid = threadIdx.x;
    const bool incremental = (flags & LCD_Q_INCREMENTAL) != 0;

    // sweep 0: decide from the indexed candidates only
    for (int i = tid; i < q; i += RBLOCK) {
        Cand c0, c1; int n;
        gather_candidates(i, flags & ~LCD_Q_NEW_WORDS_COMPARED, have_index, knn_word, knn_dist, selfdist, ld, is_new, 0, c0, c1, n);
        const bool reject = incremental && (n < 2 || c0.d > nndr * c1.d);
        is_new[i] = reject ? 1 : 0;
    }
    __syncthreads();
    if (incremental && (flags & LCD_Q_NEW_WORDS_COMPARED)) {
        for (int sweep = 0; sweep <= q; ++sweep) {
            if (tid == 0) s_changed = 0;
            __syncthreads();
            for (int i = tid; i < q; i += RBLOCK) {
                const int jmax = min(q, ((i | 63) + 1));   // same bound for the whole wave
                Cand c0, c1; int n;
                gather_candidates(i, flags, have_index, knn_word, knn_dist, selfdist, ld, is_new, jmax, c0, c1, n);
                const unsigned char v = (n < 2 || c0.d > nndr * c1.d) ? 1 : 0;
                is_next[i] = v;
                if (v != is_new[i]) s_changed = 1;
            }
            __syncthreads();
            unsigned char* t = is_new; is_new = is_next; is_next = t;
            if (!s_changed) break;
            __syncthreads();
        }
    }
    // ranks of the new words in descriptor order (getNextId() is called in that order, :1185)
    int base = 0;
    for (int i0 = 0; i0 < q; i0 += RBLOCK) {
        const int i = i0 + tid;
        const int v = (i < q && is_new[i]) ? 1 : 0;
        s_scan[tid] = v;
        __syncthreads();
        for (int off = 1; off < RBLOCK; off <<= 1) {
            const int t = tid >= off ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += t;
            __syncthreads();
        }
        if (i < q && v) out_word[i] = -(base + s_scan[tid] - 1 + 1);
        const int total = s_scan[RBLOCK - 1];
        __syncthreads();
        base += total;
    }
    if (tid == 0) out_n_new[0] = base;
    __syncthreads();
    // accepted descriptors: nearest candidate; a candidate that is itself a new word (-(j+1)) maps to that word's rank
    for (int i = tid; i < q; i += RBLOCK) {
        if (is_new[i]) continue;
        const int jmax = min(q, ((i | 63) + 1));
        Cand c0, c1; int n;
        gather_candidates(i, flags, have_index, knn_word, knn_dist, selfdist, ld, is_new, jmax, c0, c1, n);
        int w = 0;
        if (n > 0) {
            w = c0.id;
            if (w < 0) w = out_word[-w - 1];          // already final: written above
        }
        out_word[i] = w;                              // fixed dictionary without candidate: 0 ("no entry", :1211-1218)
    }
}

// findNN (:1457-1542): indexed candidates are NOT cut at the first invalid one; not-indexed candidates are
__global__ void findnn_resolve_kernel(int q, int flags, float nndr, int have_index, const int32_t* __restrict__ knn_word,
                                      const float* __restrict__ knn_dist, int have_extra, const int32_t* __restrict__ extra_word,
                                      const float* __restrict__ extra_dist, int32_t* __restrict__ out_word) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
    Cand c0, c1; int n = 0;
    c0.d = 0.f; c0.id = 0; c1.d = 0.f; c1.id = 0;
    if (have_index) {
        for (int j = 0; j < 2; ++j) {
            const float d = knn_dist[2 * i + j]; const int id = knn_word[2 * i + j];
            if (d >= 0.0f && id != 0) cand_push(c0, c1, n, d, id);
        }
    }
    if (have_extra) {
        for (int j = 0; j < 2; ++j) {
            const float d = extra_dist[2 * i + j]; const int id = extra_word[2 * i + j];
            if (d >= 0.0f && id != 0) cand_push(c0, c1, n, d, id); else break;
        }
    }
    int w = 0;
    if (flags & LCD_Q_INCREMENTAL) {
        if (n >= 2 && !(c0.d > nndr * c1.d)) w = c0.id;
    } else if (n > 0) {
        w = c0.id;
    }
    out_word[i] = w;
}

// ------------------------------------------------------------------------------------------------ vocabulary upkeep
// dst row i = src row perm[i]; a row is row_bytes/4 dwords; consecutive lanes copy consecutive dwords (coalesced)
__global__ void gather_rows_kernel(const uint32_t* __restrict__ src, const int32_t* __restrict__ src_id,
                                   const int32_t* __restrict__ perm, int n, int row_dwords,
                                   uint32_t* __restrict__ dst, int32_t* __restrict__ dst_id) {
    const size_t total = (size_t)n * row_dwords;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / row_dwords), c = (int)(e % row_dwords);
        const int sr = perm[r];
        dst[e] = src[(size_t)sr * row_dwords + c];
        if (c == 0) dst_id[r] = src_id[sr];
    }
}
__global__ void tombstone_kernel(int32_t* __restrict__ row_id, const int32_t* __restrict__ rows, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) row_id[rows[i]] = 0;
}

}  // namespace

hipError_t launch_resolve(int q, int flags, float nndr, int have_index, const int32_t* knn_word, const float* knn_dist,
                          const float* selfdist, int ld, int32_t* out_word, int32_t* out_n_new, hipStream_t s) {
    if (q <= 0) return hipSuccess;
    if (q > 8 * RBLOCK) return hipErrorInvalidValue;
    resolve_kernel<<<1, RBLOCK, (size_t)((q + 15) / 16 * 16) * 2, s>>>(q, flags, nndr, have_index, knn_word, knn_dist, selfdist, ld,
                                                                  out_word, out_n_new);
    return hipGetLastError();
}

hipError_t launch_findnn_resolve(int q, int flags, float nndr, int have_index, const int32_t* knn_word,
                                 const float* knn_dist, int have_extra, const int32_t* extra_word,
                                 const float* extra_dist, int32_t* out_word, hipStream_t s) {
    if (q <= 0) return hipSuccess;
    findnn_resolve_kernel<<<(q + 255) / 256, 256, 0, s>>>(q, flags, nndr, have_index, knn_word, knn_dist, have_extra,
                                                          extra_word, extra_dist, out_word);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const void* src, const int32_t* src_id, const int32_t* perm, int n, int row_bytes,
                              void* dst, int32_t* dst_id, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int row_dwords = row_bytes / 4;
    const size_t total = (size_t)n * row_dwords;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    gather_rows_kernel<<<blocks, 256, 0, s>>>((const uint32_t*)src, src_id, perm, n, row_dwords, (uint32_t*)dst, dst_id);
    return hipGetLastError();
}

hipError_t launch_tombstone(int32_t* row_id, const int32_t* rows, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    tombstone_kernel<<<(n + 255) / 256, 256, 0, s>>>(row_id, rows, n);
    return hipGetLastError();
}

}  // namespace lcd
