// tfidf.hip -- kernels and host-side management of the blocked inverted index (see tfidf.h for the layout).
//
// Reference behaviour reproduced (Memory.cpp:2215-2291): for every UNIQUE word id w > 0 of the query,
//   nw = refs(w).size(); logNnw = log10(N / nw) (float); skipped when nw == 0 or logNnw == 0;
//   for every (signature s, count nwi) in refs(w): ni = getNi(s); if ni != 0: L[s] += (nwi * logNnw) / ni   (all fp32).
// Every term is evaluated with exactly these fp32 operations; only the accumulation differs: the reference adds the
// terms of one signature in ascending word order in fp32, here they are added as Q15.48 integers (order-free,
// truncation error < 2^-48 per term), so results agree to ~1e-6 relative (bound 1e-4, tests/test_gpu_likelihood.py).
#include "tfidf.h"
#include "resolve_body.cuh"
#include "rowpar_body.cuh"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace lcd {
namespace {

constexpr int FW_BLOCK = 1024;   // frame_words_kernel
constexpr int SC_BLOCK = 256;    // scoring kernels

// exclusive scan, in place, of data[0..n) (LDS) by a whole workgroup; returns the total.  scratch[blockDim.x + 1] in LDS.
__device__ uint32_t block_exclusive_scan(uint32_t* data, int n, uint32_t* scratch) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = (n + nt - 1) / nt;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += data[i];
    scratch[tid] = sum;
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {
        const uint32_t t = tid >= off ? scratch[tid - off] : 0;
        __syncthreads();
        scratch[tid] += t;
        __syncthreads();
    }
    uint32_t run = scratch[tid] - sum;     // exclusive prefix of this thread's chunk
    const uint32_t total = scratch[nt - 1];
    for (int i = lo; i < hi; ++i) { const uint32_t v = data[i]; data[i] = run; run += v; }
    __syncthreads();
    return total;
}

__device__ __forceinline__ unsigned long long to_fixed(float t) {
    // t >= 0, t < 2^15.  floor(t * 2^48) in two exact halves (there is no f32 -> i64 conversion on the VALU): s = t * 2^16 is
    // exact, floor(s) < 2^31 is the high word, the fraction s - floor(s) is exact (it needs no more bits than t has) and its
    // product with 2^32 truncates to the low word.
    const float s = t * 65536.0f;
    const float fl = floorf(s);
    const float rem = s - fl;
    return ((unsigned long long)(uint32_t)fl << 32) | (uint32_t)(rem * 4294967296.0f);
}

// ---------------------------------------------------------------------------------------------- frame words
// One workgroup: reduce the frame's word slots to (unique word, count) with an LDS hash table (linear probing, atomicCAS),
// optionally append them to the open bucket as the postings of signature `slot` (nw += 1 each), and leave the
// word / count / idf lists plus the per-word idf table (idf_tab[w] = {stamp, idf}) for the scoring kernels.
// The list order is whatever the table yields: nothing downstream depends on it (integer accumulation).
__device__ __forceinline__ void frame_words_body(uint32_t* fw_smem, const int32_t* __restrict__ wslots, int n, int H, int do_register,
                                                               int32_t sig_id, long long slot, uint32_t slot_local, uint32_t ni, float N,
                                                               uint32_t stamp, uint32_t* __restrict__ nw, uint32_t* __restrict__ coo_w,
                                                               uint32_t* __restrict__ coo_pc, uint32_t* __restrict__ ne_counter,
                                                               int32_t* __restrict__ slot_sig, uint32_t* __restrict__ slot_ni,
                                                               uint32_t* __restrict__ slot_begin, uint32_t* __restrict__ slot_cnt,
                                                               uint32_t* __restrict__ q_w, uint32_t* __restrict__ q_cnt,
                                                               float* __restrict__ q_idf, uint32_t* __restrict__ q_meta,
                                                               uint2* __restrict__ idf_tab) {
    uint32_t* tkey = fw_smem;            // [H] 0xFFFFFFFF = empty
    uint32_t* tcnt = fw_smem + H;        // [H]
    uint32_t* grp = tcnt + H;            // [H / 64 + 1]
    const int tid = threadIdx.x;
    for (int i = tid; i < H; i += FW_BLOCK) { tkey[i] = 0xFFFFFFFFu; tcnt[i] = 0u; }
    __syncthreads();
    for (int i = tid; i < n; i += FW_BLOCK) {
        const int32_t ws = wslots[i];
        if (ws < 0) continue;
        const uint32_t w = (uint32_t)ws;
        uint32_t h = (w * 2654435761u) & (uint32_t)(H - 1);
        for (;;) {
            const uint32_t old = atomicCAS(&tkey[h], 0xFFFFFFFFu, w);
            if (old == 0xFFFFFFFFu || old == w) { atomicAdd(&tcnt[h], 1u); break; }
            h = (h + 1) & (uint32_t)(H - 1);
        }
    }
    __syncthreads();
    // compact the occupied table entries: ballot per 64-entry group, group offsets scanned by one thread
    const int ng = H / 64;
    for (int i0 = 0; i0 < H; i0 += FW_BLOCK) {
        const int i = i0 + tid;
        const bool occ = i < H && tkey[i] != 0xFFFFFFFFu;
        const unsigned long long bal = __ballot(occ);
        if ((tid & 63) == 0 && i < H) grp[i >> 6] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int g = 0; g < ng; ++g) { const uint32_t c = grp[g]; grp[g] = run; run += c; }
        grp[ng] = run;
    }
    __syncthreads();
    const uint32_t U = grp[ng];
    const uint32_t base = do_register ? ne_counter[0] : 0u;
    __syncthreads();
    for (int i0 = 0; i0 < H; i0 += FW_BLOCK) {
        const int i = i0 + tid;
        const bool occ = i < H && tkey[i] != 0xFFFFFFFFu;
        const unsigned long long bal = __ballot(occ);
        if (!occ) continue;
        const uint32_t u = grp[i >> 6] + (uint32_t)__popcll(bal & ((1ull << (tid & 63)) - 1ull));
        const uint32_t w = tkey[i];
        uint32_t cnt = tcnt[i];
        if (cnt > TF_CNT_MASK) cnt = TF_CNT_MASK;
        uint32_t nwv;
        if (do_register) {
            nwv = atomicAdd(&nw[w], 1u) + 1u;
            coo_w[base + u] = w;
            coo_pc[base + u] = (slot_local << TF_CNT_BITS) | cnt;
        } else {
            nwv = nw[w];
        }
        float idf = 0.0f;
        if (N > 0.0f && nwv > 0u) idf = log10f(__fdiv_rn(N, (float)nwv));   // Memory.cpp:2264-2266
        q_w[u] = w;
        q_cnt[u] = cnt;
        q_idf[u] = idf;
        idf_tab[w] = make_uint2(stamp, __float_as_uint(idf));
    }
    if (tid == 0) {
        q_meta[0] = U;
        if (do_register) {
            ne_counter[0] = base + U;
            slot_sig[slot] = sig_id;
            slot_ni[slot] = ni;
            slot_begin[slot] = base;
            slot_cnt[slot] = U;
        }
    }
}


// signatures whose retirement was requested since the last frame (Memory::disableWordsRef -> removeAllWordRef): their
// words lose one reference each and the slot is marked dead (ni = 0).  Up to 4 ride along with the next frame-words launch.
struct RetireArgs { long long slot[4]; const uint32_t* coo_w[4]; int n; };
__device__ __forceinline__ void retire_body(const RetireArgs& r, const uint32_t* __restrict__ slot_begin, const uint32_t* __restrict__ slot_cnt,
                                            uint32_t* __restrict__ nw, uint32_t* __restrict__ slot_ni, int32_t* __restrict__ slot_sig) {
    for (int p = 0; p < r.n; ++p) {
        const long long slot = r.slot[p];
        const uint32_t begin = slot_begin[slot], cnt = slot_cnt[slot];
        for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) atomicSub(&nw[r.coo_w[p][begin + k]], 1u);
        if (threadIdx.x == 0) { slot_ni[slot] = 0u; slot_sig[slot] = 0; }
    }
}

__global__ __launch_bounds__(FW_BLOCK) void frame_words_kernel(const int32_t* __restrict__ wslots, int n, int H, int do_register,
                                                               int32_t sig_id, long long slot, uint32_t slot_local, uint32_t ni, float N,
                                                               uint32_t stamp, uint32_t* __restrict__ nw, uint32_t* __restrict__ coo_w,
                                                               uint32_t* __restrict__ coo_pc, uint32_t* __restrict__ ne_counter,
                                                               int32_t* __restrict__ slot_sig, uint32_t* __restrict__ slot_ni,
                                                               uint32_t* __restrict__ slot_begin, uint32_t* __restrict__ slot_cnt,
                                                               uint32_t* __restrict__ q_w, uint32_t* __restrict__ q_cnt,
                                                               float* __restrict__ q_idf, uint32_t* __restrict__ q_meta,
                                                               uint2* __restrict__ idf_tab, RetireArgs retire) {
    extern __shared__ uint32_t fw_dyn_smem[];
    retire_body(retire, slot_begin, slot_cnt, nw, slot_ni, slot_sig);
    __syncthreads();
    frame_words_body(fw_dyn_smem, wslots, n, H, do_register, sig_id, slot, slot_local, ni, N, stamp, nw, coo_w, coo_pc, ne_counter, slot_sig,
                     slot_ni, slot_begin, slot_cnt, q_w, q_cnt, q_idf, q_meta, idf_tab);
}

#ifdef LCD_TAIL_TIMING   // timing experiment only: 100 MHz stamps between the phases of the frame tail
__device__ unsigned long long g_tail_timing[8];
#define FT_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) g_tail_timing[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FT_STAMP(i) do { } while (0)
#endif

// The single-workgroup tail of a frame in ONE launch: addNewWords decision loop (resolve_body.cuh) -> pending retirements
// -> unique words / registration / idf (frame_words_body).  Saves two dependent kernel boundaries per frame.
__global__ __launch_bounds__(FW_BLOCK) void frame_tail_kernel(ResolveArgs r, int H, int do_register, int32_t sig_id, long long slot,
                                                              uint32_t slot_local, uint32_t ni, float N, uint32_t stamp,
                                                              uint32_t* __restrict__ nw, uint32_t* __restrict__ coo_w,
                                                              uint32_t* __restrict__ coo_pc, uint32_t* __restrict__ ne_counter,
                                                              int32_t* __restrict__ slot_sig, uint32_t* __restrict__ slot_ni,
                                                              uint32_t* __restrict__ slot_begin, uint32_t* __restrict__ slot_cnt,
                                                              uint32_t* __restrict__ q_w, uint32_t* __restrict__ q_cnt,
                                                              float* __restrict__ q_idf, uint32_t* __restrict__ q_meta,
                                                              uint2* __restrict__ idf_tab, RetireArgs retire) {
    extern __shared__ uint32_t ft_dyn_smem[];
    // workgroups 1.. : the exact redo of the queries the 2-NN certificate rejected (they leave at once when there are none, which
    // is the usual case: no launch of its own for that check).  Workgroup 0 waits for them only when something was rejected.
    if (blockIdx.x > 0) { rowpar_body<64, FW_BLOCK>(r.rp, (int)blockIdx.x - 1, (int)gridDim.x - 1, r.fail_count); return; }
    if (r.rp.enabled && gridDim.x > 1) {
        if (threadIdx.x == 0 && r.fail_count[0] > 0) {
            while (__hip_atomic_load(&r.fail_count[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    FT_STAMP(0);
    resolve_body(ft_dyn_smem, r.q, r.flags, r.nndr, r.have_index, r.knn_word, r.knn_dist, r.selfdist, r.ld, r.cand_bits, r.bw, r.out_word,
                 r.out_n_new, r.knn_row, r.row_wslot, r.out_wslot);
    if (threadIdx.x == 0 && r.fail_count) { r.fail_count[0] = 0; r.fail_count[1] = 0; r.fail_count[3] = 0; }
    FT_STAMP(1);
    retire_body(retire, slot_begin, slot_cnt, nw, slot_ni, slot_sig);
    __syncthreads();      // out_wslot (global, written by this workgroup) and the LDS region are handed over
    FT_STAMP(2);
    frame_words_body(ft_dyn_smem, r.out_wslot, r.q, H, do_register, sig_id, slot, slot_local, ni, N, stamp, nw, coo_w, coo_pc, ne_counter,
                     slot_sig, slot_ni, slot_begin, slot_cnt, q_w, q_cnt, q_idf, q_meta, idf_tab);
    FT_STAMP(3);
}

// ---------------------------------------------------------------------------------------------- sealed buckets
__device__ __forceinline__ float fixed_to_float(unsigned long long v) { return (float)((double)(long long)v * (1.0 / 281474976710656.0)); }   // 2^-48

#ifdef LCD_SCORE_TIMING   // timing experiment only: 100 MHz stamps between the phases of score_sealed_body, per workgroup
__device__ unsigned long long g_score_timing[1024 * 8];
#define SC_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 1024) g_score_timing[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SC_STAMP(i) do { } while (0)
#endif

// One workgroup scores one sealed bucket for the word group g of G.  LDS (dynamic): acc[R] i64 | ni[R] | start[wg_cap] |
// scan[wg_cap + 1] | idf[wg_cap] | len[wg_cap] | scratch[SCB + 1].  With out_like != NULL (only valid for G == 1) the bucket's TF_R
// likelihood values are written straight from the LDS accumulators (no round trip through lfix, no finalize launch);
// otherwise the sums are added into lfix.
template <int SCB>
__device__ __forceinline__ void score_sealed_body(unsigned long long* sc_smem, const BucketDev* __restrict__ tab, int b, int g, int G, int wg_cap,
                                                  const uint32_t* __restrict__ q_w, const float* __restrict__ q_idf,
                                                  const uint32_t* __restrict__ q_meta, const uint32_t* __restrict__ slot_ni,
                                                  unsigned long long* __restrict__ lfix, float* __restrict__ out_like) {
    unsigned long long* acc = sc_smem;                              // [R]
    uint32_t* s_ni = (uint32_t*)(acc + TF_R);                       // [R]
    uint32_t* s_start = s_ni + TF_R;                                // [wg_cap]
    uint32_t* s_scan = s_start + wg_cap;                            // [wg_cap + 1]
    float* s_idf = (float*)(s_scan + wg_cap + 1);                   // [wg_cap]
    uint32_t* s_len = (uint32_t*)(s_idf + wg_cap);                  // [wg_cap]
    uint32_t* scratch = s_len + wg_cap;                             // [SCB + 1]
    const int tid = threadIdx.x;
    const uint32_t* __restrict__ dir = tab[b].dir;
    const uint32_t* __restrict__ ent = tab[b].ent;
    const uint32_t W = tab[b].W;
    const long long first_slot = (long long)b * TF_R;
    if (dir == nullptr) {                                           // every signature of the bucket is retired
        if (out_like) for (int i = tid; i < TF_R; i += SCB) out_like[first_slot + i] = 0.0f;
        return;
    }
    SC_STAMP(0);
    const int U = (int)q_meta[0];
    int Ug = U > g ? (U - g + G - 1) / G : 0;
    if (Ug > wg_cap) Ug = wg_cap;                                   // cannot happen: wg_cap is sized from the word count
    for (int i = tid; i < TF_R; i += SCB) { acc[i] = 0ull; s_ni[i] = slot_ni[first_slot + i]; }
    // A word's postings inside this bucket are one segment.  LONG segments (>= 64 postings, i.e. words present in a quarter or
    // more of the bucket's signatures: with a heavy-tailed vocabulary they hold most of the postings) are walked wave by wave in
    // 64-posting chunks -- the word, hence idf, is wave-uniform and no per-posting lookup is needed; the SHORT ones are walked as
    // one flattened, load-balanced list (s_scan = exclusive scan of their lengths).
    for (int k = tid; k < Ug; k += SCB) {
        const int u = g + k * G;
        const uint32_t w = q_w[u];
        const float idf = q_idf[u];
        uint32_t s = 0, e = 0;
        if (w < W && idf != 0.0f) { s = dir[w]; e = dir[w + 1]; }    // "if(logNnw)" (Memory.cpp:2267)
        s_start[k] = s;
        s_len[k] = e - s;
        s_scan[k] = (e - s) < 64u ? (e - s) : 0u;
        s_idf[k] = idf;
    }
    if (tid == 0) s_scan[Ug] = 0;
    __syncthreads();
    SC_STAMP(1);
    const uint32_t T = block_exclusive_scan(s_scan, Ug + 1, scratch);   // s_scan[Ug] == T afterwards
    // The first four short postings of every thread are located (binary search in the scanned offsets, LDS; four independent
    // chains) and REQUESTED now; they are consumed after the long segments, whose walk hides that round trip.
    uint32_t sh_addr[4], sh_e[4]; int sh_k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t t = (uint32_t)tid + (uint32_t)(u * SCB);
        int lo = 0, hi = Ug;                             // largest k with s_scan[k] <= t  (s_scan[Ug] == T > t)
        if (t < T) { while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_scan[mid] <= t) lo = mid; else hi = mid; } }
        sh_k[u] = lo;
        sh_addr[u] = t < T ? s_start[lo] + (t - s_scan[lo]) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) sh_e[u] = sh_addr[u] != 0xFFFFFFFFu ? ent[sh_addr[u]] : 0u;
    {
        // A segment has at most TF_R = 256 postings (one per signature of the bucket), i.e. four 64-posting chunks.  A wave
        // takes four words per trip and puts all (up to 16) chunk loads in flight before it touches any of them: the walk is a
        // chain of memory round trips otherwise (one per chunk).
        const int wv = tid >> 6, ln = tid & 63, nwv = SCB / 64;
        static_assert(TF_R == 256, "four chunks per segment");
        for (int k0 = wv; k0 < Ug; k0 += 4 * nwv) {
            uint32_t len[4], e[4][4];
            float idf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j * nwv;
                const uint32_t l = k < Ug ? s_len[k] : 0u;
                len[j] = l >= 64u ? l : 0u;                          // wave-uniform
                idf[j] = k < Ug ? s_idf[k] : 0.0f;
                const uint32_t start = k < Ug ? s_start[k] : 0u;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t o = (uint32_t)(c * 64 + ln);
                    e[j][c] = o < len[j] ? ent[start + o] : 0xFFFFFFFFu;      // a posting is < 2^30: the sentinel cannot occur
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t p = e[j][c];
                    if (p == 0xFFFFFFFFu) continue;
                    const uint32_t sl = p >> TF_CNT_BITS;
                    const uint32_t ni = s_ni[sl];
                    if (ni != 0u) {
                        const float term = __fdiv_rn(__fmul_rn((float)(p & TF_CNT_MASK), idf[j]), (float)ni);
                        atomicAdd(&acc[sl], to_fixed(term));
                    }
                }
            }
        }
    }
    SC_STAMP(2);
    // the short postings requested above ...
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (sh_addr[u] == 0xFFFFFFFFu) continue;
        const uint32_t sl = sh_e[u] >> TF_CNT_BITS;
        const uint32_t ni = s_ni[sl];
        if (ni != 0u) {                                              // "if(ni != 0)" (Memory.cpp:2275), 0 = retired slot
            const float term = __fdiv_rn(__fmul_rn((float)(sh_e[u] & TF_CNT_MASK), s_idf[sh_k[u]]), (float)ni);
            atomicAdd(&acc[sl], to_fixed(term));                     // ds_add_u64
        }
    }
    // ... and, for a bucket with more than 4 * SCB of them, the rest: four per thread and trip
    for (uint32_t t0 = (uint32_t)tid + 4u * SCB; t0 < T; t0 += 4 * SCB) {
        uint32_t addr[4]; int kk[4]; uint32_t e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = t0 + u * SCB;
            int lo = 0, hi = Ug;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_scan[mid] <= t) lo = mid; else hi = mid; }
            kk[u] = lo;
            addr[u] = t < T ? s_start[lo] + (t - s_scan[lo]) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = addr[u] != 0xFFFFFFFFu ? ent[addr[u]] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (addr[u] == 0xFFFFFFFFu) continue;
            const uint32_t sl = e[u] >> TF_CNT_BITS;
            const uint32_t ni = s_ni[sl];
            if (ni != 0u) {
                const float term = __fdiv_rn(__fmul_rn((float)(e[u] & TF_CNT_MASK), s_idf[kk[u]]), (float)ni);
                atomicAdd(&acc[sl], to_fixed(term));
            }
        }
    }
    __syncthreads();
    SC_STAMP(3);
    for (int i = tid; i < TF_R; i += SCB) {
        const unsigned long long v = acc[i];
        if (out_like) out_like[first_slot + i] = fixed_to_float(v);
        else if (v != 0ull) {
            if (G == 1) lfix[first_slot + i] = v;
            else atomicAdd(&lfix[first_slot + i], v);
        }
    }
}

// grid = (sealed live buckets, G word groups)
template <int SCB>
__global__ __launch_bounds__(SCB) void score_sealed_kernel(const BucketDev* __restrict__ tab, const int32_t* __restrict__ list, int G,
                                                           int wg_cap, const uint32_t* __restrict__ q_w,
                                                           const float* __restrict__ q_idf, const uint32_t* __restrict__ q_meta,
                                                           const uint32_t* __restrict__ slot_ni,
                                                           unsigned long long* __restrict__ lfix) {
    extern __shared__ unsigned long long sc_smem[];
    score_sealed_body<SCB>(sc_smem, tab, list[blockIdx.x], blockIdx.y, G, wg_cap, q_w, q_idf, q_meta, slot_ni, lfix, nullptr);
}

// ---------------------------------------------------------------------------------------------- open bucket
// The arrival-order log of the bucket that is still filling (<= TF_R signatures) is scanned once: a word-slot bitmap in
// LDS rejects the postings of words the frame does not contain, the survivors fetch idf from idf_tab.  The log is
// slot-major, so the postings a workgroup sees belong to a handful of consecutive signatures: they are summed in a
// small LDS window first and only the window is flushed with global atomics.
constexpr int OPEN_WIN = 64;
// ob / n_ob: this workgroup's index among the open-bucket workgroups.  With out_like != NULL the LAST of them to finish
// (agent-scope release / acquire around done_counter) converts the bucket's slots to float and re-zeroes lfix.
__device__ __forceinline__ void score_open_body(uint32_t* so_smem, int ob, int n_ob, const uint32_t* __restrict__ coo_w,
                                                const uint32_t* __restrict__ coo_pc, const uint32_t* __restrict__ ne_counter,
                                                long long first_slot, int n_open_slots, int bitmap_words, uint32_t stamp,
                                                const uint32_t* __restrict__ q_w, const uint32_t* __restrict__ q_meta,
                                                const uint2* __restrict__ idf_tab, const uint32_t* __restrict__ slot_ni,
                                                unsigned long long* __restrict__ lfix, float* __restrict__ out_like,
                                                int* __restrict__ done_counter) {
    uint32_t* s_bits = so_smem;                 // [bitmap_words] membership bitmap over word slots (0 words = not used)
    __shared__ unsigned long long s_win[OPEN_WIN];
    __shared__ uint32_t s_win0;
    __shared__ int s_last;
    const int nt = blockDim.x;
    const uint32_t ne = ne_counter[0];
    const uint32_t per = (ne + n_ob - 1) / n_ob;                    // contiguous chunk of the log per workgroup
    const uint32_t e0 = min((uint32_t)ob * per, ne), e1 = min(e0 + per, ne);
    if (e0 < e1) {
        const int U = (int)q_meta[0];
        for (int i = threadIdx.x; i < bitmap_words; i += nt) s_bits[i] = 0u;
        if (threadIdx.x < OPEN_WIN) s_win[threadIdx.x] = 0ull;
        if (threadIdx.x == 0) s_win0 = coo_pc[e0] >> TF_CNT_BITS;  // slot_local of the chunk's first posting
        __syncthreads();
        for (int i = threadIdx.x; i < U; i += nt) {
            const uint32_t w = q_w[i];
            if ((w >> 5) < (uint32_t)bitmap_words) atomicOr(&s_bits[w >> 5], 1u << (w & 31));
        }
        __syncthreads();
        const uint32_t win0 = s_win0;
        for (uint32_t e = e0 + threadIdx.x; e < e1; e += nt) {
            const uint32_t w = coo_w[e];
            if ((w >> 5) < (uint32_t)bitmap_words && !((s_bits[w >> 5] >> (w & 31)) & 1u)) continue;   // not a word of the frame
            const uint2 t = idf_tab[w];
            if (t.x != stamp) continue;
            const float idf = __uint_as_float(t.y);
            if (idf == 0.0f) continue;                                   // "if(logNnw)" (Memory.cpp:2267)
            const uint32_t pc = coo_pc[e];
            const uint32_t sl = pc >> TF_CNT_BITS;
            const uint32_t ni = slot_ni[first_slot + sl];
            if (ni == 0u) continue;
            const unsigned long long v = to_fixed(__fdiv_rn(__fmul_rn((float)(pc & TF_CNT_MASK), idf), (float)ni));
            if (sl - win0 < (uint32_t)OPEN_WIN) atomicAdd(&s_win[sl - win0], v);
            else atomicAdd(&lfix[first_slot + sl], v);
        }
        __syncthreads();
        if (threadIdx.x < OPEN_WIN) {
            const unsigned long long v = s_win[threadIdx.x];
            if (v != 0ull) atomicAdd(&lfix[first_slot + win0 + threadIdx.x], v);
        }
    }
    if (!out_like) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(done_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == n_ob - 1;
        if (s_last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); done_counter[0] = 0; }
    }
    __syncthreads();
    if (!s_last) return;
    for (int i = threadIdx.x; i < n_open_slots; i += nt) {
        const unsigned long long v = __hip_atomic_load(&lfix[first_slot + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out_like[first_slot + i] = fixed_to_float(v);
        if (v != 0ull) lfix[first_slot + i] = 0ull;
    }
}

__global__ __launch_bounds__(SC_BLOCK) void score_open_kernel(const uint32_t* __restrict__ coo_w, const uint32_t* __restrict__ coo_pc,
                                                              const uint32_t* __restrict__ ne_counter, long long first_slot,
                                                              int bitmap_words, uint32_t stamp, const uint32_t* __restrict__ q_w,
                                                              const uint32_t* __restrict__ q_meta, const uint2* __restrict__ idf_tab,
                                                              const uint32_t* __restrict__ slot_ni,
                                                              unsigned long long* __restrict__ lfix) {
    extern __shared__ uint32_t so_smem[];
    score_open_body(so_smem, blockIdx.x, gridDim.x, coo_w, coo_pc, ne_counter, first_slot, 0, bitmap_words, stamp, q_w, q_meta, idf_tab,
                    slot_ni, lfix, nullptr, nullptr);
}

// single-GPU fast path: every sealed bucket (retired ones included: they write zeros) and the open bucket in ONE launch,
// likelihood written directly -- no lfix round trip, no finalize launch.  blockIdx.x < n_sealed: sealed bucket list_all[x].
template <int SCB>
__global__ __launch_bounds__(SCB) void score_fused_kernel(const BucketDev* __restrict__ tab, const int32_t* __restrict__ list_all, int n_sealed,
                                                          int wg_cap, const uint32_t* __restrict__ q_w, const float* __restrict__ q_idf,
                                                          const uint32_t* __restrict__ q_meta, const uint32_t* __restrict__ slot_ni,
                                                          unsigned long long* __restrict__ lfix, float* __restrict__ out_like,
                                                          const uint32_t* __restrict__ coo_w, const uint32_t* __restrict__ coo_pc,
                                                          const uint32_t* __restrict__ ne_counter, long long open_first_slot,
                                                          int n_open_slots, int bitmap_words, uint32_t stamp,
                                                          const uint2* __restrict__ idf_tab, int* __restrict__ done_counter) {
    extern __shared__ unsigned long long sf_smem[];
    if ((int)blockIdx.x < n_sealed)
        score_sealed_body<SCB>(sf_smem, tab, list_all[blockIdx.x], 0, 1, wg_cap, q_w, q_idf, q_meta, slot_ni, lfix, out_like);
    else
        score_open_body((uint32_t*)sf_smem, (int)blockIdx.x - n_sealed, (int)gridDim.x - n_sealed, coo_w, coo_pc, ne_counter, open_first_slot,
                        n_open_slots, bitmap_words, stamp, q_w, q_meta, idf_tab, slot_ni, lfix, out_like, done_counter);
}

// fixed point -> float, and the accumulator is left zeroed for the next frame (no separate memset launch)
__global__ void finalize_kernel(long long* __restrict__ lfix, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const long long v = lfix[i];
        out[i] = (float)((double)v * (1.0 / 281474976710656.0));   // 2^-48
        if (v != 0) lfix[i] = 0;
    }
}
__global__ void gather_f32_kernel(const float* __restrict__ dense, const long long* __restrict__ slots, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const long long s = slots[i]; out[i] = s >= 0 ? dense[s] : 0.0f; }
}

// ---------------------------------------------------------------------------------------------- sealing
__global__ void seal_count_kernel(const uint32_t* __restrict__ coo_w, uint32_t ne, uint32_t* __restrict__ dir) {
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) atomicAdd(&dir[coo_w[e] + 1], 1u);
}
// inclusive scan of dir[0..n) in global memory by one workgroup (n up to a few million, sealing is rare)
__global__ __launch_bounds__(1024) void seal_scan_kernel(uint32_t* __restrict__ dir, uint32_t n) {
    __shared__ uint32_t scratch[1025];
    const int tid = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = min((uint32_t)tid * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += dir[i];
    scratch[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t t = tid >= off ? scratch[tid - off] : 0;
        __syncthreads();
        scratch[tid] += t;
        __syncthreads();
    }
    uint32_t run = scratch[tid] - sum;
    for (uint32_t i = lo; i < hi; ++i) { run += dir[i]; dir[i] = run; }
}
__global__ void seal_scatter_kernel(const uint32_t* __restrict__ coo_w, const uint32_t* __restrict__ coo_pc, uint32_t ne,
                                    const uint32_t* __restrict__ dir, uint32_t* __restrict__ cursor, uint32_t* __restrict__ ent) {
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) {
        const uint32_t w = coo_w[e];
        ent[dir[w] + atomicAdd(&cursor[w], 1u)] = coo_pc[e];
    }
}
__global__ void retire_kernel(long long slot, const uint32_t* __restrict__ coo_w, const uint32_t* __restrict__ slot_begin,
                              const uint32_t* __restrict__ slot_cnt, uint32_t* __restrict__ nw, uint32_t* __restrict__ slot_ni,
                              int32_t* __restrict__ slot_sig) {
    const uint32_t begin = slot_begin[slot], cnt = slot_cnt[slot];
    for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) atomicSub(&nw[coo_w[begin + k]], 1u);
    if (threadIdx.x == 0) { slot_ni[slot] = 0u; slot_sig[slot] = 0; }
}


// ---------------------------------------------------------------------------------------------- adjustLikelihood
// Rtabmap::adjustLikelihood (Rtabmap.cpp:5691-5760): mean / sample standard deviation over the entries > 0 after the
// virtual place (entry 0), rescale the entries above mean + stddev, then set the virtual place.  One workgroup, three
// passes over L[1..n) (0.4 MB at 100k signatures, L2-resident).  Sums are accumulated in double (the reference adds
// floats sequentially, uMean/uVariance UMath.h:419-432, 512-526): results agree to ~1e-6 relative.
__global__ __launch_bounds__(1024) void adjust_likelihood_kernel(float* __restrict__ L, int n, float ratio) {
    __shared__ double s_sum[1024];
    __shared__ unsigned int s_cnt[1024];
    __shared__ float s_max[1024];
    const int tid = threadIdx.x;
    double sum = 0.0; unsigned int cnt = 0; float mx = 0.0f;
    for (int i = 1 + tid; i < n; i += 1024) {
        const float v = L[i];
        if (v > 0.0f) { sum += (double)v; ++cnt; }
        if (v > mx) mx = v;
    }
    s_sum[tid] = sum; s_cnt[tid] = cnt; s_max[tid] = mx;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) { s_sum[tid] += s_sum[tid + off]; s_cnt[tid] += s_cnt[tid + off]; s_max[tid] = fmaxf(s_max[tid], s_max[tid + off]); }
        __syncthreads();
    }
    const unsigned int count = s_cnt[0];
    const float mean = count ? (float)(s_sum[0] / (double)count) : 0.0f;
    const float maxv = s_max[0];
    __syncthreads();
    double sq = 0.0;
    for (int i = 1 + tid; i < n; i += 1024) {
        const float v = L[i];
        if (v > 0.0f) { const float d = v - mean; sq += (double)(d * d); }
    }
    s_sum[tid] = sq;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) s_sum[tid] += s_sum[tid + off];
        __syncthreads();
    }
    const float var = count > 1 ? (float)(s_sum[0] / (double)(count - 1)) : 0.0f;
    const float stdDev = sqrtf(var);
    const float epsilon = 0.0001f;
    for (int i = 1 + tid; i < n; i += 1024) {
        const float value = L[i];
        float o = 1.0f;
        if (value > mean + stdDev) {
            if (ratio == 0.0f && mean != 0.0f) o = (value - (stdDev - epsilon)) / mean;
            else if (ratio != 0.0f && stdDev != 0.0f) o = (value - mean) / stdDev;
        }
        L[i] = o;
    }
    if (tid == 0 && n > 0) {
        float vp;
        if (ratio == 0.0f && stdDev > epsilon && maxv != 0.0f) vp = mean / stdDev + 1.0f;
        else if (ratio != 0.0f && maxv > mean) vp = stdDev / (maxv - mean) + 1.0f;
        else vp = 2.0f;
        L[0] = vp;
    }
}

inline int next_pow2(int v) { int p = 2; while (p < v) p <<= 1; return p; }

}  // namespace

hipError_t launch_adjust_likelihood(float* d_L, int n, float ratio, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    adjust_likelihood_kernel<<<1, 1024, 0, s>>>(d_L, n, ratio);
    return hipGetLastError();
}

hipError_t launch_gather_f32(const float* dense, const int64_t* slots, int n, float* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    gather_f32_kernel<<<(n + 255) / 256, 256, 0, s>>>(dense, (const long long*)slots, n, out);
    return hipGetLastError();
}

// ================================================================================================ host side
#define TF_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return e__; } while (0)

hipError_t Tfidf::init(hipStream_t s, int64_t* bytes, int64_t sig_capacity) {
    stream = s;
    bytes_device = bytes;
    TF_TRY(ensure_slots(sig_capacity > 0 ? sig_capacity : TF_R));
    TF_TRY(q_w.reserve(TF_MAX_WORDS * 4, 0, stream, bytes_device));
    TF_TRY(q_cnt.reserve(TF_MAX_WORDS * 4, 0, stream, bytes_device));
    TF_TRY(q_idf.reserve(TF_MAX_WORDS * 4, 0, stream, bytes_device));
    TF_TRY(q_meta.reserve(64, 0, stream, bytes_device));
    TF_TRY(hipMemsetAsync(q_meta.p, 0, 64, stream));
    return hipSuccess;
}

void Tfidf::destroy() {
    for (Bucket& b : buckets) { b.coo_w.release(bytes_device); b.coo_pc.release(bytes_device); b.dir.release(bytes_device); b.ent.release(bytes_device); }
    buckets.clear();
    DevBuf* all[] = {&slot_sig, &slot_ni, &slot_begin, &slot_cnt, &nw, &bkt_tab, &bkt_ne, &bkt_list, &lfix, &q_w, &q_cnt, &q_idf,
                     &q_meta, &tmp_cursor, &d_stage, &idf_tab, &bkt_list_all, &open_done};
    for (DevBuf* d : all) d->release(bytes_device);
    h_stage.release();
}

// grow a zero-initialised table
static hipError_t grow_zeroed(DevBuf& buf, size_t bytes, hipStream_t s, int64_t* total) {
    const size_t old = buf.cap;
    if (bytes <= old) return hipSuccess;
    hipError_t e = buf.reserve(bytes, old, s, total);
    if (e != hipSuccess) return e;
    return hipMemsetAsync((char*)buf.p + old, 0, buf.cap - old, s);
}

hipError_t Tfidf::ensure_slots(int64_t n) {
    TF_TRY(grow_zeroed(slot_sig, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(slot_ni, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(slot_begin, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(slot_cnt, (size_t)n * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(lfix, (size_t)n * 8, stream, bytes_device));
    return hipSuccess;
}

hipError_t Tfidf::wslot_of(int32_t word_id, int32_t* out) {
    auto it = word_wslot.find(word_id);
    if (it != word_wslot.end()) { *out = it->second; return hipSuccess; }
    const int32_t w = n_wslots++;
    word_wslot.emplace(word_id, w);
    TF_TRY(grow_zeroed(nw, (size_t)n_wslots * 4, stream, bytes_device));
    TF_TRY(grow_zeroed(idf_tab, (size_t)n_wslots * 8, stream, bytes_device));
    *out = w;
    return hipSuccess;
}

hipError_t Tfidf::upload_buckets() {
    if (!bkt_dirty) return hipSuccess;
    h_bkt.resize(buckets.size());
    std::vector<int32_t> list, list_all;
    for (size_t i = 0; i < buckets.size(); ++i) {
        Bucket& b = buckets[i];
        BucketDev d;
        d.coo_w = b.coo_w.as<uint32_t>(); d.coo_pc = b.coo_pc.as<uint32_t>();
        d.dir = b.dir.as<uint32_t>(); d.ent = b.ent.as<uint32_t>();
        d.W = b.W; d.sealed = b.sealed ? 1u : 0u; d.n_e_sealed = b.n_e_sealed; d.pad = 0;
        h_bkt[i] = d;
        if (b.sealed && b.live > 0) list.push_back((int32_t)i);
        if (b.sealed) list_all.push_back((int32_t)i);
    }
    n_list = (int)list.size();
    n_list_all = (int)list_all.size();
    if (n_list_all) {
        TF_TRY(bkt_list_all.reserve(list_all.size() * 4, 0, stream, bytes_device));
        TF_TRY(hipStreamSynchronize(stream));
        TF_TRY(hipMemcpy(bkt_list_all.p, list_all.data(), list_all.size() * 4, hipMemcpyHostToDevice));
    }
    if (!buckets.empty()) {
        TF_TRY(bkt_tab.reserve(buckets.size() * sizeof(BucketDev), 0, stream, bytes_device));
        TF_TRY(hipStreamSynchronize(stream));   // pageable source: keep it simple and ordered
        TF_TRY(hipMemcpy(bkt_tab.p, h_bkt.data(), buckets.size() * sizeof(BucketDev), hipMemcpyHostToDevice));
    }
    if (n_list) {
        TF_TRY(bkt_list.reserve(list.size() * 4, 0, stream, bytes_device));
        TF_TRY(hipMemcpy(bkt_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice));
    }
    bkt_dirty = false;
    return hipSuccess;
}

hipError_t Tfidf::seal(int bi) {
    Bucket& b = buckets[bi];
    if (b.sealed) return hipSuccess;
    uint32_t ne = 0;
    TF_TRY(hipMemcpyAsync(&ne, bkt_ne.as<uint32_t>() + bi, 4, hipMemcpyDeviceToHost, stream));
    TF_TRY(hipStreamSynchronize(stream));
    const uint32_t W = (uint32_t)n_wslots;
    TF_TRY(b.dir.reserve(((size_t)W + 1) * 4, 0, stream, bytes_device));
    TF_TRY(b.ent.reserve(std::max<size_t>(ne, 1) * 4, 0, stream, bytes_device));
    TF_TRY(tmp_cursor.reserve(((size_t)W + 1) * 4, 0, stream, bytes_device));
    TF_TRY(hipMemsetAsync(b.dir.p, 0, ((size_t)W + 1) * 4, stream));
    TF_TRY(hipMemsetAsync(tmp_cursor.p, 0, ((size_t)W + 1) * 4, stream));
    if (ne) {
        int blocks = (int)std::min<uint32_t>((ne + 255) / 256, 1024);
        seal_count_kernel<<<blocks, 256, 0, stream>>>(b.coo_w.as<uint32_t>(), ne, b.dir.as<uint32_t>());
        seal_scan_kernel<<<1, 1024, 0, stream>>>(b.dir.as<uint32_t>(), W + 1);
        seal_scatter_kernel<<<blocks, 256, 0, stream>>>(b.coo_w.as<uint32_t>(), b.coo_pc.as<uint32_t>(), ne, b.dir.as<uint32_t>(),
                                                       tmp_cursor.as<uint32_t>(), b.ent.as<uint32_t>());
        TF_TRY(hipGetLastError());
    }
    b.W = W;
    b.n_e_sealed = ne;
    b.sealed = true;
    bkt_dirty = true;
    return hipSuccess;
}

static RetireArgs take_pending(Tfidf& t) {
    RetireArgs r;
    r.n = 0;
    for (int i = 0; i < 4; ++i) { r.slot[i] = 0; r.coo_w[i] = nullptr; }
    while (r.n < 4 && !t.pending_retire.empty()) {
        const int64_t slot = t.pending_retire.back();
        t.pending_retire.pop_back();
        r.slot[r.n] = slot;
        r.coo_w[r.n] = t.buckets[(size_t)(slot / TF_R)].coo_w.as<uint32_t>();
        r.n += 1;
    }
    return r;
}

// apply every pending retirement now (stand-alone launch): needed before a bucket's memory is released
hipError_t Tfidf::flush_retire() {
    while (!pending_retire.empty()) {
        const int64_t slot = pending_retire.back();
        pending_retire.pop_back();
        retire_kernel<<<1, 256, 0, stream>>>((long long)slot, buckets[(size_t)(slot / TF_R)].coo_w.as<uint32_t>(), slot_begin.as<uint32_t>(),
                                             slot_cnt.as<uint32_t>(), nw.as<uint32_t>(), slot_ni.as<uint32_t>(), slot_sig.as<int32_t>());
        TF_TRY(hipGetLastError());
    }
    return hipSuccess;
}

static hipError_t run_frame_words(Tfidf& t, const int32_t* d_wslots, int n, bool reg, int32_t sig_id, int64_t slot, int32_t ni, float N,
                                  const ResolveArgs* resolve = nullptr) {
    const int H = next_pow2(std::max(2 * n, 128));
    size_t shmem = ((size_t)H * 2 + H / 64 + 2) * 4;
    uint32_t* coo_w = nullptr; uint32_t* coo_pc = nullptr; uint32_t* ne = nullptr;
    if (reg) {
        const int bi = (int)(slot / TF_R);
        coo_w = t.buckets[bi].coo_w.as<uint32_t>();
        coo_pc = t.buckets[bi].coo_pc.as<uint32_t>();
        ne = t.bkt_ne.as<uint32_t>() + bi;
    }
    if (t.pending_retire.size() > 4) TF_TRY(t.flush_retire());
    const RetireArgs ret = take_pending(t);
    t.stamp += 1;
    if (t.stamp == 0) t.stamp = 1;
    if (resolve) {
        const int mw = (resolve->q + 63) / 64 * 2;
        shmem = std::max(shmem, (size_t)(3 * mw + 2) * 4);
        const int n_redo = (resolve->rp.enabled && resolve->fail_count) ? (resolve->rp.n_rows + FW_BLOCK - 1) / FW_BLOCK : 0;
        frame_tail_kernel<<<1 + n_redo, FW_BLOCK, shmem, t.stream>>>(*resolve, H, reg ? 1 : 0, sig_id, (long long)slot, (uint32_t)(slot % TF_R),
                                                           (uint32_t)ni, N, t.stamp, t.nw.as<uint32_t>(), coo_w, coo_pc, ne,
                                                           t.slot_sig.as<int32_t>(), t.slot_ni.as<uint32_t>(), t.slot_begin.as<uint32_t>(),
                                                           t.slot_cnt.as<uint32_t>(), t.q_w.as<uint32_t>(), t.q_cnt.as<uint32_t>(),
                                                           t.q_idf.as<float>(), t.q_meta.as<uint32_t>(), t.idf_tab.as<uint2>(), ret);
    } else {
        frame_words_kernel<<<1, FW_BLOCK, shmem, t.stream>>>(d_wslots, n, H, reg ? 1 : 0, sig_id, (long long)slot, (uint32_t)(slot % TF_R),
                                                            (uint32_t)ni, N, t.stamp, t.nw.as<uint32_t>(), coo_w, coo_pc, ne,
                                                            t.slot_sig.as<int32_t>(), t.slot_ni.as<uint32_t>(),
                                                            t.slot_begin.as<uint32_t>(), t.slot_cnt.as<uint32_t>(),
                                                            t.q_w.as<uint32_t>(), t.q_cnt.as<uint32_t>(), t.q_idf.as<float>(),
                                                            t.q_meta.as<uint32_t>(), t.idf_tab.as<uint2>(), ret);
    }
    t.q_n_ub = n;
    return hipGetLastError();
}

hipError_t Tfidf::register_dev(int32_t sig_id, const int32_t* d_wslots, int n, int32_t ni, float N, const ResolveArgs* resolve) {
    if (n > TF_MAX_WORDS) return hipErrorInvalidValue;
    const int64_t slot = n_slots;
    TF_TRY(ensure_slots(slot + 1));
    const int bi = (int)(slot / TF_R);
    if (bi >= (int)buckets.size()) {
        if (bi > 0) TF_TRY(seal(bi - 1));
        buckets.emplace_back();
        TF_TRY(grow_zeroed(bkt_ne, (size_t)(bi + 1) * 4, stream, bytes_device));
        bkt_dirty = true;
    }
    Bucket& b = buckets[bi];
    const size_t need = (size_t)(b.ub_entries + n) * 4;
    if (need > b.coo_w.cap) {
        const size_t want = std::max(need, (size_t)TF_R * 512 * 4);
        TF_TRY(b.coo_w.reserve(want, (size_t)b.ub_entries * 4, stream, bytes_device));
        TF_TRY(b.coo_pc.reserve(want, (size_t)b.ub_entries * 4, stream, bytes_device));
        bkt_dirty = true;
    }
    TF_TRY(run_frame_words(*this, d_wslots, n, true, sig_id, slot, ni, N, resolve));
    b.ub_entries += n;
    b.n_slots += 1;
    b.live += 1;
    postings_ub += n;
    n_slots += 1;
    live_sigs += 1;
    sig_slot[sig_id] = slot;
    return hipSuccess;
}

hipError_t Tfidf::query_dev(const int32_t* d_wslots, int n, float N, const ResolveArgs* resolve) {
    if (n > TF_MAX_WORDS) return hipErrorInvalidValue;
    return run_frame_words(*this, d_wslots, n, false, 0, 0, 0, N, resolve);
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

hipError_t Tfidf::score(float* d_likelihood) {
    if (n_slots == 0) return hipSuccess;
    TF_TRY(flush_retire());
    TF_TRY(upload_buckets());
    // lfix is all zero here: zero-initialised on growth and re-zeroed by whoever consumed it last
    static const int scb = env_int("LCD_SC_BLOCK", 1024);
    static const int gforce = env_int("LCD_SC_G", 0);
    static const int fuse = env_int("LCD_SC_FUSED", 1);
    const int wcap_all = std::max(q_n_ub, 1);
    const bool has_open = !buckets.empty() && !buckets.back().sealed;
    const int bitmap_words = (n_wslots + 31) / 32;
    if (fuse && gforce <= 1 && (size_t)bitmap_words * 4 <= 32 * 1024 && (scb == 256 || scb == 512 || scb == 1024)) {
        // one launch: every sealed bucket writes its 256 likelihood values straight from LDS, the open bucket's workgroups
        // accumulate through lfix and the last of them converts its slots
        const size_t sealed_bytes = (size_t)TF_R * 8 + (size_t)TF_R * 4 + ((size_t)wcap_all * 4 + 1 + scb + 1 + 4) * 4;
        const size_t shmem = std::max(sealed_bytes, (size_t)std::max(bitmap_words, 1) * 4);
        int open_blocks = 0, bi = 0, n_open_slots = 0;
        const Bucket* ob = nullptr;
        if (has_open) {
            bi = (int)buckets.size() - 1;
            ob = &buckets[bi];
            n_open_slots = ob->n_slots;
            open_blocks = (int)std::min<int64_t>(std::max<int64_t>((ob->ub_entries + 4 * scb - 1) / (4 * scb), 1), 256);
        }
        if (n_list_all + open_blocks == 0) return hipSuccess;
        TF_TRY(grow_zeroed(open_done, 64, stream, bytes_device));
#define LCD_SCORE_FUSED(B) score_fused_kernel<B><<<n_list_all + open_blocks, B, shmem, stream>>>(bkt_tab.as<BucketDev>(), bkt_list_all.as<int32_t>(), \
            n_list_all, wcap_all, q_w.as<uint32_t>(), q_idf.as<float>(), q_meta.as<uint32_t>(), slot_ni.as<uint32_t>(), lfix.as<unsigned long long>(), \
            d_likelihood, ob ? ob->coo_w.as<uint32_t>() : nullptr, ob ? ob->coo_pc.as<uint32_t>() : nullptr, bkt_ne.as<uint32_t>() + bi, \
            (long long)bi * TF_R, n_open_slots, bitmap_words, stamp, idf_tab.as<uint2>(), open_done.as<int>())
        if (prof_b) TF_TRY(hipEventRecord(prof_b, stream));
        if (scb == 256) LCD_SCORE_FUSED(256); else if (scb == 512) LCD_SCORE_FUSED(512); else LCD_SCORE_FUSED(1024);
#undef LCD_SCORE_FUSED
        TF_TRY(hipGetLastError());
        if (prof_e) TF_TRY(hipEventRecord(prof_e, stream));
        prof_b = prof_e = nullptr;
        return hipSuccess;
    }
    TF_TRY(score_partial(lfix.as<unsigned long long>()));
    finalize_kernel<<<(unsigned)((n_slots + 255) / 256), 256, 0, stream>>>(lfix.as<long long>(), (long long)n_slots, d_likelihood);
    return hipGetLastError();
}

// fixed-point sums of q_* against every live signature, ADDED into lfix_target[0 .. n_slots) (caller provides zeros
// or a running sum); the multi-GPU path all-reduces these integers before finalize()
hipError_t Tfidf::score_partial(unsigned long long* lfix_target) {
    if (n_slots == 0) return hipSuccess;
    TF_TRY(flush_retire());
    TF_TRY(upload_buckets());
    const int wcap_all = std::max(q_n_ub, 1);
    if (n_list > 0) {
        static const int scb = env_int("LCD_SC_BLOCK", 1024);
        static const int gforce = env_int("LCD_SC_G", 0);
        int G = gforce > 0 ? gforce : (256 + n_list - 1) / n_list;     // aim at >= one workgroup per CU
        G = std::max(1, std::min(G, 8));
        const int wg_cap = (wcap_all + G - 1) / G;
        const size_t shmem = (size_t)TF_R * 8 + (size_t)TF_R * 4 + ((size_t)wg_cap * 4 + 1 + scb + 1 + 4) * 4;
        const dim3 grid(n_list, G);
#define LCD_SCORE_SEALED(B) score_sealed_kernel<B><<<grid, B, shmem, stream>>>(bkt_tab.as<BucketDev>(), bkt_list.as<int32_t>(), G, wg_cap, \
            q_w.as<uint32_t>(), q_idf.as<float>(), q_meta.as<uint32_t>(), slot_ni.as<uint32_t>(), lfix_target)
        if (scb == 256) LCD_SCORE_SEALED(256); else if (scb == 512) LCD_SCORE_SEALED(512); else LCD_SCORE_SEALED(1024);
#undef LCD_SCORE_SEALED
        TF_TRY(hipGetLastError());
    }
    if (!buckets.empty() && !buckets.back().sealed && buckets.back().ub_entries > 0) {
        const int bi = (int)buckets.size() - 1;
        const Bucket& b = buckets[bi];
        int blocks = (int)std::min<int64_t>((b.ub_entries + 4 * SC_BLOCK - 1) / (4 * SC_BLOCK), 512);
        int bitmap_words = (n_wslots + 31) / 32;
        if ((size_t)bitmap_words * 4 > 128 * 1024) bitmap_words = 0;   // > 1M word slots: idf_tab stamps alone decide
        score_open_kernel<<<blocks, SC_BLOCK, (size_t)std::max(bitmap_words, 1) * 4, stream>>>(
            b.coo_w.as<uint32_t>(), b.coo_pc.as<uint32_t>(), bkt_ne.as<uint32_t>() + bi, (long long)bi * TF_R, bitmap_words, stamp,
            q_w.as<uint32_t>(), q_meta.as<uint32_t>(), idf_tab.as<uint2>(), slot_ni.as<uint32_t>(), lfix_target);
        TF_TRY(hipGetLastError());
    }
    return hipSuccess;
}

hipError_t Tfidf::finalize(long long* lfix_src, long long n, float* d_likelihood) {
    if (n <= 0) return hipSuccess;
    finalize_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(lfix_src, n, d_likelihood);
    return hipGetLastError();
}

hipError_t Tfidf::retire(int32_t sig_id) {
    auto it = sig_slot.find(sig_id);
    if (it == sig_slot.end()) return hipErrorInvalidValue;
    const int64_t slot = it->second;
    const int bi = (int)(slot / TF_R);
    Bucket& b = buckets[bi];
    // the device side (nw -= 1 for the signature's words, ni = 0) rides along with the next frame-words launch
    pending_retire.push_back(slot);
    sig_slot.erase(it);
    live_sigs -= 1;
    b.live -= 1;
    if (b.live == 0 && b.sealed) {
        // every signature of the bucket is gone: drop its postings (pending retirements read its log: apply them first)
        TF_TRY(flush_retire());
        TF_TRY(hipStreamSynchronize(stream));
        postings_ub -= b.ub_entries;
        b.coo_w.release(bytes_device); b.coo_pc.release(bytes_device); b.dir.release(bytes_device); b.ent.release(bytes_device);
        b.W = 0; b.n_e_sealed = 0; b.ub_entries = 0;
        bkt_dirty = true;
    }
    return hipSuccess;
}

}  // namespace lcd

#ifdef LCD_SCORE_TIMING
extern "C" int lcd_debug_score_timing(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_score_timing), (size_t)n_words * 8);
}
#endif
#ifdef LCD_TAIL_TIMING
extern "C" int lcd_debug_tail_timing(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lcd::g_tail_timing), 64);
}
#endif
