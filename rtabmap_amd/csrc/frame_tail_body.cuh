// frame_tail_body.cuh -- device code of the per-frame registration: unique words of the frame (LDS hash), postings appended to the
// open bucket's log, nw / idf, retirements, and the whole frame tail (decision loop -> retirements -> registration) as ONE workgroup.
// Shared by the stand-alone kernels of tfidf.hip and by the fused pipeline launches of knn_mfma_kernels.hip (where the tail of frame
// t - 1 rides in the filter launch of frame t).
#pragma once
#include "tfidf.h"
#include "resolve_body.cuh"
#include "rowpar_body.cuh"

namespace lcd {
namespace {

// A pointer that was itself read from memory (the bucket table) has no known address space, and the compiler falls back to FLAT
// loads -- which also count on lgkmcnt, so every LDS wait would wait for them too.  These are global pointers: say so.
template <typename T> __device__ __forceinline__ T gload(const T* p) { return *(const __attribute__((address_space(1))) T*)p; }
__device__ __forceinline__ uint2 gload2(const uint2* p) {
    const unsigned long long v = *(const __attribute__((address_space(1))) unsigned long long*)p;
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

__device__ __forceinline__ uint4 gload4(const uint4* p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *(const __attribute__((address_space(1))) u32x4*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// idf -> Q5.26, round to nearest, saturating
__device__ __forceinline__ int32_t idf_to_fixed(float idf) {
    float s = idf * 67108864.0f;                        // 2^26, exact scaling
    s = fminf(fmaxf(s, -2147483520.0f), 2147483520.0f);
    return (int32_t)rintf(s);
}
// exact integer sum -> likelihood: one rounding to float, exact scaling by 2^-26, one division by ni
__device__ __forceinline__ float fixed_to_like(long long acc, uint32_t ni) {
    if (ni == 0u) return 0.0f;                          // "if(ni != 0)" (Memory.cpp:2275); 0 also marks a retired slot
    return __fdiv_rn(__ll2float_rn(acc) * 1.4901161193847656e-08f, (float)ni);
}

// float pair -> bf16 "hi" (RNE) and bf16 "lo" (the exact remainder, rounded): the split the bf16x3 filter multiplies
__device__ __forceinline__ void f16_split2_dev(float a, float b, uint32_t& hi, uint32_t& lo) {       // as f16_split2 (knn_mfma_kernels.hip)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const h2 h = __builtin_convertvector((f2){a, b}, h2);
    hi = __builtin_bit_cast(uint32_t, h);
    const f2 back = __builtin_convertvector(h, f2);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){__fsub_rn(a, back.x), __fsub_rn(b, back.y)}, h2));
}
__device__ __forceinline__ void bf16_split2_dev(float a, float b, uint32_t& hi, uint32_t& lo) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){a, b}, bf2));
    const float ra = __fsub_rn(a, __uint_as_float(hi << 16)), rb = __fsub_rn(b, __uint_as_float(hi & 0xFFFF0000u));
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){ra, rb}, bf2));
}

// ---------------------------------------------------------------------------------------------- frame words


// One workgroup: reduce the frame's word slots to (unique word, count) with an LDS hash table (linear probing, atomicCAS),
// optionally append them to the bucket log as the postings of signature `slot` (nw += 1 each), and leave the word / idf /
// dense-id lists plus the per-word idf table (idf_tab[w] = {stamp, idf}) for the scoring kernel.
// The list order is whatever the table yields: nothing downstream depends on it (integer accumulation).
// LDS: 2 * H + H / 64 + 4 words.
// SOLE: this workgroup is the only writer of the bucket's log right now (the frame path: one frame at a time on one stream), so the
// log position is read at the start and written back at the end instead of being reserved with a returning atomic in the middle.
// src_lds: the word slots in LDS (left there by the decision loop of the same kernel) instead of a.src.
template <int NT, bool SOLE>
__device__ __forceinline__ void frame_words_body(uint32_t* fw_smem, const FwArgs& a, const int32_t* src_lds = nullptr, bool have_ne0 = false,
                                                 uint32_t ne0 = 0u /* thread 0: the log position, read by the caller well ahead of time */) {
    const int H = a.H;
    uint32_t* tkey = fw_smem;            // [H] 0xFFFFFFFF = empty
    uint32_t* tcnt = fw_smem + H;        // [H]
    uint32_t* grp = tcnt + H;            // [H / 64 + 1]
    uint32_t* s_misc = grp + H / 64 + 1; // [0] log base, [1] dense list length
    const int tid = threadIdx.x;
    for (int i = tid; i < H; i += NT) { tkey[i] = 0xFFFFFFFFu; tcnt[i] = 0u; }
    if (tid == 0) { s_misc[0] = (SOLE && a.do_register) ? (have_ne0 ? ne0 : a.ne_counter[0]) : 0u; s_misc[1] = 0u; }
    if (have_ne0) lds_barrier(); else __syncthreads();
    for (int i = tid; i < a.n; i += NT) {
        int32_t ws = src_lds ? src_lds[i] : a.src[i];
        if (!src_lds && a.row_wslot) ws = ws >= 0 ? a.row_wslot[ws] : (ws <= -2 ? -ws - 2 : -1);   // (rows and marked keys: see frame_register_part; frames of more than 4 NT words)
        if (a.xlate) ws = (ws > 0 && (long long)ws < a.xlate_n) ? a.xlate[ws] : -1;
        else if (ws <= -2) ws = a.new_ws.n > 0 ? ws_runs_at_dev(a.new_ws, -ws - 2) : -1;   // the k-th new word of the frame (split tail)
        if (ws < 0) continue;
        const uint32_t w = (uint32_t)ws;
        uint32_t h = (w * 2654435761u) & (uint32_t)(H - 1);
        for (;;) {
            const uint32_t old = atomicCAS(&tkey[h], 0xFFFFFFFFu, w);
            if (old == 0xFFFFFFFFu || old == w) { atomicAdd(&tcnt[h], 1u); break; }
            h = (h + 1) & (uint32_t)(H - 1);
        }
    }
    lds_barrier();
    // compact the occupied table entries: ballot per 64-entry group, group offsets scanned by one thread
    const int ng = H / 64;
    for (int i0 = 0; i0 < H; i0 += NT) {
        const int i = i0 + tid;
        const bool occ = i < H && tkey[i] != 0xFFFFFFFFu;
        const unsigned long long bal = __ballot(occ);
        if ((tid & 63) == 0 && i < H) grp[i >> 6] = (uint32_t)__popcll(bal);
    }
    lds_barrier();
    if (tid < 64) {                                     // exclusive scan of the group counts by one wavefront
        uint32_t run = 0;
        for (int g0 = 0; g0 < ng; g0 += 64) {
            const int g = g0 + tid;
            const uint32_t c = g < ng ? grp[g] : 0u;
            uint32_t x = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (tid >= off) x += y; }
            if (g < ng) grp[g] = run + x - c;
            run += __shfl(x, 63, 64);
        }
        if (tid == 0) {
            grp[ng] = run;
            if (a.do_register && !SOLE) s_misc[0] = atomicAdd(a.ne_counter, run);   // reserve the signature's stretch of the log
        }
    }
    __syncthreads();
    const uint32_t U = grp[ng];
    const uint32_t base = s_misc[0];
    // second pass, four table entries per thread and trip: the reference count of every word comes back from a returning atomic
    // (one round trip) -- all four are in flight before the first is used
    for (int i0 = 0; i0 < H; i0 += 4 * NT) {
        uint32_t w4[4], c4[4], u4[4], n4[4], r4[4]; int32_t d4[4]; bool o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + r * NT + tid;
            const bool occ = i < H && tkey[i] != 0xFFFFFFFFu;
            const unsigned long long bal = __ballot(occ);
            o4[r] = occ; w4[r] = 0; c4[r] = 0; u4[r] = 0;
            if (occ) {
                u4[r] = grp[i >> 6] + (uint32_t)__popcll(bal & ((1ull << (tid & 63)) - 1ull));
                w4[r] = tkey[i];
                c4[r] = tcnt[i] > TF_CNT_MASK ? TF_CNT_MASK : tcnt[i];
            }
        }
        // (the answers have NO defaults and are read only where they were requested: merged with a default, each request got a register copy --
        // a wait -- right behind it, and the four round trips this loop means to overlap ran one after the other: round 6's ISA)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
#pragma clang diagnostic ignored "-Wconditional-uninitialized"
        const bool want_r = a.do_register && a.wrow;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!o4[r]) continue;
            if (want_r) r4[r] = gload(a.wrow + w4[r]);                                   // (in the same round trip as the reference counts)
            if (a.do_register) n4[r] = atomicAdd(&a.nw[w4[r]], 1u);                      // (a plain read + fire-and-forget add measured slower:
            else n4[r] = gload(a.nw + w4[r]);                                            // atomics drop the line from L2, the read then misses)
            if (a.want_q) d4[r] = gload(a.did + w4[r]);
        }
        // all four answers are awaited HERE, in front of the first store: behind a store (the branches below hide the count of operations
        // in flight from the compiler, which then waits for everything) each entry would wait for the stores of the one before it
        // (as register pins, so that the compiler's own bookkeeping knows the answers are in: behind an opaque wait it would wait again, for
        // everything, in front of every later read -- i.e. for the stores of the entry before)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(n4[r]), "+v"(d4[r]), "+v"(r4[r]));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!o4[r]) continue;
            const uint32_t w = w4[r], u = u4[r], cnt = c4[r], nwv = a.do_register ? n4[r] + 1u : n4[r];
            // a reference to a word the enqueued cleanUnusedWords tombstoned while this frame was in flight (the one documented departure
            // of the device-resident mode -- the reference's clean runs behind this frame's addNewWords and would have kept the word): counted
            if (want_r && r4[r] == 0xFFFFFFFFu && a.q_meta) atomicAdd(&a.q_meta[8], 1u);
            if (a.do_register) {
                a.coo_w[base + u] = w;
                a.coo_pc[base + u] = (a.slot_local << TF_CNT_BITS) | cnt;
            }
            if (a.want_q) {
                float idf = 0.0f;
                if (a.N > 0.0f && nwv > 0u) idf = log10f(__fdiv_rn(a.N, (float)nwv));   // Memory.cpp:2264-2266
                const int32_t idfq = idf_to_fixed(idf);
                const int32_t d = d4[r];
                a.q_w[u] = w;
                a.q_idf[u] = idfq;
                a.q_did[u] = d;
                a.idf_tab[w] = make_uint2(a.stamp, (uint32_t)idfq);
                if (d >= 0 && idfq != 0) {                                   // "if(logNnw)" (Memory.cpp:2267)
                    const uint32_t j = atomicAdd(&s_misc[1], 1u);
                    a.qd_did[j] = d;
                    a.qd_idf[j] = idfq;
                }
            }
        }
    }
#pragma clang diagnostic pop
    lds_barrier();                                      // only s_misc[1] (LDS) is awaited: the stores above need no acknowledgement here
    if (tid == 0) {
        if (a.want_q) { a.q_meta[0] = U; a.q_meta[1] = s_misc[1]; }
        if (a.do_register) {
            if (SOLE) a.ne_counter[0] = base + U;
            a.slot_sig[a.slot] = a.sig_id;
            a.slot_ni[a.slot] = a.ni;
            a.slot_begin[a.slot] = base;
            a.slot_cnt[a.slot] = U;
        }
    }
}

// signatures whose retirement was requested since the last frame (Memory::disableWordsRef -> removeAllWordRef): their
// words lose one reference each and the slot is marked dead (ni = 0).  Up to 4 ride along with the next frame-words launch.

__device__ __forceinline__ void retire_body(const RetireArgs& r, const uint32_t* __restrict__ slot_begin, const uint32_t* __restrict__ slot_cnt,
                                            uint32_t* __restrict__ nw, uint32_t* __restrict__ slot_ni, int32_t* __restrict__ slot_sig) {
    for (int p = 0; p < r.n; ++p) {
        const long long slot = r.slot[p];
        const uint32_t begin = slot_begin[slot], cnt = slot_cnt[slot];
        for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) atomicSub(&nw[r.coo_w[p][begin + k]], 1u);
        if (threadIdx.x == 0) { slot_ni[slot] = 0u; slot_sig[slot] = 0; }
    }
}
// The same with the log stretch of every retired slot (begin, cnt) already in registers (frame_register_part requests them with its other
// first reads), and the words of a slot requested two per thread before the first reference count is touched.
struct RetirePre { uint32_t begin[4], cnt[4]; };
__device__ __forceinline__ void retire_request(const RetireArgs& r, const uint32_t* __restrict__ slot_begin, const uint32_t* __restrict__ slot_cnt, RetirePre& pre) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {                                       // (slot 0 for the entries that are not there: a request without a condition)
        const long long slot = p < r.n ? r.slot[p] : 0;
        pre.begin[p] = gload(slot_begin + slot); pre.cnt[p] = gload(slot_cnt + slot);
    }
}
__device__ __forceinline__ void retire_finish(const RetireArgs& r, const RetirePre& pre, uint32_t* __restrict__ nw, uint32_t* __restrict__ slot_ni,
                                              int32_t* __restrict__ slot_sig, uint32_t k_first0 = 0u /* entries of slot 0 the caller has already released */) {
    const uint32_t nt = blockDim.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p >= r.n) continue;
        const uint32_t begin = pre.begin[p], cnt = pre.cnt[p];
        for (uint32_t k = threadIdx.x + (p == 0 ? k_first0 : 0u); k < cnt; k += 2 * nt) {
            const uint32_t k1 = min(k + nt, cnt - 1u);                  // (a clamped second request instead of one under a condition)
            uint32_t w0 = gload(r.coo_w[p] + begin + k), w1 = gload(r.coo_w[p] + begin + k1);
            asm volatile("" : "+v"(w0), "+v"(w1));                      // (both requested before either is used: the second one is not sunk under its condition)
            atomicSub(&nw[w0], 1u);
            if (k + nt < cnt) atomicSub(&nw[w1], 1u);
        }
        if (threadIdx.x == 0) { slot_ni[r.slot[p]] = 0u; slot_sig[r.slot[p]] = 0; }
    }
}

#ifdef LCD_TAIL_TIMING   // timing experiment only: 100 MHz stamps between the phases of the frame tail
__device__ unsigned long long g_tail_timing[8];
#define FT_STAMP(i) do { __builtin_amdgcn_s_barrier(); if (threadIdx.x == 0) g_tail_timing[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FT_STAMP(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------- new words -> vocabulary rows
// VWDictionary::update()'s append branch for the words the decision loop has just created (AppendArgs, tfidf.h).  `mask` / `prefix`:
// the loop's final new-word mask and its word prefix sums (LDS); descriptor i created the k-th new word iff its bit is set, k = its
// rank.  Sixteen lanes per descriptor: lane c moves the 16-byte chunk c, c + 16, ... of the row; for 64-float rows the same lanes
// produce |row|^2 (any order will do: the filter needs it to ~dim ulps) and the hi / lo bf16 split the matrix-core filter multiplies.
// one appended row, by the 16 lanes that hold it (lane c: floats [4c, 4c + 4) in x): the vocabulary row, |row|^2 (any order will do: the
// filter needs it to ~dim ulps), the hi / lo operand split the matrix-core filter multiplies; norm_max collects the largest |row|^2
__device__ __forceinline__ void append_write_row(const AppendArgs& ap, size_t row, int c, const uint4& x, float& norm_max) {
    reinterpret_cast<uint4*>(ap.vocab + row * ap.row_dwords)[c] = x;
    const float f0 = __uint_as_float(x.x), f1 = __uint_as_float(x.y), f2 = __uint_as_float(x.z), f3 = __uint_as_float(x.w);
    float s2 = fmaf(f3, f3, fmaf(f2, f2, fmaf(f1, f1, f0 * f0)));
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) s2 += __shfl_xor(s2, m, 64);
    uint2 hi, lo;                                                       // as vocab_bf16_kernel: 64 "hi" then 64 "lo" per row
    if (ap.f16) { f16_split2_dev(f0, f1, hi.x, lo.x); f16_split2_dev(f2, f3, hi.y, lo.y); }
    else { bf16_split2_dev(f0, f1, hi.x, lo.x); bf16_split2_dev(f2, f3, hi.y, lo.y); }
    reinterpret_cast<uint2*>(ap.vocab_bf + row * 64)[c] = hi;
    reinterpret_cast<uint2*>(ap.vocab_bf + row * 64 + 32)[c] = lo;
    if (c == 0) {
        ap.row_norm[2 * row] = s2; ap.row_norm[2 * row + 1] = 1.0f;
        norm_max = fmaxf(norm_max, s2);                                 // (one atomic per workgroup at the end, not one per row on one address)
    }
}
__device__ __forceinline__ void append_write_ids(const AppendArgs& ap, const WsRuns& new_ws, size_t row, int k) {
    const int32_t key = new_ws.n > 0 ? ws_runs_at_dev(new_ws, k) : -1;
    ap.row_id[row] = ap.first_id > 0 ? ap.first_id + k : (int32_t)row - ap.first_id;   // (first_id <= 0: the id follows the row)
    ap.row_wslot[row] = key;
    if (key >= 0 && ap.wrow) ap.wrow[key] = (uint32_t)row + 1u;         // the key now belongs to a row: the batched check of superseded
}                                                                       // reservations must not hand it out again
__device__ __forceinline__ void append_norm_max(const AppendArgs& ap, float norm_max) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) norm_max = fmaxf(norm_max, __shfl_xor(norm_max, m, 64));
    if ((threadIdx.x & 63) == 0 && norm_max > 0.0f) atomicMax(ap.norm_max_bits, __float_as_uint(norm_max));
}

// Deferred append, first half (the decision loop's workgroup, launch A): which descriptors became words, and how many rows there are now.
template <int NT>
__device__ __forceinline__ void append_publish(const AppendArgs& ap, int q, const uint32_t* mask, const uint32_t* prefix, int n_in_early) {
    const int n_in = n_in_early;                        // ap.cnt_in[0], read by the caller a whole decision loop ahead (a read here would wait for
                                                        // the loop's stores); < 0: no counter, nothing is appended
    const int mw = (q + 63) / 64 * 2;
    const int n_new = (int)prefix[mw];
    const int n_take = (n_in >= 0 && (long long)n_in + n_new <= ap.capacity) ? n_new : 0;
    if (n_take > 0)
        for (int i = threadIdx.x; i < q; i += NT)
            if ((mask[i >> 5] >> (i & 31)) & 1u) ap.list_out[new_rank(mask, prefix, i)] = (uint32_t)i;
    if (ap.mask_out)                                    // (zeros when nothing is appended: no shadow row is a word then)
        for (int w = threadIdx.x; w < 2 * mw + 1; w += NT) ap.mask_out[w] = n_take > 0 ? (w < mw ? mask[w] : prefix[w - mw]) : 0u;
    if (threadIdx.x == 0 && n_in >= 0) {
        ap.cnt_out[0] = n_in + n_take;
        if (ap.log_slot) ap.log_slot[0] = n_take;
        if (ap.first_out) ap.first_out[0] = ap.first_id > 0 ? ap.first_id : n_in - ap.first_id;
        if (ap.host_mirror && !ap.mirror_later) __hip_atomic_store(ap.host_mirror, ((unsigned long long)ap.tag << 32) | (unsigned long long)(uint32_t)(n_in + n_take), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// Deferred append, second half (APPEND_ROW_WGS workgroups of launch B of the same pair): workgroup `wg` writes its share of the rows.
// stage: LDS staging area (256 B per row, stage_rows of them); without one the rows travel through registers.
template <int NT>
__device__ __forceinline__ void append_rows_body(const AppendRowsArgs& A, int wg, float* stage, int stage_rows) {
    const AppendArgs& ap = A.ap;
    const int n_in = ap.cnt_in[0], n_new = ap.log_slot[0];
    const int per = (n_new + A.n_wgs - 1) / A.n_wgs;
    const int k_lo = min(wg * per, n_new), k_hi = min(k_lo + per, n_new);
    const int tid = threadIdx.x, c = tid & 15, lane = tid & 63, wave = tid >> 6;
    constexpr int G = NT / 16;
    float norm_max = 0.0f;
    if (ap.is_f32_64 && stage && stage_rows >= 4) {
        for (int k0 = k_lo; k0 < k_hi; k0 += stage_rows) {
            const int n_chunk = min(stage_rows, k_hi - k0);
            for (int i = wave; i * 4 < n_chunk; i += NT / 64) {         // one instruction = four rows = 1 KB of LDS
                const int k = k0 + min(i * 4 + (lane >> 4), n_chunk - 1);
                const float* src = ap.descriptors + (size_t)gload(ap.list_out + k) * 64 + (lane & 15) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + (size_t)i * 256), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int k = tid >> 4; k < n_chunk; k += G) {
                const uint4 x = *reinterpret_cast<const uint4*>(stage + (size_t)k * 64 + c * 4);
                append_write_row(ap, (size_t)n_in + (size_t)(k0 + k), c, x, norm_max);
                if (c == 0) append_write_ids(ap, A.new_ws, (size_t)n_in + (size_t)(k0 + k), k0 + k);
            }
            if (k0 + stage_rows < k_hi) __syncthreads();
        }
    } else {
        for (int k = k_lo + (tid >> 4); k < k_hi; k += G) {
            const size_t row = (size_t)n_in + (size_t)k;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ap.descriptors) + (size_t)gload(ap.list_out + k) * ap.row_dwords;
            if (ap.is_f32_64) append_write_row(ap, row, c, reinterpret_cast<const uint4*>(src)[c], norm_max);
            else { uint32_t* dst = ap.vocab + row * ap.row_dwords; for (int d = c; d < ap.row_dwords; d += 16) dst[d] = src[d]; }
            if (c == 0) append_write_ids(ap, A.new_ws, row, k);
        }
    }
    if (ap.is_f32_64) append_norm_max(ap, norm_max);
    // (AppendArgs::mirror_later: the decision loop left the pinned mirror to this launch)
    if (ap.mirror_later && ap.host_mirror && wg == 0 && threadIdx.x == 0)
        __hip_atomic_store(ap.host_mirror, ((unsigned long long)ap.tag << 32) | (unsigned long long)(uint32_t)(n_in + n_new), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int NT>
__device__ __forceinline__ void append_new_rows(const AppendArgs& ap, int q, const uint32_t* mask, const uint32_t* prefix, const WsRuns& new_ws,
                                                uint32_t* list /* LDS scratch, q words */,
                                                float* stage = nullptr, int stage_rows = 0 /* LDS staging area for the new rows (256 B each) */,
                                                int n_in_early = -1 /* ap.cnt_in[0], read by the caller ahead of time (a round trip less in the chain) */) {
    const int n_in = n_in_early >= 0 ? n_in_early : ap.cnt_in[0];
    const int mw = (q + 63) / 64 * 2;
    const int n_new = (int)prefix[mw];
    const int tid = threadIdx.x, c = tid & 15;
    const int n_take = (long long)n_in + n_new <= ap.capacity ? n_new : 0;
    // the descriptors that created words, in word order (the loop below then has no idle trips: ~150 of 500 descriptors create a word)
    for (int i = tid; i < q; i += NT)
        if ((mask[i >> 5] >> (i & 31)) & 1u) list[new_rank(mask, prefix, i)] = (uint32_t)i;
    __syncthreads();
    float norm_max = 0.0f;
    auto finish_row = [&](int k, const uint4& x) { append_write_row(ap, (size_t)n_in + (size_t)k, c, x, norm_max); };
    auto finish_ids = [&](int k) { append_write_ids(ap, new_ws, (size_t)n_in + (size_t)k, k); };
    constexpr int G = NT / 16;                                          // 16-lane groups
    if (ap.is_f32_64 && stage && stage_rows >= 4) {
        // Loads and stores share one in-order counter on this architecture: a loop that loads a few rows, writes them out and loads the
        // next ones waits for the WRITES of every trip before it sees the next loads (measured: 13 us for 150 rows, in the decision
        // loop's chain).  So all the rows come in first -- LDS-DMA, no registers, one round trip for a whole chunk -- and the writes
        // follow without anything ever waiting for them.
        const int lane = tid & 63, wave = tid >> 6;
        for (int k0 = 0; k0 < n_take; k0 += stage_rows) {
            const int n_chunk = min(stage_rows, n_take - k0);
            for (int i = wave; i * 4 < n_chunk; i += NT / 64) {         // one instruction = four rows = 1 KB of LDS
                const int k = k0 + min(i * 4 + (lane >> 4), n_chunk - 1);
                const float* src = ap.descriptors + (size_t)list[k] * 64 + (lane & 15) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + (size_t)i * 256), 16, 0, 0);
            }
            FT_STAMP(2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            FT_STAMP(3);
            for (int k = tid >> 4; k < n_chunk; k += G) {
                const uint4 x = *reinterpret_cast<const uint4*>(stage + (size_t)k * 64 + c * 4);
                finish_row(k0 + k, x);
                if (c == 0) finish_ids(k0 + k);
            }
            if (k0 + stage_rows < n_take) __syncthreads();              // the staging area is reused
        }
    } else {
        constexpr int UN = 4;                                           // rows per group and trip (their loads are in flight together)
        for (int k0 = tid >> 4; k0 < n_take; k0 += G * UN) {
            uint4 x[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int k = k0 + u * G;
                x[u] = make_uint4(0u, 0u, 0u, 0u);
                if (k < n_take && ap.is_f32_64) x[u] = reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(ap.descriptors) + (size_t)list[k] * ap.row_dwords)[c];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int k = k0 + u * G;
                if (k >= n_take) continue;                              // uniform over the 16 lanes of the row
                if (ap.is_f32_64) finish_row(k, x[u]);
                else {
                    uint32_t* dst = ap.vocab + ((size_t)n_in + (size_t)k) * ap.row_dwords;
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(ap.descriptors) + (size_t)list[k] * ap.row_dwords;
                    for (int d = c; d < ap.row_dwords; d += 16) dst[d] = src[d];
                }
                if (c == 0) finish_ids(k);
            }
        }
    }
    if (ap.is_f32_64 && n_take > 0) append_norm_max(ap, norm_max);      // the running maximum of |row|^2 (the filters' error bound uses it)
    if (tid == 0) {
        const int n_out = n_in + n_take;
        if (ap.is_f32_64) { ap.row_norm[2 * (size_t)n_out] = __int_as_float(0x7f800000); ap.row_norm[2 * (size_t)n_out + 1] = 1.0f; }   // sentinel
        ap.cnt_out[0] = n_out;
        if (ap.log_slot) ap.log_slot[0] = n_take;
        if (ap.first_out) ap.first_out[0] = ap.first_id > 0 ? ap.first_id : n_in - ap.first_id;
        if (ap.host_mirror) __hip_atomic_store(ap.host_mirror, ((unsigned long long)ap.tag << 32) | (unsigned long long)(uint32_t)n_out, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// a frame that appends nothing still hands the row count on (the two counters alternate from frame to frame)
__device__ __forceinline__ void append_pass_on(const AppendArgs& ap) {
    if (threadIdx.x == 0 && ap.cnt_in && ap.cnt_out) {
        const int n = ap.cnt_in[0];
        ap.cnt_out[0] = n;
        if (ap.log_slot) ap.log_slot[0] = 0;
        if (ap.first_out) ap.first_out[0] = ap.first_id > 0 ? ap.first_id : n - ap.first_id;
        if (ap.host_mirror) __hip_atomic_store(ap.host_mirror, ((unsigned long long)ap.tag << 32) | (unsigned long long)(uint32_t)n, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The single-workgroup tail of a frame in ONE launch: addNewWords decision loop (resolve_body.cuh) -> pending retirements
// -> unique words / registration / idf (frame_words_body).  Saves two dependent kernel boundaries per frame.
template <int NT>
__device__ __forceinline__ void frame_tail_body(uint32_t* ft_dyn_smem, const ResolveArgs& r, const FwArgs& a, const RetireArgs& retire, int wb, int n_wb) {
    // workgroups 1.. : the exact redo of the queries the 2-NN certificate rejected (they leave at once when there are none, which
    // is the usual case: no launch of its own for that check).  Workgroup 0 waits for them only when something was rejected.
    if (wb > 0) { rowpar_body<64, NT>(r.rp, wb - 1, n_wb - 1, r.fail_count); return; }
    if (r.rp.enabled && n_wb > 1) {
        if (threadIdx.x == 0 && r.fail_count[0] > 0) {
            while (__hip_atomic_load(&r.fail_count[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // the log position of the open bucket is needed by the registration at the very end: requested now, a whole decision loop ahead
    uint32_t ne0 = 0u;
    const bool have_ne0 = a.do_register && a.ne_counter != nullptr;
    if (threadIdx.x == 0 && have_ne0) ne0 = gload(a.ne_counter);
    int n_in_early = -1;                                // the appender's row count: the previous launch wrote it (see frame_resolve_part)
    if (r.ap.enabled && r.ap.cnt_in) n_in_early = gload(r.ap.cnt_in);
    FT_STAMP(0);
    // frames of up to 1024 descriptors: the register-resident decision loop, its result handed to the registration through LDS
    constexpr int KPT = 1024 / NT;                     // descriptors per thread of the register-resident loop: frames of up to 1024
    int32_t* lds_ws = r.q <= KPT * NT ? (int32_t*)(ft_dyn_smem + 2 * a.H + a.H / 64 + 8) : nullptr;
    const uint32_t* fmask;
    if (lds_ws) fmask = resolve_body_fast<NT, KPT>(ft_dyn_smem, lds_ws, r.q, r.flags, r.nndr, r.have_index, r.knn_word, r.knn_dist, r.selfdist, r.ld,
                                                   r.cand_bits, r.bw, r.out_word, r.out_n_new, r.knn_row, r.row_wslot, r.out_wslot, r.new_ws, r.cand_list,
                                                   r.cand_cnt, &n_in_early);
    else { asm volatile("" : "+v"(n_in_early), "+v"(ne0));
           fmask = resolve_body<NT>(ft_dyn_smem, r.q, r.flags, r.nndr, r.have_index, r.knn_word, r.knn_dist, r.selfdist, r.ld, r.cand_bits, r.bw, r.out_word,
                                    r.out_n_new, r.knn_row, r.row_wslot, r.out_wslot, r.new_ws); }
    if (threadIdx.x == 0 && r.fail_count) { r.fail_count[0] = 0; r.fail_count[1] = 0; r.fail_count[3] = 0; }
    // (before the registration reuses the LDS; the list scratch lies behind the word slots handed over in LDS)
    if (r.ap.enabled) append_new_rows<NT>(r.ap, r.q, fmask, ft_dyn_smem + 2 * ((r.q + 63) / 64 * 2), r.new_ws, ft_dyn_smem + 2 * a.H + a.H / 64 + 8 + r.q,
                                          nullptr, 0, n_in_early);
    else append_pass_on(r.ap);
    FT_STAMP(1);
    retire_body(retire, a.slot_begin, a.slot_cnt, a.nw, a.slot_ni, a.slot_sig);
    __syncthreads();      // out_wslot (global or LDS, written by this workgroup) and the LDS region are handed over
    FT_STAMP(2);
    frame_words_body<NT, true>(ft_dyn_smem, a, lds_ws, have_ne0, ne0);
    FT_STAMP(3);
}

// The same tail split in two for the pipelined handle (frame_a_kernel): the decision loop of frame t - 1 and the retirement +
// registration of frame t - 2 run as two workgroups of the SAME launch (12.9 us + 9.2 us at 256 threads used to be one 22 us chain,
// longer than the 20 us filter it was meant to hide behind).  The word slots travel through global memory (out_wslot), new words as
// the codes -(k + 2) (ResolveArgs::new_ws.n < 0) because their postings keys are reserved when the registration is prepared.
template <int NT>
__device__ __forceinline__ void frame_resolve_part(uint32_t* ft_dyn_smem, const ResolveArgs& r, int wb, int n_wb) {
    if (wb > 0) { rowpar_body<64, NT>(r.rp, wb - 1, n_wb - 1, r.fail_count); return; }
    constexpr int KPT = 1024 / NT;
    const bool fast = r.q <= KPT * NT;
    const bool helpers = r.rp.enabled && n_wb > 1;
    // (frames of up to 1 024 descriptors: the decision loop reads the helpers' counter and the appender's row count WITH its first round
    // trip -- resolve_body_fast -- instead of two round trips in front of it)
    if (helpers && !fast) {
        if (threadIdx.x == 0 && r.fail_count[0] > 0) {
            while (__hip_atomic_load(&r.fail_count[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // the row count the appender continues from was written by the previous launch: requested a whole decision loop ahead of its use
    int n_in_early = -1;
    const int32_t* cnt_in = (r.ap.enabled && r.ap.cnt_in) ? r.ap.cnt_in : nullptr;
    if (!fast && cnt_in) n_in_early = gload(cnt_in);
    FT_STAMP(0);
    const uint32_t* fmask;
    if (fast) fmask = resolve_body_fast<NT, KPT>(ft_dyn_smem, nullptr, r.q, r.flags, r.nndr, r.have_index, r.knn_word, r.knn_dist, r.selfdist,
                                                 r.ld, r.cand_bits, r.bw, r.out_word, r.out_n_new, r.knn_row, r.row_wslot, r.out_wslot, r.new_ws,
                                                 r.cand_list, r.cand_cnt, &n_in_early, cnt_in, helpers ? r.fail_count : nullptr, r.slots_are_rows != 0, r.straight != 0);
    else { asm volatile("" : "+v"(n_in_early)); fmask = resolve_body<NT>(ft_dyn_smem, r.q, r.flags, r.nndr, r.have_index, r.knn_word, r.knn_dist, r.selfdist, r.ld, r.cand_bits, r.bw, r.out_word,
                                  r.out_n_new, r.knn_row, r.row_wslot, r.out_wslot, r.new_ws, r.slots_are_rows != 0); }   // (both paths leave n_in_early awaited: the
                                                                        // compiler's wait in front of its use would otherwise cover the loop's stores)
    if (threadIdx.x == 0 && r.fail_count) { r.fail_count[0] = 0; r.fail_count[1] = 0; r.fail_count[3] = 0; }
    if (r.ap.enabled && r.ap.defer_rows) append_publish<NT>(r.ap, r.q, fmask, ft_dyn_smem + 2 * ((r.q + 63) / 64 * 2), n_in_early);
    else if (r.ap.enabled) {
        // the staging area of the new rows lies behind the appender's list, 16-byte aligned; its size comes from the launch (ap.lds_bytes)
        const int used = (3 * ((r.q + 63) / 64 * 2) + 4 + r.q + 3) & ~3;
        const int rows = (r.ap.lds_bytes / 4 - used) / 64;
        append_new_rows<NT>(r.ap, r.q, fmask, ft_dyn_smem + 2 * ((r.q + 63) / 64 * 2), r.new_ws, ft_dyn_smem + 3 * ((r.q + 63) / 64 * 2) + 4,
                            reinterpret_cast<float*>(ft_dyn_smem + used), rows > 0 ? (rows & ~3) : 0, n_in_early);
    } else append_pass_on(r.ap);
    FT_STAMP(1);
}
// The registration of a frame whose word slots are postings KEYS already (FwArgs::row_wslot == NULL: revisit phases of a pipelined stream, and every
// other caller): the word slots go to LDS after ONE round trip and the table phases of frame_words_body run while the retirement's reads
// and atomics are in flight.  (Round 6 measured both heads in both regimes, r06_ab_notes.txt 10: the two-round-trip head below wins while frames
// create words -- its second trip carries the keys the decision loop no longer gathers -- and loses 0.4-0.5 us of launch A once they only revisit.)
template <int NT>
__device__ __forceinline__ void frame_register_part_keys(uint32_t* ft_dyn_smem, const FwArgs& a, const RetireArgs& retire) {
    uint32_t ne0 = 0u;
    const bool have_ne0 = a.do_register && a.ne_counter != nullptr;
    if (threadIdx.x == 0 && have_ne0) ne0 = gload(a.ne_counter);
    // The frame's word slots (written by the decision loop one launch earlier) are requested HERE, in front of the retirement's reads:
    // they arrive with them (one in-order counter), are parked in LDS, and the table phases of frame_words_body then run on LDS alone
    // while the retirement's atomics are acknowledged -- instead of a barrier that waits for those acknowledgements and a round trip
    // for the word slots behind it.  (The barrier in front of frame_words_body's second pass still orders the reference counts.)
    constexpr int KS = 4;
    const bool pre = a.n <= KS * NT && a.xlate == nullptr && a.src != nullptr;
    int32_t ws_pre[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const int i = (int)threadIdx.x + k * NT;
        ws_pre[k] = (pre && i < a.n) ? gload(a.src + i) : -1;
    }
    FT_STAMP(4);
    retire_body(retire, a.slot_begin, a.slot_cnt, a.nw, a.slot_ni, a.slot_sig);
    if (!pre) {
        __syncthreads();
        FT_STAMP(5);
        frame_words_body<NT, true>(ft_dyn_smem, a, nullptr, have_ne0, ne0);
    } else {
        int32_t* lds_ws = (int32_t*)(ft_dyn_smem + 2 * a.H + a.H / 64 + 8);      // behind the tables (as in frame_tail_body)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int i = (int)threadIdx.x + k * NT;
            if (i < a.n) lds_ws[i] = ws_pre[k];                           // read back by the same thread
        }
        FT_STAMP(5);
        frame_words_body<NT, true>(ft_dyn_smem, a, lds_ws, have_ne0, ne0);
    }
    FT_STAMP(6);
}


template <int NT>
__device__ __forceinline__ void frame_register_part(uint32_t* ft_dyn_smem, const FwArgs& a, const RetireArgs& retire) {
    // Everything this chain reads first -- the log position of the open bucket, the frame's word slots (written by the decision loop one
    // launch earlier), the log stretches of the slots to retire -- is requested in ONE round trip and awaited once: every request WITHOUT a
    // condition, at a clamped address of an array that is always there, the values selected afterwards.  A request under a condition whose
    // result is merged with a default makes the compiler copy registers right behind the request, i.e. wait for it -- round 6's ISA had
    // eight round trips in a row here (ne0, four word slots, begin, cnt, the words).
    if (!a.row_wslot) { frame_register_part_keys<NT>(ft_dyn_smem, a, retire); return; }
    constexpr int KS = 4;
    const bool have_ne0 = a.do_register && a.ne_counter != nullptr;
    const bool pre = a.n <= KS * NT && a.xlate == nullptr && a.src != nullptr;
    const uint32_t* nep = have_ne0 ? a.ne_counter : a.nw;
    const int32_t* srcp = pre ? a.src : reinterpret_cast<const int32_t*>(a.nw);
    const int i_max = pre ? max(a.n - 1, 0) : 0;
    uint32_t ne0 = gload(nep);
    int32_t ws_pre[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) ws_pre[k] = gload(srcp + min((int)threadIdx.x + k * NT, i_max));
    RetirePre rp;
    retire_request(retire, a.slot_begin, a.slot_cnt, rp);
    asm volatile("" : "+v"(ne0), "+v"(ws_pre[0]), "+v"(ws_pre[1]), "+v"(ws_pre[2]), "+v"(ws_pre[3]));       // (awaited together, and known to be)
    asm volatile("" : "+v"(rp.begin[0]), "+v"(rp.begin[1]), "+v"(rp.begin[2]), "+v"(rp.begin[3]), "+v"(rp.cnt[0]), "+v"(rp.cnt[1]), "+v"(rp.cnt[2]), "+v"(rp.cnt[3]));
    FT_STAMP(4);
    // ---- second round trip, again without conditions: (a) the word slots are vocabulary ROWS (FwArgs::row_wslot: the decision loop of a pipelined
    // frame leaves the row of the word it chose -- the gather of its postings key used to be a round trip of THAT chain, the longest of launch A, and
    // sat in front of every wait of its sweeps): their keys; (b) the first 2 NT words of the first signature to retire (one signature of <= 2 NT
    // unique words is the usual case; the rest goes through retire_finish's loop)
    const int32_t* rwp = a.row_wslot ? a.row_wslot : reinterpret_cast<const int32_t*>(a.nw);
    const bool ret0 = retire.n >= 1 && rp.cnt[0] > 0u;
    const uint32_t* cwp = ret0 ? retire.coo_w[0] + rp.begin[0] : a.nw;
    const uint32_t cnt0 = ret0 ? rp.cnt[0] : 1u;
    int32_t key[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) key[k] = gload(rwp + (a.row_wslot ? max(ws_pre[k], 0) : 0));
    uint32_t rw0 = gload(cwp + min((uint32_t)threadIdx.x, cnt0 - 1u)), rw1 = gload(cwp + min((uint32_t)threadIdx.x + NT, cnt0 - 1u));
    asm volatile("" : "+v"(key[0]), "+v"(key[1]), "+v"(key[2]), "+v"(key[3]), "+v"(rw0), "+v"(rw1));
    if (a.row_wslot) {
#pragma unroll
        for (int k = 0; k < KS; ++k) ws_pre[k] = ws_pre[k] >= 0 ? key[k] : (ws_pre[k] <= -2 ? -ws_pre[k] - 2 : -1);   // (<= -2: the key of a word the frame created)
    }
    // the word slots are parked in LDS: the table phases of frame_words_body then run on LDS alone while the retirement's atomics are
    // acknowledged.  (The barrier in front of frame_words_body's second pass still orders the reference counts.)
    int32_t* lds_ws = pre ? (int32_t*)(ft_dyn_smem + 2 * a.H + a.H / 64 + 8) : nullptr;      // behind the tables (as in frame_tail_body)
    if (pre) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int i = (int)threadIdx.x + k * NT;
            if (i < a.n) lds_ws[i] = ws_pre[k];                           // read back by the same thread
        }
    }
    if (ret0) {
        if (threadIdx.x < rp.cnt[0]) atomicSub(&a.nw[rw0], 1u);
        if (threadIdx.x + NT < rp.cnt[0]) atomicSub(&a.nw[rw1], 1u);
    }
    retire_finish(retire, rp, a.nw, a.slot_ni, a.slot_sig, ret0 ? 2u * NT : 0u);
    if (!pre) __syncthreads();
    FT_STAMP(5);
    frame_words_body<NT, true>(ft_dyn_smem, a, lds_ws, have_ne0, ne0);
    FT_STAMP(6);
}


}  // namespace
}  // namespace lcd
