// resolve_body.cuh -- device code of the addNewWords decision loop (see resolve_kernels.hip), shared by the stand-alone
// resolve kernel and the fused per-frame tail kernel (tfidf.hip).
#pragma once
#include "lcd_kernels.h"

namespace lcd {
namespace {

constexpr int RBLOCK = 1024;
constexpr int LCD_Q_INCREMENTAL = 1;
constexpr int LCD_Q_NEW_WORDS_COMPARED = 2;

#ifdef LCD_TAIL_TIMING   // timing experiment only: 100 MHz stamps inside the fast decision loop
__device__ unsigned long long g_resolve_timing[8];
#define RB_STAMP(i) do { __builtin_amdgcn_s_barrier(); if (threadIdx.x == 0) g_resolve_timing[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned long long g_sweep_timing[32];                    // [wave][point]: per-wave stamps inside the first sweep, no barrier added
#define SW_STAMP(p) do { if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 4 && sweep == 0) g_sweep_timing[(threadIdx.x >> 6) * 8 + (p)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RB_STAMP(i) do { } while (0)
#define SW_STAMP(p) do { } while (0)
#endif

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains the wave's outstanding GLOBAL loads
// (s_waitcnt vmcnt(0)), which would end the overlap of a requested-early / consumed-late load with the work in between
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Cand { float d; int id; };   // id > 0: word id, id < 0: -(j+1) = the new word created by descriptor j

// std::multimap<float,int> insertion (equal keys keep insertion order, VWDictionary.cpp:1091) restricted to what is
// read afterwards: the two smallest entries.
__device__ __forceinline__ void cand_push(Cand& c0, Cand& c1, int& n, float d, int id) {
    if (n == 0) { c0.d = d; c0.id = id; }
    else if (d < c0.d) { c1 = c0; c0.d = d; c0.id = id; }
    else if (n == 1 || d < c1.d) { c1.d = d; c1.id = id; }
    ++n;
}

// candidates of descriptor i given the current guess (bit mask in LDS) of which earlier descriptors are new words
__device__ __forceinline__ void gather_candidates(int i, bool together, int have_index, const int32_t* __restrict__ knn_word,
                                                  const float* __restrict__ knn_dist, const float* __restrict__ selfdist,
                                                  int ld, const uint32_t* __restrict__ cand_bits, int bw,
                                                  const uint32_t* new_mask, Cand& c0, Cand& c1, int& n) {
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
    if (together) {
        // exact 2-NN (1-NN when only one exists) among the new words created before i, lowest j on ties (:1140-1160),
        // restricted to the ones that can reach the two best candidates (cand_bits)
        uint64_t b = KEY_NONE, s = KEY_NONE;
        const int wlast = i >> 5;
        for (int w = 0; w <= wlast; ++w) {
            uint32_t m = cand_bits[(size_t)i * bw + w] & new_mask[w];
            if (w == wlast) m &= (1u << (i & 31)) - 1u;           // only j < i
            while (m) {
                const int j = (w << 5) + __builtin_ctz(m);
                m &= m - 1;
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

// rank of descriptor j among the new words = number of mask bits below j (word prefix sums in LDS)
__device__ __forceinline__ int new_rank(const uint32_t* mask, const uint32_t* prefix, int j) {
    return (int)(prefix[j >> 5] + __popc(mask[j >> 5] & ((1u << (j & 31)) - 1u)));
}

// The same decision loop for frames of at most KPT * NT descriptors (KPT descriptors per thread, NT = workgroup size),
// latency-trimmed: the kernel this runs in is ONE workgroup per frame, so what counts is the number of dependent global round trips.
//   * the indexed neighbours, their vocabulary rows and the descriptor's candidate-bit row are read ONCE, all loads in flight
//     together; the postings keys of both neighbours are requested before the sweeps and consumed after them;
//   * the bit row is kept as at most four non-zero (word, bits) pairs in registers (rows with more fall back to memory): a
//     descriptor whose row is empty -- nearly all of them in a mature vocabulary -- has its sweep-0 decision for good and a sweep
//     costs it nothing; the winner of a sweep lives in a register, not in out_word;
//   * the result is also left in LDS (lds_wslot, may be NULL) for the registration that follows in the same kernel.
// Same results as resolve_body (tests drive both).  rs_smem as below.
template <int NT, int KPT>
__device__ __forceinline__ const uint32_t* resolve_body_fast(uint32_t* rs_smem, int32_t* lds_wslot, int q, int flags, float nndr, int have_index,
                                                  const int32_t* __restrict__ knn_word, const float* __restrict__ knn_dist,
                                                  const float* __restrict__ selfdist, int ld,
                                                  const uint32_t* __restrict__ cand_bits, int bw,
                                                  int32_t* __restrict__ out_word, int32_t* __restrict__ out_n_new,
                                                  const int32_t* __restrict__ knn_row, const int32_t* __restrict__ row_wslot,
                                                  int32_t* __restrict__ out_wslot, const WsRuns& new_ws,
                                                  const uint2* __restrict__ cand_list = nullptr, const int32_t* __restrict__ cand_cnt = nullptr,
                                                  int* keep_in_reg = nullptr, const int32_t* early_ptr = nullptr, int32_t* helpers_fail = nullptr,
                                                  bool slots_are_rows = false, bool straight_ok = true) {
    const int mw = (q + 63) / 64 * 2;
    uint32_t* mask_cur = rs_smem;
    uint32_t* mask_next = rs_smem + mw;
    uint32_t* prefix = rs_smem + 2 * mw;
    __shared__ int s_changed_f;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool incremental = (flags & LCD_Q_INCREMENTAL) != 0;
    const bool together = incremental && (flags & LCD_Q_NEW_WORDS_COMPARED) && cand_bits != nullptr;
    const int qpad = mw * 32;
    struct Dsc {
        int w0, w1; int32_t ws_a, ws_b;
        Cand b0, b1; int nb;
        int cj[4]; float cd[4]; int nc; bool overflow;     // the same-frame candidates below the descriptor (at most four kept)
        uint32_t rb[16];                                   // overflow: the descriptor's bit row (frames of up to 512 descriptors)
        bool reject; int win;
    };
    Dsc st[KPT];
    RB_STAMP(0);
    // ---- round trip 1: indexed neighbours + their rows, the same-frame candidates (count + compact list left by the re-rank; descriptors
    //      with more than four, and paths without the lists, walk their bit row in every sweep instead) and the bit row itself (frames of up
    //      to 512 descriptors: 64 bytes) -- all descriptors of the thread in flight together
    float d0[KPT], d1[KPT]; int r0[KPT], r1[KPT];
    int cn[KPT];
    uint4 cl_lo[KPT], cl_hi[KPT], rbq[KPT][4];
    // `straight`: every array of the round trip is there (a pipelined frame's decision loop always) -> the requests are issued
    // UNCONDITIONALLY at clamped addresses, in one straight line, and the values are selected afterwards.  With a request under a
    // condition the compiler merges the loaded registers with their defaults right behind the request -- a register copy that needs the
    // data: round 6's ISA had `s_waitcnt vmcnt(2)` behind the first list request of EVERY descriptor of the thread, i.e. four round trips
    // in a row where the source says one (stamps: 5.1 us from the entry to the reject mask; one round trip is ~2.5 us in that launch).
    // A bit row is read as four 16-byte pieces whatever its length: the bytes behind a short row are the next rows and, behind the last
    // row, the compact lists of the same buffer (cand_bits_layout) -- checked here -- and words >= bw are cleared below.
    const bool straight = straight_ok && have_index && (out_wslot || lds_wslot) && together && cand_cnt && cand_list && bw >= 2 && bw <= 16 && q >= 8 &&
                          reinterpret_cast<const uint32_t*>(cand_list) == cand_bits + (size_t)q * bw;
    // early_ptr: a word the caller wants in *keep_in_reg (the appender's row count), requested BEHIND the round trip's requests: in front
    // of them the compiler's wait for it (the straight block reuses registers the other block loads into) was a round trip of its own.
    // helpers_fail: the exact-redo helpers' counters ([0] = queries the 2-NN certificate rejected, [3] = helpers done).  The count is
    // read WITH the round trip, not in front of it (it used to be a scalar load + barrier ahead of everything); the rare frame with
    // rejected queries waits for the helpers and reads everything again.
    if (!straight && helpers_fail) {
        if (tid == 0 && helpers_fail[0] > 0) {
            while (__hip_atomic_load(&helpers_fail[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (!straight && early_ptr && keep_in_reg) *keep_in_reg = *(const __attribute__((address_space(1))) int32_t*)early_ptr;
    if (straight) {
        const int32_t* fcp = helpers_fail ? helpers_fail : out_n_new;      // (always a readable word: the loads below have no condition)
        const int32_t* ep = early_ptr ? early_ptr : out_n_new;
        int32_t fc = 0, ev = -1;
        auto round_trip = [&]() __attribute__((always_inline)) {
            float2 dd[KPT]; int2 ww[KPT], rr[KPT]; int cr[KPT];
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const int ic = min(tid + k * NT, q - 1);
                dd[k] = *reinterpret_cast<const float2*>(knn_dist + 2 * ic);
                ww[k] = *reinterpret_cast<const int2*>(knn_word + 2 * ic);
                rr[k] = *reinterpret_cast<const int2*>(knn_row + 2 * ic);
                cr[k] = cand_cnt[ic];
                cl_lo[k] = *reinterpret_cast<const uint4*>(cand_list + (size_t)ic * 4);
                cl_hi[k] = *reinterpret_cast<const uint4*>(cand_list + (size_t)ic * 4 + 2);
                const uint32_t* row = cand_bits + (size_t)ic * bw;
#pragma unroll
                for (int u = 0; u < 4; ++u) rbq[k][u] = *reinterpret_cast<const uint4*>(row + 4 * u);
            }
            fc = *(const __attribute__((address_space(1))) int32_t*)fcp;
            ev = *(const __attribute__((address_space(1))) int32_t*)ep;
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const bool in = tid + k * NT < q;
                d0[k] = in ? dd[k].x : -1.0f; d1[k] = in ? dd[k].y : -1.0f;
                st[k].w0 = in ? ww[k].x : 0; st[k].w1 = in ? ww[k].y : 0;
                r0[k] = in ? rr[k].x : -1; r1[k] = in ? rr[k].y : -1;
                cn[k] = in ? cr[k] : 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) if (4 * u >= bw) rbq[k][u] = make_uint4(0u, 0u, 0u, 0u);
            }
        };
        round_trip();
        if (helpers_fail && fc > 0) {                                     // (the same word for every thread: the branch is uniform)
            if (tid == 0) {
                while (__hip_atomic_load(&helpers_fail[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            round_trip();
        }
        if (early_ptr && keep_in_reg) *keep_in_reg = ev;
    } else {
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int i = tid + k * NT;
        d0[k] = -1.0f; d1[k] = -1.0f; st[k].w0 = 0; st[k].w1 = 0; r0[k] = -1; r1[k] = -1;
        if (i < q && have_index) {
            const float2 dd = *reinterpret_cast<const float2*>(knn_dist + 2 * i);
            const int2 ww = *reinterpret_cast<const int2*>(knn_word + 2 * i);
            d0[k] = dd.x; d1[k] = dd.y; st[k].w0 = ww.x; st[k].w1 = ww.y;
            if (out_wslot || lds_wslot) { const int2 rr = *reinterpret_cast<const int2*>(knn_row + 2 * i); r0[k] = rr.x; r1[k] = rr.y; }
        }
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int i = tid + k * NT;
        cn[k] = 0;
        cl_lo[k] = make_uint4(0u, 0u, 0u, 0u); cl_hi[k] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int u = 0; u < 4; ++u) rbq[k][u] = make_uint4(0u, 0u, 0u, 0u);
        if (together && i < q) {
            cn[k] = cand_cnt ? cand_cnt[i] : 5;
            if (cand_cnt) {
                cl_lo[k] = *reinterpret_cast<const uint4*>(cand_list + (size_t)i * 4);
                cl_hi[k] = *reinterpret_cast<const uint4*>(cand_list + (size_t)i * 4 + 2);
            }
            if (bw <= 16) {
                const uint4* row = reinterpret_cast<const uint4*>(cand_bits + (size_t)i * bw);
#pragma unroll
                for (int u = 0; u < 4; ++u) if (4 * u < bw) rbq[k][u] = row[u];
            }
        }
    }
    }
    // ---- round trip 2: postings keys of both neighbours (consumed after the sweeps).  row_wslot == NULL: knn_row IS the word slot -- the sharded
    //      frame's keys, or, on a pipelined handle, the ROW, whose key the registration looks up one launch later (PipeOpts::slots_from_rows): this
    //      chain, the longest of launch A, then has no second round trip -- and none in flight in front of the sweeps' waits
    if (row_wslot) {
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int32_t a = row_wslot[max(r0[k], 0)], b = row_wslot[max(r1[k], 0)];
            st[k].ws_a = r0[k] >= 0 ? a : -1;
            st[k].ws_b = r1[k] >= 0 ? b : -1;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KPT; ++k) { st[k].ws_a = r0[k]; st[k].ws_b = r1[k]; }     // (-1 where there is no neighbour)
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int i = tid + k * NT;
        Dsc& S = st[k];
        S.nc = 0; S.overflow = cn[k] > 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { S.cj[e] = 0; S.cd[e] = 0.0f; }
        if (cn[k] > 0 && cn[k] <= 4) {
            const uint4 lo = cl_lo[k];
            const uint4 hi = cn[k] > 2 ? cl_hi[k] : make_uint4(0u, 0u, 0u, 0u);
            S.cj[0] = (int)lo.x; S.cd[0] = __uint_as_float(lo.y); S.cj[1] = (int)lo.z; S.cd[1] = __uint_as_float(lo.w);
            S.cj[2] = (int)hi.x; S.cd[2] = __uint_as_float(hi.y); S.cj[3] = (int)hi.z; S.cd[3] = __uint_as_float(hi.w);
            S.nc = cn[k];
        }
        // more than four candidates (a word that occurs many times in the frame makes all its descriptors candidates of each other):
        // keep the whole bit row in registers; a sweep then ANDs it with the new-word mask and touches memory only for the hits
#pragma unroll
        for (int u = 0; u < 16; ++u) S.rb[u] = 0u;
        if (S.overflow && bw <= 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint4 v = rbq[k][u]; S.rb[4 * u] = v.x; S.rb[4 * u + 1] = v.y; S.rb[4 * u + 2] = v.z; S.rb[4 * u + 3] = v.w; }
            const int wlast = i >> 5;                            // only j < i
#pragma unroll
            for (int u = 0; u < 16; ++u) { if (u == wlast) S.rb[u] &= (1u << (i & 31)) - 1u; else if (u > wlast) S.rb[u] = 0u; }
        }
    }
    // ---- the indexed candidates (they do not change from sweep to sweep), :1092-1137: stop at the first invalid neighbour;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int i = tid + k * NT;
        Dsc& S = st[k];
        S.nb = 0; S.b0.d = 0.f; S.b0.id = 0; S.b1.d = 0.f; S.b1.id = 0;
        if (i < q && have_index) {
            if (d0[k] >= 0.0f && S.w0 != 0) { cand_push(S.b0, S.b1, S.nb, d0[k], S.w0); if (d1[k] >= 0.0f && S.w1 != 0) cand_push(S.b0, S.b1, S.nb, d1[k], S.w1); }
        }
        S.reject = i < q && incremental && (S.nb < 2 || S.b0.d > nndr * S.b1.d);
        S.win = S.nb > 0 ? S.b0.id : 0;
    }
    RB_STAMP(1);
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int i = tid + k * NT;
        if (i < qpad) {
            const unsigned long long bal = __ballot(st[k].reject);
            if (lane == 0) { mask_cur[i >> 5] = (uint32_t)bal; mask_cur[(i >> 5) + 1] = (uint32_t)(bal >> 32); }
        }
    }
    lds_barrier();
    RB_STAMP(2);
    if (together) {
        for (int sweep = 0; sweep <= q; ++sweep) {
            SW_STAMP(0);
            if (tid == 0) s_changed_f = 0;
            lds_barrier();
            SW_STAMP(1);
            // the new-word mask of this sweep in registers (frames of up to 512 descriptors): the bit-row path below then ANDs registers
            // -- sixteen dependent LDS round trips per descriptor with a long candidate list cost a wave ~2 us per sweep
            uint32_t mreg[16];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                                     // mw is even: 8-byte reads are always aligned
                uint2 v = make_uint2(0u, 0u);
                if (bw <= 16 && 2 * u < mw) v = *reinterpret_cast<const uint2*>(mask_cur + 2 * u);
                mreg[2 * u] = v.x; mreg[2 * u + 1] = v.y;
            }
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const int i = tid + k * NT;
                Dsc& S = st[k];
                if (i < q && (S.nc > 0 || S.overflow)) {                   // only these descriptors can change their mind
                    Cand c0 = S.b0, c1 = S.b1; int n = S.nb;
                    uint64_t b = KEY_NONE, sk = KEY_NONE;
                    if (!S.overflow) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = S.cj[e];
                            if (e < S.nc && ((mask_cur[j >> 5] >> (j & 31)) & 1u)) {         // j became a new word
                                const uint64_t key = ((uint64_t)__float_as_uint(S.cd[e]) << 32) | (uint32_t)j;
                                const uint64_t hi = b > key ? b : key;
                                b = b < key ? b : key;
                                sk = sk < hi ? sk : hi;
                            }
                        }
                    } else if (bw <= 16) {
                        // (the descriptors with long lists are the copies of a place's popular words: their candidates are each other, none of
                        // them a new word -- one branch-free test says so and the walk below, 16 words x a data-dependent loop, is skipped: it was
                        // most of the 0.4-1.1 us a descriptor cost a sweep in round 6's stamps)
                        // (finding the first hit of all the thread's descriptors first and requesting their distances together -- one wait per sweep
                        // instead of one per descriptor with a hit -- was measured and lost: launch A 12.4 -> 12.8 us, profiles/r06_ab_notes.txt item 9)
                        uint32_t any = 0u;
#pragma unroll
                        for (int w = 0; w < 16; ++w) any |= S.rb[w] & mreg[w];
                        if (any)
#pragma unroll
                        for (int w = 0; w < 16; ++w) {
                            uint32_t m = S.rb[w] & mreg[w];
                            while (m) {
                                const int j = (w << 5) + __builtin_ctz(m);
                                m &= m - 1;
                                const uint64_t key = ((uint64_t)__float_as_uint(selfdist[(size_t)j * ld + i]) << 32) | (uint32_t)j;
                                const uint64_t hi = b > key ? b : key;
                                b = b < key ? b : key;
                                sk = sk < hi ? sk : hi;
                            }
                        }
                    } else {
                        const int wlast = i >> 5;
                        for (int w = 0; w <= wlast; ++w) {
                            uint32_t m = cand_bits[(size_t)i * bw + w] & mask_cur[w];
                            if (w == wlast) m &= (1u << (i & 31)) - 1u;
                            while (m) {
                                const int j = (w << 5) + __builtin_ctz(m);
                                m &= m - 1;
                                const uint64_t key = ((uint64_t)__float_as_uint(selfdist[(size_t)j * ld + i]) << 32) | (uint32_t)j;
                                const uint64_t hi = b > key ? b : key;
                                b = b < key ? b : key;
                                sk = sk < hi ? sk : hi;
                            }
                        }
                    }
                    if (b != KEY_NONE) cand_push(c0, c1, n, __uint_as_float((uint32_t)(b >> 32)), -((int)(uint32_t)b + 1));
                    if (sk != KEY_NONE) cand_push(c0, c1, n, __uint_as_float((uint32_t)(sk >> 32)), -((int)(uint32_t)sk + 1));
                    S.reject = n < 2 || c0.d > nndr * c1.d;
                    S.win = n > 0 ? c0.id : 0;
                }
                SW_STAMP(2 + k);
                if (i < qpad) {
                    const unsigned long long bal = __ballot(S.reject);
                    if (lane == 0) {
                        const uint32_t lo = (uint32_t)bal, hi = (uint32_t)(bal >> 32);
                        mask_next[i >> 5] = lo; mask_next[(i >> 5) + 1] = hi;
                        if (lo != mask_cur[i >> 5] || hi != mask_cur[(i >> 5) + 1]) s_changed_f = 1;
                    }
                }
            }
            SW_STAMP(6);
            lds_barrier();
            SW_STAMP(7);
            uint32_t* t = mask_cur; mask_cur = mask_next; mask_next = t;
            if (!s_changed_f) break;
            lds_barrier();
        }
    }
    RB_STAMP(3);
    // word prefix sums of the final mask -> ranks of the new words in descriptor order (getNextId() order, :1185): one wavefront
    if (tid < 64) {
        uint32_t run = 0;
        for (int w0 = 0; w0 < mw; w0 += 64) {
            const int w = w0 + tid;
            const uint32_t c = w < mw ? (uint32_t)__popc(mask_cur[w]) : 0u;
            uint32_t x = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
            if (w < mw) prefix[w] = run + x - c;
            run += __shfl(x, 63, 64);
        }
        if (tid == 0) prefix[mw] = run;                              // (out_n_new is stored with the word ids below: no store in front of a read)
    }
    lds_barrier();
    RB_STAMP(4);
    // every descriptor's word and postings key first, the stores behind them: the keys of new words are looked up in the launch
    // arguments (memory reads), and a read that follows a store waits for the store's acknowledgement (one in-order counter) --
    // store / look-up / store / ... was four round trips at the end of this chain
    int wv_[KPT]; int32_t wsv_[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int i = tid + k * NT;
        const Dsc& S = st[k];
        wv_[k] = 0; wsv_[k] = -1;
        if (i >= q) continue;
        const bool is_new = (mask_cur[i >> 5] >> (i & 31)) & 1u;
        int w;
        if (is_new) w = -(new_rank(mask_cur, prefix, i) + 1);
        else {
            w = S.win;
            if (w < 0) w = -(new_rank(mask_cur, prefix, -w - 1) + 1);   // matched a same-frame new word
        }
        int32_t ws = -1;
        if (w < 0 && new_ws.n > 0) { ws = ws_runs_at_dev(new_ws, -w - 1); if (slots_are_rows) ws = ws >= 0 ? -(ws + 2) : -1; }   // (a key among rows: marked)
        if (w < 0 && new_ws.n < 0) ws = w - 1;                          // split tail: code -(k + 2), translated by the registration workgroup
        if (w > 0) { if (S.w0 == w) ws = S.ws_a; else if (S.w1 == w) ws = S.ws_b; }
        wv_[k] = w; wsv_[k] = ws;
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) asm volatile("" : "+v"(wv_[k]), "+v"(wsv_[k]));
    if (keep_in_reg) asm volatile("" : "+v"(*keep_in_reg));              // a value the caller reads right after the stores below
    if (tid == 0) out_n_new[0] = (int32_t)prefix[mw];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int i = tid + k * NT;
        if (i >= q) continue;
        out_word[i] = wv_[k];
        if (lds_wslot) lds_wslot[i] = wsv_[k];
        else if (out_wslot) out_wslot[i] = wsv_[k];
    }
    RB_STAMP(5);
    return mask_cur;                                   // the final new-word mask (its word prefix sums are at rs_smem + 2 * mw)
}

// The whole decision loop for one frame, executed by ONE workgroup of NT threads.  rs_smem: 3 * mw + 2 words of LDS,
// mw = ceil(q / 64) * 2.
template <int NT>
__device__ __forceinline__ const uint32_t* resolve_body(uint32_t* rs_smem, int q, int flags, float nndr, int have_index,
                                                         const int32_t* __restrict__ knn_word, const float* __restrict__ knn_dist,
                                                         const float* __restrict__ selfdist, int ld,
                                                         const uint32_t* __restrict__ cand_bits, int bw,
                                                         int32_t* __restrict__ out_word, int32_t* __restrict__ out_n_new,
                                                         const int32_t* __restrict__ knn_row, const int32_t* __restrict__ row_wslot,
                                                         int32_t* __restrict__ out_wslot, const WsRuns& new_ws, bool slots_are_rows = false) {
    // rs_smem: mask_a[mw] | mask_b[mw] | prefix[mw + 1]
    const int mw = (q + 63) / 64 * 2;                 // mask words (a whole number of waves)
    uint32_t* mask_cur = rs_smem;
    uint32_t* mask_next = rs_smem + mw;
    uint32_t* prefix = rs_smem + 2 * mw;
    __shared__ int s_changed;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool incremental = (flags & LCD_Q_INCREMENTAL) != 0;
    const bool together = incremental && (flags & LCD_Q_NEW_WORDS_COMPARED) && cand_bits != nullptr;
    const int qpad = mw * 32;

    // sweep 0: decide from the indexed candidates only; out_word holds the current winner of every descriptor
    for (int i = tid; i < qpad; i += NT) {
        bool reject = false;
        if (i < q) {
            Cand c0, c1; int n;
            gather_candidates(i, false, have_index, knn_word, knn_dist, selfdist, ld, cand_bits, bw, mask_cur, c0, c1, n);
            reject = incremental && (n < 2 || c0.d > nndr * c1.d);
            out_word[i] = n > 0 ? c0.id : 0;
        }
        const unsigned long long bal = __ballot(reject);
        if (lane == 0) { mask_cur[(i >> 5)] = (uint32_t)bal; mask_cur[(i >> 5) + 1] = (uint32_t)(bal >> 32); }
    }
    __syncthreads();
    if (together) {
        for (int sweep = 0; sweep <= q; ++sweep) {
            if (tid == 0) s_changed = 0;
            __syncthreads();
            for (int i = tid; i < qpad; i += NT) {
                bool reject = false;
                if (i < q) {
                    Cand c0, c1; int n;
                    gather_candidates(i, true, have_index, knn_word, knn_dist, selfdist, ld, cand_bits, bw, mask_cur, c0, c1, n);
                    reject = n < 2 || c0.d > nndr * c1.d;
                    out_word[i] = n > 0 ? c0.id : 0;
                }
                const unsigned long long bal = __ballot(reject);
                if (lane == 0) {
                    const uint32_t lo = (uint32_t)bal, hi = (uint32_t)(bal >> 32);
                    mask_next[i >> 5] = lo; mask_next[(i >> 5) + 1] = hi;
                    if (lo != mask_cur[i >> 5] || hi != mask_cur[(i >> 5) + 1]) s_changed = 1;
                }
            }
            __syncthreads();
            uint32_t* t = mask_cur; mask_cur = mask_next; mask_next = t;
            if (!s_changed) break;
            __syncthreads();
        }
    }
    // word prefix sums of the final mask -> ranks of the new words in descriptor order (getNextId() order, :1185)
    if (tid == 0) {
        uint32_t run = 0;
        for (int w = 0; w < mw; ++w) { prefix[w] = run; run += __popc(mask_cur[w]); }
        prefix[mw] = run;
        out_n_new[0] = (int32_t)run;
    }
    __syncthreads();
    for (int i = tid; i < q; i += NT) {
        const bool is_new = (mask_cur[i >> 5] >> (i & 31)) & 1u;
        int w;
        if (is_new) w = -(new_rank(mask_cur, prefix, i) + 1);
        else {
            w = out_word[i];                           // winner of the last sweep (own entry: no cross-thread read)
            if (w < 0) w = -(new_rank(mask_cur, prefix, -w - 1) + 1);   // matched a same-frame new word
        }
        out_word[i] = w;                               // fixed dictionary without candidate: 0 ("no entry", :1211-1218)
        if (out_wslot) {
            // postings key of the chosen EXISTING word: it is one of the descriptor's two indexed neighbours
            // (row_wslot == NULL: knn_row already holds the postings key of each neighbour -- sharded mode)
            // A NEW word (created by this descriptor or by an earlier one of the frame it matched) references the frame's signature
            // too -- the VisualWord constructor does addRef(signatureId), VWDictionary.cpp:1185 -- under the k-th key the caller reserved
            // for the frame's new words; without a reservation new words get no posting.
            int32_t ws = -1;
            if (w < 0 && new_ws.n > 0) { ws = ws_runs_at_dev(new_ws, -w - 1); if (slots_are_rows) ws = ws >= 0 ? -(ws + 2) : -1; }
            if (w < 0 && new_ws.n < 0) ws = w - 1;                      // split tail: code -(k + 2) (see resolve_body_fast)
            if (w > 0) {
                if (knn_word[2 * i] == w) ws = row_wslot ? row_wslot[knn_row[2 * i]] : knn_row[2 * i];
                else if (knn_word[2 * i + 1] == w) ws = row_wslot ? row_wslot[knn_row[2 * i + 1]] : knn_row[2 * i + 1];
            }
            out_wslot[i] = ws;
        }
    }
    return mask_cur;
}


}  // namespace
}  // namespace lcd
