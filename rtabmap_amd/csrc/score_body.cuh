// score_body.cuh -- device code of the TF-IDF scoring launch (see tfidf.h for the index layout): one workgroup per sealed bucket,
// one wavefront per signature of the bucket that is still filling.  Shared by the stand-alone score_kernel of tfidf.hip and by the
// fused pipeline launch of knn_mfma_kernels.hip (where the scoring of frame t - 1 rides in the re-rank launch of frame t).
#pragma once
#include "frame_tail_body.cuh"

namespace lcd {
namespace {

#ifdef LCD_SCORE_TIMING   // timing experiment only: 100 MHz stamps between the phases of score_sealed_body, per workgroup
__device__ unsigned long long g_score_timing[1024 * 8];
#define SC_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 1024) g_score_timing[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SC_STAMP(i) do { } while (0)
#endif



// One workgroup scores one sealed bucket (256 signatures).
//   dense rows : wavefront v takes the frame's dense words v, v + NWV, ...; a lane reads the four counts of its four signatures
//                with one 4-byte load (the wavefront reads the 256-byte row in one coalesced request) and keeps four 64-bit
//                sums in registers; all loads of a trip are issued before any is consumed;
//   sparse part: one thread per frame word looks the word up in the bucket's directory (one 8-byte read; a second one for the
//                offsets when the word is present).  The segments of the 64 words of a wavefront are then walked by that
//                wavefront alone: lane-wise inclusive scan of the lengths, every lane finds the segment of "its" posting with six
//                cross-lane reads (no LDS arrays, no workgroup barrier, no per-workgroup scan) and adds count x idf with an LDS
//                64-bit atomic;
//   output     : acc / ni, written straight from LDS.
// The whole body is a chain of dependent global reads (word list -> directory block -> segment offsets -> postings; dense list
// -> rows): every wavefront issues ALL independent loads of a stage before it consumes any of them (loads return in order, so
// waiting for an older one leaves the younger ones in flight).  LDS: acc[256] i64 | ni[256] (3 KB, static).
template <int SCB>
__device__ __forceinline__ void score_segments(const uint32_t* __restrict__ sp_ent, unsigned long long* acc, uint32_t start, uint32_t len, int32_t idf) {
    const int ln = threadIdx.x & 63;
    uint32_t incl = len;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(incl, off, 64); if (ln >= off) incl += y; }
    const uint32_t Tw = __shfl(incl, 63, 64);                       // wave-uniform
    for (uint32_t t0 = 0; t0 < Tw; t0 += 128) {
        uint32_t e[2]; int32_t f[2]; bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t t = t0 + (uint32_t)(u * 64 + ln);
            int pos = 0;                                            // number of lanes whose inclusive sum is <= t = the owner of posting t
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) { const uint32_t v = __shfl(incl, pos + step - 1, 64); if (v <= t) pos += step; }
            if (pos > 63) pos = 63;
            const uint32_t i_o = __shfl(incl, pos, 64), l_o = __shfl(len, pos, 64), s_o = __shfl(start, pos, 64);
            f[u] = __shfl(idf, pos, 64);
            ok[u] = t < Tw;
            e[u] = ok[u] ? gload(sp_ent + s_o + (t - (i_o - l_o))) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
            const long long term = (long long)(int)(e[u] & TF_CNT_MASK) * (long long)f[u];
            atomicAdd(&acc[e[u] >> TF_CNT_BITS], (unsigned long long)term);      // ds_add_u64
        }
    }
}

template <int SCB>
__device__ __forceinline__ void score_sealed_body(const ScoreArgs& A, int b) {
    __shared__ unsigned long long acc[TF_R];
    const int tid = threadIdx.x;
    const BucketDev B = A.tab[b];
    const long long first_slot = (long long)b * TF_R;
    // A bucket whose signatures are all retired has no rows and no directory below (D = 0, W = 0: every sum stays 0, and ni = 0 writes
    // 0 like the reference's "if(ni != 0)").  No early exit on B.state: the branch would put the bucket record's round trip in front
    // of every other load of the workgroup.
    const bool live = B.state == 1u;
    SC_STAMP(0);
    constexpr int NWV = SCB / 64;
    // dense rows per wavefront and trip.  (24 rows for the 4-wave workgroups of the fused launch -- one trip instead of two for the ~95
    // dense words of a frame -- measured SLOWER on the same box: launch B 21.1 us against 18.7; more loads in flight per wave, fewer
    // waves resident.  -DLCD_SCORE_DR4=24 rebuilds the experiment.)
#ifndef LCD_SCORE_DR4
#define LCD_SCORE_DR4 16
#endif
    constexpr int DR = NWV == 4 ? LCD_SCORE_DR4 : (128 / NWV > 16 ? 16 : 128 / NWV);
    constexpr int KW = SCB >= 512 ? 1 : 512 / SCB;                  // frame words per thread in the fused first pass
    constexpr int NI = TF_R / SCB > 0 ? TF_R / SCB : 1;             // signatures whose ni a thread carries
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    const uint32_t D = live ? A.bkt_D[b] : 0u;
    const uint32_t BW = live ? B.W : 0u;
    const uint32_t flags = A.bkt_flags[b];
    const int U = (int)A.q_meta[0];
    const int Ud = (int)A.q_meta[1];
    // ---- stage A: the frame's lists (every workgroup reads the same few KB), ni.  The lists are read WITHOUT looking at their lengths
    //      first (the buffers hold TF_MAX_WORDS entries; what lies beyond U / Ud is masked afterwards): one round trip for the bucket
    //      record, the lengths and the lists instead of two.
    static_assert(KW * SCB <= TF_MAX_WORDS && DR * NWV <= TF_MAX_WORDS, "the unconditional list reads stay inside the buffers");
    uint32_t w[KW]; int32_t idf[KW], did[KW]; bool look[KW];
#pragma unroll
    for (int u = 0; u < KW; ++u) {
        const int k = tid + u * SCB;
        w[u] = A.q_w[k]; idf[u] = A.q_idf[k]; did[u] = A.q_did[k];
    }
    int32_t dj[DR], fj[DR];
#pragma unroll
    for (int u = 0; u < DR; ++u) {
        const int j = wv + u * NWV;                                  // wave-uniform: scalar loads
        dj[u] = A.qd_did[j];
        fj[u] = A.qd_idf[j];
    }
#pragma unroll
    for (int u = 0; u < KW; ++u) { if (tid + u * SCB >= U) { w[u] = 0; idf[u] = 0; did[u] = -1; } }
#pragma unroll
    for (int u = 0; u < DR; ++u) { if (wv + u * NWV >= Ud) { dj[u] = -1; fj[u] = 0; } }
    uint32_t ni_v[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) { const int i = tid + u * SCB; ni_v[u] = i < TF_R ? A.slot_ni[first_slot + i] : 0u; }
    // ---- stage B: directory blocks of the sparse words, dense rows
    uint4 r0[KW], r1[KW];                                              // the word's record of the word-major directory (32 bytes)
#pragma unroll
    for (int u = 0; u < KW; ++u) {
        const bool dense_here = did[u] >= 0 && (uint32_t)did[u] < D;
        look[u] = idf[u] != 0 && w[u] < BW && (!dense_here || (flags & 1u));   // a dense word has sparse postings only for counts > 255
        r0[u] = make_uint4(0u, 0u, 0u, 0u); r1[u] = r0[u];
        if (look[u]) {
            const uint4* rec = reinterpret_cast<const uint4*>(A.dir2 + ((size_t)(w[u] >> 5) * A.dir2_stride + (uint32_t)b) * TF_DIR2_DWORDS);
            r0[u] = gload4(rec); r1[u] = gload4(rec + 1);
        }
    }
    uint32_t c[DR];
#pragma unroll
    for (int u = 0; u < DR; ++u) c[u] = (dj[u] >= 0 && (uint32_t)dj[u] < D) ? gload((const uint32_t*)(B.dense + (size_t)dj[u] * TF_R + 4 * ln)) : 0u;
    // ---- stage C: start and length of the words' posting segments, from the record alone (a block with a count that does not
    //      fit its 5-bit field goes through the per-bucket directory: a dependent lookup, rare)
    uint32_t start[KW], len[KW];
#pragma unroll
    for (int u = 0; u < KW; ++u) {
        start[u] = 0; len[u] = 0;
        if (!look[u]) continue;
        const uint32_t f[6] = {r0[u].z, r0[u].w, r1[u].x, r1[u].y, r1[u].z, r1[u].w};
        const uint32_t p = w[u] & 31u, pd = p / 6u, ps = 5u * (p % 6u);
        if (!(r0[u].y & TF_DIR2_SAT)) {
            uint32_t before = 0;
#pragma unroll
            for (uint32_t d = 0; d < 6u; ++d) {
                const uint32_t x = d < pd ? f[d] : (d == pd ? (f[d] & ((1u << ps) - 1u)) : 0u);
                before += dir2_sum6(x);
            }
            uint32_t fd = f[0];
#pragma unroll
            for (uint32_t d = 1; d < 6u; ++d) fd = d == pd ? f[d] : fd;
            len[u] = (fd >> ps) & 31u;
            start[u] = r0[u].x + before;
        } else {
            const uint2 blk = gload2(B.dirb + (w[u] >> 5));
            const uint32_t bit = 1u << p;
            if (blk.x & bit) {
                const uint32_t r = blk.y + (uint32_t)__popc(blk.x & (bit - 1u));
                const uint32_t s0 = gload(B.sp_off + r), s1 = gload(B.sp_off + r + 1);
                start[u] = s0; len[u] = s1 - s0;
            }
        }
    }
    // the dense rows: a lane owns four signatures
    long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int u = 0; u < DR; ++u) {
        const long long f64 = (long long)fj[u];
        a0 += (long long)(int)(c[u] & 255u) * f64;
        a1 += (long long)(int)((c[u] >> 8) & 255u) * f64;
        a2 += (long long)(int)((c[u] >> 16) & 255u) * f64;
        a3 += (long long)(int)(c[u] >> 24) * f64;
    }
    for (int j0 = wv + DR * NWV; j0 < Ud; j0 += DR * NWV) {           // frames with more than 128 dense words
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            const int j = j0 + u * NWV;
            dj[u] = j < Ud ? A.qd_did[j] : -1;
            fj[u] = j < Ud ? A.qd_idf[j] : 0;
        }
#pragma unroll
        for (int u = 0; u < DR; ++u) c[u] = (dj[u] >= 0 && (uint32_t)dj[u] < D) ? gload((const uint32_t*)(B.dense + (size_t)dj[u] * TF_R + 4 * ln)) : 0u;
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            const long long f64 = (long long)fj[u];
            a0 += (long long)(int)(c[u] & 255u) * f64;
            a1 += (long long)(int)((c[u] >> 8) & 255u) * f64;
            a2 += (long long)(int)((c[u] >> 16) & 255u) * f64;
            a3 += (long long)(int)(c[u] >> 24) * f64;
        }
    }
    for (int i = tid; i < TF_R; i += SCB) acc[i] = 0ull;
    __syncthreads();                                                 // accumulators zeroed
    SC_STAMP(1);
    {
        const int ln4 = ln * 4;
        if (a0) atomicAdd(&acc[ln4 + 0], (unsigned long long)a0);
        if (a1) atomicAdd(&acc[ln4 + 1], (unsigned long long)a1);
        if (a2) atomicAdd(&acc[ln4 + 2], (unsigned long long)a2);
        if (a3) atomicAdd(&acc[ln4 + 3], (unsigned long long)a3);
    }
    SC_STAMP(2);
    // ---- sparse postings, wavefront by wavefront
#pragma unroll
    for (int u = 0; u < KW; ++u) score_segments<SCB>(B.sp_ent, acc, start[u], len[u], idf[u]);
    for (int k0 = KW * SCB; k0 < U; k0 += SCB) {                     // frames with more than 512 unique words
        const int k = k0 + tid;
        uint32_t st2 = 0, ln2 = 0; int32_t idf2 = 0;
        if (k < U) {
            const uint32_t w2 = A.q_w[k];
            idf2 = A.q_idf[k];
            const int32_t d2 = A.q_did[k];
            const bool dh = d2 >= 0 && (uint32_t)d2 < D;
            if (idf2 != 0 && w2 < BW && (!dh || (flags & 1u))) {
                const uint2 bk = gload2(B.dirb + (w2 >> 5));
                const uint32_t bit = 1u << (w2 & 31);
                if (bk.x & bit) {
                    const uint32_t r = bk.y + (uint32_t)__popc(bk.x & (bit - 1u));
                    st2 = gload(B.sp_off + r);
                    ln2 = gload(B.sp_off + r + 1) - st2;
                }
            }
        }
        score_segments<SCB>(B.sp_ent, acc, st2, ln2, idf2);
    }
    __syncthreads();
    SC_STAMP(3);
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + u * SCB;
        if (i >= TF_R) continue;
        const long long v = (long long)acc[i];
        if (A.out_like) A.out_like[first_slot + i] = fixed_to_like(v, ni_v[u]);
        else A.out_fix[first_slot + i] = ni_v[u] ? v : 0;
    }
}

// The bucket that is still filling (<= 256 signatures): one wavefront per signature walks the signature's own stretch of the
// arrival-order log, keeps the postings whose word belongs to the frame (idf_tab stamp) and reduces them inside the wave --
// no atomics, no second pass, every slot written exactly once.
template <int SCB>
__device__ __forceinline__ void score_open_body(const ScoreArgs& A, int ob) {
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int sl = ob * (SCB / 64) + wv;
    if (sl >= A.n_open_slots) return;
    const BucketDev B = A.tab[A.n_closed];
    const long long slot = (long long)A.n_closed * TF_R + sl;
    const uint32_t begin = A.slot_begin[slot], cnt = A.slot_cnt[slot], ni = A.slot_ni[slot];
    long long acc = 0;
    if (ni != 0u) {
        for (uint32_t e0 = 0; e0 < cnt; e0 += 4 * 64) {
            uint32_t w[4], pc[4]; uint2 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t e = e0 + u * 64 + ln;
                w[u] = e < cnt ? gload(B.coo_w + begin + e) : 0xFFFFFFFFu;
                pc[u] = e < cnt ? gload(B.coo_pc + begin + e) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = A.idf_tab[w[u] != 0xFFFFFFFFu ? w[u] : 0u];      // unconditional: four loads in flight
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (w[u] != 0xFFFFFFFFu && t[u].x == A.stamp) acc += (long long)(int)(pc[u] & TF_CNT_MASK) * (long long)(int32_t)t[u].y;
        }
    }
    int lo = (int)(uint32_t)acc, hi = (int)(acc >> 32);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int olo = __shfl_xor(lo, off, 64), ohi = __shfl_xor(hi, off, 64);
        const long long s = (((long long)hi << 32) | (uint32_t)lo) + (((long long)ohi << 32) | (uint32_t)olo);
        lo = (int)(uint32_t)s; hi = (int)(s >> 32);
    }
    if (ln == 0) {
        const long long v = ((long long)hi << 32) | (uint32_t)lo;
        if (A.out_like) A.out_like[slot] = fixed_to_like(v, ni);
        else A.out_fix[slot] = ni ? v : 0;
    }
}


}  // namespace
}  // namespace lcd
