// rowpar_body.cuh -- the exact redo of the queries the MFMA-filter certificate rejected (device code shared by the stand-alone
// knn_rowpar_kernel and by the extra workgroups of the fused frame tail, tfidf.hip), plus the small key helpers both use.
#pragma once
#include "lcd_kernels.h"

namespace lcd {
namespace {

__device__ __forceinline__ void top2_push(uint64_t& best, uint64_t& second, uint64_t k) {
    const uint64_t hi = best > k ? best : k;
    best = best < k ? best : k;
    second = second < hi ? second : hi;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}


// Row `qi` of the candidate bit matrix of the addNewWords resolution (knn2_kernels.hip: bit r = dist(r, qi) < distance of qi's
// second indexed neighbour) from the already computed same-frame distance matrix, which is symmetric bit for bit.  Called by
// whole waves (n_threads a multiple of 64); writes all bw words of the row.
__device__ __forceinline__ void cand_bits_row(const CandBits& cb, int qi, float thr, int tid, int n_threads) {
    int below = 0;                                                  // set bits below qi so far (called by ONE wave: n_threads == 64)
    for (int base = (tid >> 6) * 64; base < cb.ld; base += n_threads) {
        const int r = base + (tid & 63);
        const float d = r < cb.nq ? cb.selfdist[(size_t)qi * cb.ld + r] : __int_as_float(0x7f800000);
        const unsigned long long m = __ballot(d < thr);
        if ((tid & 63) == 0) *reinterpret_cast<unsigned long long*>(cb.bits + (size_t)qi * cb.bw + (base >> 5)) = m;
        if (cb.cnt && n_threads == 64) {
            const unsigned long long m2 = __ballot(d < thr && r < qi);
            const int pos = below + (int)__popcll(m2 & ((1ull << (tid & 63)) - 1ull));
            if (d < thr && r < qi && pos < CAND_LIST) cb.list[(size_t)qi * CAND_LIST + pos] = make_uint2((uint32_t)r, __float_as_uint(d));
            below += (int)__popcll(m2);
        }
    }
    if (cb.cnt && n_threads == 64 && (tid & 63) == 0) cb.cnt[qi] = below;
}


// The queries the certificate rejects (usually none, sometimes a handful) are redone exactly with the WHOLE chip on each of
// them: one lane per vocabulary row (the row stays in VGPRs), the listed queries are looped over (query broadcast from LDS),
// every workgroup reduces to its two best keys per query and the LAST workgroup to arrive (agent-scope release / acquire around
// a counter, cdna_hip_programming.md guideline 16) merges them, writes the result into the query's own slot (and its row of
// candidate bits) and raises fail_count[3] for whoever waits for the redo.  Leaves at once when the list is empty.
//   fail_count: [0] rejected queries, [1] arrival counter, [3] done flag.   wb / n_wb: this workgroup's index / the number of
//   workgroups walking the rows; the rows are walked in chunks of NT (chunk c by workgroup c % n_wb): a launch may bring fewer
//   workgroups than there are chunks -- the fused frame launch does (REDO_WGS_MAX): its helpers queue for compute-unit slots behind the
//   filter's workgroups, ~400 of them kept the launch open 1.5-2 us after everything else had finished (round 6's stamps).
// Returns 0: nothing to redo (every workgroup of the launch sees that); 1: this workgroup's share is done; 2: this workgroup was the last,
// it merged, and the results of the whole search are final.
template <int DIM, int NT>
__device__ __forceinline__ int rowpar_body(const RowparArgs& a, int wb, int n_wb, int32_t* __restrict__ fail_count) {
    const int nf = fail_count[0];
    if (nf <= 0) return 0;
    constexpr int NW = NT / 64;
    __shared__ float s_q[DIM];
    __shared__ uint64_t s_k[NW][2];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_chunks = (a.n_rows + NT - 1) / NT;
    const int n_rows_now = a.n_rows_dev ? min(a.n_rows_dev[0], a.n_rows) : a.n_rows;
    for (int ch = wb; ch < n_chunks; ch += n_wb) {
    const int row = ch * NT + (int)threadIdx.x;
    const bool live = row < n_rows_now && a.row_id[min(row, a.n_rows - 1)] != 0;
    float v[DIM];
    {
        const float4* src = reinterpret_cast<const float4*>(a.vocab + (size_t)min(row, a.n_rows - 1) * DIM);
#pragma unroll
        for (int g = 0; g < DIM / 4; ++g) { const float4 x = src[g]; v[4 * g] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w; }
    }
    for (int f = 0; f < nf; ++f) {
        __syncthreads();
        if (threadIdx.x < DIM) s_q[threadIdx.x] = a.queries[(size_t)a.fail_list[f] * DIM + threadIdx.x];
        __syncthreads();
        float res = 0.0f;                              // rtflann::L2 (dist.h:150-177), a = row, b = query
#pragma unroll
        for (int g = 0; g + 3 < DIM; g += 4) {
            const float d0 = __fsub_rn(v[g + 0], s_q[g + 0]);
            const float d1 = __fsub_rn(v[g + 1], s_q[g + 1]);
            const float d2 = __fsub_rn(v[g + 2], s_q[g + 2]);
            const float d3 = __fsub_rn(v[g + 3], s_q[g + 3]);
            float t = __fmul_rn(d0, d0);
            t = __fadd_rn(t, __fmul_rn(d1, d1));
            t = __fadd_rn(t, __fmul_rn(d2, d2));
            t = __fadd_rn(t, __fmul_rn(d3, d3));
            res = __fadd_rn(res, t);
        }
        uint64_t best = live ? (((uint64_t)__float_as_uint(res) << 32) | (uint32_t)row) : KEY_NONE, second = KEY_NONE;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint64_t ob = shfl_xor_u64(best, m), os = shfl_xor_u64(second, m);
            top2_push(best, second, ob);
            top2_push(best, second, os);
        }
        if (lane == 0) { s_k[wave][0] = best; s_k[wave][1] = second; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < NW; ++w) { top2_push(best, second, s_k[w][0]); top2_push(best, second, s_k[w][1]); }
            a.partial[((size_t)f * n_chunks + ch) * 2 + 0] = best;
            a.partial[((size_t)f * n_chunks + ch) * 2 + 1] = second;
        }
    }
    }
    // publish this workgroup's keys, find out whether it is the last one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(&fail_count[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == n_wb - 1;
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return 1;
    // last workgroup: one wave per listed query merges the n_wb * 2 keys
    const int n_keys = n_chunks * 2;
    for (int f = wave; f < nf; f += NW) {
        uint64_t best = KEY_NONE, second = KEY_NONE;
        for (int c = lane; c < n_keys; c += 64) top2_push(best, second, a.partial[(size_t)f * n_keys + c]);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint64_t ob = shfl_xor_u64(best, m), os = shfl_xor_u64(second, m);
            top2_push(best, second, ob);
            top2_push(best, second, os);
        }
        const int qo = a.fail_list[f];
        if (lane == 0) {
            const uint64_t k[2] = {best, second};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (k[j] == KEY_NONE) { a.out_row[2 * qo + j] = -1; a.out_word[2 * qo + j] = 0; a.out_dist[2 * qo + j] = -1.0f; }
                else {
                    const uint32_t r = (uint32_t)k[j];
                    a.out_row[2 * qo + j] = (int32_t)r;
                    a.out_word[2 * qo + j] = a.row_id[r];
                    a.out_dist[2 * qo + j] = __uint_as_float((uint32_t)(k[j] >> 32));
                }
            }
        }
        if (a.cb.bits) {                                              // the redone query's candidate bits
            const float thr = (a.cb.have_index && second != KEY_NONE) ? __uint_as_float((uint32_t)(second >> 32)) : __int_as_float(0x7f800000);
            cand_bits_row(a.cb, qo, thr, lane, 64);
        }
    }
    // the redo is complete: tell the waiting frame tail (if any)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&fail_count[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return 2;
}

}  // namespace
}  // namespace lcd
