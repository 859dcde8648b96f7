// bayes.hip -- BayesFilter::computePosterior on the device (see bayes.h for the layout).
//
// The arithmetic follows the reference's statements in their types (float matrix elements, the double comparisons its mixed
// float/double expressions promote to); two things are evaluated differently, both inside the float rounding the reference itself
// leaves open:  (1) prior = prediction * posterior is cv::gemm in the reference (OpenCV, not in its tree); here every row is a
// sum in double over the row's non-zeros, rounded to float once;  (2) the reference adds the posterior's entries into a float
// one by one (:205-218) before dividing; here the sum is taken in double over per-workgroup partials in a fixed order.
#include "bayes.h"

#include <algorithm>
#include <cstring>

namespace lcd {
namespace {

constexpr int BY_BLOCK = 256;
constexpr uint32_t SLOT_MASK = (1u << BAYES_SLOT_BITS) - 1u;

struct ColS { float scale; float delta; float fill; uint32_t flags; };   // flags: 1 = renormalised, 2 = the list holds the column's own slot
struct Part1 { double s_in; double fill; long long n_in; long long pad; };

__device__ __forceinline__ bool in_set(long long s, long long n_cons, const int32_t* __restrict__ slot_sig) {
    return s < n_cons && slot_sig[s] != 0;
}

// deterministic workgroup sum (fixed tree), result in every thread
__device__ __forceinline__ double block_sum(double v, double* s_red) {
    const int tid = threadIdx.x;
    s_red[tid] = v;
    __syncthreads();
    for (int off = BY_BLOCK / 2; off > 0; off >>= 1) {
        if (tid < off) s_red[tid] += s_red[tid + off];
        __syncthreads();
    }
    const double r = s_red[0];
    __syncthreads();
    return r;
}

// number of signatures taking part (only needed ahead of the column pass when the prediction has an "all other places" fill)
__global__ __launch_bounds__(BY_BLOCK) void bayes_count_kernel(long long n_cons, const int32_t* __restrict__ slot_sig, Part1* __restrict__ part) {
    __shared__ double s_red[BY_BLOCK];
    long long n = 0;
    for (long long c = (long long)blockIdx.x * BY_BLOCK + threadIdx.x; c < n_cons; c += (long long)gridDim.x * BY_BLOCK) n += slot_sig[c] != 0;
    const double t = block_sum((double)n, s_red);
    if (threadIdx.x == 0) part[blockIdx.x].n_in = (long long)t;
}

// pass 1, one thread per column c: the sum addNeighborProb returns (:237-270) and what normalize() derives from it (:434-500);
// the posterior the column is multiplied with (updatePosterior :709-736) goes to pin[1 + c]
__global__ __launch_bounds__(BY_BLOCK) void bayes_column_kernel(BayesParams prm, long long n_slots, long long n_cons, const int32_t* __restrict__ slot_sig,
                                                                const uint32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, long long cap, int K,
                                                                const uint8_t* __restrict__ was_in, const float* __restrict__ post, int empty,
                                                                int cols_known, ColS* __restrict__ col, float* __restrict__ pin, Part1* __restrict__ part) {
    __shared__ double s_red[BY_BLOCK];
    long long cols = 0;
    if (cols_known) {                                       // bayes_count_kernel ran: cols = 1 + signatures taking part
        long long n = 0;
        for (int b = 0; b < BAYES_GRID; ++b) n += part[b].n_in;
        cols = n + 1;
    }
    double s_in = 0.0, s_fill = 0.0;
    long long n_in = 0;
    for (long long c = (long long)blockIdx.x * BY_BLOCK + threadIdx.x; c < n_slots; c += (long long)gridDim.x * BY_BLOCK) {
        ColS cs = {1.0f, 0.0f, 0.0f, 0u};
        float p = 0.0f;
        if (in_set(c, n_cons, slot_sig)) {
            p = empty ? 1.0f : (was_in[c] ? post[1 + c] : 0.0f);
            float sum = 0.0f, self_v = 0.0f;
            int nz = 0;
            bool has_self = false;
            const int n = min(cnt[c], K);
            for (int k = 0; k < n; ++k) {
                const uint32_t e = nbr[(size_t)k * cap + c];
                const long long r = e & SLOT_MASK;
                if (!in_set(r, n_cons, slot_sig)) continue;
                const float v = prm.lc[(e >> BAYES_SLOT_BITS) + 1];
                sum += v;
                if (r == c) { has_self = true; self_v = v; }
                else if (v != 0.0f) ++nz;
            }
            if ((double)sum < (double)prm.total - prm.lc0) {                       // the neighbours that were not found go to the loop closure itself
                cs.delta = (float)((double)prm.total - prm.lc0 - (double)sum);
                sum += cs.delta;
            }
            if (self_v + cs.delta != 0.0f) ++nz;                                   // the diagonal element
            if (prm.all_other > 0.0f && cols > 1) {                                // every element still 0 gets a small value (:455-465)
                const float value = prm.all_other / (float)(cols - 1);
                const long long n_zero = (cols - 1) - nz;
                // the reference adds `value` n_zero times into the float; one rounded product here (non-default PredictionLC only)
                sum = (float)((double)sum + (double)value * (double)n_zero);
                cs.fill = value;
            }
            if ((double)sum < (double)prm.max_norm - 0.0001 || (double)sum > (double)prm.max_norm + 0.0001) {
                cs.scale = prm.max_norm / sum;
                cs.flags |= 1u;
                cs.fill = cs.fill * cs.scale;
                if (cs.fill < prm.eps) cs.fill = 0.0f;
            }
            if (has_self) cs.flags |= 2u;
            s_in += (double)p;
            s_fill += (double)cs.fill * (double)p;
            ++n_in;
        }
        col[c] = cs;
        pin[1 + c] = p;
    }
    const double a = block_sum(s_in, s_red), b = block_sum(s_fill, s_red), n = block_sum((double)n_in, s_red);
    if (threadIdx.x == 0) { part[blockIdx.x].s_in = a; part[blockIdx.x].fill = b; part[blockIdx.x].n_in = (long long)n; }
    if (blockIdx.x == 0 && threadIdx.x == 0) pin[0] = empty ? 1.0f : post[0];      // the virtual place is in every update
}

// one element of the prediction matrix as normalize() leaves it: v = the value addNeighborProb stored (+ delta on the diagonal)
__device__ __forceinline__ float finish_element(float v, const ColS& cs, float eps) {
    if (cs.flags & 1u) { v = v * cs.scale; if (v < eps) v = 0.0f; }
    return v;
}

// pass 2, one thread per row i: prior[i] = sum over the columns that hold i (its own neighbour list: the lists are symmetric),
// then STEP 2 (:205-218): posterior = likelihood * prior, not yet normalised
__global__ __launch_bounds__(BY_BLOCK) void bayes_row_kernel(BayesParams prm, long long n_slots, long long n_cons, const int32_t* __restrict__ slot_sig,
                                                             const uint32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, long long cap, int K,
                                                             const ColS* __restrict__ col, const float* __restrict__ pin, const Part1* __restrict__ part,
                                                             const float* __restrict__ like, float* __restrict__ post, double* __restrict__ part2) {
    __shared__ double s_red[BY_BLOCK];
    double s_in = 0.0, s_fill = 0.0;
    long long n_in = 0;
    for (int b = 0; b < BAYES_GRID; ++b) { s_in += part[b].s_in; s_fill += part[b].fill; n_in += part[b].n_in; }
    const long long cols = n_in + 1;
    const float pin_vp = pin[0];
    // the virtual place's column (:376-411): its value in every row >= 1
    float vp_col = 0.0f, p00 = 1.0f;
    if (prm.vp_prior > 0.0f) {
        if (cols > 1) { vp_col = (float)((1.0 - prm.vp_prior) / (double)(cols - 1)); p00 = prm.vp_prior; }
    } else if (cols > 1) { vp_col = (float)(1.0 / (double)cols); p00 = vp_col; }
    const double from_vp = (double)vp_col * (double)pin_vp;
    double usum = 0.0;
    for (long long i = (long long)blockIdx.x * BY_BLOCK + threadIdx.x; i < n_slots; i += (long long)gridDim.x * BY_BLOCK) {
        float u = 0.0f;
        if (in_set(i, n_cons, slot_sig)) {
            double acc = 0.0;
            bool has_self = false;
            const int n = min(cnt[i], K);
            for (int k = 0; k < n; ++k) {
                const uint32_t e = nbr[(size_t)k * cap + i];
                const long long c = e & SLOT_MASK;
                if (!in_set(c, n_cons, slot_sig)) continue;
                const ColS cs = col[c];
                float v = prm.lc[(e >> BAYES_SLOT_BITS) + 1];
                if (c == i) { v = v + cs.delta; has_self = true; }
                if (v == 0.0f) continue;                                           // an element left at 0: it holds the column's fill value
                v = finish_element(v, cs, prm.eps);
                acc += ((double)v - (double)cs.fill) * (double)pin[1 + c];
            }
            if (!has_self) {                                                       // diagonal of a column whose list does not hold itself: 0 + delta
                const ColS cs = col[i];
                if (cs.delta != 0.0f) acc += ((double)finish_element(cs.delta, cs, prm.eps) - (double)cs.fill) * (double)pin[1 + i];
            }
            const float prior = (float)(acc + s_fill + from_vp);
            u = like[1 + i] * prior;
            usum += (double)u;
        }
        post[1 + i] = u;
    }
    const double t = block_sum(usum, s_red);
    if (threadIdx.x == 0) part2[blockIdx.x] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // row 0: the virtual place's own value + Bayes/PredictionLC[0] from every other column (:486-490)
        const float prior0 = (float)((double)p00 * (double)pin_vp + (double)(float)prm.lc0 * s_in);
        post[0] = like[0] * prior0;
    }
}

// pass 3: normalise (:221-230), remember who took part, best hypothesis per workgroup (Rtabmap.cpp:2147-2158: ids > 0, highest
// posterior, the higher id on equal values)
__global__ __launch_bounds__(BY_BLOCK) void bayes_normalize_kernel(long long n_slots, long long n_cons, const int32_t* __restrict__ slot_sig,
                                                                   const double* __restrict__ part2, float* __restrict__ post, uint8_t* __restrict__ was_in,
                                                                   float* __restrict__ d_posterior, unsigned long long* __restrict__ part3) {
    __shared__ unsigned long long s_key[BY_BLOCK], s_slot[BY_BLOCK];
    double t = 0.0;
    for (int b = 0; b < BAYES_GRID; ++b) t += part2[b];
    const float u0 = post[0];
    const float sum = (float)(t + (double)u0);
    unsigned long long key = 0ull, slot = ~0ull;
    for (long long i = (long long)blockIdx.x * BY_BLOCK + threadIdx.x; i < n_slots; i += (long long)gridDim.x * BY_BLOCK) {
        const bool in = in_set(i, n_cons, slot_sig);
        float p = 0.0f;
        if (in) {
            p = post[1 + i];
            if (sum != 0.0f) p = p / sum;
            post[1 + i] = p;
            if (p > 0.0f) {
                const unsigned long long k = ((unsigned long long)__float_as_uint(p) << 32) | (unsigned long long)(uint32_t)slot_sig[i];
                if (k > key) { key = k; slot = (unsigned long long)i; }
            }
        }
        was_in[i] = in ? 1 : 0;
        if (d_posterior) d_posterior[1 + i] = p;
    }
    const int tid = threadIdx.x;
    s_key[tid] = key; s_slot[tid] = slot;
    __syncthreads();
    for (int off = BY_BLOCK / 2; off > 0; off >>= 1) {
        if (tid < off && s_key[tid + off] > s_key[tid]) { s_key[tid] = s_key[tid + off]; s_slot[tid] = s_slot[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) { part3[2 * blockIdx.x] = s_key[0]; part3[2 * blockIdx.x + 1] = s_slot[0]; }
}

// pass 4 (one workgroup): the virtual place's posterior and the best hypothesis
__global__ __launch_bounds__(BY_BLOCK) void bayes_result_kernel(const double* __restrict__ part2, const Part1* __restrict__ part, const unsigned long long* __restrict__ part3,
                                                                float* __restrict__ post, float* __restrict__ d_posterior, BayesOut* __restrict__ out) {
    __shared__ unsigned long long s_key[BY_BLOCK], s_slot[BY_BLOCK];
    const int tid = threadIdx.x;
    s_key[tid] = tid < BAYES_GRID ? part3[2 * tid] : 0ull;
    s_slot[tid] = tid < BAYES_GRID ? part3[2 * tid + 1] : ~0ull;
    __syncthreads();
    for (int off = BY_BLOCK / 2; off > 0; off >>= 1) {
        if (tid < off && s_key[tid + off] > s_key[tid]) { s_key[tid] = s_key[tid + off]; s_slot[tid] = s_slot[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        double t = 0.0;
        long long n_in = 0;
        for (int b = 0; b < BAYES_GRID; ++b) { t += part2[b]; n_in += part[b].n_in; }
        const float u0 = post[0];
        const float sum = (float)(t + (double)u0);
        const float p0 = sum != 0.0f ? u0 / sum : u0;
        post[0] = p0;
        if (d_posterior) d_posterior[0] = p0;
        if (out) {
            BayesOut o;
            const unsigned long long k = s_key[0];
            o.sig_id = k ? (int32_t)(uint32_t)k : 0;
            o.slot = k ? (int32_t)s_slot[0] : -1;
            o.posterior = __uint_as_float((uint32_t)(k >> 32));
            o.value = 1 - p0;
            o.virtual_place = p0;
            o.n_considered = (int32_t)n_in;
            o.sum = sum;
            o.reserved = 0;
            *out = o;
        }
    }
}
static_assert(BAYES_GRID <= BY_BLOCK, "the result pass reduces one partial per thread");

// neighbour lists: enter (b, margin) into a's list and (a, margin) into b's; an entry for the same neighbour is replaced
// (uInsert, BayesFilter.cpp:589).  The triples are unique, so no two threads enter the same neighbour into the same list.
__device__ __forceinline__ void list_insert(uint32_t* __restrict__ nbr, int32_t* __restrict__ cnt, long long cap, int K, int32_t a, int32_t b, uint32_t margin,
                                            unsigned long long* __restrict__ overflow) {
    const uint32_t e = (margin << BAYES_SLOT_BITS) | (uint32_t)b;
    const int n = min(__hip_atomic_load(&cnt[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), K);
    for (int k = 0; k < n; ++k) {
        if ((nbr[(size_t)k * cap + a] & SLOT_MASK) == (uint32_t)b) { nbr[(size_t)k * cap + a] = e; return; }
    }
    const int pos = atomicAdd(&cnt[a], 1);
    if (pos < K) nbr[(size_t)pos * cap + a] = e;
    else { atomicSub(&cnt[a], 1); atomicAdd(overflow, 1ull); }
}
__global__ void bayes_link_kernel(const int32_t* __restrict__ triples, int n, uint32_t* __restrict__ nbr, int32_t* __restrict__ cnt, long long cap, int K,
                                  unsigned long long* __restrict__ overflow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t a = triples[3 * i], b = triples[3 * i + 1];
    const uint32_t m = (uint32_t)triples[3 * i + 2];
    list_insert(nbr, cnt, cap, K, a, b, m, overflow);
    if (a != b) list_insert(nbr, cnt, cap, K, b, a, m, overflow);
}

// a few lists (one signature entering the working memory): the triples travel as kernel arguments, no staging copy
constexpr int LINK_SMALL = 256;
struct LinkArgs { int32_t t[3 * LINK_SMALL]; };
__global__ void bayes_link_small_kernel(LinkArgs a, int n, uint32_t* __restrict__ nbr, int32_t* __restrict__ cnt, long long cap, int K,
                                        unsigned long long* __restrict__ overflow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t x = a.t[3 * i], y = a.t[3 * i + 1];
    const uint32_t m = (uint32_t)a.t[3 * i + 2];
    list_insert(nbr, cnt, cap, K, x, y, m, overflow);
    if (x != y) list_insert(nbr, cnt, cap, K, y, x, m, overflow);
}

}  // namespace

#define BY_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return e__; } while (0)

void Bayes::destroy() {
    for (DevBuf* b : {&nbr, &cnt, &post, &was_in, &col, &tmp, &partial, &scal, &pairs, &overflow}) b->release(bytes);
    cap = 0;
}

hipError_t Bayes::configure(const double* lc, int n, float vp_prior) {
    // BayesFilter::setPredictionLC (:77-122) on values that are already parsed, and parseParameters (:64-74)
    if (n < 2 || n > BAYES_MAX_LC || vp_prior < 0.0f || vp_prior > 1.0f) return hipErrorInvalidValue;
    for (int i = 0; i < n; ++i) if (!(lc[i] >= 0.0 && lc[i] <= 1.0)) return hipErrorInvalidValue;
    BayesParams p{};
    p.n_lc = n;
    float total = 0.0f, eps = 0.0f;
    for (int j = 0; j < n; ++j) {
        p.lc[j] = (float)lc[j];
        total += lc[j];                                     // float += double, as the reference accumulates
        if (j == 0 || lc[j] < eps) eps = lc[j];
    }
    p.total = total;
    p.lc0 = lc[0];
    p.eps = eps;
    p.vp_prior = vp_prior;
    p.max_norm = 1 - lc[0];
    p.all_other = total < 1 ? 1.0f - total : 0.0f;
    prm = p;
    configured = true;
    return hipSuccess;
}

hipError_t Bayes::ensure(int64_t n_slots, int k_needed) {
    int nk = K;
    while (nk < k_needed) nk *= 2;
    if (nk > BAYES_MAX_K) return hipErrorInvalidValue;
    if (n_slots <= cap && cap > 0 && nk == K) return hipSuccess;
    int64_t ncap = cap ? cap : 4096;
    while (ncap < n_slots) ncap *= 2;
    // neighbour table: k-major, so growing the slot dimension is a pitched copy and growing K appends rows
    DevBuf nn;
    BY_TRY(nn.reserve((size_t)nk * ncap * 4, 0, stream, bytes));
    BY_TRY(hipMemsetAsync(nn.p, 0xFF, (size_t)nk * ncap * 4, stream));
    if (cap > 0) BY_TRY(hipMemcpy2DAsync(nn.p, (size_t)ncap * 4, nbr.p, (size_t)cap * 4, (size_t)cap * 4, (size_t)K, hipMemcpyDeviceToDevice, stream));
    if (nbr.p) { BY_TRY(hipStreamSynchronize(stream)); nbr.release(bytes); }
    nbr = nn;
    K = nk;
    if (ncap == cap) return hipSuccess;
    auto grow = [&](DevBuf& b, size_t old_bytes, size_t new_bytes) -> hipError_t {
        const size_t had = b.cap;
        BY_TRY(b.reserve(new_bytes, old_bytes, stream, bytes));
        if (b.cap > had) BY_TRY(hipMemsetAsync((char*)b.p + (had < old_bytes ? had : old_bytes), 0, b.cap - (had < old_bytes ? had : old_bytes), stream));
        return hipSuccess;
    };
    BY_TRY(grow(cnt, (size_t)cap * 4, (size_t)ncap * 4));
    BY_TRY(grow(post, cap ? (size_t)(cap + 1) * 4 : 0, (size_t)(ncap + 1) * 4));
    BY_TRY(grow(was_in, (size_t)cap, (size_t)ncap));
    BY_TRY(grow(col, 0, (size_t)ncap * sizeof(ColS)));
    BY_TRY(grow(tmp, 0, (size_t)(ncap + 1) * 4));
    BY_TRY(grow(partial, 0, (size_t)BAYES_GRID * (sizeof(Part1) + sizeof(double) + 16)));
    BY_TRY(grow(overflow, 0, 256));
    cap = ncap;
    cnt_ub.resize((size_t)cap, 0);
    return hipSuccess;
}

hipError_t Bayes::reset() {
    empty = true;                                            // BayesFilter::reset (:138-143); the neighbour lists (_neighborsIndex) go too
    if (cap > 0) {
        BY_TRY(hipMemsetAsync(was_in.p, 0, (size_t)cap, stream));
        BY_TRY(hipMemsetAsync(cnt.p, 0, (size_t)cap * 4, stream));
        BY_TRY(hipMemsetAsync(nbr.p, 0xFF, (size_t)K * cap * 4, stream));
    }
    std::fill(cnt_ub.begin(), cnt_ub.end(), 0);
    return hipSuccess;
}

hipError_t Bayes::link(const std::vector<int32_t>& triples) {
    const int n = (int)(triples.size() / 3);
    if (n == 0) return hipSuccess;
    // room for every entry this call may add (an entry that replaces an existing one is counted again: the bound only grows)
    int64_t top = 0;
    int k_needed = 0;
    for (int i = 0; i < n; ++i) top = std::max<int64_t>(top, std::max(triples[3 * i], triples[3 * i + 1]));
    BY_TRY(ensure(std::max<int64_t>(cap, top + 1)));
    for (int i = 0; i < n; ++i) {
        const int32_t a = triples[3 * i], b = triples[3 * i + 1];
        k_needed = std::max(k_needed, ++cnt_ub[a]);
        if (a != b) k_needed = std::max(k_needed, ++cnt_ub[b]);
    }
    if (k_needed > K) BY_TRY(ensure(cap, k_needed));
    if (n <= LINK_SMALL) {
        LinkArgs a;
        memcpy(a.t, triples.data(), triples.size() * 4);
        bayes_link_small_kernel<<<(n + 63) / 64, 64, 0, stream>>>(a, n, nbr.as<uint32_t>(), cnt.as<int32_t>(), (long long)cap, K, overflow.as<unsigned long long>());
        return hipGetLastError();
    }
    BY_TRY(pairs.reserve(triples.size() * 4, 0, stream, bytes));
    // pageable source: the copy is staged by the runtime before the call returns
    BY_TRY(hipMemcpyAsync(pairs.p, triples.data(), triples.size() * 4, hipMemcpyHostToDevice, stream));
    BY_TRY(hipStreamSynchronize(stream));
    bayes_link_kernel<<<(n + 255) / 256, 256, 0, stream>>>(pairs.as<int32_t>(), n, nbr.as<uint32_t>(), cnt.as<int32_t>(), (long long)cap, K,
                                                          overflow.as<unsigned long long>());
    return hipGetLastError();
}

hipError_t Bayes::update(const float* d_adjusted, const int32_t* slot_sig, int64_t n_slots, int64_t n_cons, float* d_posterior, BayesOut* d_out) {
    if (!configured) return hipErrorNotReady;
    if (n_cons < 0) n_cons = 0;
    if (n_cons > n_slots) n_cons = n_slots;
    BY_TRY(ensure(std::max<int64_t>(n_slots, 1)));
    Part1* part = partial.as<Part1>();
    double* part2 = (double*)(part + BAYES_GRID);
    unsigned long long* part3 = (unsigned long long*)(part2 + BAYES_GRID);
    const int cols_known = prm.all_other > 0.0f ? 1 : 0;
    if (cols_known) bayes_count_kernel<<<BAYES_GRID, BY_BLOCK, 0, stream>>>((long long)n_cons, slot_sig, part);
    bayes_column_kernel<<<BAYES_GRID, BY_BLOCK, 0, stream>>>(prm, (long long)n_slots, (long long)n_cons, slot_sig, nbr.as<uint32_t>(), cnt.as<int32_t>(),
                                                           (long long)cap, K, was_in.as<uint8_t>(), post.as<float>(), empty ? 1 : 0, cols_known,
                                                           col.as<ColS>(), tmp.as<float>(), part);
    bayes_row_kernel<<<BAYES_GRID, BY_BLOCK, 0, stream>>>(prm, (long long)n_slots, (long long)n_cons, slot_sig, nbr.as<uint32_t>(), cnt.as<int32_t>(),
                                                        (long long)cap, K, col.as<ColS>(), tmp.as<float>(), part, d_adjusted, post.as<float>(), part2);
    bayes_normalize_kernel<<<BAYES_GRID, BY_BLOCK, 0, stream>>>((long long)n_slots, (long long)n_cons, slot_sig, part2, post.as<float>(), was_in.as<uint8_t>(),
                                                              d_posterior, part3);
    bayes_result_kernel<<<1, BY_BLOCK, 0, stream>>>(part2, part, part3, post.as<float>(), d_posterior, d_out);
    BY_TRY(hipGetLastError());
    empty = false;
    return hipSuccess;
}

hipError_t Bayes::read_overflow(int64_t* out) {
    *out = 0;
    if (!overflow.p) return hipSuccess;
    unsigned long long v = 0;
    BY_TRY(hipMemcpyAsync(&v, overflow.p, 8, hipMemcpyDeviceToHost, stream));
    BY_TRY(hipStreamSynchronize(stream));
    *out = (int64_t)v;
    return hipSuccess;
}

}  // namespace lcd
