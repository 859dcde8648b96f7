// bayes.hip -- the decision stage behind a frame's likelihood, on the device: Rtabmap::adjustLikelihood (Rtabmap.cpp:5691-5760),
// BayesFilter::computePosterior (BayesFilter.cpp:145-235) and the selection of the highest hypothesis (Rtabmap.cpp:2147-2158).
// See bayes.h for the layout.  Two passes over the signature slots, each followed by a one-workgroup fold of the per-workgroup
// partials in a fixed order -- an update is bit-reproducible:
//   pass 1  likelihood statistics (sum, sum of squares, count, best raw likelihood)            [adjustLikelihood's uMean / uVariance]
//           + per column of the prediction matrix: what addNeighborProb / normalize derive from its neighbour list   [Bayes]
//   pass 2  adjusted likelihood per slot; prior = prediction x posterior as a gather over the slot's own (symmetric) neighbour
//           list; posterior = likelihood x prior, not yet normalised                                                    [Bayes]
//           + its sum and the best hypothesis; the posterior stays unnormalised in HBM, readers divide by the sum        [Bayes]
//   (pass 3, only when the caller wants the posterior as a vector: the division, written out)
//
// The arithmetic follows the reference's statements in their types (float matrix elements, the double comparisons its mixed
// float/double expressions promote to).  Evaluated differently, inside the float rounding the reference itself leaves open:
//  (1) prior = prediction * posterior is cv::gemm in the reference (OpenCV, not in its tree); here every row is a sum in double
//      over the row's non-zeros, rounded to float once;
//  (2) the reference adds the posterior's entries into a float one by one (:205-218) before dividing; here the sum is taken in
//      double;  (3) uMean / uVariance add floats one by one (UMath.h:419-432, 512-526); here sum and sum of squares are taken
//      in double in one pass, the variance as (S2 - 2 m S1 + n m^2) / (n - 1) with m the float mean the reference subtracts.
#include "bayes.h"

#include <algorithm>
#include <unordered_map>
#include <cstring>

namespace lcd {
namespace {

constexpr int DC_BLOCK = 256;
// Measured at 100 000 slots (round 5, same box, with_bayes ms per step; profiles/r05_ab_notes.txt): 256 / 512 / 1024 / 2048 / 4096 workgroups per pass
// -> 0.107 / 0.083 / 0.076 / 0.091 / 0.130.  Also measured and NOT kept: every list row of three steps requested up front, then every gather
// (two round trips per workgroup instead of one per step): 114 / 128-capped registers instead of 86 / 100 -> pass 1 15.7 -> 17.7 us, pass 2 20.5 ->
// 59.5 us -- the passes live on resident waves, not on the length of one workgroup's chain.
constexpr int DC_MAX_GRID = 1024;
constexpr uint32_t SLOT_MASK = (1u << BAYES_SLOT_BITS) - 1u;
constexpr int ROWS = 6;                   // list rows (8 entries each) requested up front by the passes: 48 entries cover a chain neighbourhood (33)
constexpr int TILE = 8;                   // slots per tile of the neighbour table: nbr[(slot / 8) * K + k][slot % 8]

// per column: scale < 0 marks a renormalised column (|scale| = maxNorm / sum); scale == 0: the slot does not take part
struct ColS { float scale; float delta; float fill; float pin; };

struct Part1 { double s1, s2, s_in, s_fill; unsigned long long key; long long cnt, n_in; long long pad; };
struct Part2 { double usum; unsigned long long key, slot; long long pad; };
struct Scal {
    // pass 1
    double s_in, s_fill;
    long long n_in, cnt_pos;
    float mean, stddev, vp_adj, maxv;
    unsigned long long best_key;
    // pass 2 (kept until the next update: the posterior stays unnormalised in post[], divided by `sum` when it is read)
    float sum, p0, u0, pad0;
    unsigned int ticket1, ticket2, ticket3, pad1;
};

__device__ __forceinline__ size_t tile_at(long long c, int k, int K) { return ((size_t)(c >> 3) * K + k) * TILE + (size_t)(c & 7); }
// slot_sig == NULL: every slot below n_cons takes part (a plain likelihood vector, lcd_adjust_likelihood)
__device__ __forceinline__ bool in_set(long long s, long long n_cons, const int32_t* __restrict__ slot_sig) { return s < n_cons && (slot_sig == nullptr || slot_sig[s] != 0); }

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __shfl_xor((unsigned)u, m, 64), hi = __shfl_xor((unsigned)(u >> 32), m, 64);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long u, int m) {
    const unsigned lo = __shfl_xor((unsigned)u, m, 64), hi = __shfl_xor((unsigned)(u >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}
// One workgroup reduction for everything a pass accumulates: four double sums, two counts, one (key, payload) maximum.  Wave
// butterfly, one LDS slot per wave, ONE barrier; thread 0 then adds the waves in order (fixed order: bit-reproducible).  The
// result is valid in thread 0 only.
struct Red { double d0, d1, d2, d3; long long c0, c1; unsigned long long key, pay; };
template <int NW>
__device__ __forceinline__ Red block_reduce(Red r, Red* s_r) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        r.d0 += shfl_xor_d(r.d0, m); r.d1 += shfl_xor_d(r.d1, m); r.d2 += shfl_xor_d(r.d2, m); r.d3 += shfl_xor_d(r.d3, m);
        r.c0 += (long long)shfl_xor_u64((unsigned long long)r.c0, m); r.c1 += (long long)shfl_xor_u64((unsigned long long)r.c1, m);
        const unsigned long long ok = shfl_xor_u64(r.key, m), op = shfl_xor_u64(r.pay, m);
        if (ok > r.key) { r.key = ok; r.pay = op; }
    }
    if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const Red o = s_r[w];
            r.d0 += o.d0; r.d1 += o.d1; r.d2 += o.d2; r.d3 += o.d3; r.c0 += o.c0; r.c1 += o.c1;
            if (o.key > r.key) { r.key = o.key; r.pay = o.pay; }
        }
    }
    return r;
}

__device__ __forceinline__ float adjusted_value(float value, float mean, float stdDev, float ratio) {   // Rtabmap.cpp:5722-5745
    float o = 1.0f;
    if (value > mean + stdDev) {
        if (ratio == 0.0f && mean != 0.0f) o = (value - (stdDev - 0.0001f)) / mean;
        else if (ratio != 0.0f && stdDev != 0.0f) o = (value - mean) / stdDev;
    }
    return o;
}

struct Pass1Args {
    BayesParams prm;
    long long n_slots, n_cons;
    const int32_t* slot_sig;
    const float* like;             // raw likelihood per slot, or NULL (the caller supplies the adjusted vector)
    float ratio;
    HypothesisOut* hyp;            // may be NULL
    // Bayes
    const uint32_t* nbr; const int32_t* cnt; int K;
    const uint8_t* was_in; const float* post; int empty;
    long long cols;                // < 0: counted ahead of the pass into scal->cnt_pos (the prediction fills "all other places")
    ColS* col;
    Part1* part; Scal* scal;
};

// number of signatures taking part, ahead of pass 1: only the fill variant needs it there
__global__ __launch_bounds__(DC_BLOCK) void decide_count_kernel(long long n_cons, const int32_t* __restrict__ slot_sig, long long* __restrict__ out) {
    __shared__ Red s_r[4];
    Red r = {0.0, 0.0, 0.0, 0.0, 0, 0, 0ull, 0ull};
    for (long long c = (long long)blockIdx.x * DC_BLOCK + threadIdx.x; c < n_cons; c += (long long)gridDim.x * DC_BLOCK) r.c0 += slot_sig[c] != 0;
    r = block_reduce<4>(r, s_r);
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)out, (unsigned long long)r.c0);     // integer: order-free
}

// BAYES: 8 lanes per slot walk its neighbour list (a wave covers a tile of 8 slots x 8 entries per step: 256 B coalesced);
// otherwise one lane per slot for the likelihood statistics alone
template <bool BAYES>
__global__ __launch_bounds__(DC_BLOCK) void decide_pass1_kernel(Pass1Args a) {
    constexpr int LPS = BAYES ? 8 : 1;
    constexpr int SPB = DC_BLOCK / LPS;                     // slots per workgroup step
    __shared__ Red s_r[4];
    __shared__ float s_lc[BAYES_MAX_LC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (BAYES) { if (tid < BAYES_MAX_LC) s_lc[tid] = a.prm.lc[tid]; __syncthreads(); }
    const int slot_in_wave = BAYES ? (lane & 7) : lane, k_sub = BAYES ? (lane >> 3) : 0;
    long long cols = a.cols;
    if (BAYES && cols < 0) cols = a.scal->cnt_pos + 1;      // left there by decide_count_kernel
    const float sum_prev = BAYES ? a.scal->sum : 0.0f;      // normalisation constant of the last update
    if (BAYES && blockIdx.x == 0 && tid == 0) {             // the virtual place's last posterior rides behind the last column
        ColS v = {0.0f, 0.0f, 0.0f, a.empty ? 1.0f : a.scal->p0};
        a.col[a.n_slots] = v;
    }
    Red acc = {0.0, 0.0, 0.0, 0.0, 0, 0, 0ull, 0ull};       // d0 = sum v, d1 = sum v^2, d2 = sum of the last posterior, d3 = its fill share; c0 = positives, c1 = taking part
    const long long stride = (long long)gridDim.x * SPB;
    uint32_t en[ROWS];                                      // the next step's list rows, requested one step ahead
#pragma unroll
    for (int j = 0; j < ROWS; ++j) en[j] = 0xFFFFFFFFu;
    if (BAYES) {
        const long long c0 = (long long)blockIdx.x * SPB + wave * (64 / LPS) + slot_in_wave;
        if (c0 < a.n_slots) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) en[j] = a.nbr[tile_at(c0, k_sub + 8 * j, a.K)];    // an unused entry reads 0xFFFFFFFF: no count needed
        }
    }
    for (long long base = (long long)blockIdx.x * SPB; base < a.n_slots; base += stride) {
        const long long c = base + wave * (64 / LPS) + slot_in_wave;
        const bool valid = c < a.n_slots;
        // everything that does not depend on another load is requested first
        uint32_t e[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { e[j] = en[j]; en[j] = 0xFFFFFFFFu; }
        if (BAYES && c + stride < a.n_slots) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) en[j] = a.nbr[tile_at(c + stride, k_sub + 8 * j, a.K)];
        }
        const bool in = valid && in_set(c, a.n_cons, a.slot_sig);
        float lv = 0.0f, pold = 0.0f;
        uint8_t wi = 0;
        if (k_sub == 0 && valid) {
            if (a.like) lv = a.like[c];
            if (BAYES && !a.empty) { wi = a.was_in[c]; pold = a.post[1 + c]; }
        }
        if (a.like && in && k_sub == 0 && lv > 0.0f) {
            acc.d0 += (double)lv; acc.d1 += (double)lv * (double)lv; ++acc.c0;
            const unsigned long long k = ((unsigned long long)__float_as_uint(lv) << 32) | (unsigned long long)(uint32_t)(c + 1);
            if (k > acc.key) acc.key = k;
        }
        if (BAYES) {
            float sum = 0.0f, self_v = 0.0f;
            int nz = 0;
            int k0 = 0;
            while (true) {
                int32_t sg[ROWS];
#pragma unroll
                for (int j = 0; j < ROWS; ++j) { const long long r = e[j] & SLOT_MASK; sg[j] = (in && e[j] != 0xFFFFFFFFu && r < a.n_cons) ? a.slot_sig[r] : 0; }
#pragma unroll
                for (int j = 0; j < ROWS; ++j) {
                    if (sg[j] == 0) continue;
                    const float v = s_lc[(e[j] >> BAYES_SLOT_BITS) + 1];
                    sum += v;
                    if ((long long)(e[j] & SLOT_MASK) == c) self_v = v;
                    else if (v != 0.0f) ++nz;
                }
                k0 += 8 * ROWS;
                if (k0 >= a.K || !__any(e[ROWS - 1] != 0xFFFFFFFFu)) break;                       // lists are filled front to back: an empty last row ends them
#pragma unroll
                for (int j = 0; j < ROWS; ++j) { const int k = k0 + k_sub + 8 * j; e[j] = (valid && k < a.K) ? a.nbr[tile_at(c, k, a.K)] : 0xFFFFFFFFu; }
            }
#pragma unroll
            for (int m = 8; m <= 32; m <<= 1) {                        // the 8 lanes of a slot: fixed order
                sum += __shfl_xor(sum, m, 64);
                self_v = fmaxf(self_v, __shfl_xor(self_v, m, 64));
                nz += __shfl_xor(nz, m, 64);
            }
            if (k_sub == 0 && valid) {
                ColS cs = {0.0f, 0.0f, 0.0f, 0.0f};
                if (in) {
                    float p = 1.0f;                                                            // updatePosterior :709-736
                    if (!a.empty) { p = wi ? pold : 0.0f; if (sum_prev != 0.0f) p = p / sum_prev; }   // the division of :221-230, made on reading
                    cs.scale = 1.0f;
                    if ((double)sum < (double)a.prm.total - a.prm.lc0) {                       // neighbours not found go to the loop closure itself (:440-445)
                        cs.delta = (float)((double)a.prm.total - a.prm.lc0 - (double)sum);
                        sum += cs.delta;
                    }
                    if (self_v + cs.delta != 0.0f) ++nz;                                       // the diagonal element
                    if (a.prm.all_other > 0.0f && cols > 1) {                                  // every element still 0 gets a small value (:455-465)
                        const float value = a.prm.all_other / (float)(cols - 1);
                        // the reference adds `value` once per zero element into the float; one rounded product here (non-default PredictionLC only)
                        sum = (float)((double)sum + (double)value * (double)((cols - 1) - nz));
                        cs.fill = value;
                    }
                    if ((double)sum < (double)a.prm.max_norm - 0.0001 || (double)sum > (double)a.prm.max_norm + 0.0001) {   // :467-477
                        const float sc = a.prm.max_norm / sum;
                        cs.scale = -sc;
                        cs.fill = cs.fill * sc;
                        if (cs.fill < a.prm.eps) cs.fill = 0.0f;
                    }
                    cs.pin = p;
                    acc.d2 += (double)p;
                    acc.d3 += (double)cs.fill * (double)p;
                    ++acc.c1;
                }
                a.col[c] = cs;
            }
        }
    }
    acc.pay = acc.key;
    acc = block_reduce<4>(acc, s_r);
    if (tid == 0) { Part1 p = {acc.d0, acc.d1, acc.d2, acc.d3, acc.key, acc.c0, acc.c1, 0}; a.part[blockIdx.x] = p; }
}

// fold of pass 1: thread t takes partials t, t + NT, ... in order, then the fixed tree; thread 0 derives adjustLikelihood's
// statistics.  Either its own one-workgroup launch or the prologue of EVERY pass-2 workgroup (same inputs, same order: same bits)
// -- a release/acquire hand-off inside pass 1 would cost each workgroup an L2 write-back on this multi-die part.
struct Fold1 { double s_in, s_fill; long long n_in, cnt_pos; float mean, stddev, vp_adj, maxv; unsigned long long best_key; };
template <int NT>
__device__ __forceinline__ void fold1(const Part1* __restrict__ part, int n_part, bool have_like, float ratio, Red* s_r, Fold1* s_out) {
    constexpr int PER = DC_MAX_GRID / NT;
    const int tid = threadIdx.x;
    Red r = {0.0, 0.0, 0.0, 0.0, 0, 0, 0ull, 0ull};
    if (n_part <= NT) {                                          // the usual case: one partial per thread
        if (tid < n_part) { const Part1 p = part[tid]; r.d0 = p.s1; r.d1 = p.s2; r.d2 = p.s_in; r.d3 = p.s_fill; r.c0 = p.cnt; r.c1 = p.n_in; r.key = p.key; }
    } else {
        Part1 p[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {                          // all loads in flight together
            const int b = tid + j * NT;
            if (b < n_part) p[j] = part[b];
            else { p[j].s1 = p[j].s2 = p[j].s_in = p[j].s_fill = 0.0; p[j].key = 0ull; p[j].cnt = p[j].n_in = 0; }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            r.d0 += p[j].s1; r.d1 += p[j].s2; r.d2 += p[j].s_in; r.d3 += p[j].s_fill; r.c0 += p[j].cnt; r.c1 += p[j].n_in;
            if (p[j].key > r.key) r.key = p[j].key;
        }
    }
    r.pay = r.key;
    r = block_reduce<NT / 64>(r, s_r);
    if (tid == 0) {
        Fold1 f;
        f.s_in = r.d2; f.s_fill = r.d3; f.n_in = r.c1; f.cnt_pos = r.c0; f.best_key = r.key;
        const double S1 = r.d0, S2 = r.d1;
        const long long CP = r.c0;
        float mean = 0.0f, stdDev = 0.0f, vp = 2.0f;
        const float maxv = __uint_as_float((uint32_t)(r.key >> 32));
        if (have_like) {
            mean = CP ? (float)(S1 / (double)CP) : 0.0f;                                       // uMean
            double var = 0.0;
            if (CP > 1) var = (S2 - 2.0 * (double)mean * S1 + (double)CP * (double)mean * (double)mean) / (double)(CP - 1);   // uVariance around the float mean
            stdDev = sqrtf((float)fmax(var, 0.0));
            if (ratio == 0.0f && stdDev > 0.0001f && maxv != 0.0f) vp = mean / stdDev + 1.0f;     // Rtabmap.cpp:5747-5758
            else if (ratio != 0.0f && maxv > mean) vp = stdDev / (maxv - mean) + 1.0f;
        }
        f.mean = mean; f.stddev = stdDev; f.vp_adj = vp; f.maxv = maxv;
        *s_out = f;
    }
    __syncthreads();
}
// what the fold leaves in HBM: the scalars (pass 2's fold and the next update read them) and the likelihood's best candidate
__device__ __forceinline__ void publish_fold1(const Fold1& f, Scal* sc, HypothesisOut* hyp, bool have_like, float ratio, const int32_t* __restrict__ slot_sig) {
    sc->s_in = f.s_in; sc->s_fill = f.s_fill; sc->n_in = f.n_in; sc->cnt_pos = f.cnt_pos; sc->best_key = f.best_key;
    sc->mean = f.mean; sc->stddev = f.stddev; sc->vp_adj = f.vp_adj; sc->maxv = f.maxv;
    if (hyp && have_like) {
        HypothesisOut h;
        const long long slot = (long long)(uint32_t)f.best_key - 1;
        h.slot = (int32_t)slot;
        h.sig_id = slot >= 0 ? slot_sig[slot] : 0;
        h.likelihood = slot >= 0 ? f.maxv : 0.0f;
        h.adjusted = slot >= 0 ? adjusted_value(f.maxv, f.mean, f.stddev, ratio) : 0.0f;
        h.virtual_place = f.vp_adj; h.mean = f.mean; h.stddev = f.stddev; h.n_positive = (int32_t)f.cnt_pos;
        *hyp = h;
    }
}
constexpr int DC_FOLD = 1024;
__global__ __launch_bounds__(DC_FOLD) void decide_fold1_kernel(Pass1Args a, int n_part) {
    __shared__ Red s_r[DC_FOLD / 64];
    __shared__ Fold1 s_f;
    fold1<DC_FOLD>(a.part, n_part, a.like != nullptr, a.ratio, s_r, &s_f);
    if (threadIdx.x == 0) publish_fold1(s_f, a.scal, a.hyp, a.like != nullptr, a.ratio, a.slot_sig);
}

struct Pass2Args {
    BayesParams prm;
    long long n_slots, n_cons;
    const int32_t* slot_sig;
    const float* like; float ratio;   // raw likelihood (adjusted here), or NULL: adj_in holds the adjusted vector
    const float* adj_in;
    float* adj_out;                   // may be NULL
    const uint32_t* nbr; const int32_t* cnt; int K;
    const ColS* col;
    float* post;                      // [1 + slot]: unnormalised posterior out
    uint8_t* was_in;
    Part2* part2; Scal* scal;
    BayesOut* out;                    // may be NULL
    const Part1* part1; int n_part1;  // pass 1's partials: every workgroup folds them itself
    HypothesisOut* hyp;               // may be NULL
};

// one element of the prediction matrix as normalize() leaves it: v = the value addNeighborProb stored (+ delta on the diagonal)
__device__ __forceinline__ float finish_element(float v, float scale, float eps) {
    if (scale < 0.0f) { v = v * -scale; if (v < eps) v = 0.0f; }
    return v;
}

// pass 2: adjusted likelihood; prior row by row; STEP 2; the sum that normalises (:221-230) and the highest hypothesis
// (Rtabmap.cpp:2147-2158: ids > 0, highest posterior, the higher id on equal values; its value is 1 - the virtual place's
// posterior).  The posterior stays unnormalised in post[]: whoever reads it divides by scal->sum -- the same float division.
// The hypothesis is picked on the unnormalised values (two that differ by an ulp could round to the same quotient).
template <bool BAYES>
__global__ __launch_bounds__(DC_BLOCK) void decide_pass2_kernel(Pass2Args a) {
    constexpr int LPS = BAYES ? 8 : 1;
    constexpr int SPB = DC_BLOCK / LPS;
    __shared__ Red s_r[4];
    __shared__ float s_lc[BAYES_MAX_LC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (BAYES) { if (tid < BAYES_MAX_LC) s_lc[tid] = a.prm.lc[tid]; __syncthreads(); }
    const int slot_in_wave = BAYES ? (lane & 7) : lane, k_sub = BAYES ? (lane >> 3) : 0;
    const long long stride = (long long)gridDim.x * SPB;
    uint32_t en[ROWS];                                                 // the first step's list rows: in flight while pass 1 is folded
#pragma unroll
    for (int j = 0; j < ROWS; ++j) en[j] = 0xFFFFFFFFu;
    if (BAYES) {
        const long long c0 = (long long)blockIdx.x * SPB + wave * (64 / LPS) + slot_in_wave;
        if (c0 < a.n_slots) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) en[j] = a.nbr[tile_at(c0, k_sub + 8 * j, a.K)];    // an unused entry reads 0xFFFFFFFF: no count needed
        }
    }
    const float pin_vp = BAYES ? a.col[a.n_slots].pin : 0.0f;          // the virtual place's last posterior rides behind the last column
    __shared__ Fold1 s_f;
    fold1<DC_BLOCK>(a.part1, a.n_part1, a.like != nullptr, a.ratio, s_r, &s_f);
    const Fold1 sc = s_f;
    if (blockIdx.x == 0 && tid == 0) publish_fold1(sc, a.scal, a.hyp, a.like != nullptr, a.ratio, a.slot_sig);
    const long long cols = sc.n_in + 1;
    float vp_col = 0.0f;                                               // the virtual place's column in the rows >= 1 (:376-411; row 0: fold 2)
    if (a.prm.vp_prior > 0.0f) {
        if (cols > 1) vp_col = (float)((1.0 - a.prm.vp_prior) / (double)(cols - 1));
    } else if (cols > 1) vp_col = (float)(1.0 / (double)cols);
    const double from_vp = (double)vp_col * (double)pin_vp;
    double usum = 0.0;
    unsigned long long key = 0ull, kslot = ~0ull;
    for (long long base = (long long)blockIdx.x * SPB; base < a.n_slots; base += stride) {
        const long long i = base + wave * (64 / LPS) + slot_in_wave;
        const bool valid = i < a.n_slots;
        uint32_t e[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { e[j] = en[j]; en[j] = 0xFFFFFFFFu; }
        if (BAYES && i + stride < a.n_slots) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) en[j] = a.nbr[tile_at(i + stride, k_sub + 8 * j, a.K)];
        }
        const bool in = valid && in_set(i, a.n_cons, a.slot_sig);
        float o = 0.0f;
        if (k_sub == 0 && valid) {
            if (a.like) o = in ? adjusted_value(a.like[i], sc.mean, sc.stddev, a.ratio) : 0.0f;
            else o = in ? a.adj_in[1 + i] : 0.0f;
            if (a.adj_out) a.adj_out[1 + i] = o;
        }
        if (BAYES) {
            double acc = 0.0;
            int has_self = 0;
            int k0 = 0;
            while (true) {
                ColS cs[ROWS];
#pragma unroll
                for (int j = 0; j < ROWS; ++j) {
                    if (in && e[j] != 0xFFFFFFFFu) cs[j] = a.col[e[j] & SLOT_MASK];
                    else { cs[j].scale = 0.0f; cs[j].delta = 0.0f; cs[j].fill = 0.0f; cs[j].pin = 0.0f; }
                }
#pragma unroll
                for (int j = 0; j < ROWS; ++j) {
                    if (cs[j].scale == 0.0f) continue;                                 // the column's signature does not take part
                    float v = s_lc[(e[j] >> BAYES_SLOT_BITS) + 1];
                    if ((long long)(e[j] & SLOT_MASK) == i) { v = v + cs[j].delta; has_self = 1; }
                    if (v == 0.0f) continue;                                           // an element left at 0: it holds the column's fill value
                    v = finish_element(v, cs[j].scale, a.prm.eps);
                    acc += ((double)v - (double)cs[j].fill) * (double)cs[j].pin;
                }
                k0 += 8 * ROWS;
                if (k0 >= a.K || !__any(e[ROWS - 1] != 0xFFFFFFFFu)) break;                   // lists are filled front to back: an empty last row ends them
#pragma unroll
                for (int j = 0; j < ROWS; ++j) { const int k = k0 + k_sub + 8 * j; e[j] = (valid && k < a.K) ? a.nbr[tile_at(i, k, a.K)] : 0xFFFFFFFFu; }
            }
#pragma unroll
            for (int m = 8; m <= 32; m <<= 1) { acc += shfl_xor_d(acc, m); has_self |= __shfl_xor(has_self, m, 64); }
            if (k_sub == 0 && valid) {
                float u = 0.0f;
                if (in) {
                    if (!has_self) {                                                   // diagonal of a column whose list does not hold itself: 0 + delta
                        const ColS c0 = a.col[i];
                        if (c0.delta != 0.0f) acc += ((double)finish_element(c0.delta, c0.scale, a.prm.eps) - (double)c0.fill) * (double)c0.pin;
                    }
                    const float prior = (float)(acc + sc.s_fill + from_vp);
                    u = o * prior;                                                     // STEP 2 (:205-218)
                    usum += (double)u;
                    if (u > 0.0f) {
                        const unsigned long long k = ((unsigned long long)__float_as_uint(u) << 32) | (unsigned long long)(uint32_t)a.slot_sig[i];
                        if (k > key) { key = k; kslot = (unsigned long long)i; }
                    }
                }
                a.post[1 + i] = u;
                a.was_in[i] = in ? 1 : 0;
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0 && a.adj_out && a.like) a.adj_out[0] = sc.vp_adj;
    if (!BAYES) return;
    Red r = {usum, 0.0, 0.0, 0.0, 0, 0, key, kslot};
    r = block_reduce<4>(r, s_r);
    if (tid == 0) { Part2 p = {r.d0, r.key, r.pay, 0}; a.part2[blockIdx.x] = p; }
}

// fold of pass 2: the sum that normalises, the virtual place's posterior, the highest hypothesis
__global__ __launch_bounds__(DC_FOLD) void decide_fold2_kernel(Pass2Args a, int n_part) {
    __shared__ Red s_r[DC_FOLD / 64];
    const int tid = threadIdx.x;
    Part2 p[DC_MAX_GRID / DC_FOLD];
#pragma unroll
    for (int j = 0; j < DC_MAX_GRID / DC_FOLD; ++j) {
        const int b = tid + j * DC_FOLD;
        if (b < n_part) p[j] = a.part2[b];
        else { p[j].usum = 0.0; p[j].key = 0ull; p[j].slot = ~0ull; }
    }
    Red r = {0.0, 0.0, 0.0, 0.0, 0, 0, 0ull, ~0ull};
#pragma unroll
    for (int j = 0; j < DC_MAX_GRID / DC_FOLD; ++j) { r.d0 += p[j].usum; if (p[j].key > r.key) { r.key = p[j].key; r.pay = p[j].slot; } }
    r = block_reduce<DC_FOLD / 64>(r, s_r);
    const double T = r.d0;
    const unsigned long long key = r.key, kslot = r.pay;
    if (tid == 0) {
        const Scal sc = *a.scal;
        const long long cols = sc.n_in + 1;
        const float pin_vp = a.col[a.n_slots].pin;
        float p00 = 1.0f;
        if (a.prm.vp_prior > 0.0f) { if (cols > 1) p00 = a.prm.vp_prior; }
        else if (cols > 1) p00 = (float)(1.0 / (double)cols);
        // row 0: the virtual place's own value + Bayes/PredictionLC[0] from every other column (:486-490)
        const float like0 = a.like ? sc.vp_adj : a.adj_in[0];
        const float prior0 = (float)((double)p00 * (double)pin_vp + (double)(float)a.prm.lc0 * sc.s_in);
        const float u0 = like0 * prior0;
        const float sum = (float)(T + (double)u0);
        const float p0 = sum != 0.0f ? u0 / sum : u0;
        a.scal->sum = sum; a.scal->u0 = u0; a.scal->p0 = p0;
        if (a.out) {
            BayesOut o;
            const float ub = __uint_as_float((uint32_t)(key >> 32));
            o.sig_id = key ? (int32_t)(uint32_t)key : 0;
            o.slot = key ? (int32_t)kslot : -1;
            o.posterior = sum != 0.0f ? ub / sum : ub;
            o.value = 1 - p0;
            o.virtual_place = p0;
            o.n_considered = (int32_t)sc.n_in;
            o.sum = sum;
            o.reserved = 0;
            *a.out = o;
        }
    }
}

// optional pass 3: the normalised posterior as a vector, for a caller that asked for it
__global__ __launch_bounds__(DC_BLOCK) void decide_posterior_kernel(long long n_slots, const float* __restrict__ post, const uint8_t* __restrict__ was_in,
                                                                    const Scal* __restrict__ scal, float* __restrict__ d_posterior) {
    const float sum = scal->sum;
    for (long long i = (long long)blockIdx.x * DC_BLOCK + threadIdx.x; i < n_slots; i += (long long)gridDim.x * DC_BLOCK) {
        float p = was_in[i] ? post[1 + i] : 0.0f;
        if (sum != 0.0f) p = p / sum;
        d_posterior[1 + i] = p;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) d_posterior[0] = scal->p0;
}

// neighbour lists: enter (b, margin) into a's list and (a, margin) into b's; an entry for the same neighbour is replaced
// (uInsert, BayesFilter.cpp:589).  One wavefront per (triple, direction): the lanes search the list side by side.  The triples
// of a call are unique, so no two wavefronts enter the same neighbour into the same list.
__device__ __forceinline__ void list_insert(uint32_t* __restrict__ nbr, int32_t* __restrict__ cnt, int K, int32_t a, int32_t b, uint32_t margin,
                                            unsigned long long* __restrict__ overflow, int lane) {
    const uint32_t e = (margin << BAYES_SLOT_BITS) | (uint32_t)b;
    const int n = min(__hip_atomic_load(&cnt[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), K);
    bool found = false;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        const bool hit = k < n && (nbr[tile_at(a, k, K)] & SLOT_MASK) == (uint32_t)b;
        if (hit) nbr[tile_at(a, k, K)] = e;
        if (__ballot(hit)) { found = true; break; }
    }
    if (found || lane != 0) return;
    const int pos = atomicAdd(&cnt[a], 1);
    if (pos < K) nbr[tile_at(a, pos, K)] = e;
    else { atomicSub(&cnt[a], 1); atomicAdd(overflow, 1ull); }
}
__global__ __launch_bounds__(256) void bayes_link_kernel(const int32_t* __restrict__ triples, int n, uint32_t* __restrict__ nbr, int32_t* __restrict__ cnt, int K,
                                                         unsigned long long* __restrict__ overflow) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (w >= 2 * n) return;
    const int i = w >> 1;
    const int32_t a = triples[3 * i], b = triples[3 * i + 1];
    const uint32_t m = (uint32_t)triples[3 * i + 2];
    if ((w & 1) == 0) list_insert(nbr, cnt, K, a, b, m, overflow, lane);
    else if (a != b) list_insert(nbr, cnt, K, b, a, m, overflow, lane);
}
// the lists of the signatures a call names start over (the reference's _neighborsIndex entry of a new id IS what getNeighborsId
// returned, BayesFilter.cpp:581-592): every entry back to "unused", one wavefront per list
__global__ __launch_bounds__(256) void bayes_clear_kernel(const int32_t* __restrict__ slots, int n, uint32_t* __restrict__ nbr, int32_t* __restrict__ cnt, int K) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (w >= n) return;
    const int32_t s = slots[w];
    for (int k = lane; k < K; k += 64) nbr[tile_at(s, k, K)] = 0xFFFFFFFFu;
    if (lane == 0) cnt[s] = 0;
}
struct ClearArgs { int32_t s[64]; };
__global__ __launch_bounds__(256) void bayes_clear_small_kernel(ClearArgs a, int n, uint32_t* __restrict__ nbr, int32_t* __restrict__ cnt, int K) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (w >= n) return;
    const int32_t s = a.s[w];
    for (int k = lane; k < K; k += 64) nbr[tile_at(s, k, K)] = 0xFFFFFFFFu;
    if (lane == 0) cnt[s] = 0;
}
// a few lists (one signature entering the working memory): the triples travel as kernel arguments, no staging copy
constexpr int LINK_SMALL = 256;
struct LinkArgs { int32_t t[3 * LINK_SMALL]; };
__global__ __launch_bounds__(256) void bayes_link_small_kernel(LinkArgs t, int n, uint32_t* __restrict__ nbr, int32_t* __restrict__ cnt, int K,
                                                               unsigned long long* __restrict__ overflow) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (w >= 2 * n) return;
    const int i = w >> 1;
    const int32_t a = t.t[3 * i], b = t.t[3 * i + 1];
    const uint32_t m = (uint32_t)t.t[3 * i + 2];
    if ((w & 1) == 0) list_insert(nbr, cnt, K, a, b, m, overflow, lane);
    else if (a != b) list_insert(nbr, cnt, K, b, a, m, overflow, lane);
}

}  // namespace

#define BY_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return e__; } while (0)

void Bayes::destroy() {
    for (DevBuf* b : {&nbr, &cnt, &post, &was_in, &col, &partial, &scal, &pairs, &overflow}) b->release(bytes);
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

hipError_t Bayes::ensure_scratch() {
    if (partial.p) return hipSuccess;
    const size_t need = (size_t)DC_MAX_GRID * (sizeof(Part1) + sizeof(Part2));
    BY_TRY(partial.reserve(need, 0, stream, bytes));
    BY_TRY(scal.reserve(256, 0, stream, bytes));
    BY_TRY(hipMemsetAsync(scal.p, 0, 256, stream));
    BY_TRY(overflow.reserve(256, 0, stream, bytes));
    BY_TRY(hipMemsetAsync(overflow.p, 0, 256, stream));
    return hipSuccess;
}

hipError_t Bayes::ensure(int64_t n_slots, int k_needed) {
    BY_TRY(ensure_scratch());
    int nk = K;
    while (nk < k_needed) nk *= 2;
    if (nk > BAYES_MAX_K) return hipErrorInvalidValue;
    if (n_slots <= cap && cap > 0 && nk == K) return hipSuccess;
    int64_t ncap = cap ? cap : 4096;
    while (ncap < n_slots) ncap *= 2;
    // neighbour table in tiles of 8 slots x K entries: more slots append tiles, a larger K widens every tile (pitched copy)
    DevBuf nn;
    BY_TRY(nn.reserve((size_t)nk * ncap * 4, 0, stream, bytes));
    BY_TRY(hipMemsetAsync(nn.p, 0xFF, (size_t)nk * ncap * 4, stream));
    if (cap > 0) BY_TRY(hipMemcpy2DAsync(nn.p, (size_t)nk * TILE * 4, nbr.p, (size_t)K * TILE * 4, (size_t)K * TILE * 4, (size_t)(cap / TILE), hipMemcpyDeviceToDevice, stream));
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
    BY_TRY(grow(col, 0, (size_t)(ncap + 1) * sizeof(ColS)));
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

hipError_t Bayes::link(const std::vector<int32_t>& triples, const std::vector<int32_t>& restart) {
    // lists that start over; one that has never held an entry needs no launch (the usual case: a signature's list is passed once)
    std::vector<int32_t> dirty;
    for (int32_t s : restart) {
        if (s < 0 || s >= (int64_t)cnt_ub.size() || cnt_ub[s] == 0) continue;
        cnt_ub[s] = 0;
        dirty.push_back(s);
    }
    if (!dirty.empty()) {
        const int nd = (int)dirty.size();
        const int blocks = (int)(((int64_t)nd * 64 + 255) / 256);
        if (nd <= 64) {
            ClearArgs a;
            memcpy(a.s, dirty.data(), dirty.size() * 4);
            bayes_clear_small_kernel<<<blocks, 256, 0, stream>>>(a, nd, nbr.as<uint32_t>(), cnt.as<int32_t>(), K);
        } else {
            BY_TRY(pairs.reserve(dirty.size() * 4, 0, stream, bytes));
            BY_TRY(hipMemcpyAsync(pairs.p, dirty.data(), dirty.size() * 4, hipMemcpyHostToDevice, stream));
            BY_TRY(hipStreamSynchronize(stream));
            bayes_clear_kernel<<<blocks, 256, 0, stream>>>(pairs.as<int32_t>(), nd, nbr.as<uint32_t>(), cnt.as<int32_t>(), K);
        }
        BY_TRY(hipGetLastError());
    }
    const int n = (int)(triples.size() / 3);
    if (n == 0) return hipSuccess;
    // room for every entry this call may add.  cnt_ub counts an entry that only replaces an existing one again, so it can run ahead
    // of the true lengths: before the table is widened -- or the call refused -- the true lengths are read back (rare, synchronises)
    int64_t top = 0;
    for (int i = 0; i < n; ++i) top = std::max<int64_t>(top, std::max(triples[3 * i], triples[3 * i + 1]));
    BY_TRY(ensure(std::max<int64_t>(cap, top + 1)));
    std::unordered_map<int32_t, int> inc;
    inc.reserve((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        const int32_t a = triples[3 * i], b = triples[3 * i + 1];
        inc[a] += 1;
        if (a != b) inc[b] += 1;
    }
    auto needed = [&]() { int k = 0; for (const auto& e : inc) k = std::max(k, cnt_ub[(size_t)e.first] + e.second); return k; };
    int k_needed = needed();
    if (k_needed > K) {
        std::vector<int32_t> true_cnt((size_t)cap);
        BY_TRY(hipMemcpyAsync(true_cnt.data(), cnt.p, (size_t)cap * 4, hipMemcpyDeviceToHost, stream));   // behind the clears above
        BY_TRY(hipStreamSynchronize(stream));
        for (const auto& e : inc) cnt_ub[(size_t)e.first] = std::min(cnt_ub[(size_t)e.first], true_cnt[(size_t)e.first]);
        k_needed = needed();
    }
    if (k_needed > K) BY_TRY(ensure(cap, k_needed));               // (nothing has been committed if this fails)
    for (const auto& e : inc) cnt_ub[(size_t)e.first] += e.second;
    const int blocks = (int)(((int64_t)2 * n * 64 + 255) / 256);
    if (n <= LINK_SMALL) {
        LinkArgs a;
        memcpy(a.t, triples.data(), triples.size() * 4);
        bayes_link_small_kernel<<<blocks, 256, 0, stream>>>(a, n, nbr.as<uint32_t>(), cnt.as<int32_t>(), K, overflow.as<unsigned long long>());
        return hipGetLastError();
    }
    BY_TRY(pairs.reserve(triples.size() * 4, 0, stream, bytes));
    // pageable source: the copy is staged by the runtime before the call returns
    BY_TRY(hipMemcpyAsync(pairs.p, triples.data(), triples.size() * 4, hipMemcpyHostToDevice, stream));
    BY_TRY(hipStreamSynchronize(stream));
    bayes_link_kernel<<<blocks, 256, 0, stream>>>(pairs.as<int32_t>(), n, nbr.as<uint32_t>(), cnt.as<int32_t>(), K, overflow.as<unsigned long long>());
    return hipGetLastError();
}

hipError_t Bayes::decide(const DecideArgs& d, const int32_t* slot_sig, int64_t n_slots, int64_t n_cons) {
    if (d.bayes && !configured) return hipErrorNotReady;
    if (n_cons < 0) n_cons = 0;
    if (n_cons > n_slots) n_cons = n_slots;
    BY_TRY(ensure_scratch());
    if (d.bayes) BY_TRY(ensure(std::max<int64_t>(n_slots, 1)));
    Part1* part = partial.as<Part1>();
    Part2* part2 = (Part2*)(part + DC_MAX_GRID);
    Scal* sc = scal.as<Scal>();
    const int spb = d.bayes ? DC_BLOCK / 8 : DC_BLOCK;
    const int grid = (int)std::min<int64_t>(DC_MAX_GRID, std::max<int64_t>(1, (n_slots + spb - 1) / spb));
    Pass1Args a1{};
    a1.prm = prm; a1.n_slots = n_slots; a1.n_cons = n_cons; a1.slot_sig = slot_sig; a1.like = d.like; a1.ratio = d.ratio; a1.hyp = d.hyp;
    a1.part = part; a1.scal = sc; a1.cols = 0;
    if (d.bayes) {
        a1.nbr = nbr.as<uint32_t>(); a1.cnt = cnt.as<int32_t>(); a1.K = K; a1.was_in = was_in.as<uint8_t>(); a1.post = post.as<float>();
        a1.empty = empty ? 1 : 0; a1.col = col.as<ColS>();
        if (prm.all_other > 0.0f) {                              // the fill value needs the number of columns ahead of the pass
            BY_TRY(hipMemsetAsync(&sc->cnt_pos, 0, 8, stream));
            decide_count_kernel<<<std::min(grid, 256), DC_BLOCK, 0, stream>>>((long long)n_cons, slot_sig, (long long*)&sc->cnt_pos);
            a1.cols = -1;
        }
        decide_pass1_kernel<true><<<grid, DC_BLOCK, 0, stream>>>(a1);
    } else {
        decide_pass1_kernel<false><<<grid, DC_BLOCK, 0, stream>>>(a1);
    }
    if (!(d.bayes || d.adj_out)) decide_fold1_kernel<<<1, DC_FOLD, 0, stream>>>(a1, grid);      // otherwise pass 2 folds in its prologue
    if (d.bayes || d.adj_out) {
        Pass2Args a2{};
        a2.prm = prm; a2.n_slots = n_slots; a2.n_cons = n_cons; a2.slot_sig = slot_sig; a2.like = d.like; a2.ratio = d.ratio;
        a2.adj_in = d.adj_in; a2.adj_out = d.adj_out; a2.part2 = part2; a2.scal = sc; a2.part1 = part; a2.n_part1 = grid; a2.hyp = d.hyp;
        if (d.bayes) {
            a2.nbr = nbr.as<uint32_t>(); a2.cnt = cnt.as<int32_t>(); a2.K = K; a2.col = col.as<ColS>(); a2.post = post.as<float>();
            a2.was_in = was_in.as<uint8_t>(); a2.out = d.d_bayes;
            decide_pass2_kernel<true><<<grid, DC_BLOCK, 0, stream>>>(a2);
            decide_fold2_kernel<<<1, DC_FOLD, 0, stream>>>(a2, grid);
        } else {
            decide_pass2_kernel<false><<<grid, DC_BLOCK, 0, stream>>>(a2);
        }
    }
    if (d.bayes) {
        if (d.d_posterior) {
            const int g3 = (int)std::min<int64_t>(DC_MAX_GRID, std::max<int64_t>(1, (n_slots + DC_BLOCK - 1) / DC_BLOCK));
            decide_posterior_kernel<<<g3, DC_BLOCK, 0, stream>>>((long long)n_slots, post.as<float>(), was_in.as<uint8_t>(), sc, d.d_posterior);
        }
        empty = false;
    }
    return hipGetLastError();
}

hipError_t Bayes::read_posterior(int64_t n, std::vector<float>* p, std::vector<uint8_t>* in) {
    p->assign((size_t)n + 1, 0.0f);
    in->assign((size_t)n + 1, 0);
    if (!post.p || empty || !scal.p) return hipSuccess;
    Scal sc;
    BY_TRY(hipMemcpyAsync(p->data(), post.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, stream));
    if (n > 0) BY_TRY(hipMemcpyAsync(in->data() + 1, was_in.p, (size_t)n, hipMemcpyDeviceToHost, stream));
    BY_TRY(hipMemcpyAsync(&sc, scal.p, sizeof(Scal), hipMemcpyDeviceToHost, stream));
    BY_TRY(hipStreamSynchronize(stream));
    (*in)[0] = 1;
    (*p)[0] = sc.p0;
    for (int64_t i = 1; i <= n; ++i) {
        float v = (*in)[(size_t)i] ? (*p)[(size_t)i] : 0.0f;
        if (sc.sum != 0.0f) v = v / sc.sum;                   // the division of BayesFilter.cpp:221-230
        (*p)[(size_t)i] = v;
    }
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
