// bayes.h -- the recursive Bayes filter over the signatures of the working memory, on the device.
//
// Replaces BayesFilter::computePosterior (reference corelib/src/BayesFilter.cpp:145-235) and the hypothesis selection that
// follows it (Rtabmap.cpp:2147-2158).  The reference multiplies a dense m x m prediction matrix (BayesFilter.cpp:313, m = working
// memory + 1) with the last posterior: 40 GB at 100k signatures.  The matrix is banded by graph neighbourhood -- column c holds
// Bayes/PredictionLC[margin + 1] at the graph neighbours of c (addNeighborProb :237-270), the virtual place's value in row 0, a
// constant elsewhere (normalize :434-500) -- so the device keeps the neighbour lists (what Memory::getNeighborsId answered, as the
// reference's incremental updatePrediction :581-592 caches them in _neighborsIndex) and evaluates each column's few non-zeros on
// the fly: O(m * neighbours) per frame.
//
// Layout in HBM (per signature slot s, the slots of the inverted index):
//   nbr         uint32  tiles of 8 slots x K entries, entry k of slot s at ((s / 8) * K + k) * 8 + s % 8: margin << 27 | neighbour slot
//                       (8 lanes walk one slot's list, so a wave reads 8 entries of 8 slots = 256 contiguous bytes per step)
//   cnt[s]      int32   entries in use
//   post[1 + s] float   posterior of the last update BEFORE the division by its sum (kept beside it: readers divide)
//   was_in[s]   uint8   s took part in the last update (BayesFilter::updatePosterior :709-736: others restart at 0)
// Three launches per update, shared with Rtabmap::adjustLikelihood (statistics + columns -> adjusted value + rows -> normalise +
// arg-max); every reduction is in a fixed order (per-workgroup partials folded by the last workgroup), so an update is
// bit-reproducible.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "devbuf.h"

namespace lcd {

constexpr int BAYES_MAX_LC = 32;          // Bayes/PredictionLC values (the default string has 18)
constexpr int BAYES_SLOT_BITS = 27;       // neighbour slot in the low bits of an entry, margin above
constexpr int BAYES_GRID = 256;           // workgroups of the column / row passes (grid-stride; = number of partial sums)
constexpr int BAYES_MAX_K = 8192;         // longest neighbour list accepted

struct BayesParams {                      // kernel argument: BayesFilter's members, in the types the reference computes with
    float lc[BAYES_MAX_LC];               // (float)_predictionLC[k]: what addNeighborProb stores into the float matrix
    int n_lc;
    float total;                          // _totalPredictionLCValues (float accumulator of the double values, :109-117)
    double lc0;                           // _predictionLC[0]
    float eps;                            // _predictionEpsilon (smallest value, :112-115)
    float vp_prior;                       // _virtualPlacePrior
    float max_norm;                       // (float)(1 - _predictionLC[0])   (:466)
    float all_other;                      // _totalPredictionLCValues < 1 ? 1.0f - total : 0   (:448-452)
};

struct HypothesisOut {                    // == lcd_hypothesis (include/lcd.h)
    int32_t sig_id; int32_t slot; float likelihood; float adjusted; float virtual_place; float mean; float stddev; int32_t n_positive;
};

struct BayesOut {                         // == lcd_bayes_result (include/lcd.h)
    int32_t sig_id; int32_t slot; float posterior; float value; float virtual_place; int32_t n_considered; float sum; int32_t reserved;
};

// one decision stage: what goes in and which outputs are wanted
struct DecideArgs {
    const float* like = nullptr;          // raw likelihood over the slots (adjusted on the fly), or NULL: adj_in is the adjusted vector
    float ratio = 0.0f;                   // Rtabmap/VirtualPlaceLikelihoodRatio
    const float* adj_in = nullptr;        // [1 + n_slots], entry 0 = virtual place (only when like == NULL)
    float* adj_out = nullptr;             // out, may be NULL: the adjusted vector
    HypothesisOut* hyp = nullptr;         // out, may be NULL: best raw likelihood + adjustLikelihood's statistics
    bool bayes = false;                   // run the filter
    float* d_posterior = nullptr;         // out, may be NULL
    BayesOut* d_bayes = nullptr;          // out, may be NULL
};

struct Bayes {
    hipStream_t stream = nullptr;
    int64_t* bytes = nullptr;
    BayesParams prm{};
    bool configured = false;
    bool empty = true;                    // BayesFilter::_posterior is empty: the entries of the next update start at 1 (:717-720)
    int K = 64;                           // neighbour capacity per signature: doubles whenever a list could outgrow it
    int64_t cap = 0;                      // slots allocated
    std::vector<int32_t> cnt_ub;          // per slot: upper bound of its list length (every entry ever entered counts once)
    DevBuf nbr, cnt, post, was_in, col, partial, scal, pairs, overflow;
    std::string err;

    void init(hipStream_t s, int64_t* b) { stream = s; bytes = b; }
    void destroy();
    hipError_t configure(const double* lc, int n, float vp_prior);
    hipError_t ensure(int64_t n_slots, int k_needed = 0);
    hipError_t reset();
    // pairs: (slot a, slot b, margin) triples, canonical (a <= b) and unique; both directions are entered.  restart: slots whose own
    // list is emptied first (the signatures the call lists: their list becomes what the call says, plus what later calls enter)
    hipError_t link(const std::vector<int32_t>& triples, const std::vector<int32_t>& restart = std::vector<int32_t>());
    hipError_t ensure_scratch();
    // adjustLikelihood / filter update / hypotheses over the slots [0, n_cons) that are live (slot_sig != 0)
    hipError_t decide(const DecideArgs& d, const int32_t* slot_sig, int64_t n_slots, int64_t n_cons);
    hipError_t read_overflow(int64_t* out);     // synchronises
    // normalised posterior of the first n slots ([0] = virtual place) and who took part in the last update; synchronises
    hipError_t read_posterior(int64_t n, std::vector<float>* p, std::vector<uint8_t>* in);
};

}  // namespace lcd
