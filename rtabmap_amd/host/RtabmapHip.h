// RtabmapHip.h -- the loop-closure detection block of rtabmap::Rtabmap::process for an appearance-only frame, over the host mirrors.
//
// Mirrors (reference corelib/src/Rtabmap.cpp): the likelihood over the working memory :2046-2118 (every signature of the working
// memory and the virtual place; GPS and intermediate-node filters are out of scope), adjustLikelihood :5691-5757, the posterior
// :2131, the highest hypothesis :2147-2158, its acceptance against Rtabmap/LoopThr and Rtabmap/LoopRatio :2162-2222, and the
// global loop-closure link :3129-3186.  Same member and parameter names.  Everything else of Rtabmap::process (odometry, graph
// optimisation, retrieval, memory management, statistics) is out of scope.
// A signature without words never reaches the device's inverted index (VWDictionaryHip::flushReferences), so a working memory that
// holds one makes computePosterior refuse (error logged, the last posterior stays).  The reference keeps such "bad signatures" in
// the working memory with likelihood 0 unless Mem/BadSignaturesIgnored drops them (Memory.cpp:2448, :2800; default false): frames
// without features are the caller's to skip here.
// Each step is one device call behind the mirrors: MemoryHip::update (lcd_quantize), MemoryHip::computeLikelihood
// (lcd_likelihood), adjustLikelihood (lcd_adjust_likelihood), BayesFilterHip::computePosterior (lcd_bayes_*).
#pragma once
#include <map>
#include <utility>
#include <vector>

#include "BayesFilterHip.h"
#include "MemoryHip.h"

namespace rtabmap_amd {

class RtabmapHip {
public:
    explicit RtabmapHip(const ParametersMap& parameters = ParametersMap(), int device = 0);
    ~RtabmapHip();
    void parseParameters(const ParametersMap& parameters);   // Rtabmap/LoopThr, Rtabmap/LoopRatio, Rtabmap/VirtualPlaceLikelihoodRatio

    // one frame of descriptors; true when the frame was processed
    bool process(const Mat& descriptors);

    int getLoopClosureId() const { return _loopClosureHypothesis.first; }
    float getLoopClosureValue() const { return _loopClosureHypothesis.second; }
    int getHighestHypothesisId() const { return _highestHypothesis.first; }
    float getHighestHypothesisValue() const { return _highestHypothesis.second; }
    int getLastLocationId() const { return _lastLocationId; }
    float getLoopThr() const { return _loopThr; }
    float getLoopRatio() const { return _loopRatio; }
    const MemoryHip* getMemory() const { return _memory; }
    MemoryHip* getMemory() { return _memory; }
    BayesFilterHip* getBayesFilter() { return _bayesFilter; }
    // the vectors of the last frame (Statistics::likelihood / rawLikelihood / posterior, Rtabmap.cpp:4168-4180)
    const std::map<int, float>& getRawLikelihood() const { return _rawLikelihood; }
    const std::map<int, float>& getLikelihood() const { return _likelihood; }
    const std::map<int, float>& getPosterior() const { return _posterior; }
    const std::vector<int>& getLastWordIds() const { return _lastWordIds; }

    void adjustLikelihood(std::map<int, float>& likelihood) const;

private:
    MemoryHip* _memory;
    BayesFilterHip* _bayesFilter;
    float _loopThr, _loopRatio;
    int _virtualPlaceLikelihoodRatio;
    std::pair<int, float> _loopClosureHypothesis, _highestHypothesis;
    int _lastLocationId;
    std::map<int, float> _rawLikelihood, _likelihood, _posterior;
    std::vector<int> _lastWordIds;
};

}  // namespace rtabmap_amd
