// RtabmapHip.cpp -- see RtabmapHip.h.
#include "RtabmapHip.h"

#include <cstdio>
#include <cstdlib>
#include <list>

namespace rtabmap_amd {

RtabmapHip::RtabmapHip(const ParametersMap& parameters, int device)
    : _memory(new MemoryHip(parameters, device)), _bayesFilter(new BayesFilterHip(parameters)), _loopThr(0.11f), _loopRatio(0.0f),
      _virtualPlaceLikelihoodRatio(0), _loopClosureHypothesis(0, 0.0f), _highestHypothesis(0, 0.0f), _lastLocationId(0) {
    this->parseParameters(parameters);   // Parameters.h:197-200 defaults above
}

RtabmapHip::~RtabmapHip() {
    delete _bayesFilter;
    delete _memory;
}

void RtabmapHip::parseParameters(const ParametersMap& parameters) {
    ParametersMap::const_iterator it;
    if ((it = parameters.find("Rtabmap/LoopThr")) != parameters.end()) _loopThr = uStr2Float(it->second);
    if ((it = parameters.find("Rtabmap/LoopRatio")) != parameters.end()) _loopRatio = uStr2Float(it->second);
    if ((it = parameters.find("Rtabmap/VirtualPlaceLikelihoodRatio")) != parameters.end()) _virtualPlaceLikelihoodRatio = atoi(it->second.c_str());
}

// Rtabmap::adjustLikelihood :5691-5757: the statistics, the rescaling and the virtual place's value are one device pass over the
// values in map order (the first entry is the virtual place); an empty likelihood is left alone (:5696-5699)
void RtabmapHip::adjustLikelihood(std::map<int, float>& likelihood) const {
    if (likelihood.size() == 0) return;
    lcd_engine* engine = _memory->getVWDictionary()->engine();
    if (!engine) { fprintf(stderr, "[ERROR] adjustLikelihood: no device engine\n"); return; }
    std::vector<float> values;
    values.reserve(likelihood.size());
    for (std::map<int, float>::const_iterator i = likelihood.begin(); i != likelihood.end(); ++i) values.push_back(i->second);
    if (lcd_adjust_likelihood(engine, values.data(), (int)values.size(), (float)_virtualPlaceLikelihoodRatio) != LCD_OK) {
        fprintf(stderr, "[ERROR] adjustLikelihood: %s\n", lcd_last_error(engine));
        return;
    }
    size_t k = 0;
    for (std::map<int, float>::iterator i = likelihood.begin(); i != likelihood.end(); ++i, ++k) i->second = values[k];
}

bool RtabmapHip::process(const Mat& descriptors) {
    // :1263-1265
    _loopClosureHypothesis = std::pair<int, float>(0, 0.0f);
    const std::pair<int, float> lastHighestHypothesis = _highestHypothesis;
    _highestHypothesis = std::pair<int, float>(0, 0.0f);
    _rawLikelihood.clear(); _likelihood.clear(); _posterior.clear();

    _lastLocationId = _memory->update(descriptors, -1, _lastWordIds);       // Memory::update :1470-1477
    if (_lastLocationId <= 0) return false;
    if (descriptors.rows == 0) return true;   // a signature without features is a bad signature: "Ignoring likelihood and loop closure hypotheses" (:2234-2237)
    const std::set<int>& wm = _memory->getWorkingMem();
    if (wm.size() > 1) {                                                    // getWorkingMemSize(): signatures besides the virtual place
        // :2046-2118: every location of the working memory + the virtual one
        std::list<int> signaturesToCompare(wm.begin(), wm.end());
        _rawLikelihood = _memory->computeLikelihood(_lastLocationId, signaturesToCompare);
        _likelihood = _rawLikelihood;
        this->adjustLikelihood(_likelihood);                                // :2121
        _posterior = _bayesFilter->computePosterior(_memory, _likelihood);  // :2131
        if (_posterior.size() && _bayesFilter->lastUpdateOk()) {            // :2147-2158 (the device selected it in the same pass); a failed
            _highestHypothesis = _bayesFilter->getHighestHypothesis();      // update leaves (0, 0): no acceptance on a stale posterior
        }
        if (_highestHypothesis.first > 0) {                                 // :2162-2222 without the RGB-D and epipolar branches
            const float loopThr = _loopThr;
            if (_highestHypothesis.second >= loopThr) {
                if (_posterior.size() <= 2 && loopThr > 0.0f) {
                    // rejected hypothesis: single hypothesis
                } else if (_loopRatio > 0.0f && lastHighestHypothesis.second && _highestHypothesis.second < _loopRatio * lastHighestHypothesis.second) {
                    fprintf(stderr, "[ WARN] rejected hypothesis: not satisfying hypothesis ratio (%f < %f * %f)\n", _highestHypothesis.second, _loopRatio,
                            lastHighestHypothesis.second);
                } else if (_loopRatio > 0.0f && lastHighestHypothesis.second == 0) {
                    fprintf(stderr, "[ WARN] rejected hypothesis: last closure hypothesis is null (loop ratio is on)\n");
                } else {
                    _loopClosureHypothesis = _highestHypothesis;
                }
            }
        }
    }
    // :3129-3186: the global loop closure becomes a link of the graph (no transform in appearance-only mode)
    if (_loopClosureHypothesis.first > 0 && !_memory->addLink(_lastLocationId, _loopClosureHypothesis.first, MemoryHip::kGlobalClosure))
        _loopClosureHypothesis.first = 0;
    return true;
}

}  // namespace rtabmap_amd
