// BayesFilterHip.h -- host-side mirror of rtabmap::BayesFilter over the C-ABI of include/lcd.h (lcd_bayes_*).
//
// Same public interface as the reference class (corelib/include/rtabmap/core/BayesFilter.h:43-86): parseParameters,
// computePosterior(memory, likelihood), reset, setPredictionLC, getPosterior, getVirtualPlacePrior, getPredictionLC(Str).
// The reference builds an m x m prediction matrix on the host (generatePrediction :273-420, or patches the previous one,
// updatePrediction :502-706) and multiplies it with the last posterior; here the device keeps the neighbour lists the matrix is
// made of and evaluates the product from them (rtabmap_amd/csrc/bayes.hip), so generatePrediction() -- a cv::Mat of 40 GB at
// 100 000 signatures -- has no counterpart.  The host does what updatePrediction does with Memory: for every id that is new in
// the likelihood it asks getNeighborsId(id, _predictionLC.size() - 1, ...) once (:581-592) and hands the list to the device,
// which enters it into the neighbours' lists as well (the reference's _neighborsIndex); Bayes/FullPredictionUpdate = true asks
// for every id on every call (:328-352) -- the same lists unless loop closures have shortened a path since.
// No filter arithmetic runs on the host; without the engine every call fails loudly.
#pragma once
#include <map>
#include <set>
#include <string>
#include <vector>

#include "MemoryHip.h"

namespace rtabmap_amd {

class BayesFilterHip {
public:
    explicit BayesFilterHip(const ParametersMap& parameters = ParametersMap());
    virtual ~BayesFilterHip();
    virtual void parseParameters(const ParametersMap& parameters);   // Bayes/PredictionLC, Bayes/VirtualPlacePriorThr, Bayes/FullPredictionUpdate
    const std::map<int, float>& computePosterior(const MemoryHip* memory, const std::map<int, float>& likelihood);
    void reset();

    void setPredictionLC(const std::string& prediction);

    const std::map<int, float>& getPosterior() const { return _posterior; }
    float getVirtualPlacePrior() const { return _virtualPlacePrior; }
    bool isFullPredictionUpdate() const { return _fullPredictionUpdate; }
    const std::vector<double>& getPredictionLC() const { return _predictionLC; }   // {Vp, Lc, l1, l2, l3, l4...}
    std::string getPredictionLCStr() const;

    // what Rtabmap.cpp:2147-2158 reads off the posterior, computed by the same device pass: (signature id, 1 - virtual place)
    std::pair<int, float> getHighestHypothesis() const { return _highestHypothesis; }
    // false when the last computePosterior failed (the reference logs UERROR and returns the old posterior, BayesFilter.cpp:150-166; here
    // the device can fail as well): the highest hypothesis is (0, 0) then, so that no caller accepts a stale one
    bool lastUpdateOk() const { return _lastUpdateOk; }
    const std::string& lastError() const { return _lastError; }

private:
    bool configureDevice(lcd_engine* engine);
    const std::map<int, float>& failed() { _lastUpdateOk = false; _highestHypothesis = std::pair<int, float>(0, 0.0f); return _posterior; }

private:
    std::map<int, float> _posterior;
    float _virtualPlacePrior;
    std::vector<double> _predictionLC;   // {Vp, Lc, l1, l2, l3, l4...}
    bool _fullPredictionUpdate;
    // device side
    lcd_engine* _engine;                 // the engine the filter state lives on (the dictionary's, seen at the first update)
    bool _deviceConfigured;              // it holds the current _predictionLC / prior (false: configure before the next update)
    std::set<int> _listedIds;            // ids whose neighbour list the device has (the keys of the reference's _neighborsIndex)
    std::pair<int, float> _highestHypothesis;
    bool _lastUpdateOk = true;
    std::string _lastError;
};

}  // namespace rtabmap_amd
