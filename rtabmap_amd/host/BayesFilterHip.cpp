// BayesFilterHip.cpp -- see BayesFilterHip.h.
#include "BayesFilterHip.h"

#include <cstdio>
#include <cstdlib>
#include <sstream>

namespace rtabmap_amd {

namespace {
// Parameters.h:362-364
const char* kDefaultPredictionLC = "0.1 0.36 0.30 0.16 0.062 0.0151 0.00255 0.000324 2.5e-05 1.3e-06 4.8e-08 1.2e-09 1.9e-11 2.2e-13 1.7e-15 8.5e-18 2.9e-20 6.9e-23";
const float kDefaultVirtualPlacePrior = 0.9f;

bool parseBool(const std::string& s) { return s == "true" || s == "True" || s == "TRUE" || s == "1"; }
// uStr2Float: the decimal mark may be '.' or ',' whatever the locale
float str2Float(const std::string& s) { return uStr2Float(s); }   // classic ("C") locale whatever LC_NUMERIC says (VWDictionaryHip.h)
void logError(const std::string& m) { fprintf(stderr, "[ERROR] %s\n", m.c_str()); }
}  // namespace

BayesFilterHip::BayesFilterHip(const ParametersMap& parameters)
    : _virtualPlacePrior(kDefaultVirtualPlacePrior), _fullPredictionUpdate(false), _engine(0), _deviceConfigured(false), _highestHypothesis(0, 0.0f) {
    this->setPredictionLC(kDefaultPredictionLC);
    this->parseParameters(parameters);
}

BayesFilterHip::~BayesFilterHip() {}

void BayesFilterHip::parseParameters(const ParametersMap& parameters) {   // BayesFilter.cpp:56-69
    ParametersMap::const_iterator iter;
    if ((iter = parameters.find("Bayes/PredictionLC")) != parameters.end()) this->setPredictionLC(iter->second);
    if ((iter = parameters.find("Bayes/VirtualPlacePriorThr")) != parameters.end()) {
        const float v = str2Float(iter->second);
        if (v >= 0.0f && v <= 1.0f) _virtualPlacePrior = v;               // the reference asserts the range (:68)
        else logError("Bayes/VirtualPlacePriorThr must be in [0, 1]: \"" + iter->second + "\" ignored");
        _deviceConfigured = false;
    }
    if ((iter = parameters.find("Bayes/FullPredictionUpdate")) != parameters.end()) _fullPredictionUpdate = parseBool(iter->second);
}

// format = {Virtual place, Loop closure, level1, level2, l3, l4...}; BayesFilter.cpp:72-122: fewer than two values or a value
// outside [0, 1] leaves the current values in place (error logged); each value goes through a float on its way into the doubles
void BayesFilterHip::setPredictionLC(const std::string& prediction) {
    std::vector<std::string> strValues;
    std::istringstream in(prediction);
    for (std::string tok; std::getline(in, tok, ' ');) if (!tok.empty()) strValues.push_back(tok);   // uSplit drops empty tokens
    if (strValues.size() < 2) {
        logError("The number of values < 2 (prediction=\"" + prediction + "\")");
        return;
    }
    if (strValues.size() > 32) {          // the device keeps the pattern in a 32-entry table (the margin field of a list entry has 5 bits)
        logError("Bayes/PredictionLC: more than 32 values are not supported by the device filter (prediction=\"" + prediction + "\")");
        return;
    }
    std::vector<double> tmpValues(strValues.size());
    for (size_t i = 0; i < strValues.size(); ++i) {
        tmpValues[i] = str2Float(strValues[i]);
        if (tmpValues[i] < 0.0 || tmpValues[i] > 1.0) {
            logError("The prediction is not valid (values must be between >0 && <=1) prediction=\"" + prediction + "\"");
            return;
        }
    }
    _predictionLC = tmpValues;
    _deviceConfigured = false;           // _totalPredictionLCValues / _predictionEpsilon (:109-117) are derived on the device side
}

std::string BayesFilterHip::getPredictionLCStr() const {   // :130-142
    std::string values;
    for (unsigned int i = 0; i < _predictionLC.size(); ++i) {
        std::ostringstream s;
        s << _predictionLC[i];
        values.append(s.str());
        if (i + 1 < _predictionLC.size()) values.append(" ");
    }
    return values;
}

void BayesFilterHip::reset() {   // :138-143
    _posterior.clear();
    _listedIds.clear();
    _highestHypothesis = std::pair<int, float>(0, 0.0f);
    if (_engine && lcd_bayes_reset(_engine) != LCD_OK) logError(lcd_last_error(_engine));
}

bool BayesFilterHip::configureDevice(lcd_engine* engine) {
    if (_engine == engine && _deviceConfigured) return true;
    if (_predictionLC.size() > 32) { _lastError = "Bayes/PredictionLC: at most 32 values"; return false; }
    if (lcd_bayes_configure(engine, _predictionLC.data(), (int)_predictionLC.size(), _virtualPlacePrior) != LCD_OK) {
        _lastError = lcd_last_error(engine);
        return false;
    }
    if (_engine && _engine != engine) { _posterior.clear(); _listedIds.clear(); }   // another device state altogether
    _engine = engine;
    _deviceConfigured = true;
    return true;
}

const std::map<int, float>& BayesFilterHip::computePosterior(const MemoryHip* memory, const std::map<int, float>& likelihood) {
    // the reference's three refusals (:149-165): the last posterior is returned unchanged
    if (!memory) { logError("Memory is Null!"); return failed(); }
    if (!likelihood.size()) { logError("likelihood is empty!"); return failed(); }
    if (_predictionLC.size() < 2) { logError("Prediction is not valid!"); return failed(); }
    lcd_engine* engine = const_cast<MemoryHip*>(memory)->getVWDictionary()->engine();
    if (!engine) { _lastError = "no device engine (the dictionary has not seen a descriptor yet)"; logError(_lastError); return failed(); }
    if (!this->configureDevice(engine)) { logError(_lastError); return failed(); }
    // the device scores registered signatures: references added since the last likelihood are sent now
    if (!const_cast<MemoryHip*>(memory)->flushReferences()) { _lastError = memory->lastError(); logError(_lastError); return failed(); }

    // STEP 1 of the reference -- the prediction -- is the neighbour lists.  updatePrediction :560-592: ids that are new in the
    // likelihood bring their neighbourhood; removed ids need nothing (a signature that left the memory is skipped on the device).
    std::vector<int32_t> listIds, nbrIds, nbrMargins;
    std::vector<int64_t> offsets(1, 0);
    std::set<int> present;
    for (std::map<int, float>::const_iterator i = likelihood.begin(); i != likelihood.end(); ++i) {
        if (i->first <= 0) continue;
        present.insert(i->first);
        if (!_fullPredictionUpdate && _listedIds.count(i->first)) continue;
        const std::map<int, int> neighbors = memory->getNeighborsId(i->first, (int)_predictionLC.size() - 1);
        listIds.push_back(i->first);
        // (neighbours still in the short-term memory stay in the list, as in _neighborsIndex: they are not in the likelihood, so
        // they take no part -- addNeighborProb :244-262 / the filter of :338-349 -- until they reach the working memory)
        for (std::map<int, int>::const_iterator n = neighbors.begin(); n != neighbors.end(); ++n) {
            nbrIds.push_back(n->first);
            nbrMargins.push_back(n->second);
        }
        offsets.push_back((int64_t)nbrIds.size());
    }
    if (!listIds.empty() &&
        lcd_bayes_set_neighbors(engine, (int)listIds.size(), listIds.data(), offsets.data(), nbrIds.data(), nbrMargins.data()) != LCD_OK) {
        _lastError = lcd_last_error(engine);
        logError(_lastError);
        return failed();
    }
    _listedIds.swap(present);            // the keys of _neighborsIndex after updatePrediction: exactly the ids of this likelihood

    // STEPS 1b-2 + normalisation (:172-232) on the device
    std::vector<int32_t> ids;
    std::vector<float> values;
    ids.reserve(likelihood.size());
    values.reserve(likelihood.size());
    for (std::map<int, float>::const_iterator i = likelihood.begin(); i != likelihood.end(); ++i) { ids.push_back(i->first); values.push_back(i->second); }
    lcd_bayes_result r;
    if (lcd_bayes_update(engine, ids.data(), values.data(), (int)ids.size(), &r) != LCD_OK) {
        _lastError = lcd_last_error(engine);
        logError(_lastError);
        return failed();
    }
    std::vector<float> post(ids.size(), 0.0f);
    if (lcd_bayes_posterior(engine, ids.data(), (int)ids.size(), post.data()) != LCD_OK) {
        _lastError = lcd_last_error(engine);
        logError(_lastError);
        return failed();
    }
    _posterior.clear();                  // updatePosterior :709-736: the posterior holds exactly the ids of the likelihood
    for (size_t k = 0; k < ids.size(); ++k) _posterior.insert(_posterior.end(), std::pair<int, float>(ids[k], post[k]));
    _highestHypothesis = std::pair<int, float>(r.sig_id, r.value);
    _lastUpdateOk = true;
    return _posterior;
}

}  // namespace rtabmap_amd
