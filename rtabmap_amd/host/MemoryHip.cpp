// MemoryHip.cpp -- see MemoryHip.h.
#include "MemoryHip.h"

#include "DbLoaderHip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>

namespace rtabmap_amd {

const int MemoryHip::kIdVirtual = -1;

namespace {
// a roctx range on the engine's track (lcd_set_option "roctx"; a no-op otherwise) + the elapsed milliseconds, like the UTimer the
// reference wraps around the same stages
struct Stage {
    lcd_engine* eng; std::chrono::steady_clock::time_point t0;
    Stage(VWDictionaryHip* vwd, const char* name) : eng(vwd->engine()), t0(std::chrono::steady_clock::now()) { if (eng) lcd_trace_push(eng, name); }
    ~Stage() { if (eng) lcd_trace_pop(eng); }
    float ms() const { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

MemoryHip::MemoryHip(const ParametersMap& parameters, int device)
    : _vwd(new VWDictionaryHip(parameters, device)), _idCount(0), _maxStMemSize(10), _deviceFrames(true), _likeSig(0), _likeSortedValid(false) {
    ParametersMap::const_iterator it = parameters.find("Mem/STMSize");
    if (it != parameters.end()) _maxStMemSize = atoi(it->second.c_str());
    if (_maxStMemSize < 0) _maxStMemSize = 0;
    _workingMem.insert(kIdVirtual);             // Memory.cpp:592
    static const char* names[] = {"TimingMem/Pre_update/ms", "TimingMem/Joining_dictionary_update/ms", "TimingMem/Add_new_words/ms",
                                  "Timing/Likelihood_computation/ms", "Timing/Forgetting/ms", "Keypoint/Dictionary_size/words",
                                  "Keypoint/Current_frame/words", "Keypoint/Indexed_words/words", "Keypoint/Index_memory_usage/KB"};
    for (size_t i = 0; i < sizeof(names) / sizeof(names[0]); ++i) _stats[names[i]] = 0.0f;   // (Statistics::_defaultData: every key exists from the start)
}

void MemoryHip::refreshEngineStatistics() {
    _stats["Keypoint/Dictionary_size/words"] = (float)_vwd->getVisualWords().size();
    _stats["Keypoint/Indexed_words/words"] = (float)_vwd->getIndexedWordsCount();
    lcd_stats st;
    if (_vwd->engine() && lcd_get_stats(_vwd->engine(), &st) == LCD_OK) _stats["Keypoint/Index_memory_usage/KB"] = (float)(st.bytes_device / 1024);
}
MemoryHip::~MemoryHip() { delete _vwd; }

void MemoryHip::cleanUnusedWords() {   // Memory.cpp:6899-6920 (no database: the words are deleted)
    std::vector<VisualWord*> removedWords = _vwd->getUnusedWords();
    if (removedWords.size()) {
        _vwd->removeWords(removedWords);
        for (unsigned int i = 0; i < removedWords.size(); ++i) delete removedWords[i];
    }
}

void MemoryHip::preUpdate() {   // Memory.cpp:1004-1016; with Kp/Parallelized the update() runs on PreUpdateThread, joined before addNewWords
    if (_vwd->isIncremental()) this->cleanUnusedWords();
    _vwd->update();
}

int MemoryHip::update(const Mat& descriptors, int nQuantized, std::vector<int>& outIds) {
    Stage whole(_vwd, "Memory::update");
    {
        Stage st(_vwd, "Memory::preUpdate");
        this->preUpdate();
        _stats["TimingMem/Pre_update/ms"] = st.ms();
    }
    Stage quant(_vwd, "VWDictionary::addNewWords");
    const int id = ++_idCount;
    std::list<int> wordIds;
    const int rows = descriptors.rows;
    int nq = nQuantized < 0 || nQuantized > rows ? rows : nQuantized;
    _likeSig = 0;
    bool fast = false;
    if (rows && nq == rows && _deviceFrames && _vwd->isIncremental()) {
        // quantisation + the signature's references + update()'s append + the likelihood of Rtabmap.cpp:2117, one device call
        fast = _vwd->addNewWordsAndScore(descriptors, id, (float)(_signatures.size() + 1), [this](int s) { return this->getNi(s); }, wordIds, _likeSlots);
        if (fast && !_likeSlots.empty()) { _likeSig = id; _likeSortedValid = false; }
    }
    if (rows && !fast) {
        if (nq > 0) {
            Mat forQuantization = nq == rows ? descriptors : Mat(nq, descriptors.cols, descriptors.type(), descriptors.data.data());
            wordIds = _vwd->addNewWords(forQuantization, id);
            if ((int)wordIds.size() < rows) {   // Memory.cpp:6029-6046: ids -1,-2,.. for features without a word
                std::vector<int> all(rows, -1);
                int i = 0;
                for (std::list<int>::iterator it = wordIds.begin(); it != wordIds.end(); ++it) all[i++] = *it;
                int neg = -1;
                for (i = 0; i < rows; ++i) if (all[i] < 0) all[i] = neg--;
                wordIds.assign(all.begin(), all.end());
            }
        } else {
            int neg = -1;
            for (int i = 0; i < rows; ++i) wordIds.push_back(neg--);
        }
    }
    _stats["TimingMem/Add_new_words/ms"] = quant.ms();          // (on the device-frame path this also holds the frame's likelihood: one call)
    _stats["Keypoint/Current_frame/words"] = (float)wordIds.size();
    _stats["Keypoint/Dictionary_size/words"] = (float)_vwd->getVisualWords().size();
    _stats["Keypoint/Indexed_words/words"] = (float)_vwd->getIndexedWordsCount();
    _signatures[id] = std::vector<int>(wordIds.begin(), wordIds.end());
    outIds.assign(wordIds.begin(), wordIds.end());
    this->addSignatureToStm(id);
    return id;
}

// Memory::addSignatureToStm :1146-1230 (the neighbour link to the newest signature of the short-term memory; poses and covariances
// are out of scope) followed by the transfer loop of Memory::update :1112-1135 (no intermediate nodes here: every signature counts)
void MemoryHip::addSignatureToStm(int id) {
    if (_stMem.size()) {
        const int last = *_stMem.rbegin();
        _links[id][last] = kNeighbor;
        _links[last][id] = kNeighbor;
    }
    _stMem.insert(id);
    while (_stMem.size() && _maxStMemSize > 0 && (int)_stMem.size() > _maxStMemSize) {
        const int oldest = *_stMem.begin();     // moveSignatureToWMFromSTM :1442-1500 without graph reduction
        _workingMem.insert(oldest);
        _stMem.erase(oldest);
    }
}

// Memory::addLink :3877-3935 for a loop closure between two signatures in memory: nothing to do (true) when they are linked
// already, false + error when one of them is not in the working / short-term memory (weights are out of scope)
bool MemoryHip::addLink(int from, int to, LinkType type) {
    if (!_signatures.count(from) || !_signatures.count(to)) {
        if (!_signatures.count(from)) fprintf(stderr, "[ERROR] from=%d, to=%d, Signature %d not found in working/st memories\n", from, to, from);
        if (!_signatures.count(to)) fprintf(stderr, "[ERROR] from=%d, to=%d, Signature %d not found in working/st memories\n", from, to, to);
        return false;
    }
    if (from == to) return false;
    std::map<int, std::map<int, LinkType> >::const_iterator l = _links.find(to);
    if (l != _links.end() && l->second.count(from)) return true;
    _links[from][to] = type;
    _links[to][from] = type;
    return true;
}

// Memory::getNeighborsId(id, maxGraphDepth, 0, false, false, true, true) :1703-1893.  Level by level over the neighbour links; a
// loop-closure link puts its far end on the SAME level as the node it leaves from (incrementMarginOnLoop = false) and a node is
// taken that way only once per call; nodes that are not in memory are neither reported nor expanded (maxCheckedInDatabase = 0).
// Intermediate nodes, landmarks and local-space closures do not exist in this subset.  maxGraphDepth = 0: no limit.
std::map<int, int> MemoryHip::getNeighborsId(int signatureId, int maxGraphDepth) const {
    std::map<int, int> found;
    if (signatureId <= 0 || maxGraphDepth < 0) return found;
    std::set<int> nextLevel, viaClosure;
    nextLevel.insert(signatureId);
    for (int margin = 0; (maxGraphDepth == 0 || margin < maxGraphDepth) && !nextLevel.empty(); ++margin) {
        std::list<int> level(nextLevel.rbegin(), nextLevel.rend());       // most recent first, as the reference walks a level
        nextLevel.clear();
        for (std::list<int>::iterator n = level.begin(); n != level.end(); ++n) {
            if (found.count(*n) || !_signatures.count(*n)) continue;
            found[*n] = margin;
            std::map<int, std::map<int, LinkType> >::const_iterator links = _links.find(*n);
            if (links == _links.end()) continue;
            for (std::map<int, LinkType>::const_iterator l = links->second.begin(); l != links->second.end(); ++l) {
                if (found.count(l->first)) continue;
                if (l->second == kNeighbor) nextLevel.insert(l->first);
                else if (viaClosure.insert(l->first).second) level.push_back(l->first);
            }
        }
    }
    return found;
}

int MemoryHip::addSignature(const std::vector<int>& wordIds, int id) {
    _likeSig = 0;
    if (id == 0) id = ++_idCount;
    if (_signatures.count(id)) return 0;
    for (size_t k = 0; k < wordIds.size(); ++k) if (wordIds[k] > 0) _vwd->addWordRef(wordIds[k], id);
    _signatures[id] = wordIds;
    if (id > _idCount) _idCount = id;
    _workingMem.insert(id);                     // Memory::loadDataFromDb: loaded signatures enter the working memory (:447-480)
    return id;
}

int MemoryHip::loadDataFromDb(const std::string& path, bool lastStateOnly) {
    DbLoaderHip db;
    DbDictionary dict;
    DbSignatures sigs;
    _loadError.clear();
    if (!db.open(path) || !db.loadSignatureWords(sigs, lastStateOnly) || !db.loadDictionary(dict, sigs.sigIds.empty() && _vwd->isIncremental())) {
        _loadError = db.lastError();
        return -1;
    }
    // which words to keep: an incremental dictionary only what the loaded signatures reference (Memory.cpp:394-424)
    std::set<int> referenced;
    if (_vwd->isIncremental() && !sigs.sigIds.empty())
        for (size_t k = 0; k < sigs.wordIds.size(); ++k) if (sigs.wordIds[k] > 0) referenced.insert(sigs.wordIds[k]);
    const bool filter = _vwd->isIncremental() && !sigs.sigIds.empty();
    const size_t rowBytes = (size_t)dict.cols * (dict.type == MAT_32F ? 4 : 1);
    // every check that can fail comes BEFORE the first change of state: a refused load leaves the memory as it was
    {
        std::set<int> have(dict.wordIds.begin(), dict.wordIds.end()), nodes;
        const std::map<int, VisualWord*>& old = _vwd->getVisualWords();
        for (size_t s = 0; s < sigs.sigIds.size(); ++s) {
            if (_signatures.count(sigs.sigIds[s]) || !nodes.insert(sigs.sigIds[s]).second) {
                _loadError = "node " + std::to_string(sigs.sigIds[s]) + " is in memory already";
                return -1;
            }
            for (int64_t k = sigs.offsets[s]; k < sigs.offsets[s + 1]; ++k) {
                const int w = sigs.wordIds[(size_t)k];
                if (w > 0 && !have.count(w) && !old.count(w)) {
                    _loadError = "The dictionary is empty or missing some words from nodes in WM (word " + std::to_string(w) + " of node " +
                                 std::to_string(sigs.sigIds[s]) + ")";
                    fprintf(stderr, "[ERROR] %s\n", _loadError.c_str());
                    return -1;
                }
            }
        }
    }
    for (size_t k = 0; k < dict.wordIds.size(); ++k) {
        if (filter && !referenced.count(dict.wordIds[k])) continue;
        VisualWord* vw = new VisualWord(dict.wordIds[k], Mat(1, dict.cols, dict.type, dict.rows.data() + k * rowBytes));
        vw->setSaved(true);
        _vwd->addWord(vw);
    }
    if (!dict.wordIds.empty() || dict.lastWordId > 0) _vwd->setLastWordId(dict.lastWordId);   // DBDriverSqlite3::loadQuery :3617-3619: fixed dictionaries too
    _vwd->update();
    if (_vwd->getVisualWords().size() && !_vwd->isAvailable()) { _loadError = _vwd->lastError(); return -1; }
    int loaded = 0;
    for (size_t s = 0; s < sigs.sigIds.size(); ++s) {
        const std::vector<int> words(sigs.wordIds.begin() + sigs.offsets[s], sigs.wordIds.begin() + sigs.offsets[s + 1]);
        for (size_t k = 0; k < words.size(); ++k)
            if (words[k] > 0 && !_vwd->getWord(words[k])) {
                _loadError = "The dictionary is empty or missing some words from nodes in WM (word " + std::to_string(words[k]) + " of node " +
                             std::to_string(sigs.sigIds[s]) + ")";
                fprintf(stderr, "[ERROR] %s\n", _loadError.c_str());
                return -1;
            }
        if (this->addSignature(words, sigs.sigIds[s]) != sigs.sigIds[s]) { _loadError = "node " + std::to_string(sigs.sigIds[s]) + " is in memory already"; return -1; }
        loaded += 1;
    }
    if (!_vwd->flushReferencesBulk([this](int id) { return this->getNi(id); })) { _loadError = _vwd->lastError(); return -1; }
    return loaded;
}

int MemoryHip::getNi(int signatureId) const {   // Memory.cpp:4955-4968
    std::map<int, std::vector<int> >::const_iterator it = _signatures.find(signatureId);
    if (it != _signatures.end()) return (int)it->second.size();
    std::map<int, int>::const_iterator d = _dbNi.find(signatureId);
    return d == _dbNi.end() ? 0 : d->second;
}

void MemoryHip::forget(int signatureId) {
    std::map<int, std::vector<int> >::iterator it = _signatures.find(signatureId);
    if (it == _signatures.end()) return;
    Stage st(_vwd, "Memory::forget");
    struct Put { std::map<std::string, float>& m; Stage& s; ~Put() { m["Timing/Forgetting/ms"] = s.ms(); } } put{_stats, st};
    _likeSig = 0;                               // N and the references change: the frame's likelihood is no longer Memory::computeLikelihood's
    std::set<int> keys(it->second.begin(), it->second.end());   // uUniqueKeys (Memory.cpp:6885)
    _vwd->removeAllWordRefs(keys, signatureId);
    _dbNi[signatureId] = (int)it->second.size();
    _signatures.erase(it);
    _stMem.erase(signatureId);                  // the links stay (they are in the database): getNeighborsId stops at the missing node
    _workingMem.erase(signatureId);
}

std::vector<int> MemoryHip::signatureIds() const {
    std::vector<int> v;
    for (std::map<int, std::vector<int> >::const_iterator i = _signatures.begin(); i != _signatures.end(); ++i) v.push_back(i->first);
    return v;
}

std::map<int, float> MemoryHip::computeLikelihood(const std::list<int>& wordIds, const std::list<int>& ids) {
    Stage st(_vwd, "Memory::computeLikelihood");
    const float N = (float)_signatures.size();   // Memory.cpp:2248: every signature in memory, not only `ids`
    std::map<int, float> L = _vwd->computeLikelihood(wordIds, ids, N, [this](int s) { return this->getNi(s); });
    _stats["Timing/Likelihood_computation/ms"] = st.ms();
    return L;
}

// (signature id, likelihood) of every signature registered on the device, ascending id, from the frame's slot-indexed result
const std::vector<std::pair<int, float> >& MemoryHip::sortedLikelihood() {
    if (_likeSortedValid) return _likeSorted;
    const std::vector<int>& slotSig = _vwd->slotSignatures();
    _likeSorted.clear();
    const size_t n = std::min(slotSig.size(), _likeSlots.size());
    _likeSorted.reserve(n);
    bool ascending = true;
    for (size_t s = 0; s < n; ++s) {
        if (slotSig[s] == 0) continue;
        if (!_likeSorted.empty() && slotSig[s] < _likeSorted.back().first) ascending = false;
        _likeSorted.push_back(std::pair<int, float>(slotSig[s], _likeSlots[s]));
    }
    if (!ascending) std::sort(_likeSorted.begin(), _likeSorted.end());   // (re-registered signatures sit in later slots)
    _likeSortedValid = true;
    return _likeSorted;
}

static float lookupSorted(const std::vector<std::pair<int, float> >& v, size_t& cursor, int id) {
    // ids usually arrive ascending (Rtabmap builds the list from a std::map): a cursor that only moves forward; else a binary search
    if (cursor < v.size() && v[cursor].first <= id) {
        while (cursor < v.size() && v[cursor].first < id) ++cursor;
        return cursor < v.size() && v[cursor].first == id ? v[cursor].second : 0.0f;
    }
    std::vector<std::pair<int, float> >::const_iterator it = std::lower_bound(v.begin(), v.end(), std::pair<int, float>(id, -1e30f));
    cursor = (size_t)(it - v.begin());
    return it != v.end() && it->first == id ? it->second : 0.0f;
}

std::map<int, float> MemoryHip::computeLikelihood(int signatureId, const std::list<int>& ids) {
    std::map<int, std::vector<int> >::const_iterator it = _signatures.find(signatureId);
    if (it == _signatures.end()) return std::map<int, float>();   // "The signature is null" (Memory.cpp:2222)
    if (_likeSig != 0 && _likeSig == signatureId) {
        Stage st(_vwd, "Memory::computeLikelihood");
        struct Put { std::map<std::string, float>& m; Stage& s; ~Put() { m["Timing/Likelihood_computation/ms"] = s.ms(); } } put{_stats, st};
        std::map<int, float> likelihood;
        if (ids.empty()) { fprintf(stderr, "[WARN] ids list is empty\n"); return likelihood; }   // :2227-2231
        const std::vector<std::pair<int, float> >& v = sortedLikelihood();
        size_t cursor = 0;
        bool sorted = true; int last = 0; bool first = true;
        for (std::list<int>::const_iterator i = ids.begin(); i != ids.end(); ++i) { if (!first && *i <= last) { sorted = false; break; } last = *i; first = false; }
        if (sorted) for (std::list<int>::const_iterator i = ids.begin(); i != ids.end(); ++i) likelihood.insert(likelihood.end(), std::pair<int, float>(*i, lookupSorted(v, cursor, *i)));
        else for (std::list<int>::const_iterator i = ids.begin(); i != ids.end(); ++i) likelihood[*i] = lookupSorted(v, cursor, *i);
        return likelihood;
    }
    return computeLikelihood(std::list<int>(it->second.begin(), it->second.end()), ids);
}

void MemoryHip::computeLikelihood(int signatureId, const std::list<int>& ids, std::map<int, float>& likelihood) {
    if (_likeSig == 0 || _likeSig != signatureId || !_signatures.count(signatureId) || ids.empty()) { likelihood = this->computeLikelihood(signatureId, ids); return; }
    Stage st(_vwd, "Memory::computeLikelihood");
    struct Put { std::map<std::string, float>& m; Stage& s; ~Put() { m["Timing/Likelihood_computation/ms"] = s.ms(); } } put{_stats, st};
    const std::vector<std::pair<int, float> >& v = sortedLikelihood();
    size_t cursor = 0;
    std::map<int, float>::iterator m = likelihood.begin();
    int last = 0; bool first = true;
    for (std::list<int>::const_iterator i = ids.begin(); i != ids.end(); ++i) {
        if (!first && *i <= last) { likelihood = this->computeLikelihood(signatureId, ids); return; }   // unsorted ids: the plain way
        last = *i; first = false;
        while (m != likelihood.end() && m->first < *i) m = likelihood.erase(m);
        const float val = lookupSorted(v, cursor, *i);
        if (m != likelihood.end() && m->first == *i) { m->second = val; ++m; }
        else m = ++likelihood.insert(m, std::pair<int, float>(*i, val));
    }
    likelihood.erase(m, likelihood.end());
}

bool MemoryHip::computeLikelihoodFlat(int signatureId, std::vector<int>& sigIds, std::vector<float>& values) {
    if (_likeSig == 0 || _likeSig != signatureId) return false;
    if (!_likeSortedValid) {
        // the usual case -- signatures registered in ascending id, so the device's slot order IS ascending id -- in ONE pass over the slots, without
        // the (id, value) pairs the map overloads look things up in: 100 000 slots are 1.2 MB read and 0.8 MB written
        Stage st(_vwd, "Memory::computeLikelihood");
        const std::vector<int>& slotSig = _vwd->slotSignatures();
        const size_t n = std::min(slotSig.size(), _likeSlots.size());
        sigIds.resize(n);
        values.resize(n);
        size_t m = 0;
        int last = 0;
        bool ascending = true;
        for (size_t k = 0; k < n; ++k) {
            const int id = slotSig[k];
            if (id == 0) continue;
            if (id < last) { ascending = false; break; }
            last = id;
            sigIds[m] = id; values[m] = _likeSlots[k];
            ++m;
        }
        if (ascending) {
            sigIds.resize(m);
            values.resize(m);
            _stats["Timing/Likelihood_computation/ms"] = st.ms();
            return true;
        }
    }
    const std::vector<std::pair<int, float> >& v = sortedLikelihood();
    sigIds.resize(v.size());
    values.resize(v.size());
    for (size_t k = 0; k < v.size(); ++k) { sigIds[k] = v[k].first; values[k] = v[k].second; }
    return true;
}

}  // namespace rtabmap_amd
