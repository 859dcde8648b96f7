// MemoryHip.cpp -- see MemoryHip.h.
#include "MemoryHip.h"

#include <set>

namespace rtabmap_amd {

MemoryHip::MemoryHip(const ParametersMap& parameters, int device) : _vwd(new VWDictionaryHip(parameters, device)), _idCount(0) {}
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
    this->preUpdate();
    const int id = ++_idCount;
    std::list<int> wordIds;
    const int rows = descriptors.rows;
    int nq = nQuantized < 0 || nQuantized > rows ? rows : nQuantized;
    if (rows) {
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
    _signatures[id] = std::vector<int>(wordIds.begin(), wordIds.end());
    outIds.assign(wordIds.begin(), wordIds.end());
    return id;
}

int MemoryHip::addSignature(const std::vector<int>& wordIds, int id) {
    if (id == 0) id = ++_idCount;
    if (_signatures.count(id)) return 0;
    for (size_t k = 0; k < wordIds.size(); ++k) if (wordIds[k] > 0) _vwd->addWordRef(wordIds[k], id);
    _signatures[id] = wordIds;
    if (id > _idCount) _idCount = id;
    return id;
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
    std::set<int> keys(it->second.begin(), it->second.end());   // uUniqueKeys (Memory.cpp:6885)
    for (std::set<int>::iterator k = keys.begin(); k != keys.end(); ++k) _vwd->removeAllWordRef(*k, signatureId);
    _dbNi[signatureId] = (int)it->second.size();
    _signatures.erase(it);
}

std::vector<int> MemoryHip::signatureIds() const {
    std::vector<int> v;
    for (std::map<int, std::vector<int> >::const_iterator i = _signatures.begin(); i != _signatures.end(); ++i) v.push_back(i->first);
    return v;
}

std::map<int, float> MemoryHip::computeLikelihood(const std::list<int>& wordIds, const std::list<int>& ids) {
    const float N = (float)_signatures.size();   // Memory.cpp:2248: every signature in memory, not only `ids`
    return _vwd->computeLikelihood(wordIds, ids, N, [this](int s) { return this->getNi(s); });
}

std::map<int, float> MemoryHip::computeLikelihood(int signatureId, const std::list<int>& ids) {
    std::map<int, std::vector<int> >::const_iterator it = _signatures.find(signatureId);
    if (it == _signatures.end()) return std::map<int, float>();   // "The signature is null" (Memory.cpp:2222)
    return computeLikelihood(std::list<int>(it->second.begin(), it->second.end()), ids);
}

}  // namespace rtabmap_amd
