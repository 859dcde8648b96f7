// MemoryHip.h -- the hot-path subset of rtabmap::Memory that drives VWDictionary per frame, over VWDictionaryHip.
//
// Mirrors (reference corelib/src/Memory.cpp): preUpdate :1004-1016 + cleanUnusedWords :6899-6920, the quantisation glue
// of createSignature :5941-6059 (features not sent to quantisation get ids -1,-2,.. and still count in ni),
// getNi :4955-4968, disableWordsRef :6877-6897 (WM -> LTM transfer) and computeLikelihood :2177-2292 (TF-IDF branch).
// Everything else of Memory (graph, database, sensors) is out of scope.
#pragma once
#include <list>
#include <map>
#include <vector>

#include "VWDictionaryHip.h"

namespace rtabmap_amd {

class MemoryHip {
public:
    explicit MemoryHip(const ParametersMap& parameters = ParametersMap(), int device = 0);
    ~MemoryHip();
    VWDictionaryHip* getVWDictionary() { return _vwd; }

    // Memory::update -> createSignature: quantise rows [0, nQuantized) of `descriptors` (all if < 0); returns the new
    // signature id, wordIds receives one id per descriptor (unquantised ones -1,-2,..)
    int update(const Mat& descriptors, int nQuantized, std::vector<int>& wordIds);
    // a signature given directly by its word ids (database replay, Memory.cpp:447-480)
    int addSignature(const std::vector<int>& wordIds, int id = 0);
    void forget(int signatureId);                   // moveToTrash -> disableWordsRef; the node leaves _signatures
    int getNi(int signatureId) const;
    size_t signaturesSize() const { return _signatures.size(); }
    std::vector<int> signatureIds() const;
    std::map<int, float> computeLikelihood(int signatureId, const std::list<int>& ids);
    std::map<int, float> computeLikelihood(const std::list<int>& wordIds, const std::list<int>& ids);

private:
    void preUpdate();
    void cleanUnusedWords();
    VWDictionaryHip* _vwd;
    std::map<int, std::vector<int> > _signatures;   // id -> words in keypoint order (Signature::getWords keys)
    std::map<int, int> _dbNi;                       // DBDriver::getInvertedIndexNi of transferred nodes
    int _idCount;
};

}  // namespace rtabmap_amd
