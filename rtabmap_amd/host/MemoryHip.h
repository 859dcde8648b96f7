// MemoryHip.h -- the hot-path subset of rtabmap::Memory that drives VWDictionary per frame, over VWDictionaryHip.
//
// Mirrors (reference corelib/src/Memory.cpp): preUpdate :1004-1016 + cleanUnusedWords :6899-6920, the quantisation glue
// of createSignature :5941-6059 (features not sent to quantisation get ids -1,-2,.. and still count in ni),
// getNi :4955-4968, disableWordsRef :6877-6897 (WM -> LTM transfer) and computeLikelihood :2177-2292 (TF-IDF branch).
// For the Bayes filter it also keeps what BayesFilter asks a Memory: the short-term / working memory
// split (addSignatureToStm :1146-1230, the transfer loop of update() :1112-1135, moveSignatureToWMFromSTM :1442), neighbour and
// loop-closure links between signatures, and getNeighborsId :1703-1893 restricted to the arguments BayesFilter passes
// (BayesFilter.cpp:329: maxCheckedInDatabase = 0, loop closures stay on the margin of the node they leave from).
// Everything else of Memory (poses, database, sensors, rehearsal) is out of scope.
#pragma once
#include <list>
#include <map>
#include <set>
#include <string>
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
    // Memory::loadDataFromDb (Memory.cpp:392-480) from a RTAB-Map database file: the nodes of the last saved state (or all of them) become
    // signatures of the working memory, the dictionary is loaded -- an incremental one only with the words those signatures reference
    // (DBDriver::loadWords), a fixed one whole (DBDriver::load) --, indexed by ONE update(), the references added, and the device
    // receives them with ONE bulk registration (DbLoaderHip.h).  Returns the number of signatures loaded, -1 on error (lastError()).
    // A signature that references a word the database does not hold makes the load fail (the reference would rebuild the dictionary
    // from the nodes' own descriptors, Memory.cpp:481-565: out of scope).
    int loadDataFromDb(const std::string& path, bool lastStateOnly = true);
    void forget(int signatureId);                   // moveToTrash -> disableWordsRef; the node leaves _signatures
    int getNi(int signatureId) const;
    size_t signaturesSize() const { return _signatures.size(); }
    std::vector<int> signatureIds() const;
    std::map<int, float> computeLikelihood(int signatureId, const std::list<int>& ids);
    std::map<int, float> computeLikelihood(const std::list<int>& wordIds, const std::list<int>& ids);
    // The device-resident frame (round 5).  update() quantises, registers the signature, appends its new words to the device vocabulary
    // and scores it against every registered signature in ONE device call (VWDictionaryHip::addNewWordsAndScore -> lcd_frame_host); the
    // computeLikelihood(signatureId, ids) that Rtabmap::process makes next (Rtabmap.cpp:2117) is answered from that result when nothing
    // changed in between (same N, same references) -- otherwise, and for any other signature, it runs lcd_likelihood as before.
    // setDeviceFrames(false) gives the call-by-call path of rounds 1-4 back (lcd_quantize, lcd_sig_add, lcd_likelihood: five
    // synchronisations per frame).  Results are the same (tests run both).
    void setDeviceFrames(bool on) { _deviceFrames = on; _likeSig = 0; }
    bool deviceFrames() const { return _deviceFrames; }
    // the same answer into a caller-owned map: entries whose keys are already there are overwritten in place (Rtabmap asks for nearly the
    // same ids frame after frame: a std::map of 100 000 fresh nodes costs milliseconds of allocation), others inserted, keys not in
    // `ids` erased
    void computeLikelihood(int signatureId, const std::list<int>& ids, std::map<int, float>& likelihood);
    // ... and as two parallel vectors over EVERY signature registered on the device, ascending id (what a caller that feeds
    // adjustLikelihood / the Bayes filter from flat arrays wants).  false: no result of update() is at hand for that signature.
    bool computeLikelihoodFlat(int signatureId, std::vector<int>& sigIds, std::vector<float>& values);

    // ---- what BayesFilter::computePosterior asks
    enum LinkType { kNeighbor = 0, kGlobalClosure = 1 };   // Link::Type (Link.h:42-56), the two kinds this subset creates
    static const int kIdVirtual;                           // -1 (Memory.cpp:71)
    // loop-closure link between two signatures in memory (Memory::addLink :3500-3590: refused when either is missing, when
    // from == to or when the two are already linked)
    bool addLink(int from, int to, LinkType type = kGlobalClosure);
    std::map<int, int> getNeighborsId(int signatureId, int maxGraphDepth) const;
    bool isInSTM(int signatureId) const { return _stMem.find(signatureId) != _stMem.end(); }
    bool isInWM(int signatureId) const { return _workingMem.find(signatureId) != _workingMem.end(); }
    const std::set<int>& getStMem() const { return _stMem; }
    const std::set<int>& getWorkingMem() const { return _workingMem; }   // kIdVirtual included, as in the reference
    int getMaxStMemSize() const { return _maxStMemSize; }
    // every signature's references are on the device (the Bayes filter works on registered signatures)
    bool flushReferences() { return _vwd->flushReferences([this](int s) { return this->getNi(s); }); }
    // the same after many addSignature calls (a memory replayed without a database): ONE bulk registration
    bool flushReferencesBulk() { return _vwd->flushReferencesBulk([this](int s) { return this->getNi(s); }); }
    // The statistics of the last update() / computeLikelihood() / forget() under the reference's names (Statistics.h:178,189,190,200,201,
    // 209-212; emitted at Memory.cpp:5931,6062 and Rtabmap.cpp:4357,4367): "TimingMem/Pre_update/ms", "TimingMem/Joining_dictionary_update/ms"
    // (0: update() runs in line, there is no PreUpdateThread to join), "TimingMem/Add_new_words/ms", "Timing/Likelihood_computation/ms",
    // "Timing/Forgetting/ms", "Keypoint/Dictionary_size/words", "Keypoint/Current_frame/words", "Keypoint/Indexed_words/words",
    // "Keypoint/Index_memory_usage/KB" (the engine's HBM, lcd_stats.bytes_device -- the reference reports FLANN's memory here).
    // The three Keypoint/ engine figures are refreshed by refreshEngineStatistics() (lcd_get_stats synchronises the stream: not per frame).
    const std::map<std::string, float>& getStatistics() const { return _stats; }
    void refreshEngineStatistics();
    const std::string& lastError() const { return _vwd->lastError(); }
    const std::string& loadError() const { return _loadError; }        // of the last loadDataFromDb that returned -1

private:
    void preUpdate();
    void cleanUnusedWords();
    VWDictionaryHip* _vwd;
    std::string _loadError;                         // of the last loadDataFromDb
    std::map<int, std::vector<int> > _signatures;   // id -> words in keypoint order (Signature::getWords keys)
    std::map<int, int> _dbNi;                       // DBDriver::getInvertedIndexNi of transferred nodes
    int _idCount;
    void addSignatureToStm(int id);
    std::set<int> _stMem, _workingMem;
    int _maxStMemSize;                              // Mem/STMSize (Parameters.h: 10)
    std::map<int, std::map<int, LinkType> > _links; // Signature::getLinks(): id -> (other id -> type)
    // the likelihood update() brought back with the frame: by device slot, for signature _likeSig (0: none / stale)
    bool _deviceFrames;
    int _likeSig;
    std::vector<float> _likeSlots;
    std::vector<std::pair<int, float> > _likeSorted;   // (signature id, value) ascending id, built on demand from _likeSlots
    bool _likeSortedValid;
    const std::vector<std::pair<int, float> >& sortedLikelihood();
    std::map<std::string, float> _stats;
};

}  // namespace rtabmap_amd
