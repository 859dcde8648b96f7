// VWDictionaryHip.h -- host-side mirror of rtabmap::VWDictionary / VisualWord over the C-ABI of include/lcd.h.
//
// The reference is compiled C++ (corelib/include/rtabmap/core/VWDictionary.h:46-156, VisualWord.h:38-64) and its
// toolchain dependencies (OpenCV) are absent from this image, so this is a stand-alone C++ class with the SAME method
// names, argument meaning and error behaviour as the reference class, on a minimal matrix type instead of cv::Mat.
// It is what a maintainer's `Kp/NNStrategy = 5 (kNNBruteForceHIP)` branch inside the real VWDictionary.cpp would do
// (see INTEGRATION.md): the host keeps the authoritative word / reference bookkeeping exactly like the reference, the
// three hot spots call the device:
//     update()        -> lcd_vocab_append | lcd_vocab_remove + lcd_vocab_rebuild      (VWDictionary.cpp:475-701)
//     addNewWords()   -> lcd_quantize (2-NN + same-frame words + NNDR on the device)  (VWDictionary.cpp:913-1229)
//     findNN()        -> lcd_find_nn                                                  (VWDictionary.cpp:1273-1552)
// and Memory::computeLikelihood's TF-IDF branch (Memory.cpp:2215-2291) -> lcd_likelihood through computeLikelihood().
// No search or scoring arithmetic is done on the host: if the engine cannot be created every call fails loudly.
#pragma once
#include <cstdlib>
#include <functional>
#include <list>
#include <map>
#include <set>
#include <locale>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/lcd.h"

namespace rtabmap_amd {

// cv::Mat stand-in: row-major, CV_32F (type 5) or CV_8U (type 0) single channel
enum { MAT_8U = 0, MAT_32F = 5 };
struct Mat {
    int rows = 0, cols = 0, type_ = -1;
    std::vector<unsigned char> data;
    Mat() {}
    Mat(int r, int c, int t, const void* src = nullptr);
    int type() const { return type_; }
    bool empty() const { return rows == 0 || cols == 0; }
    size_t elemSize() const { return type_ == MAT_32F ? 4 : 1; }
    size_t rowBytes() const { return (size_t)cols * elemSize(); }
    const unsigned char* ptr(int r) const { return data.data() + (size_t)r * rowBytes(); }
    Mat row(int r) const { return Mat(1, cols, type_, ptr(r)); }
};

typedef std::map<std::string, std::string> ParametersMap;
// uStr2Float (UConversion.cpp:138): the decimal mark may be '.' or ','
// and the number is read in the classic ("C") locale whatever LC_NUMERIC the process runs under, as the reference's imbued stream does
inline float uStr2Float(const std::string& s) {
    std::string v = s;
    for (size_t i = 0; i < v.size(); ++i) if (v[i] == ',') v[i] = '.';
    std::istringstream in(v);
    in.imbue(std::locale::classic());
    float value = 0.0f;                     // extracted as a float, like the reference: out of range reads as +-FLT_MAX, not as infinity
    in >> value;
    return value;
}

// reference VisualWord.h:38-64, VisualWord.cpp:36-70.  Same interface, same answers; the container behind it differs (round 6): the postings
// (signature id, occurrences) live in a vector sorted by signature id -- 8 bytes each instead of a 48-byte red-black node, the newest signature
// appended at the back, the one Memory forgets taken from the front (an offset moves; the dead prefix is dropped when it outgrows the live part)
// -- and the std::map<int, int> the reference's getReferences() returns is built from it ON DEMAND and kept until the next change.  The hot
// path (addRef per descriptor, removeAllRef per forgotten signature, "any reference left?") never builds it: at 100 000 signatures the maps of
// the popular words held ~10^5 nodes each and their upkeep was ~1 ms per frame of the mirror's time.
class VisualWord {
public:
    VisualWord(int id, const Mat& descriptor, int signatureId = 0);
    void addRef(int signatureId);
    int removeAllRef(int signatureId);
    int getTotalReferences() const { return _totalReferences; }
    int id() const { return _id; }
    const Mat& getDescriptor() const { return _descriptor; }
    const std::map<int, int>& getReferences() const;                  // materialised on demand (see above)
    size_t getReferencesCount() const { return _refs.size() - _head; }   // getReferences().size() without the map
    bool isSaved() const { return _saved; }
    void setSaved(bool saved) { _saved = saved; }
private:
    int _id;
    Mat _descriptor;
    bool _saved;
    int _totalReferences;
    std::vector<std::pair<int, int> > _refs;   // (signature id , occurrence in the signature), ascending id, live from _head on
    size_t _head;
    mutable std::map<int, int> _references;    // getReferences()'s answer, valid while _mapValid
    mutable bool _mapValid;
    std::vector<std::pair<int, int> >::iterator findRef(int signatureId);
};

class VWDictionaryHip {
public:
    enum NNStrategy { kNNFlannNaive, kNNFlannKdTree, kNNFlannLSH, kNNBruteForce, kNNBruteForceGPU, kNNBruteForceHIP, kNNUndef };
    static const int ID_START;     // 1
    static const int ID_INVALID;   // 0

    explicit VWDictionaryHip(const ParametersMap& parameters = ParametersMap(), int device = 0);
    virtual ~VWDictionaryHip();

    virtual void parseParameters(const ParametersMap& parameters);   // Kp/* keys of Parameters.h:243-266
    virtual void update();
    virtual std::list<int> addNewWords(const Mat& descriptors, int signatureId);
    virtual void addWord(VisualWord* vw);   // takes ownership

    std::vector<int> findNN(const std::list<VisualWord*>& vws) const;
    std::vector<int> findNN(const Mat& descriptors) const;

    bool addWordRef(int wordId, int signatureId);
    void removeAllWordRef(int wordId, int signatureId);
    void removeAllWordRefs(const std::set<int>& wordIds, int signatureId);   // the same for every unique word of a signature (Memory::disableWordsRef)
    const VisualWord* getWord(int id) const;
    VisualWord* getUnusedWord(int id) const;
    void setLastWordId(int id) { _lastWordId = id; }
    const std::map<int, VisualWord*>& getVisualWords() const { return _visualWords; }
    float getNndrRatio() const { return _nndrRatio; }
    bool isNewWordsComparedTogether() const { return _newWordsComparedTogether; }
    unsigned int getNotIndexedWordsCount() const { return (unsigned int)_notIndexedWords.size(); }
    int getLastIndexedWordId() const;
    int getTotalActiveReferences() const { return _totalActiveReferences; }
    unsigned int getIndexedWordsCount() const { return (unsigned int)_mapIndexId.size(); }
    bool isIncremental() const { return _incrementalDictionary; }
    void setIncrementalDictionary();
    void setFixedDictionary(const std::string& dictionaryPath);   // text format only (VWDictionary.cpp:181-257)
    void exportDictionary(const char* fileNameReferences, const char* fileNameDescriptors) const;

    void clear(bool printWarningsIfNotEmpty = true);
    std::vector<VisualWord*> getUnusedWords() const;
    std::vector<int> getUnusedWordIds() const;
    unsigned int getUnusedWordsSize() const { return (unsigned int)_unusedWords.size(); }
    void removeWords(const std::vector<VisualWord*>& words);   // caller must delete the words
    void deleteUnusedWords();

    // ---- Memory::computeLikelihood(signature, ids), TF-IDF branch (Memory.cpp:2215-2291).
    // wordIds: the keys of signature->getWords(); N = Memory::getSignatures().size(); getNi = Memory::getNi.
    std::map<int, float> computeLikelihood(const std::list<int>& wordIds, const std::list<int>& ids, float N,
                                           const std::function<int(int)>& getNi);

    // ---- Memory::update's quantisation AND Memory::computeLikelihood of the new signature in ONE device call (lcd_frame_host, ABI v5):
    // addNewWords(descriptors, signatureId) with the bookkeeping of :1162-1219 -- plus, on the device, the signature's references (no
    // flushReferences for it later), the words it creates appended to the device vocabulary (the append branch of the update() that
    // Memory::preUpdate runs in front of the NEXT frame, :571-609: that update() then only moves them to the indexed words) and the
    // TF-IDF likelihood of the new signature against every registered signature, by device slot (slotSignatures()).  One
    // synchronisation instead of five.  N / getNi as for computeLikelihood (N counts the new signature).
    // Returns false -- nothing changed -- when the fast path does not apply (fixed dictionary, words waiting for update(), rows the
    // device pads, no engine): the caller runs addNewWords() + computeLikelihood() instead.
    bool addNewWordsAndScore(const Mat& descriptors, int signatureId, float N, const std::function<int(int)>& getNi, std::list<int>& wordIds,
                             std::vector<float>& likelihoodBySlot);
    // device slot -> signature id (0: the slot's signature was removed); slots are handed out in registration order
    const std::vector<int>& slotSignatures() const { return _slotSig; }
    // host time spent INSIDE lcd_frame_host by addNewWordsAndScore (copies, launches, the one synchronisation) and the number of calls:
    // what is left of a caller's update() time is the mirror's own std::map bookkeeping
    void fastFrameStats(long long* deviceCallNs, long long* calls) const { *deviceCallNs = _fastDeviceNs; *calls = _fastCalls; }

    // send the references added / removed since the last call to the device's inverted index (computeLikelihood does it itself)
    bool flushReferences(const std::function<int(int)>& getNi);
    // the same for a memory that has just been loaded (Memory::loadDataFromDb, Memory.cpp:447-480): every signature that is not on the
    // device yet goes there with ONE lcd_sig_add_bulk instead of one lcd_sig_add each; what is left (signatures whose references
    // changed after they were registered) is flushed as usual
    bool flushReferencesBulk(const std::function<int(int)>& getNi);

    // The engine is a cache of this object's state (failure row; the reference repairs its dictionary on load,
    // Memory.cpp:481-565, and rebuilds a bad FLANN index, VWDictionary.cpp:835-838): after a device fault -- or whenever the caller
    // wants a fresh device state -- destroy the handle, create a new one and replay the indexed words in row order and every
    // signature's references from the host maps.  Returns false (lastError) when no device can be opened.
    bool rebuildEngine();

    // ids of the indexed rows in device row order (the tie-break order); for tests
    std::vector<int> getIndexedWordIds() const;
    bool isAvailable() const { return _engine != nullptr; }
    const std::string& lastError() const { return _lastError; }
    lcd_engine* engine() const { return _engine; }

protected:
    int getNextId() { return ++_lastWordId; }
    bool ensureEngine(int type, int cols) const;
    void markDirty(int signatureId);

protected:
    std::map<int, VisualWord*> _visualWords;
    int _totalActiveReferences;

private:
    bool _incrementalDictionary;
    float _nndrRatio;
    std::string _dictionaryPath, _newDictionaryPath;
    bool _newWordsComparedTogether;
    int _lastWordId;
    NNStrategy _strategy;
    std::map<int, int> _mapIndexId, _mapIdIndex;
    std::map<int, VisualWord*> _unusedWords;
    std::set<int> _notIndexedWords;
    std::set<int> _removedIndexedWords;
    // device side
    int _device;
    mutable lcd_engine* _engine;
    mutable int _engineType, _engineCols;
    mutable std::string _lastError;
    // signature-granular mirror of the references for the device inverted index
    std::map<int, std::vector<int> > _sigWords;   // signature -> word ids referenced (one entry per addWordRef)
    std::set<int> _dirtySigs;
    std::set<int> _deviceSigs;
    // mirror of the engine's slot table (lcd.h: slots in registration order, never reused)
    // id -> word, beside _visualWords (the reference's std::map stays the authority and the iteration order; this is the O(1) way to it
    // for the per-descriptor bookkeeping of a frame)
    std::vector<VisualWord*> _byId;
    void indexWord(VisualWord* vw) { if (vw->id() >= 0) { if ((size_t)vw->id() >= _byId.size()) _byId.resize((size_t)vw->id() + 1 + _byId.size() / 2, nullptr); _byId[(size_t)vw->id()] = vw; } }
    VisualWord* lookupWord(int id) const {
        if (id >= 0 && (size_t)id < _byId.size()) return _byId[(size_t)id];
        std::map<int, VisualWord*>::const_iterator it = _visualWords.find(id);
        return it == _visualWords.end() ? nullptr : it->second;
    }
    std::vector<int> _slotSig;
    std::map<int, int> _sigSlot;
    void slotAdd(int signatureId) { _sigSlot[signatureId] = (int)_slotSig.size(); _slotSig.push_back(signatureId); }
    void slotRetire(int signatureId) { std::map<int, int>::iterator i = _sigSlot.find(signatureId); if (i != _sigSlot.end()) { _slotSig[(size_t)i->second] = 0; _sigSlot.erase(i); } }
    // words a frame created that are ALREADY rows of the device vocabulary (addNewWordsAndScore with append_new_words): still in
    // _notIndexedWords on the host until update() runs
    std::set<int> _deviceRows;
    long long _fastDeviceNs = 0, _fastCalls = 0;
};

}  // namespace rtabmap_amd
