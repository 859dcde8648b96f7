// oracle/lcd_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A from-scratch CPU restatement of the loop-closure hot path of introlab/rtabmap (reference v0.23.8), written
// to be the CHECKER for the HIP engine in rtabmap_amd/.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product (rtabmap_amd/csrc, liblcd_hip.so) never links, imports or
// calls it and has no CPU fallback.
//
// What is restated, and from where (paths relative to /root/reference):
//   * distance functors           corelib/src/rtflann/algorithms/dist.h:150-177 (L2), :211-238 (L1), :555-579 (Hamming)
//   * exact k-NN + tie-break      corelib/src/rtflann/algorithms/linear_index.h:129-144, util/result_set.h:151-171
//   * VisualWord                  corelib/src/VisualWord.cpp:51-70
//   * VWDictionary                corelib/src/VWDictionary.cpp: update :475-701, addWordRef :880, removeAllWordRef :899,
//                                 addNewWords :913-1229, findNN :1273-1552, addWord :1554, removeWords :1595,
//                                 fixed-dictionary text reader :181-257, exportDictionary :1619-1696
//   * Memory::computeLikelihood   corelib/src/Memory.cpp:2215-2291 (TF-IDF branch), getNi :4955-4968,
//                                 preUpdate :1004-1016, cleanUnusedWords :6899-6920, disableWordsRef :6877-6897,
//                                 createSignature quantisation glue :5941-6059 (ids -1,-2,.. for unquantised features)
//   * Rtabmap::adjustLikelihood   corelib/src/Rtabmap.cpp:5691-5760 with uMean/uVariance (utilite UMath.h:419-432, 512-526)
//   * BayesFilter                 corelib/src/BayesFilter.cpp:77-122, 145-270, 273-420, 434-500, 709-736; hypothesis
//                                 selection corelib/src/Rtabmap.cpp:2147-2158
//
// Pinning (see tests/test_oracle_*.py): the distance functors and the exact 2-NN are checked against the reference's
// own vendored rtflann compiled in place (oracle/_ref/librtflann_ref.so); the TF-IDF restatement is checked against
// the reference's MATLAB known-answer vectors (archive/2010-LoopClosure/Tests/TestComputeLikelihood.m:23-27,
// TestUpdateCommonSignature.m:24-27) via tests/golden/loopclosure2010.npz.  The cv::BFMatcher side searches
// (VWDictionary.cpp:1027,1143,1354,1449) are OpenCV code that is NOT in the reference tree: they are restated as
// exact scans with lowest-row tie-break and are "parity unpinned" (SURVEY.md section 8c).
//
// The containers are the reference's own (std::map / std::set / std::multimap) so iteration orders and the
// equal-key insertion order of std::multimap (VWDictionary.cpp:1091) are reproduced by construction.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <list>
#include <map>
#include <set>
#include <locale>
#include <sstream>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

enum { T_F32 = 0, T_U8 = 1 };
// Kp/NNStrategy values (VWDictionary.h:49-55)
enum { kNNFlannNaive = 0, kNNFlannKdTree = 1, kNNFlannLSH = 2, kNNBruteForce = 3, kNNBruteForceGPU = 4 };
enum { METRIC_L2 = 0, METRIC_HAMMING = 1, METRIC_L1 = 2, METRIC_HAMMING_CV = 3 };
static inline bool metric_is_u8(int metric) { return metric == METRIC_HAMMING || metric == METRIC_HAMMING_CV; }

// ---------------------------------------------------------------------------------------------- distances
// rtflann::L2<float>::operator()  dist.h:150-177 (worst_dist = -1: no early exit)
static inline float dist_l2(const float* a, const float* b, size_t size) {
    float result = 0.0f, d0, d1, d2, d3;
    const float* last = a + size;
    const float* lastgroup = last - 3;
    while (a < lastgroup) {
        d0 = a[0] - b[0]; d1 = a[1] - b[1]; d2 = a[2] - b[2]; d3 = a[3] - b[3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        a += 4; b += 4;
    }
    while (a < last) { d0 = *a++ - *b++; result += d0 * d0; }
    return result;
}
// rtflann::L1<float>::operator()  dist.h:211-238
static inline float dist_l1(const float* a, const float* b, size_t size) {
    float result = 0.0f, d0, d1, d2, d3;
    const float* last = a + size;
    const float* lastgroup = last - 3;
    while (a < lastgroup) {
        d0 = std::fabs(a[0] - b[0]); d1 = std::fabs(a[1] - b[1]);
        d2 = std::fabs(a[2] - b[2]); d3 = std::fabs(a[3] - b[3]);
        result += d0 + d1 + d2 + d3;
        a += 4; b += 4;
    }
    while (a < last) { d0 = std::fabs(*a++ - *b++); result += d0; }
    return result;
}
// rtflann::Hamming<unsigned char>::operator()  dist.h:555-579 (64-bit words, SWAR popcount)
static inline unsigned popcnt64(uint64_t n) {
    n -= ((n >> 1) & 0x5555555555555555ULL);
    n = (n & 0x3333333333333333ULL) + ((n >> 2) & 0x3333333333333333ULL);
    return (unsigned)((((n + (n >> 4)) & 0x0f0f0f0f0f0f0f0fULL) * 0x0101010101010101ULL) >> 56);
}
static inline unsigned dist_hamming(const unsigned char* a, const unsigned char* b, size_t size) {
    unsigned result = 0;
    size_t words = size / 8;   // like the reference, trailing bytes beyond a multiple of 8 are ignored
    for (size_t i = 0; i < words; ++i) {
        uint64_t x, y;
        std::memcpy(&x, a + 8 * i, 8); std::memcpy(&y, b + 8 * i, 8);
        result += popcnt64(x ^ y);
    }
    return result;
}
// cv::NORM_HAMMING as cv::BFMatcher / cv::cuda::DescriptorMatcher compute it (VWDictionary.cpp:1027-1075, :1143): the bits of
// EVERY byte.  It differs from rtflann::Hamming above only for descriptor sizes that are not a multiple of 8 bytes (AKAZE's
// 61): the reference's FLANN strategies then ignore the tail, its brute-force strategies and the same-frame comparison do not.
// OpenCV's source is not in the reference tree: this is the documented definition of NORM_HAMMING.
static inline unsigned dist_hamming_cv(const unsigned char* a, const unsigned char* b, size_t size) {
    unsigned result = dist_hamming(a, b, size);
    for (size_t i = size / 8 * 8; i < size; ++i) result += popcnt64((uint64_t)(a[i] ^ b[i]));
    return result;
}
static inline float dist_any(int metric, const void* a, const void* b, size_t cols) {
    if (metric == METRIC_HAMMING) return (float)dist_hamming((const unsigned char*)a, (const unsigned char*)b, cols);
    if (metric == METRIC_HAMMING_CV) return (float)dist_hamming_cv((const unsigned char*)a, (const unsigned char*)b, cols);
    if (metric == METRIC_L1) return dist_l1((const float*)a, (const float*)b, cols);
    return dist_l2((const float*)a, (const float*)b, cols);
}

// KNNSimpleResultSet::addPoint  result_set.h:151-171 (FLANN_FIRST_MATCH undefined: an equal distance never displaces
// an earlier element => lower scan position wins ties)
struct Top2 {
    float d[2]; long idx[2]; int count;
    Top2() { d[0] = d[1] = std::numeric_limits<float>::max(); idx[0] = idx[1] = -1; count = 0; }
    inline void add(float dist, long index) {
        float worst = d[1];
        if (dist >= worst) return;
        if (count < 2) ++count;
        int i;
        for (i = count - 1; i > 0; --i) {
            if (d[i - 1] > dist) { d[i] = d[i - 1]; idx[i] = idx[i - 1]; } else break;
        }
        d[i] = dist; idx[i] = index;
    }
};

// ---------------------------------------------------------------------------------------------- matrix
struct Mat {           // the tiny stand-in for cv::Mat: row-major, CV_32F or CV_8U
    int rows, cols, type;
    std::vector<unsigned char> data;
    Mat() : rows(0), cols(0), type(-1) {}
    size_t elem() const { return type == T_F32 ? 4 : 1; }
    size_t rowBytes() const { return (size_t)cols * elem(); }
    const unsigned char* row(int r) const { return data.data() + (size_t)r * rowBytes(); }
    bool empty() const { return rows == 0; }
    void push_back(const unsigned char* r, int c, int t) {
        if (rows == 0) { cols = c; type = t; }
        data.insert(data.end(), r, r + (size_t)c * (t == T_F32 ? 4 : 1));
        ++rows;
    }
    void clear() { rows = 0; cols = 0; type = -1; data.clear(); }
};

// ---------------------------------------------------------------------------------------------- VisualWord
struct VisualWord {    // VisualWord.h:38-64, VisualWord.cpp:51-70
    int id; int cols; int type;
    std::vector<unsigned char> desc;
    std::map<int, int> references;   // <signature id, occurrence in the signature>
    int totalReferences;
    VisualWord(int id_, const unsigned char* d, int c, int t, int signatureId = 0)
        : id(id_), cols(c), type(t), desc(d, d + (size_t)c * (t == T_F32 ? 4 : 1)), totalReferences(0) {
        if (signatureId) addRef(signatureId);
    }
    void addRef(int signatureId) {
        std::map<int, int>::iterator it = references.find(signatureId);
        if (it != references.end()) it->second += 1;
        else references.insert(references.end(), std::make_pair(signatureId, 1));
        ++totalReferences;
    }
    int removeAllRef(int signatureId) {   // uTake(_references, signatureId, 0)
        int removed = 0;
        std::map<int, int>::iterator it = references.find(signatureId);
        if (it != references.end()) { removed = it->second; references.erase(it); }
        totalReferences -= removed;
        return removed;
    }
};

// exact k<=2 scan == cv::BFMatcher::knnMatch restated (OpenCV source absent: lowest-row tie-break assumed)
static void scan_top2(int metric, const Mat& train, const unsigned char* q, int k, Top2& out,
                      const std::vector<char>* removed = 0) {
    Top2 t;
    if (k == 1) {   // a 1-NN keeps the first minimum
        float best = std::numeric_limits<float>::max(); long bi = -1;
        for (int r = 0; r < train.rows; ++r) {
            if (removed && (*removed)[r]) continue;
            float d = dist_any(metric, train.row(r), q, train.cols);
            if (d < best) { best = d; bi = r; }
        }
        if (bi >= 0) { t.d[0] = best; t.idx[0] = bi; t.count = 1; }
    } else {
        for (int r = 0; r < train.rows; ++r) {
            if (removed && (*removed)[r]) continue;
            t.add(dist_any(metric, train.row(r), q, train.cols), r);
        }
    }
    out = t;
}

// ---------------------------------------------------------------------------------------------- VWDictionary
struct VWDictionary {
    // parameters (Parameters.h:243-266)
    int strategy; bool incrementalDictionary; bool incrementalFlann; float nndrRatio; bool newWordsComparedTogether;
    // state (VWDictionary.h:124-156)
    std::map<int, VisualWord*> visualWords;
    std::map<int, VisualWord*> unusedWords;
    std::set<int> notIndexedWords;
    std::set<int> removedIndexedWords;
    int totalActiveReferences;
    int lastWordId;
    bool useDistanceL1;
    // search structure.  For strategies >= kNNBruteForce this is _dataTree; for kNNFlannNaive it models the rtflann
    // LINEAR index: rows in insertion order, a removed flag per row (nn_index.h removePoint), index == row position.
    Mat dataTree;
    std::vector<char> rowRemoved;
    bool flannBuilt;
    std::map<int, int> mapIndexId, mapIdIndex;
    std::string lastError;

    VWDictionary() : strategy(kNNBruteForce), incrementalDictionary(true), incrementalFlann(true), nndrRatio(0.8f),
                     newWordsComparedTogether(true), totalActiveReferences(0), lastWordId(0), useDistanceL1(false),
                     flannBuilt(false) {}
    ~VWDictionary() { clear(); }

    bool isFlann() const { return strategy < kNNBruteForce; }
    int metricFor(int type) const {   // VWDictionary.cpp:1027, 1143: HAMMING for CV_8U else (L1 if useDistanceL1_) L2SQR
        if (type == T_U8) return METRIC_HAMMING_CV;               // cv::BFMatcher(NORM_HAMMING)
        return useDistanceL1 ? METRIC_L1 : METRIC_L2;
    }

    void clear() {   // :843-873
        for (std::map<int, VisualWord*>::iterator i = visualWords.begin(); i != visualWords.end(); ++i) delete i->second;
        visualWords.clear(); notIndexedWords.clear(); removedIndexedWords.clear(); unusedWords.clear();
        totalActiveReferences = 0; lastWordId = 0; dataTree.clear(); rowRemoved.clear(); flannBuilt = false;
        mapIndexId.clear(); mapIdIndex.clear(); useDistanceL1 = false;
    }

    bool addWordRef(int wordId, int signatureId) {   // :880-897
        std::map<int, VisualWord*>::iterator it = visualWords.find(wordId);
        if (it != visualWords.end()) {
            it->second->addRef(signatureId);
            totalActiveReferences += 1;
            unusedWords.erase(wordId);
            return true;
        }
        return false;
    }
    void removeAllWordRef(int wordId, int signatureId) {   // :899-911
        std::map<int, VisualWord*>::iterator it = visualWords.find(wordId);
        if (it != visualWords.end()) {
            totalActiveReferences -= it->second->removeAllRef(signatureId);
            if (it->second->references.size() == 0) unusedWords.insert(std::make_pair(wordId, it->second));
        }
    }
    void addWord(VisualWord* vw) {   // :1554-1573 (takes ownership)
        if (!vw) return;
        visualWords.insert(visualWords.end(), std::make_pair(vw->id, vw));
        notIndexedWords.insert(notIndexedWords.end(), vw->id);
        if (vw->references.size()) {
            int s = 0;
            for (std::map<int, int>::iterator i = vw->references.begin(); i != vw->references.end(); ++i) s += i->second;
            totalActiveReferences += s;
        } else {
            unusedWords.insert(unusedWords.end(), std::make_pair(vw->id, vw));
        }
        if (lastWordId < vw->id) lastWordId = vw->id;
    }
    void removeWords(const std::vector<int>& ids, bool del) {   // :1595-1607 (+ the caller's delete)
        for (size_t i = 0; i < ids.size(); ++i) {
            std::map<int, VisualWord*>::iterator it = visualWords.find(ids[i]);
            VisualWord* vw = it != visualWords.end() ? it->second : 0;
            visualWords.erase(ids[i]);
            unusedWords.erase(ids[i]);
            if (notIndexedWords.erase(ids[i]) == 0) removedIndexedWords.insert(ids[i]);
            if (del && vw) delete vw;
        }
    }
    std::vector<int> getUnusedWordIds() const {
        std::vector<int> v;
        for (std::map<int, VisualWord*>::const_iterator i = unusedWords.begin(); i != unusedWords.end(); ++i) v.push_back(i->first);
        return v;
    }

    // ------------------------------------------------------------------------------------------ update() :475-701
    void update() {
        if (!incrementalDictionary && !notIndexedWords.size()) return;   // fixed dictionary already indexed (:482-489)
        if (notIndexedWords.size() || visualWords.size() == 0 || removedIndexedWords.size()) {
            bool firstUpdate = removedIndexedWords.empty() && visualWords.size() == notIndexedWords.size();
            if (!firstUpdate && incrementalFlann && strategy < kNNBruteForce && visualWords.size()) {
                // incremental FLANN (:499-570): removePoint, then addPoints one word at a time in ascending id
                for (std::set<int>::iterator it = removedIndexedWords.begin(); it != removedIndexedWords.end(); ++it) {
                    int idx = mapIdIndex.at(*it);
                    rowRemoved[idx] = 1;
                    mapIndexId.erase(idx);
                    mapIdIndex.erase(*it);
                }
                for (std::set<int>::iterator it = notIndexedWords.begin(); it != notIndexedWords.end(); ++it) {
                    VisualWord* w = visualWords.at(*it);
                    if (w->type == T_U8) useDistanceL1 = true;
                    int index = 0;
                    if (!flannBuilt) {   // buildIndex with the first descriptor (:545-552): nextIndex restarts at 0
                        dataTree.clear(); rowRemoved.clear();
                        dataTree.push_back(w->desc.data(), w->cols, w->type); rowRemoved.push_back(0);
                        flannBuilt = true; index = 0;
                    } else {
                        index = dataTree.rows;   // FlannIndex::addPoints returns nextIndex_++ (FlannIndex.cpp:650-656)
                        dataTree.push_back(w->desc.data(), w->cols, w->type); rowRemoved.push_back(0);
                    }
                    mapIndexId.insert(std::make_pair(index, w->id));
                    mapIdIndex.insert(std::make_pair(w->id, index));
                }
            } else if (strategy >= kNNBruteForce && notIndexedWords.size() && removedIndexedWords.size() == 0 &&
                       visualWords.size()) {
                // brute-force append (:571-609)
                int i = dataTree.rows;
                for (std::set<int>::iterator it = notIndexedWords.begin(); it != notIndexedWords.end(); ++it) {
                    VisualWord* w = visualWords.at(*it);
                    dataTree.push_back(w->desc.data(), w->cols, w->type); rowRemoved.push_back(0);
                    mapIndexId.insert(mapIndexId.end(), std::make_pair(i, w->id));
                    mapIdIndex.insert(std::make_pair(w->id, i));
                    ++i;
                }
            } else {
                // full rebuild in ascending word id (:610-690)
                mapIndexId.clear(); mapIdIndex.clear(); dataTree.clear(); rowRemoved.clear(); flannBuilt = false;
                if (visualWords.size()) {
                    if (visualWords.begin()->second->type == T_U8) useDistanceL1 = true;
                    int i = 0;
                    for (std::map<int, VisualWord*>::iterator it = visualWords.begin(); it != visualWords.end(); ++it, ++i) {
                        VisualWord* w = it->second;
                        dataTree.push_back(w->desc.data(), w->cols, w->type); rowRemoved.push_back(0);
                        mapIndexId.insert(mapIndexId.end(), std::make_pair(i, w->id));
                        mapIdIndex.insert(mapIdIndex.end(), std::make_pair(w->id, i));
                    }
                    flannBuilt = isFlann();   // _flannIndex->buildIndex is called for every strategy; only FLANN ones search it
                }
            }
        }
        notIndexedWords.clear();
        removedIndexedWords.clear();
    }

    bool searchable() const {   // :1015 / :1347
        if (isFlann()) return flannBuilt;
        return !dataTree.empty() && dataTree.rows >= 2;
    }
    // the indexed 2-NN: FlannIndex::knnSearch(LINEAR) for strategy 0 (removed rows skipped, linear_index.h:131-137),
    // cv::BFMatcher::knnMatch over _dataTree for strategy >= 3.  Missing neighbours are reported as (idx -1, d -1).
    void indexed2nn(const unsigned char* q, int type, long idx[2], float d[2]) const {
        int metric;
        if (isFlann()) metric = (type == T_U8) ? METRIC_HAMMING : (useDistanceL1 ? METRIC_L1 : METRIC_L2);  // FlannIndex.cpp:727-744
        else metric = (type == T_U8) ? METRIC_HAMMING_CV : METRIC_L2;                                        // :1027 (cv::BFMatcher)
        Top2 t;
        scan_top2(metric, dataTree, q, 2, t, isFlann() ? &rowRemoved : 0);
        for (int j = 0; j < 2; ++j) {
            if (j < t.count) { idx[j] = t.idx[j]; d[j] = t.d[j]; } else { idx[j] = -1; d[j] = -1.0f; }
        }
    }
    int idOfIndex(long index) const {   // uValue(_mapIndexId, index) -> 0 when unknown
        if (index < 0) return 0;
        std::map<int, int>::const_iterator it = mapIndexId.find((int)index);
        return it == mapIndexId.end() ? 0 : it->second;
    }
    bool checkInput(int rows, int cols, int type) {
        if (rows == 0 || cols == 0) { lastError = "Descriptors size is null!"; return false; }                // :920
        if (!incrementalDictionary && visualWords.empty()) { lastError = "fixed dictionary is empty"; return false; }  // :926
        if (visualWords.size()) {
            VisualWord* f = visualWords.begin()->second;
            if (f->cols != cols) { lastError = "descriptor size mismatch"; return false; }                    // :948
            if (f->type != type) { lastError = "descriptor type mismatch"; return false; }                    // :953
        }
        if (dataTree.rows) {
            if (dataTree.cols != cols) { lastError = "descriptor size mismatch (index)"; return false; }      // :986
            if (dataTree.type != type) { lastError = "descriptor type mismatch (index)"; return false; }      // :992
        }
        return true;
    }

    // ------------------------------------------------------------------------------------------ addNewWords :913-1229
    // returns false on the reference's "log error, return empty list" paths
    bool addNewWords(const unsigned char* desc, int rows, int cols, int type, int signatureId, std::list<int>& wordIds) {
        wordIds.clear();
        if (strategy == kNNFlannKdTree || strategy == kNNFlannLSH) { lastError = "strategy not restated (approximate)"; return false; }
        if (!checkInput(rows, cols, type)) return false;
        if (type == T_U8) useDistanceL1 = true;   // :963
        const size_t rb = (size_t)cols * (type == T_F32 ? 4 : 1);
        Mat newWords; std::vector<int> newWordsId;
        bool searched = searchable();
        for (int i = 0; i < rows; ++i) {
            const unsigned char* q = desc + (size_t)i * rb;
            std::multimap<float, int> fullResults;
            if (searched) {
                long idx[2]; float d[2];
                indexed2nn(q, type, idx, d);
                for (int j = 0; j < 2; ++j) {   // :1092-1137: stop at the first invalid neighbour
                    int id = idOfIndex(idx[j]);
                    if (d[j] >= 0.0f && id != 0) fullResults.insert(std::make_pair(d[j], id)); else break;
                }
            }
            if (newWordsComparedTogether && newWords.rows) {   // :1140-1160
                Top2 t;
                scan_top2(metricFor(type), newWords, q, newWords.rows > 1 ? 2 : 1, t);
                for (int j = 0; j < t.count; ++j) {
                    float d = t.d[j]; int id = newWordsId[t.idx[j]];
                    if (d >= 0.0f && id != 0) fullResults.insert(std::make_pair(d, id)); else break;
                }
            }
            if (incrementalDictionary) {   // :1162-1209
                bool badDist = false;
                if (fullResults.size() == 0) badDist = true;
                if (!badDist) {
                    if (fullResults.size() >= 2) {
                        if (fullResults.begin()->first > nndrRatio * (++fullResults.begin())->first) badDist = true;
                    } else badDist = true;
                }
                if (badDist) {
                    VisualWord* vw = new VisualWord(++lastWordId, q, cols, type, signatureId);
                    visualWords.insert(visualWords.end(), std::make_pair(vw->id, vw));
                    notIndexedWords.insert(notIndexedWords.end(), vw->id);
                    newWords.push_back(q, cols, type);
                    newWordsId.push_back(vw->id);
                    wordIds.push_back(vw->id);
                } else {
                    addWordRef(fullResults.begin()->second, signatureId);
                    wordIds.push_back(fullResults.begin()->second);
                }
            } else if (fullResults.size()) {   // :1211-1218 fixed dictionary: nearest word, or no entry at all
                addWordRef(fullResults.begin()->second, signatureId);
                wordIds.push_back(fullResults.begin()->second);
            }
        }
        totalActiveReferences += (int)notIndexedWords.size();   // :1227 (sic: the whole not-indexed set)
        return true;
    }

    // ------------------------------------------------------------------------------------------ findNN :1273-1552
    std::vector<int> findNN(const unsigned char* query, int rows, int cols, int type) {
        std::vector<int> resultIds(rows, 0);
        if (!(visualWords.size() && rows)) return resultIds;
        VisualWord* f = visualWords.begin()->second;
        if (f->cols != cols || f->type != type) { lastError = "descriptor size/type mismatch"; return resultIds; }
        if (dataTree.rows && (dataTree.cols != cols || dataTree.type != type)) { lastError = "descriptor size/type mismatch (index)"; return resultIds; }
        const size_t rb = (size_t)cols * (type == T_F32 ? 4 : 1);
        bool searched = searchable();
        // not-indexed words, ascending id (:1416-1451)
        Mat dataNotIndexed; std::vector<int> notIndexedIds;
        for (std::set<int>::iterator it = notIndexedWords.begin(); it != notIndexedWords.end(); ++it) {
            VisualWord* vw = visualWords.at(*it);
            dataNotIndexed.push_back(vw->desc.data(), vw->cols, vw->type);
            notIndexedIds.push_back(vw->id);
        }
        for (int i = 0; i < rows; ++i) {
            const unsigned char* q = query + (size_t)i * rb;
            std::multimap<float, int> fullResults;
            if (searched) {
                long idx[2]; float d[2];
                indexed2nn(q, type, idx, d);
                for (int j = 0; j < 2; ++j) {   // :1457-1495: NOT cut at the first invalid neighbour
                    int id = idOfIndex(idx[j]);
                    if (d[j] >= 0.0f && id != 0) fullResults.insert(std::make_pair(d[j], id));
                }
            }
            if (dataNotIndexed.rows) {
                Top2 t;
                scan_top2(metricFor(type), dataNotIndexed, q, dataNotIndexed.rows > 1 ? 2 : 1, t);
                for (int j = 0; j < t.count; ++j) {
                    float d = t.d[j]; int id = notIndexedIds[t.idx[j]];
                    if (d >= 0.0f && id != 0) fullResults.insert(std::make_pair(d, id)); else break;
                }
            }
            if (incrementalDictionary) {   // :1515-1542
                bool badDist = false;
                if (fullResults.size() == 0) badDist = true;
                if (!badDist) {
                    if (fullResults.size() >= 2) {
                        if (fullResults.begin()->first > nndrRatio * (++fullResults.begin())->first) badDist = true;
                    } else badDist = true;
                }
                if (!badDist) resultIds[i] = fullResults.begin()->second;
            } else if (fullResults.size()) {
                resultIds[i] = fullResults.begin()->second;
            }
        }
        return resultIds;
    }

    // ------------------------------------------------------------------------------------------ text dictionary
    // reader :181-257 (header "WordID Descriptors...<dim>", then "id v0 .. v(dim-1)"), writer :1619-1696 ("%d " / "%f ")
    int loadFixedText(const char* path) {
        FILE* f = fopen(path, "r");
        if (!f) { lastError = "cannot open dictionary"; return -1; }
        std::string line; int c; int dim = 0; bool header = true; int n = 0;
        std::vector<std::string> lines;
        std::string cur;
        while ((c = fgetc(f)) != EOF) { if (c == '\n') { lines.push_back(cur); cur.clear(); } else cur.push_back((char)c); }
        if (!cur.empty()) lines.push_back(cur);
        fclose(f);
        for (size_t li = 0; li < lines.size(); ++li) {
            const std::string& s = lines[li];
            if (header) {
                // uSplitNumChar: first all-digit token of the header is the dimension (UStl.h:670)
                size_t p = 0; dim = 0;
                while (p < s.size()) {
                    if (isdigit((unsigned char)s[p])) { size_t e = p; while (e < s.size() && isdigit((unsigned char)s[e])) ++e; dim = atoi(s.substr(p, e - p).c_str()); break; }
                    ++p;
                }
                if (dim <= 0 || dim > 1000) { lastError = "bad dictionary header"; return -1; }
                header = false; continue;
            }
            std::vector<std::string> tok; std::string t;   // uSplit on ' ' dropping empty tokens (UStl.h:564)
            for (size_t p = 0; p <= s.size(); ++p) {
                if (p == s.size() || s[p] == ' ') { if (!t.empty()) tok.push_back(t); t.clear(); } else t.push_back(s[p]);
            }
            if ((int)tok.size() != dim + 1) continue;   // malformed line skipped with a warning (:228-231)
            int id = atoi(tok[0].c_str());
            std::vector<float> v(dim);
            for (int k = 0; k < dim; ++k) {   // uStr2Float: ',' -> '.', C locale (UConversion.cpp:138)
                std::string x = tok[k + 1]; std::replace(x.begin(), x.end(), ',', '.');
                std::istringstream in(x);             // as the reference: a float extracted from a stream in the C locale (one rounding;
                in.imbue(std::locale::classic());     // out of range reads as +-FLT_MAX) -- pinned by tests/test_oracle_umath.py
                float value = 0.0f;
                in >> value;
                v[k] = value;
            }
            VisualWord* vw = new VisualWord(id, (const unsigned char*)v.data(), dim, T_F32);
            visualWords.insert(visualWords.end(), std::make_pair(id, vw));
            notIndexedWords.insert(notIndexedWords.end(), id);
            unusedWords.insert(unusedWords.end(), std::make_pair(id, vw));
            if (lastWordId < id) lastWordId = id;
            ++n;
        }
        incrementalDictionary = false;
        update();
        return n;
    }
    int exportText(const char* refsPath, const char* descPath) const {
        if (visualWords.empty()) return -1;
        if (visualWords.begin()->second->type != T_F32) return -2;   // "Exporting binary descriptors is not implemented!"
        FILE* fr = refsPath && refsPath[0] ? fopen(refsPath, "w") : 0;
        FILE* fd = descPath && descPath[0] ? fopen(descPath, "w") : 0;
        if (fr) fprintf(fr, "WordID SignaturesID...\n");
        if (fd) fprintf(fd, "WordID Descriptors...%d\n", visualWords.begin()->second->cols);
        for (std::map<int, VisualWord*>::const_iterator it = visualWords.begin(); it != visualWords.end(); ++it) {
            if (fr) {
                fprintf(fr, "%d ", it->first);
                for (std::map<int, int>::const_iterator j = it->second->references.begin(); j != it->second->references.end(); ++j)
                    for (int k = 0; k < j->second; ++k) fprintf(fr, "%d ", j->first);
                fprintf(fr, "\n");
            }
            if (fd) {
                fprintf(fd, "%d ", it->first);
                const float* d = (const float*)it->second->desc.data();
                for (int k = 0; k < it->second->cols; ++k) fprintf(fd, "%f ", d[k]);
                fprintf(fd, "\n");
            }
        }
        if (fr) fclose(fr);
        if (fd) fclose(fd);
        return 0;
    }
};

// ---------------------------------------------------------------------------------------------- Memory (hot-path subset)
struct Signature {
    int id; bool enabled;
    std::multimap<int, int> words;   // <word id, keypoint index> (Signature.h)
};

static bool g_log10_double = false;          // see computeLikelihood
struct Memory {
    VWDictionary vwd;
    std::map<int, Signature*> signatures;     // STM + WM, what Memory::getSignatures() returns
    std::map<int, int> dbNi;                  // stand-in for DBDriver::getInvertedIndexNi of transferred nodes
    int idCount;
    int maxFeatures;                          // Kp/MaxFeatures emulation for the -1,-2,.. glue (0 = quantise all)
    Memory() : idCount(0), maxFeatures(0) {}
    ~Memory() { for (std::map<int, Signature*>::iterator i = signatures.begin(); i != signatures.end(); ++i) delete i->second; }

    void cleanUnusedWords() {   // Memory.cpp:6899-6920
        std::vector<int> ids = vwd.getUnusedWordIds();
        if (ids.size()) vwd.removeWords(ids, true);
    }
    void preUpdate() {          // Memory.cpp:1004-1016 with Kp/Parallelized: update() still precedes addNewWords
        if (vwd.incrementalDictionary) cleanUnusedWords();
        vwd.update();
    }
    // Memory::update -> createSignature quantisation block (:5941-6059) -> addSignatureToStm
    // `quantise` rows [0, nq) are sent to addNewWords; the remaining features get ids -1,-2,...
    int update(const unsigned char* desc, int rows, int cols, int type, int nq, std::vector<int>& outIds) {
        preUpdate();
        int id = ++idCount;
        std::list<int> wordIds;
        if (nq > rows) nq = rows;
        if (rows) {
            if (nq > 0) {
                vwd.addNewWords(desc, nq, cols, type, id, wordIds);
                if ((int)wordIds.size() < rows) {
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
        Signature* s = new Signature();
        s->id = id; s->enabled = true;
        int k = 0;
        for (std::list<int>::iterator it = wordIds.begin(); it != wordIds.end(); ++it, ++k) s->words.insert(std::make_pair(*it, k));
        signatures.insert(signatures.end(), std::make_pair(id, s));
        outIds.assign(wordIds.begin(), wordIds.end());
        return id;
    }
    int getNi(int signatureId) const {   // :4955-4968
        std::map<int, Signature*>::const_iterator it = signatures.find(signatureId);
        if (it != signatures.end()) return (int)it->second->words.size();
        std::map<int, int>::const_iterator d = dbNi.find(signatureId);
        return d == dbNi.end() ? 0 : d->second;
    }
    void disableWordsRef(int signatureId) {   // :6877-6897
        std::map<int, Signature*>::iterator it = signatures.find(signatureId);
        if (it == signatures.end() || !it->second->enabled) return;
        int last = 0; bool first = true;
        for (std::multimap<int, int>::iterator w = it->second->words.begin(); w != it->second->words.end(); ++w) {
            if (first || w->first != last) { vwd.removeAllWordRef(w->first, signatureId); last = w->first; first = false; }
        }
        it->second->enabled = false;
    }
    void forget(int signatureId) {   // WM -> LTM transfer: moveToTrash -> disableWordsRef, signature leaves _signatures
        std::map<int, Signature*>::iterator it = signatures.find(signatureId);
        if (it == signatures.end()) return;
        disableWordsRef(signatureId);
        dbNi[signatureId] = (int)it->second->words.size();
        delete it->second;
        signatures.erase(it);
    }
    // computeLikelihood, TF-IDF branch :2215-2291.  words = the query signature's word ids (any order, duplicates ok)
    bool computeLikelihood(const std::vector<int>& sigWords, const std::vector<int>& ids, std::map<int, float>& likelihood) const {
        likelihood.clear();
        if (ids.empty()) return false;
        for (size_t i = 0; i < ids.size(); ++i) likelihood.insert(likelihood.end(), std::make_pair(ids[i], 0.0f));
        std::set<int> uniq(sigWords.begin(), sigWords.end());   // uUniqueKeys: ascending unique keys
        float nwi, ni, nw, N, logNnw;
        N = (float)signatures.size();
        if (N) {
            for (std::set<int>::iterator i = uniq.begin(); i != uniq.end(); ++i) {
                if (*i > 0) {
                    std::map<int, VisualWord*>::const_iterator w = vwd.visualWords.find(*i);
                    if (w == vwd.visualWords.end()) return false;   // UASSERT in the reference
                    const std::map<int, int>& refs = w->second->references;
                    nw = (float)refs.size();
                    if (nw) {
                        // Memory.cpp:2266 writes the unqualified `log10(N/nw)` on floats.  Memory.cpp includes <cmath> through its
                        // headers; with libstdc++ >= 6 / libc++ the global namespace then holds the float overload (the <math.h>
                        // wrapper does `using std::log10`) and the call is log10f; an older library only declares ::log10(double) and
                        // the ratio is promoted, the result rounded back to float on assignment.  g_log10_double selects the second
                        // reading for tests/test_oracle_golden.py, which checks that both stay within the parity bound of each other.
                        logNnw = g_log10_double ? (float)log10((double)(N / nw)) : log10f(N / nw);
                        if (logNnw) {
                            for (std::map<int, int>::const_iterator j = refs.begin(); j != refs.end(); ++j) {
                                std::map<int, float>::iterator it = likelihood.find(j->first);
                                if (it != likelihood.end()) {
                                    nwi = (float)j->second;
                                    ni = (float)getNi(j->first);
                                    if (ni != 0) it->second += (nwi * logNnw) / ni;
                                }
                            }
                        }
                    }
                }
            }
        }
        return true;
    }
};

// Rtabmap::adjustLikelihood :5691-5760.  L[0] is the virtual place.
static void adjustLikelihood(float* L, int n, float virtualPlaceLikelihoodRatio) {
    if (n == 0) return;
    std::list<float> values;
    for (int i = 1; i < n; ++i) if (L[i] > 0) values.push_back(L[i]);
    float mean = 0;   // uMean UMath.h:419-432
    if (values.size()) { for (std::list<float>::iterator i = values.begin(); i != values.end(); ++i) mean += *i; mean /= values.size(); }
    float var = 0;    // uVariance UMath.h:512-526 (sample variance, n-1)
    if (values.size() > 1) { float sum = 0; for (std::list<float>::iterator i = values.begin(); i != values.end(); ++i) sum += (*i - mean) * (*i - mean); var = sum / (values.size() - 1); }
    float stdDev = std::sqrt(var);
    float epsilon = 0.0001f, max = 0.0f;
    for (int i = 1; i < n; ++i) {
        float value = L[i];
        L[i] = 1.0f;
        if (value > mean + stdDev) {
            if (virtualPlaceLikelihoodRatio == 0 && mean) L[i] = (value - (stdDev - epsilon)) / mean;
            else if (virtualPlaceLikelihoodRatio != 0 && stdDev) L[i] = (value - mean) / stdDev;
        }
        if (value > max) max = value;
    }
    if (virtualPlaceLikelihoodRatio == 0 && stdDev > epsilon && max) L[0] = mean / stdDev + 1.0f;
    else if (virtualPlaceLikelihoodRatio != 0 && max > mean) L[0] = stdDev / (max - mean) + 1.0f;
    else L[0] = 2.0f;
}


// ================================================================================================ BayesFilter
// corelib/src/BayesFilter.cpp: setPredictionLC :77-122, computePosterior :145-235, addNeighborProb :237-270,
// generatePrediction :273-420 (the full update; the incremental updatePrediction :502-706 rebuilds the same columns from
// cached neighbour maps), normalize :434-500, updatePosterior :709-736.  Hypothesis selection: Rtabmap.cpp:2147-2158.
// Memory is replaced by what the filter asks it: getNeighborsId(id, LC.size() - 1, 0, false, false, true, true) answers
// (`graph`, set by the test harness) and isInSTM (`stm`).  The matrix product prior = prediction * posterior is cv::gemm on
// CV_32F data -- OpenCV is not in the reference tree: restated as a row-by-column sum accumulated in double and rounded to
// float once (OpenCV's generic GEMM kernel for float uses double accumulators); "parity unpinned" for that rounding.
// Two evaluations of the same arithmetic: `dense` allocates the m x m matrix and follows the reference statement by statement
// (small m only); the sparse one keeps the non-zero elements of each column and is valid when _totalPredictionLCValues >= 1
// (no "all other places" fill, true for the default Bayes/PredictionLC), for the 100k-signature tests.
struct BayesFilter {
    std::vector<double> predictionLC;
    float virtualPlacePrior = 0.9f;            // Parameters.h Bayes/VirtualPlacePriorThr
    float totalPredictionLCValues = 0.0f;
    float predictionEpsilon = 0.0f;
    std::map<int, float> posterior;
    std::map<int, std::map<int, int> > graph;  // id -> (neighbour id -> margin), the harness' Memory::getNeighborsId
    std::set<int> stm;
    // Bayes/FullPredictionUpdate = false (the reference's default): the matrix of the last call and the neighbour cache it is patched from
    bool fullPredictionUpdate = true;
    std::vector<float> prediction; int predictionCols = 0;
    std::map<int, std::map<int, int> > neighborsIndex;

    void setPredictionLC(const double* v, int n) {                       // :77-122 (values already parsed)
        predictionLC.assign(v, v + n);
        totalPredictionLCValues = 0.0f;
        for (unsigned int j = 0; j < predictionLC.size(); ++j) {
            totalPredictionLCValues += predictionLC[j];
            if (j == 0 || predictionLC[j] < predictionEpsilon) predictionEpsilon = predictionLC[j];
        }
    }
    void reset() { posterior.clear(); prediction.clear(); predictionCols = 0; neighborsIndex.clear(); }   // :138-143
    std::map<int, int> getNeighborsId(int id) const {                   // Memory::getNeighborsId as the harness answers it
        std::map<int, std::map<int, int> >::const_iterator g = graph.find(id);
        return g == graph.end() ? std::map<int, int>() : g->second;
    }

    // ---- dense, literal
    float addNeighborProb(std::vector<float>& P, int cols, unsigned int col, const std::map<int, int>& neighbors,
                          const std::map<int, int>& idToIndex) {        // :237-270
        float sum = 0.0f;
        for (std::map<int, int>::const_iterator iter = neighbors.begin(); iter != neighbors.end(); ++iter) {
            if (iter->first >= 0) {
                std::map<int, int>::const_iterator jter = idToIndex.find(iter->first);
                if (jter != idToIndex.end()) sum += P[col + (size_t)jter->second * cols] = predictionLC[iter->second + 1];
            }
        }
        return sum;
    }
    void normalize(std::vector<float>& P, int cols, unsigned int index, float addedProbabilitiesSum, bool virtualPlaceUsed) {   // :434-500
        if (addedProbabilitiesSum < totalPredictionLCValues - predictionLC[0]) {
            float delta = totalPredictionLCValues - predictionLC[0] - addedProbabilitiesSum;
            P[index + (size_t)index * cols] += delta;
            addedProbabilitiesSum += delta;
        }
        float allOtherPlacesValue = 0;
        if (totalPredictionLCValues < 1) allOtherPlacesValue = 1.0f - totalPredictionLCValues;
        if (allOtherPlacesValue > 0 && cols > 1) {
            float value = allOtherPlacesValue / float(cols - 1);
            for (int j = virtualPlaceUsed ? 1 : 0; j < cols; ++j) {
                if (P[index + (size_t)j * cols] == 0) { P[index + (size_t)j * cols] = value; addedProbabilitiesSum += P[index + (size_t)j * cols]; }
            }
        }
        float maxNorm = 1 - (virtualPlaceUsed ? predictionLC[0] : 0);
        if (addedProbabilitiesSum < maxNorm - 0.0001 || addedProbabilitiesSum > maxNorm + 0.0001) {
            for (int j = virtualPlaceUsed ? 1 : 0; j < cols; ++j) {
                P[index + (size_t)j * cols] *= maxNorm / addedProbabilitiesSum;
                if (P[index + (size_t)j * cols] < predictionEpsilon) P[index + (size_t)j * cols] = 0.0f;
            }
            addedProbabilitiesSum = maxNorm;
        }
        if (virtualPlaceUsed) { P[index] = predictionLC[0]; addedProbabilitiesSum += P[index]; }
    }
    std::map<int, int> neighborsNotInStm(int id) const {                // :330-352 (the filter part)
        std::map<int, int> neighbors;
        std::map<int, std::map<int, int> >::const_iterator g = graph.find(id);
        if (g != graph.end()) neighbors = g->second;
        for (std::map<int, int>::iterator iter = neighbors.begin(); iter != neighbors.end();) {
            if (stm.count(iter->first)) neighbors.erase(iter++); else ++iter;
        }
        return neighbors;
    }
    bool generatePredictionDense(const std::vector<int>& ids, std::vector<float>& P) {   // :273-420
        const int cols = (int)ids.size();
        std::map<int, int> idToIndexMap;
        for (unsigned int i = 0; i < ids.size(); ++i) if (ids[i] > 0) idToIndexMap[ids[i]] = i;
        P.assign((size_t)cols * cols, 0.0f);
        std::set<int> idsDone;
        for (unsigned int i = 0; i < ids.size(); ++i) {
            if (idsDone.find(ids[i]) != idsDone.end()) continue;
            if (ids[i] > 0) {
                if (!fullPredictionUpdate) neighborsIndex[ids[i]] = getNeighborsId(ids[i]);     // uInsert :332-335 (before the STM filter)
                std::map<int, int> neighbors = neighborsNotInStm(ids[i]);
                std::list<int> idsLoopMargin;
                for (std::map<int, int>::iterator iter = neighbors.begin(); iter != neighbors.end(); ++iter)
                    if (iter->second == 0 && idToIndexMap.find(iter->first) != idToIndexMap.end()) idsLoopMargin.push_back(iter->first);
                if (idsLoopMargin.size() == 0) return false;            // UFATAL :357
                for (std::list<int>::iterator iter = idsLoopMargin.begin(); iter != idsLoopMargin.end(); ++iter) {
                    if (!fullPredictionUpdate) neighborsIndex[*iter] = neighbors;                  // uInsert :364-367 (the filtered map)
                    float sum = 0.0f;
                    int index = idToIndexMap.at(*iter);
                    sum += addNeighborProb(P, cols, index, neighbors, idToIndexMap);
                    idsDone.insert(*iter);
                    normalize(P, cols, index, sum, ids[0] < 0);
                }
            } else {
                if (virtualPlacePrior > 0) {
                    if (cols > 1) {
                        P[i] = virtualPlacePrior;
                        float val = (1.0 - virtualPlacePrior) / (cols - 1);
                        for (int j = 1; j < cols; j++) P[i + (size_t)j * cols] = val;
                    } else if (cols > 0) P[i] = 1;
                } else {
                    if (cols > 1) { float val = 1.0 / cols; for (int j = 0; j < cols; j++) P[i + (size_t)j * cols] = val; }
                    else if (cols > 0) P[i] = 1;
                }
            }
        }
        return true;
    }
    // updatePrediction :502-706: the old matrix patched for the ids that came and went, columns rebuilt from _neighborsIndex
    bool updatePredictionDense(const std::vector<float>& oldPrediction, const std::vector<int>& oldIds, const std::vector<int>& newIds,
                               std::vector<float>& P) {
        const int oldCols = (int)oldIds.size(), cols = (int)newIds.size();
        P.assign((size_t)cols * cols, 0.0f);
        std::set<int> oldIdsSet(oldIds.begin(), oldIds.end());
        std::map<int, int> newIdToIndexMap;
        for (unsigned int i = 0; i < newIds.size(); ++i) if (newIds[i] > 0) newIdToIndexMap[newIds[i]] = i;
        std::set<int> removedIds;                                        // :543-553
        for (unsigned int i = 0; i < oldIds.size(); ++i) {
            if (oldIds[i] > 0 && newIdToIndexMap.find(oldIds[i]) == newIdToIndexMap.end()) { removedIds.insert(removedIds.end(), oldIds[i]); neighborsIndex.erase(oldIds[i]); }
        }
        bool oldAllCopied = false;                                       // :556-564
        if (removedIds.empty() && newIds.size() > oldIds.size() && std::equal(oldIds.begin(), oldIds.end(), newIds.begin())) {
            for (int r = 0; r < oldCols; ++r) for (int c = 0; c < oldCols; ++c) P[c + (size_t)r * cols] = oldPrediction[c + (size_t)r * oldCols];
            oldAllCopied = true;
        }
        std::set<int> idsToUpdate;                                       // :566-621
        for (unsigned int i = 0; i < oldIds.size() || i < newIds.size(); ++i) {
            if (i < oldIds.size()) {
                if (removedIds.find(oldIds[i]) != removedIds.end()) {
                    for (unsigned int j = 0; j < (unsigned int)oldCols; ++j)
                        if (j != i && removedIds.find(oldIds[j]) == removedIds.end()) idsToUpdate.insert(oldIds[j]);
                }
            }
            if (i < newIds.size() && oldIdsSet.find(newIds[i]) == oldIdsSet.end()) {
                if (neighborsIndex.find(newIds[i]) == neighborsIndex.end()) {
                    std::map<int, int> neighbors = getNeighborsId(newIds[i]);
                    for (std::map<int, int>::iterator iter = neighbors.begin(); iter != neighbors.end(); ++iter) {
                        std::map<int, std::map<int, int> >::iterator jter = neighborsIndex.find(iter->first);
                        if (jter != neighborsIndex.end()) jter->second[newIds[i]] = iter->second;                     // uInsert :589
                    }
                    neighborsIndex.insert(std::make_pair(newIds[i], neighbors));
                }
                const std::map<int, int>& neighbors = neighborsIndex.at(newIds[i]);
                float sum = addNeighborProb(P, cols, i, neighbors, newIdToIndexMap);
                normalize(P, cols, i, sum, newIds[0] < 0);
                for (std::map<int, int>::const_iterator iter = neighbors.begin(); iter != neighbors.end(); ++iter)
                    if (oldIdsSet.find(iter->first) != oldIdsSet.end() && removedIds.find(iter->first) == removedIds.end()) idsToUpdate.insert(iter->first);
            }
        }
        for (std::set<int>::iterator iter = idsToUpdate.begin(); iter != idsToUpdate.end(); ++iter) {              // :628-650
            int id = *iter;
            if (id > 0) {
                int index = newIdToIndexMap.at(id);
                std::map<int, std::map<int, int> >::iterator kter = neighborsIndex.find(id);
                if (kter == neighborsIndex.end()) return false;          // UASSERT_MSG :636
                float sum = addNeighborProb(P, cols, index, kter->second, newIdToIndexMap);
                normalize(P, cols, index, sum, newIds[0] < 0);
            }
        }
        if (!oldAllCopied) {                                             // :654-682: copy the columns that did not change
            for (unsigned int i = 0; i < oldIds.size(); ++i) {
                if (oldIds[i] > 0 && removedIds.find(oldIds[i]) == removedIds.end() && idsToUpdate.find(oldIds[i]) == idsToUpdate.end()) {
                    for (int j = 0; j < oldCols; ++j) {
                        if (oldIds[j] > 0 && removedIds.find(oldIds[j]) == removedIds.end()) {
                            float v = oldPrediction[i + (size_t)j * oldCols];
                            int ii = newIdToIndexMap.at(oldIds[i]);
                            int jj = newIdToIndexMap.at(oldIds[j]);
                            P[ii + (size_t)jj * cols] = v;
                        }
                    }
                }
            }
        }
        if (newIds[0] < 0) {                                             // :685-700: the virtual place
            if (cols > 1) {
                P[0] = virtualPlacePrior;
                float val = (1.0 - virtualPlacePrior) / (cols - 1);
                for (int j = 1; j < cols; j++) { P[(size_t)j * cols] = val; P[j] = predictionLC[0]; }
            } else if (cols > 0) P[0] = 1;
        }
        return true;
    }
    // generatePrediction's dispatch :273-286 for Bayes/FullPredictionUpdate = false
    bool generatePredictionIncremental(const std::vector<int>& ids, std::vector<float>& P) {
        std::vector<int> oldIds;
        for (std::map<int, float>::const_iterator i = posterior.begin(); i != posterior.end(); ++i) oldIds.push_back(i->first);
        if (oldIds == ids && !prediction.empty()) { P = prediction; return true; }
        bool ok;
        if (!prediction.empty()) ok = updatePredictionDense(prediction, oldIds, ids, P);
        else ok = generatePredictionDense(ids, P);
        if (ok) { prediction = P; predictionCols = (int)ids.size(); }
        return ok;
    }
    void updatePosterior(const std::vector<int>& likelihoodIds) {       // :709-736
        std::map<int, float> newPosterior;
        for (std::vector<int>::const_iterator i = likelihoodIds.begin(); i != likelihoodIds.end(); ++i) {
            std::map<int, float>::iterator post = posterior.find(*i);
            if (post == posterior.end()) newPosterior.insert(std::pair<int, float>(*i, posterior.size() == 0 ? 1.0f : 0.0f));
            else newPosterior.insert(std::pair<int, float>(post->first, post->second));
        }
        posterior = newPosterior;
    }
    // ---- sparse evaluation of the same columns (totalPredictionLCValues >= 1 only)
    struct Column { std::vector<std::pair<int, float> > rows; float row0; };   // rows: matrix row index >= 1 -> value; row0 = P[0][c]
    bool columnSparse(int c_index, const std::map<int, int>& neighbors, const std::map<int, int>& idToIndex, bool virtualPlaceUsed, int cols, Column& out) {
        std::map<int, float> col;                                          // row -> value, like the dense column's non-zeros
        float sum = 0.0f;
        for (std::map<int, int>::const_iterator iter = neighbors.begin(); iter != neighbors.end(); ++iter) {
            if (iter->first >= 0) {
                std::map<int, int>::const_iterator jter = idToIndex.find(iter->first);
                if (jter != idToIndex.end()) sum += col[jter->second] = predictionLC[iter->second + 1];
            }
        }
        if (sum < totalPredictionLCValues - predictionLC[0]) {
            float delta = totalPredictionLCValues - predictionLC[0] - sum;
            col[c_index] += delta;
            sum += delta;
        }
        if (totalPredictionLCValues < 1 && cols > 1) return false;         // needs the dense fill
        float maxNorm = 1 - (virtualPlaceUsed ? predictionLC[0] : 0);
        if (sum < maxNorm - 0.0001 || sum > maxNorm + 0.0001) {
            for (std::map<int, float>::iterator e = col.begin(); e != col.end(); ++e) {
                if (virtualPlaceUsed && e->first == 0) continue;
                e->second *= maxNorm / sum;
                if (e->second < predictionEpsilon) e->second = 0.0f;
            }
        }
        out.rows.assign(col.begin(), col.end());
        out.row0 = virtualPlaceUsed ? (float)predictionLC[0] : 0.0f;
        return true;
    }

    // computePosterior :145-235.  likelihood: ids ascending (std::map order), ids[0] may be the virtual place (-1).
    // returns 0 ok, -1 invalid input, -2 a signature without a 0-margin neighbour (UFATAL), -3 sparse evaluation not valid
    int computePosterior(const int* ids_in, const float* like, int m, int dense, float* out) {
        if (m <= 0 || predictionLC.size() < 2) return -1;
        std::vector<int> ids(ids_in, ids_in + m);
        const int cols = m;
        std::vector<float> prior(m, 0.0f);
        std::vector<float> post(m);
        if (dense) {
            std::vector<float> P;
            fullPredictionUpdate = dense != 2;                           // dense == 2: the reference's default incremental mode
            if (!(dense == 2 ? generatePredictionIncremental(ids, P) : generatePredictionDense(ids, P))) return -2;
            updatePosterior(ids);
            int j = 0;
            for (std::map<int, float>::const_iterator i = posterior.begin(); i != posterior.end(); ++i) post[j++] = i->second;
            for (int r = 0; r < m; ++r) {
                double acc = 0.0;
                for (int c = 0; c < m; ++c) acc += (double)P[c + (size_t)r * cols] * (double)post[c];
                prior[r] = (float)acc;
            }
        } else {
            std::map<int, int> idToIndexMap;
            for (unsigned int i = 0; i < ids.size(); ++i) if (ids[i] > 0) idToIndexMap[ids[i]] = i;
            const bool vp = ids[0] < 0;
            std::vector<Column> columns(m);
            std::vector<char> done(m, 0);
            for (int i = 0; i < m; ++i) {
                if (done[i] || ids[i] <= 0) continue;
                std::map<int, int> neighbors = neighborsNotInStm(ids[i]);
                std::list<int> idsLoopMargin;
                for (std::map<int, int>::iterator iter = neighbors.begin(); iter != neighbors.end(); ++iter)
                    if (iter->second == 0 && idToIndexMap.find(iter->first) != idToIndexMap.end()) idsLoopMargin.push_back(iter->first);
                if (idsLoopMargin.size() == 0) return -2;
                for (std::list<int>::iterator iter = idsLoopMargin.begin(); iter != idsLoopMargin.end(); ++iter) {
                    int index = idToIndexMap.at(*iter);
                    if (!columnSparse(index, neighbors, idToIndexMap, vp, cols, columns[index])) return -3;
                    done[index] = 1;
                }
            }
            updatePosterior(ids);
            int j = 0;
            for (std::map<int, float>::const_iterator i = posterior.begin(); i != posterior.end(); ++i) post[j++] = i->second;
            std::vector<double> acc(m, 0.0);
            for (int c = 0; c < m; ++c) {
                if (ids[c] <= 0) {                                         // the virtual place's column :376-411
                    if (virtualPlacePrior > 0) {
                        if (cols > 1) {
                            acc[c] += (double)virtualPlacePrior * (double)post[c];
                            float val = (1.0 - virtualPlacePrior) / (cols - 1);
                            for (int r = 1; r < cols; ++r) acc[r] += (double)val * (double)post[c];
                        } else acc[c] += (double)post[c];
                    } else {
                        if (cols > 1) { float val = 1.0 / cols; for (int r = 0; r < cols; ++r) acc[r] += (double)val * (double)post[c]; }
                        else acc[c] += (double)post[c];
                    }
                    continue;
                }
                const Column& col = columns[c];
                if (vp) acc[0] += (double)col.row0 * (double)post[c];
                for (size_t e = 0; e < col.rows.size(); ++e) acc[col.rows[e].first] += (double)col.rows[e].second * (double)post[c];
            }
            for (int r = 0; r < m; ++r) prior[r] = (float)acc[r];
        }
        // STEP 2 :205-233
        float sum = 0;
        int j = 0;
        for (int i = 0; i < m; ++i) {
            std::map<int, float>::iterator p = posterior.find(ids[i]);
            if (p != posterior.end()) { p->second = like[i] * prior[j++]; sum += p->second; }
        }
        if (sum != 0) for (std::map<int, float>::iterator i = posterior.begin(); i != posterior.end(); ++i) i->second /= sum;
        j = 0;
        for (std::map<int, float>::const_iterator i = posterior.begin(); i != posterior.end(); ++i) out[j++] = i->second;
        return 0;
    }
};

}  // namespace orc

// =============================================================================================== C entry points (ctypes)
using namespace orc;
extern "C" {

float orc_dist_l2(const float* a, const float* b, size_t n) { return dist_l2(a, b, n); }
float orc_dist_l1(const float* a, const float* b, size_t n) { return dist_l1(a, b, n); }
unsigned orc_dist_hamming(const unsigned char* a, const unsigned char* b, size_t n) { return dist_hamming(a, b, n); }

// exact k<=2 linear scan (linear_index.h:129-144 + result_set.h:151-171).  removed may be NULL.  idx -1 / dist -1 = none.
// threads > 1 parallelises over queries (the "generous" CPU baseline; results are per-query so unchanged).
void orc_knn2_linear(int metric, const void* train, long n, int cols, const unsigned char* removed,
                     const void* queries, long nq, long* idx, float* dist, int threads) {
    const size_t rb = (size_t)cols * (metric_is_u8(metric) ? 1 : 4);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (long q = 0; q < nq; ++q) {
        const unsigned char* qp = (const unsigned char*)queries + (size_t)q * rb;
        Top2 t;
        for (long r = 0; r < n; ++r) {
            if (removed && removed[r]) continue;
            t.add(dist_any(metric, (const unsigned char*)train + (size_t)r * rb, qp, cols), r);
        }
        for (int j = 0; j < 2; ++j) {
            if (j < t.count) { idx[2 * q + j] = t.idx[j]; dist[2 * q + j] = t.d[j]; }
            else { idx[2 * q + j] = -1; dist[2 * q + j] = -1.0f; }
        }
    }
}
// full nq x n distance matrix (row-major), for the self-distance checks
void orc_dist_matrix(int metric, const void* a, long na, const void* b, long nb, int cols, float* out) {
    const size_t rb = (size_t)cols * (metric_is_u8(metric) ? 1 : 4);
    for (long i = 0; i < na; ++i)
        for (long j = 0; j < nb; ++j)
            out[i * nb + j] = dist_any(metric, (const unsigned char*)a + (size_t)i * rb, (const unsigned char*)b + (size_t)j * rb, cols);
}

// ---- VWDictionary handle
void* orc_vwd_create(int strategy, int incremental, float nndr, int newWordsComparedTogether, int incrementalFlann) {
    VWDictionary* d = new VWDictionary();
    d->strategy = strategy; d->incrementalDictionary = incremental != 0; d->nndrRatio = nndr;
    d->newWordsComparedTogether = newWordsComparedTogether != 0; d->incrementalFlann = incrementalFlann != 0;
    return d;
}
void orc_vwd_destroy(void* h) { delete (VWDictionary*)h; }
const char* orc_vwd_last_error(void* h) { return ((VWDictionary*)h)->lastError.c_str(); }
int orc_vwd_add_new_words(void* h, const void* desc, int rows, int cols, int type, int sigId, int* out, int cap) {
    std::list<int> ids;
    if (!((VWDictionary*)h)->addNewWords((const unsigned char*)desc, rows, cols, type, sigId, ids)) return -1;
    int n = 0;
    for (std::list<int>::iterator i = ids.begin(); i != ids.end() && n < cap; ++i) out[n++] = *i;
    return (int)ids.size();
}
int orc_vwd_find_nn(void* h, const void* desc, int rows, int cols, int type, int* out) {
    std::vector<int> r = ((VWDictionary*)h)->findNN((const unsigned char*)desc, rows, cols, type);
    for (int i = 0; i < rows; ++i) out[i] = r[i];
    return rows;
}
void orc_vwd_update(void* h) { ((VWDictionary*)h)->update(); }
void orc_vwd_add_word(void* h, int id, const void* desc, int cols, int type) {
    ((VWDictionary*)h)->addWord(new VisualWord(id, (const unsigned char*)desc, cols, type));
}
int orc_vwd_add_word_ref(void* h, int wordId, int sigId) { return ((VWDictionary*)h)->addWordRef(wordId, sigId) ? 1 : 0; }
void orc_vwd_remove_all_word_ref(void* h, int wordId, int sigId) { ((VWDictionary*)h)->removeAllWordRef(wordId, sigId); }
int orc_vwd_get_unused_word_ids(void* h, int* out, int cap) {
    std::vector<int> v = ((VWDictionary*)h)->getUnusedWordIds();
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}
void orc_vwd_remove_words(void* h, const int* ids, int n) { ((VWDictionary*)h)->removeWords(std::vector<int>(ids, ids + n), true); }
void orc_vwd_delete_unused_words(void* h) { VWDictionary* d = (VWDictionary*)h; d->removeWords(d->getUnusedWordIds(), true); }
void orc_vwd_clear(void* h) { ((VWDictionary*)h)->clear(); }
// which: 0 visualWords, 1 notIndexed, 2 indexed rows (live), 3 totalActiveReferences, 4 lastWordId, 5 unused, 6 removedIndexed
long orc_vwd_stat(void* h, int which) {
    VWDictionary* d = (VWDictionary*)h;
    switch (which) {
        case 0: return (long)d->visualWords.size();
        case 1: return (long)d->notIndexedWords.size();
        case 2: return (long)d->mapIndexId.size();
        case 3: return d->totalActiveReferences;
        case 4: return d->lastWordId;
        case 5: return (long)d->unusedWords.size();
        case 6: return (long)d->removedIndexedWords.size();
    }
    return -1;
}
int orc_vwd_get_word_refs(void* h, int wordId, int* sigs, int* counts, int cap) {
    VWDictionary* d = (VWDictionary*)h;
    std::map<int, VisualWord*>::iterator it = d->visualWords.find(wordId);
    if (it == d->visualWords.end()) return -1;
    int n = 0;
    for (std::map<int, int>::iterator j = it->second->references.begin(); j != it->second->references.end(); ++j, ++n)
        if (n < cap) { sigs[n] = j->first; counts[n] = j->second; }
    return n;
}
int orc_vwd_word_ids(void* h, int* out, int cap) {
    VWDictionary* d = (VWDictionary*)h; int n = 0;
    for (std::map<int, VisualWord*>::iterator i = d->visualWords.begin(); i != d->visualWords.end(); ++i, ++n) if (n < cap) out[n] = i->first;
    return n;
}
// ids of the indexed rows in scan (tie-break) order
int orc_vwd_index_ids(void* h, int* out, int cap) {
    VWDictionary* d = (VWDictionary*)h; int n = 0;
    for (std::map<int, int>::iterator i = d->mapIndexId.begin(); i != d->mapIndexId.end(); ++i, ++n) if (n < cap) out[n] = i->second;
    return n;
}
int orc_vwd_load_fixed_text(void* h, const char* path) { return ((VWDictionary*)h)->loadFixedText(path); }
int orc_vwd_export_text(void* h, const char* refs, const char* desc) { return ((VWDictionary*)h)->exportText(refs, desc); }

// ---- Memory handle
void* orc_mem_create(int strategy, int incremental, float nndr, int newWordsComparedTogether, int incrementalFlann) {
    Memory* m = new Memory();
    m->vwd.strategy = strategy; m->vwd.incrementalDictionary = incremental != 0; m->vwd.nndrRatio = nndr;
    m->vwd.newWordsComparedTogether = newWordsComparedTogether != 0; m->vwd.incrementalFlann = incrementalFlann != 0;
    return m;
}
void orc_mem_destroy(void* h) { delete (Memory*)h; }
void* orc_mem_vwd(void* h) { return &((Memory*)h)->vwd; }
int orc_mem_update(void* h, const void* desc, int rows, int cols, int type, int nq, int* outIds) {
    std::vector<int> ids;
    int id = ((Memory*)h)->update((const unsigned char*)desc, rows, cols, type, nq, ids);
    for (size_t i = 0; i < ids.size(); ++i) outIds[i] = ids[i];
    return id;
}
// bulk construction used by the benchmark: a signature with given word ids whose refs are added with addWordRef
int orc_mem_add_signature(void* h, const int* wordIds, int n) {
    Memory* m = (Memory*)h;
    int id = ++m->idCount;
    Signature* s = new Signature(); s->id = id; s->enabled = true;
    for (int k = 0; k < n; ++k) { s->words.insert(std::make_pair(wordIds[k], k)); if (wordIds[k] > 0) m->vwd.addWordRef(wordIds[k], id); }
    m->signatures.insert(m->signatures.end(), std::make_pair(id, s));
    return id;
}
// The same state as n_sigs calls of orc_mem_add_signature (rows of `q` word ids), built fast: the new signature ids are larger
// than every id already referenced, so each std::map insertion happens at the end (O(1) with a hint) and the words are found
// through a direct index instead of the id map.  Test glue for the 100k-signature memories of bench.py / the headline-size test
// (the reference fills its maps the slow way, through addWordRef, Memory.cpp:447-480); tests/test_oracle_bulk.py pins it
// against the call-by-call construction.  Returns the first new signature id.
int orc_mem_add_signatures_bulk(void* h, const int* wordIds, int n_sigs, int q) {
    Memory* m = (Memory*)h;
    int maxId = 0;
    for (std::map<int, VisualWord*>::iterator i = m->vwd.visualWords.begin(); i != m->vwd.visualWords.end(); ++i) if (i->first > maxId) maxId = i->first;
    std::vector<VisualWord*> byId((size_t)maxId + 1, (VisualWord*)0);
    for (std::map<int, VisualWord*>::iterator i = m->vwd.visualWords.begin(); i != m->vwd.visualWords.end(); ++i) byId[i->first] = i->second;
    const int first = m->idCount + 1;
    std::vector<std::pair<int, int> > sorted((size_t)q);
    for (int s = 0; s < n_sigs; ++s) {
        const int id = ++m->idCount;
        const int* w = wordIds + (size_t)s * q;
        Signature* sg = new Signature(); sg->id = id; sg->enabled = true;
        for (int k = 0; k < q; ++k) sorted[k] = std::make_pair(w[k], k);
        std::stable_sort(sorted.begin(), sorted.end());          // multimap order: by word id, equal ids in insertion order
        for (int k = 0; k < q; ++k) sg->words.insert(sg->words.end(), sorted[k]);
        for (int k = 0; k < q;) {
            int e = k;
            while (e < q && sorted[e].first == sorted[k].first) ++e;
            const int word = sorted[k].first;
            if (word > 0 && word <= maxId && byId[word]) {
                VisualWord* vw = byId[word];
                if (!vw->references.empty() && vw->references.rbegin()->first >= id) {       // not the fast case: fall back
                    for (int r = k; r < e; ++r) m->vwd.addWordRef(word, id);
                } else {
                    vw->references.insert(vw->references.end(), std::make_pair(id, e - k));
                    vw->totalReferences += e - k;
                    m->vwd.totalActiveReferences += e - k;
                    if (vw->references.size() == 1) m->vwd.unusedWords.erase(word);
                }
            }
            k = e;
        }
        m->signatures.insert(m->signatures.end(), std::make_pair(id, sg));
    }
    return first;
}
// same, with an explicit signature id (fixtures: the virtual place is id -1, Memory.cpp:71)
int orc_mem_add_signature_with_id(void* h, int id, const int* wordIds, int n) {
    Memory* m = (Memory*)h;
    if (m->signatures.count(id)) return 0;
    Signature* s = new Signature(); s->id = id; s->enabled = true;
    for (int k = 0; k < n; ++k) { s->words.insert(std::make_pair(wordIds[k], k)); if (wordIds[k] > 0) m->vwd.addWordRef(wordIds[k], id); }
    m->signatures.insert(std::make_pair(id, s));
    if (id > m->idCount) m->idCount = id;
    return id;
}
void orc_mem_forget(void* h, int sigId) { ((Memory*)h)->forget(sigId); }
int orc_mem_get_ni(void* h, int sigId) { return ((Memory*)h)->getNi(sigId); }
long orc_mem_num_signatures(void* h) { return (long)((Memory*)h)->signatures.size(); }
int orc_mem_signature_ids(void* h, int* out, int cap) {
    Memory* m = (Memory*)h; int n = 0;
    for (std::map<int, Signature*>::iterator i = m->signatures.begin(); i != m->signatures.end(); ++i, ++n) if (n < cap) out[n] = i->first;
    return n;
}
// likelihood of `words` (a signature's word ids) against `ids`; out[i] pairs with ids sorted ascending, as std::map iterates
int orc_mem_compute_likelihood(void* h, const int* words, int nwords, const int* ids, int nids, int* outIds, float* out) {
    std::map<int, float> L;
    if (!((Memory*)h)->computeLikelihood(std::vector<int>(words, words + nwords), std::vector<int>(ids, ids + nids), L)) return -1;
    int n = 0;
    for (std::map<int, float>::iterator i = L.begin(); i != L.end(); ++i, ++n) { outIds[n] = i->first; out[n] = i->second; }
    return n;
}
void orc_adjust_likelihood(float* L, int n, float ratio) { adjustLikelihood(L, n, ratio); }
void orc_set_log10_double(int on) { g_log10_double = on != 0; }

// ---- BayesFilter
void* orc_bayes_create(const double* lc, int n, float virtualPlacePrior) {
    BayesFilter* b = new BayesFilter();
    b->setPredictionLC(lc, n);
    b->virtualPlacePrior = virtualPlacePrior;
    return b;
}
void orc_bayes_destroy(void* h) { delete (BayesFilter*)h; }
void orc_bayes_reset(void* h) { ((BayesFilter*)h)->reset(); }
// what Memory::getNeighborsId(id, LC.size() - 1, 0, false, false, true, true) returns from now on
void orc_bayes_set_neighbors(void* h, int id, const int* nbr, const int* margin, int n) {
    std::map<int, int>& g = ((BayesFilter*)h)->graph[id];
    g.clear();
    for (int i = 0; i < n; ++i) g[nbr[i]] = margin[i];
}
void orc_bayes_set_stm(void* h, const int* ids, int n) { BayesFilter* b = (BayesFilter*)h; b->stm.clear(); b->stm.insert(ids, ids + n); }
int orc_bayes_compute_posterior(void* h, const int* ids, const float* like, int m, int dense, float* out) {
    return ((BayesFilter*)h)->computePosterior(ids, like, m, dense, out);
}
// Rtabmap.cpp:2147-2158: walk from the highest id down, strict comparison, ids > 0 only; value = 1 - posterior of the first entry
void orc_bayes_hypothesis(const int* ids, const float* post, int m, int* out_id, float* out_value) {
    int best = 0; float v = 0.0f;
    for (int i = m - 1; i >= 0; --i) if (ids[i] > 0 && post[i] > v) { best = ids[i]; v = post[i]; }
    *out_id = best;
    *out_value = m > 0 ? 1 - post[0] : 0.0f;
}

// ---- the "generous" CPU baseline of bench.py ONLY (never a parity reference): Memory::computeLikelihood's TF-IDF sum over FLAT
// postings (word-major CSR: the postings of word k are [word_off[k], word_off[k + 1]) of post_sig / post_cnt) instead of the
// reference's std::map per word, with OpenMP over the query's words and one partial vector per thread -- what a CPU implementation
// that gave up the reference's containers could do.  Same arithmetic as Memory.cpp:2264-2277 per posting; the summation order
// differs (partial sums per thread).
void orc_flat_tfidf(const long long* word_off, const int* post_sig, const int* post_cnt, const int* ni, int n_sig, int n_words, float N, int threads,
                    float* out) {
    if (threads < 1) threads = 1;
    std::vector<float> part((size_t)threads * (size_t)n_sig, 0.0f);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
    for (int k = 0; k < n_words; ++k) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        float* acc = part.data() + (size_t)tid * (size_t)n_sig;
        const long long b = word_off[k], e = word_off[k + 1];
        const float nw = (float)(e - b);
        if (!(nw > 0.0f)) continue;
        const float logNnw = log10f(N / nw);
        if (logNnw == 0.0f) continue;
        for (long long p = b; p < e; ++p) {
            const int s = post_sig[p];
            const float nis = (float)ni[s];
            if (nis != 0.0f) acc[s] += ((float)post_cnt[p] * logNnw) / nis;
        }
    }
#pragma omp parallel for num_threads(threads)
    for (int s = 0; s < n_sig; ++s) {
        float v = 0.0f;
        for (int t = 0; t < threads; ++t) v += part[(size_t)t * (size_t)n_sig + s];
        out[s] = v;
    }
}

}  // extern "C"
