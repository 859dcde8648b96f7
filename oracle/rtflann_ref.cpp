// oracle/rtflann_ref.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" harness around the reference's OWN vendored FLANN fork, compiled in place from
// /root/reference/corelib/src/rtflann (no reference source is copied into this repository).
// It is what oracle/_ref/librtflann_ref.so is built from (see oracle/Makefile) and gives the tests
//   * the real reference arithmetic for L2 / L1 / Hamming      (rtflann/algorithms/dist.h:150,211,555)
//   * the real exact 2-NN with its tie-break                   (linear_index.h:129-144, result_set.h:151-171)
//   * the reference's default approximate kd-tree (speed only) (kdtree_index.h, FlannIndex.cpp:298)
//   * the statistics helpers Rtabmap::adjustLikelihood calls   (utilite UMath.h: uMean(list) :419-432, uVariance(list, mean) :512-526)
//     and its number parser                                    (utilite UConversion.cpp: uStr2Float)
// so that the hand-written restatement in lcd_oracle.cpp can be pinned against reference code, and so
// that bench.py can time the reference CPU search on the GPU box's host cores.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <list>
#include <map>
#include <string>
#include <vector>
#include "rtflann/flann.hpp"
#include "rtabmap/utilite/UConversion.h"
#include "rtabmap/utilite/UMath.h"
#include "rtabmap/utilite/UStl.h"

namespace {

enum { METRIC_L2 = 0, METRIC_HAMMING = 1, METRIC_L1 = 2 };
enum { ALGO_LINEAR = 0, ALGO_KDTREE = 1 };

struct RefIndexBase {
    virtual ~RefIndexBase() {}
    virtual void add(const void* rows, size_t n, float rebuild) = 0;
    virtual void remove(size_t id) = 0;
    virtual int knn(const void* q, size_t nq, size_t k, int checks, int cores, size_t* idx, float* dist) = 0;
    virtual size_t size() const = 0;
};

template <typename Dist>
struct RefIndex : RefIndexBase {
    typedef typename Dist::ElementType E;
    typedef typename Dist::ResultType R;
    size_t dim;
    // rtflann keeps pointers into the caller's matrices (as FlannIndex.cpp:560 "addedDescriptors_" does),
    // so every block of rows handed to it is kept alive here.
    std::vector<std::vector<E>*> blocks;
    rtflann::Index<Dist>* index;

    RefIndex(const void* rows, size_t n, size_t dim_, int algo, int trees) : dim(dim_), index(0) {
        std::vector<E>* b = new std::vector<E>((const E*)rows, (const E*)rows + n * dim);
        blocks.push_back(b);
        rtflann::Matrix<E> m(b->data(), n, dim);
        if (algo == ALGO_KDTREE) index = new rtflann::Index<Dist>(m, rtflann::KDTreeIndexParams(trees));
        else                     index = new rtflann::Index<Dist>(m, rtflann::LinearIndexParams());
        index->buildIndex();
    }
    ~RefIndex() { delete index; for (auto* b : blocks) delete b; }
    void add(const void* rows, size_t n, float rebuild) override {
        std::vector<E>* b = new std::vector<E>((const E*)rows, (const E*)rows + n * dim);
        blocks.push_back(b);
        index->addPoints(rtflann::Matrix<E>(b->data(), n, dim), rebuild);
    }
    void remove(size_t id) override { index->removePoint(id); }
    size_t size() const override { return index->size(); }
    int knn(const void* q, size_t nq, size_t k, int checks, int cores, size_t* idx, float* dist) override {
        rtflann::Matrix<E> qm((E*)q, nq, dim);
        rtflann::Matrix<size_t> im(idx, nq, k);
        std::vector<R> d(nq * k);
        rtflann::Matrix<R> dm(d.data(), nq, k);
        for (size_t i = 0; i < nq * k; ++i) { idx[i] = (size_t)-1; d[i] = (R)0; dist[i] = -1.0f; }
        rtflann::SearchParams p(checks, 0.0f, true);   // FlannIndex.cpp:720
        p.cores = cores;                               // reference leaves this at 1 (params.h:62-69)
        int found = index->knnSearch(qm, im, dm, k, p);
        // rtflann leaves unfilled slots untouched: flag them (idx==-1 -> dist -1) for the caller.
        for (size_t i = 0; i < nq * k; ++i) dist[i] = (idx[i] == (size_t)-1) ? -1.0f : (float)d[i];
        return found;
    }
};

}  // namespace

extern "C" {

void* ref_index_create(int metric, int algo, int trees, const void* rows, size_t n, size_t dim) {
    try {
        if (metric == METRIC_L2)      return new RefIndex<rtflann::L2<float> >(rows, n, dim, algo, trees);
        if (metric == METRIC_L1)      return new RefIndex<rtflann::L1<float> >(rows, n, dim, algo, trees);
        if (metric == METRIC_HAMMING) return new RefIndex<rtflann::Hamming<unsigned char> >(rows, n, dim, algo, trees);
    } catch (...) {}
    return 0;
}
void ref_index_destroy(void* h) { delete (RefIndexBase*)h; }
void ref_index_add(void* h, const void* rows, size_t n, float rebuild) { ((RefIndexBase*)h)->add(rows, n, rebuild); }
void ref_index_remove(void* h, size_t id) { ((RefIndexBase*)h)->remove(id); }
size_t ref_index_size(void* h) { return ((RefIndexBase*)h)->size(); }
int ref_index_knn(void* h, const void* q, size_t nq, size_t k, int checks, int cores, size_t* idx, float* dist) {
    return ((RefIndexBase*)h)->knn(q, nq, k, checks, cores, idx, dist);
}
// direct access to the reference distance functors
float ref_dist_l2(const float* a, const float* b, size_t n) { return rtflann::L2<float>()(a, b, n); }
float ref_dist_l1(const float* a, const float* b, size_t n) { return rtflann::L1<float>()(a, b, n); }
unsigned ref_dist_hamming(const unsigned char* a, const unsigned char* b, size_t n) {
    return rtflann::Hamming<unsigned char>()(a, b, n);
}
// the reference's own templates on a std::list<float>, as Rtabmap::adjustLikelihood (Rtabmap.cpp:5717-5720) instantiates them
float ref_umean_list(const float* v, size_t n) { std::list<float> l(v, v + n); return uMean(l); }
float ref_uvariance_list(const float* v, size_t n, float mean) { std::list<float> l(v, v + n); return uVariance(l, mean); }
float ref_ustr2float(const char* s) { return uStr2Float(std::string(s)); }
// The text dictionary reader of VWDictionary::setFixedDictionary (VWDictionary.cpp:189-243) and the descriptor writer of
// exportDictionary (:1650-1690), statement by statement on a std::map of float rows instead of VisualWord / cv::Mat, with the reference's
// OWN tokenisers and number parser (uSplitNumChar, uIsDigit, uSplit, uStr2Float): what the restated loaders must reproduce on awkward
// files (repeated spaces, decimal commas, short lines, repeated ids).  Returns the number of words written, < 0 on error.
int ref_dictionary_text_roundtrip(const char* in, const char* out) {
    std::ifstream file;
    file.open(in, std::ifstream::in);
    if (!file.good()) return -1;
    std::string str;
    std::list<std::string> strList;
    std::getline(file, str);
    strList = uSplitNumChar(str);
    int dimension = 0;
    for (std::list<std::string>::iterator iter = strList.begin(); iter != strList.end(); ++iter) {
        if (uIsDigit(iter->at(0))) { dimension = std::atoi(iter->c_str()); break; }
    }
    if (dimension <= 0 || dimension > 1000) return -2;
    std::map<int, std::vector<float> > words;
    while (file.good()) {
        std::getline(file, str);
        strList = uSplit(str);
        if ((int)strList.size() == dimension + 1) {
            std::list<std::string>::iterator iter = strList.begin();
            int id = std::atoi(iter->c_str());
            std::vector<float> descriptor(dimension);
            ++iter;
            int i = 0;
            for (; i < dimension && iter != strList.end(); ++i, ++iter) descriptor[i] = uStr2Float(*iter);
            words.insert(words.end(), std::pair<int, std::vector<float> >(id, descriptor));   // (an id seen before is ignored, as by _visualWords.insert)
        }
    }
    file.close();
    FILE* fo = fopen(out, "w");
    if (!fo) return -3;
    if (words.size() == 0) fprintf(fo, "WordID Descriptors...\n");
    else fprintf(fo, "WordID Descriptors...%d\n", (int)words.begin()->second.size());
    for (std::map<int, std::vector<float> >::const_iterator iter = words.begin(); iter != words.end(); ++iter) {
        fprintf(fo, "%d ", iter->first);
        for (size_t i = 0; i < iter->second.size(); i++) fprintf(fo, "%f ", iter->second[i]);
        fprintf(fo, "\n");
    }
    fclose(fo);
    return (int)words.size();
}
// the version comparison of the database driver's schema switches (utilite UStl.h:717-790)
int ref_ustrnumcmp(const char* a, const char* b) { return uStrNumCmp(std::string(a), std::string(b)); }

}  // extern "C"
