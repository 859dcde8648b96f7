// c_shim.cpp -- flat C entry points over VWDictionaryHip / MemoryHip so that the Python parity tests can drive the C++
// host mirror exactly as they drive the oracle (same call sequence, same argument meaning).  Not part of lcd.h.
#include <chrono>
#include <cstring>

#include "BayesFilterHip.h"
#include "DbLoaderHip.h"
#include "MemoryHip.h"
#include "RtabmapHip.h"

using namespace rtabmap_amd;

static ParametersMap make_params(int strategy, int incremental, float nndr, int together, const char* dictPath) {
    ParametersMap p;
    p["Kp/NNStrategy"] = std::to_string(strategy);
    p["Kp/IncrementalDictionary"] = incremental ? "true" : "false";
    p["Kp/NndrRatio"] = std::to_string(nndr);
    p["Kp/NewWordsComparedTogether"] = together ? "true" : "false";
    if (dictPath && dictPath[0]) p["Kp/DictionaryPath"] = dictPath;
    return p;
}
static Mat make_mat(const void* data, int rows, int cols, int type) { return Mat(rows, cols, type == 0 ? MAT_32F : MAT_8U, data); }

extern "C" {

void* hvwd_create(int strategy, int incremental, float nndr, int together, const char* dictPath, int device) {
    return new VWDictionaryHip(make_params(strategy, incremental, nndr, together, dictPath), device);
}
void hvwd_destroy(void* h) { delete (VWDictionaryHip*)h; }
int hvwd_available(void* h) { return ((VWDictionaryHip*)h)->isAvailable() ? 1 : 0; }
const char* hvwd_last_error(void* h) { return ((VWDictionaryHip*)h)->lastError().c_str(); }
int hvwd_add_new_words(void* h, const void* desc, int rows, int cols, int type, int sigId, int* out, int cap) {
    std::list<int> ids = ((VWDictionaryHip*)h)->addNewWords(make_mat(desc, rows, cols, type), sigId);
    int n = 0;
    for (std::list<int>::iterator i = ids.begin(); i != ids.end() && n < cap; ++i) out[n++] = *i;
    return (int)ids.size();
}
int hvwd_find_nn(void* h, const void* desc, int rows, int cols, int type, int* out) {
    std::vector<int> r = ((VWDictionaryHip*)h)->findNN(make_mat(desc, rows, cols, type));
    for (int i = 0; i < rows; ++i) out[i] = r[i];
    return rows;
}
void hvwd_update(void* h) { ((VWDictionaryHip*)h)->update(); }
void hvwd_add_word(void* h, int id, const void* desc, int cols, int type) {
    ((VWDictionaryHip*)h)->addWord(new VisualWord(id, make_mat(desc, 1, cols, type)));
}
int hvwd_add_word_ref(void* h, int wordId, int sigId) { return ((VWDictionaryHip*)h)->addWordRef(wordId, sigId) ? 1 : 0; }
void hvwd_remove_all_word_ref(void* h, int wordId, int sigId) { ((VWDictionaryHip*)h)->removeAllWordRef(wordId, sigId); }
int hvwd_get_unused_word_ids(void* h, int* out, int cap) {
    std::vector<int> v = ((VWDictionaryHip*)h)->getUnusedWordIds();
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}
void hvwd_delete_unused_words(void* h) { ((VWDictionaryHip*)h)->deleteUnusedWords(); }
void hvwd_clear(void* h) { ((VWDictionaryHip*)h)->clear(false); }
int hvwd_rebuild_engine(void* h) { return ((VWDictionaryHip*)h)->rebuildEngine() ? 1 : 0; }
// which: 0 visualWords, 1 notIndexed, 2 indexed, 3 totalActiveReferences, 4 lastIndexedWordId, 5 unused
long hvwd_stat(void* h, int which) {
    VWDictionaryHip* d = (VWDictionaryHip*)h;
    switch (which) {
        case 0: return (long)d->getVisualWords().size();
        case 1: return (long)d->getNotIndexedWordsCount();
        case 2: return (long)d->getIndexedWordsCount();
        case 3: return d->getTotalActiveReferences();
        case 4: return d->getLastIndexedWordId();
        case 5: return (long)d->getUnusedWordsSize();
    }
    return -1;
}
int hvwd_get_word_refs(void* h, int wordId, int* sigs, int* counts, int cap) {
    const VisualWord* vw = ((VWDictionaryHip*)h)->getWord(wordId);
    if (!vw) return -1;
    int n = 0;
    for (std::map<int, int>::const_iterator j = vw->getReferences().begin(); j != vw->getReferences().end(); ++j, ++n)
        if (n < cap) { sigs[n] = j->first; counts[n] = j->second; }
    return n;
}
int hvwd_index_ids(void* h, int* out, int cap) {
    std::vector<int> v = ((VWDictionaryHip*)h)->getIndexedWordIds();
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}
int hvwd_export_text(void* h, const char* refs, const char* desc) { ((VWDictionaryHip*)h)->exportDictionary(refs, desc); return 0; }

void* hmem_create(int strategy, int incremental, float nndr, int together, const char* dictPath, int device) {
    return new MemoryHip(make_params(strategy, incremental, nndr, together, dictPath), device);
}
void* hmem_create_stm(int strategy, int incremental, float nndr, int together, const char* dictPath, int device, int stmSize) {
    ParametersMap p = make_params(strategy, incremental, nndr, together, dictPath);
    p["Mem/STMSize"] = std::to_string(stmSize);
    return new MemoryHip(p, device);
}
void hmem_destroy(void* h) { delete (MemoryHip*)h; }
void* hmem_vwd(void* h) { return ((MemoryHip*)h)->getVWDictionary(); }
int hmem_update(void* h, const void* desc, int rows, int cols, int type, int nq, int* outIds) {
    std::vector<int> ids;
    const int id = ((MemoryHip*)h)->update(make_mat(desc, rows, cols, type), nq, ids);
    for (size_t i = 0; i < ids.size(); ++i) outIds[i] = ids[i];
    return id;
}
int hmem_add_signature(void* h, int id, const int* wordIds, int n) {
    return ((MemoryHip*)h)->addSignature(std::vector<int>(wordIds, wordIds + n), id);
}
void hmem_forget(void* h, int sigId) { ((MemoryHip*)h)->forget(sigId); }
// the statistics under the reference's names (MemoryHip::getStatistics): value of `name`, -1 if there is no such key; name == NULL: the
// number of keys; refresh != 0: the engine figures are read first (synchronises)
float hmem_statistic(void* h, const char* name, int refresh) {
    MemoryHip* m = (MemoryHip*)h;
    if (refresh) m->refreshEngineStatistics();
    if (!name) return (float)m->getStatistics().size();
    std::map<std::string, float>::const_iterator it = m->getStatistics().find(name);
    return it == m->getStatistics().end() ? -1.0f : it->second;
}
int hmem_set_engine_option(void* h, const char* key, long value) {
    VWDictionaryHip* d = ((MemoryHip*)h)->getVWDictionary();
    return d->engine() ? lcd_set_option(d->engine(), key, (int64_t)value) : -1;
}
int hmem_get_ni(void* h, int sigId) { return ((MemoryHip*)h)->getNi(sigId); }
long hmem_num_signatures(void* h) { return (long)((MemoryHip*)h)->signaturesSize(); }
int hmem_compute_likelihood(void* h, const int* words, int nwords, const int* ids, int nids, int* outIds, float* out) {
    std::map<int, float> L = ((MemoryHip*)h)->computeLikelihood(std::list<int>(words, words + nwords), std::list<int>(ids, ids + nids));
    int n = 0;
    for (std::map<int, float>::iterator i = L.begin(); i != L.end(); ++i, ++n) { outIds[n] = i->first; out[n] = i->second; }
    return n;
}

// The per-frame loop a caller of the reference's interface runs, timed inside C++ (bench.py's `cpp_interface_ms_per_step`): Memory::update
// (-> VWDictionary::addNewWords) of frame i, Memory::computeLikelihood of the new signature against every signature in memory
// (Rtabmap.cpp:2046-2117 builds that list), the oldest signature forgotten (WM -> LTM).  Returns the mean milliseconds per frame.
double hmem_time_loop(void* h, const void* descs, int n_frames, int rows, int cols, int type, int steps) {
    MemoryHip* m = (MemoryHip*)h;
    const size_t frame_bytes = (size_t)rows * cols * (type == 0 ? 4 : 1);
    std::vector<int> ids;
    double total = 0.0;
    for (int i = 0; i < steps + 2; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        const Mat d = make_mat((const char*)descs + (size_t)(i % n_frames) * frame_bytes, rows, cols, type);
        const int id = m->update(d, -1, ids);
        const std::vector<int> all = m->signatureIds();
        std::list<int> lst(all.begin(), all.end());
        const std::map<int, float> L = m->computeLikelihood(id, lst);
        if (!all.empty()) m->forget(all.front());
        if (L.empty()) return -1.0;
        if (i >= 2) total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return total / steps;
}

void hmem_set_device_frames(void* h, int on) { ((MemoryHip*)h)->setDeviceFrames(on != 0); }
// mean milliseconds a device-resident update() spent inside lcd_frame_host so far (-1: none ran)
double hmem_fast_frame_device_ms(void* h) {
    long long ns = 0, calls = 0;
    ((MemoryHip*)h)->getVWDictionary()->fastFrameStats(&ns, &calls);
    return calls ? 1e-6 * (double)ns / (double)calls : -1.0;
}
// n signatures of q words each (ids first_id, first_id + 1, ...) through addSignature, then ONE bulk registration on the device
int hmem_add_signatures_bulk(void* h, const int* words, int n, int q, int first_id) {
    MemoryHip* m = (MemoryHip*)h;
    for (int s = 0; s < n; ++s)
        if (m->addSignature(std::vector<int>(words + (size_t)s * q, words + (size_t)(s + 1) * q), first_id + s) != first_id + s) return -1;
    return m->flushReferencesBulk() ? n : -1;
}
// computeLikelihoodFlat of the signature update() has just created: returns the number of entries (-1: no result at hand)
int hmem_compute_likelihood_flat(void* h, int sigId, int* outIds, float* out, int cap) {
    std::vector<int> ids; std::vector<float> v;
    if (!((MemoryHip*)h)->computeLikelihoodFlat(sigId, ids, v)) return -1;
    for (size_t i = 0; i < ids.size() && (int)i < cap; ++i) { outIds[i] = ids[i]; out[i] = v[i]; }
    return (int)ids.size();
}
// Memory::computeLikelihood(signature id, ids) -- the overload Rtabmap::process calls (Rtabmap.cpp:2117): after update() it is answered
// from the frame's own device call
int hmem_compute_likelihood_of(void* h, int sigId, const int* ids, int nids, int* outIds, float* out) {
    std::map<int, float> L = ((MemoryHip*)h)->computeLikelihood(sigId, std::list<int>(ids, ids + nids));
    int n = 0;
    for (std::map<int, float>::iterator i = L.begin(); i != L.end(); ++i, ++n) { outIds[n] = i->first; out[n] = i->second; }
    return n;
}

// hmem_time_loop with the three ways a caller can take the likelihood (bench.py's cpp_interface_* keys), the caller's own list of ids
// kept from frame to frame (one push, one pop) instead of rebuilt: mode 0 = std::map by value (the reference's signature), 1 = into a
// caller-owned std::map updated in place, 2 = flat vectors.  out4 = mean ms per frame of {whole step, update, computeLikelihood, forget}.
int hmem_time_loop_modes(void* h, const void* descs, int n_frames, int rows, int cols, int type, int steps, int mode, double* out4) {
    MemoryHip* m = (MemoryHip*)h;
    const size_t frame_bytes = (size_t)rows * cols * (type == 0 ? 4 : 1);
    std::vector<int> ids;
    const std::vector<int> all0 = m->signatureIds();
    std::list<int> lst(all0.begin(), all0.end());
    std::map<int, float> keep;
    std::vector<int> fi; std::vector<float> fv;
    double t[4] = {0, 0, 0, 0};
    for (int i = 0; i < steps + 2; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        const Mat d = make_mat((const char*)descs + (size_t)(i % n_frames) * frame_bytes, rows, cols, type);
        const int id = m->update(d, -1, ids);
        lst.push_back(id);
        const auto t1 = std::chrono::steady_clock::now();
        size_t n_out = 0;
        if (mode == 0) { const std::map<int, float> L = m->computeLikelihood(id, lst); n_out = L.size(); }
        else if (mode == 1) { m->computeLikelihood(id, lst, keep); n_out = keep.size(); }
        else { if (!m->computeLikelihoodFlat(id, fi, fv)) return -1; n_out = fi.size(); }
        const auto t2 = std::chrono::steady_clock::now();
        if (!lst.empty()) { m->forget(lst.front()); lst.pop_front(); }
        const auto t3 = std::chrono::steady_clock::now();
        if (n_out == 0) return -1;
        if (i >= 2) {
            t[0] += std::chrono::duration<double, std::milli>(t3 - t0).count();
            t[1] += std::chrono::duration<double, std::milli>(t1 - t0).count();
            t[2] += std::chrono::duration<double, std::milli>(t2 - t1).count();
            t[3] += std::chrono::duration<double, std::milli>(t3 - t2).count();
        }
    }
    for (int k = 0; k < 4; ++k) out4[k] = t[k] / steps;
    return 0;
}

// the values the mirror's classes take when no parameter is given (tests/test_parameter_defaults.py compares them with the defaults of
// the reference's Parameters.h): out[0..7] = Kp/NndrRatio, Kp/IncrementalDictionary, Kp/NewWordsComparedTogether, Mem/STMSize,
// Rtabmap/LoopThr, Rtabmap/LoopRatio, Bayes/VirtualPlacePriorThr, Bayes/FullPredictionUpdate; then the Bayes/PredictionLC values.
// Returns the number of PredictionLC values.  No device call is made (the engine is created on first use).
int hparams_defaults(double* out, int cap) {
    RtabmapHip r((ParametersMap()));
    const VWDictionaryHip* d = r.getMemory()->getVWDictionary();
    out[0] = d->getNndrRatio(); out[1] = d->isIncremental() ? 1 : 0; out[2] = d->isNewWordsComparedTogether() ? 1 : 0;
    out[3] = r.getMemory()->getMaxStMemSize(); out[4] = r.getLoopThr(); out[5] = r.getLoopRatio();
    out[6] = r.getBayesFilter()->getVirtualPlacePrior(); out[7] = r.getBayesFilter()->isFullPredictionUpdate() ? 1 : 0;
    const std::vector<double>& lc = r.getBayesFilter()->getPredictionLC();
    for (size_t i = 0; i < lc.size() && 8 + (int)i < cap; ++i) out[8 + i] = lc[i];
    return (int)lc.size();
}

// uStr2Float as the mirror's parameter parsing uses it (VWDictionaryHip.h; the reference's: utilite UConversion.cpp)
float hutil_str2float(const char* s) { return uStr2Float(std::string(s ? s : "")); }

// ---- the database reader (DbLoaderHip.h): no device call is made by the hdb_* entries
struct HDb { DbLoaderHip db; DbDictionary dict; DbSignatures sigs; };
void* hdb_open(const char* path) {
    HDb* h = new HDb();
    if (!h->db.open(path ? path : "")) { delete h; return nullptr; }
    return h;
}
void hdb_close(void* h) { delete (HDb*)h; }
const char* hdb_version(void* h) { return ((HDb*)h)->db.version().c_str(); }
const char* hdb_last_error(void* h) { return ((HDb*)h)->db.lastError().c_str(); }
// out4: type (0 = CV_8U, 5 = CV_32F, -1 none), descriptor size, last word id, bytes of the rows; returns the number of words, -1 on error
int hdb_load_dictionary(void* h, int lastStateOnly, int* out4) {
    HDb* d = (HDb*)h;
    if (!d->db.loadDictionary(d->dict, lastStateOnly != 0)) return -1;
    out4[0] = d->dict.type; out4[1] = d->dict.cols; out4[2] = d->dict.lastWordId; out4[3] = (int)d->dict.rows.size();
    return (int)d->dict.wordIds.size();
}
void hdb_dictionary_copy(void* h, int* wordIds, unsigned char* rows) {
    HDb* d = (HDb*)h;
    if (!d->dict.wordIds.empty()) std::memcpy(wordIds, d->dict.wordIds.data(), d->dict.wordIds.size() * sizeof(int));
    if (!d->dict.rows.empty()) std::memcpy(rows, d->dict.rows.data(), d->dict.rows.size());
}
// returns the number of signatures (-1 on error); *nWords = total word entries
int hdb_load_signatures(void* h, int lastStateOnly, long long* nWords) {
    HDb* d = (HDb*)h;
    if (!d->db.loadSignatureWords(d->sigs, lastStateOnly != 0)) return -1;
    *nWords = (long long)d->sigs.wordIds.size();
    return (int)d->sigs.sigIds.size();
}
void hdb_signatures_copy(void* h, int* sigIds, long long* offsets, int* wordIds, int* ni) {
    HDb* d = (HDb*)h;
    const size_t n = d->sigs.sigIds.size();
    if (n) { std::memcpy(sigIds, d->sigs.sigIds.data(), n * sizeof(int)); std::memcpy(ni, d->sigs.ni.data(), n * sizeof(int)); }
    for (size_t i = 0; i < d->sigs.offsets.size(); ++i) offsets[i] = (long long)d->sigs.offsets[i];
    if (!d->sigs.wordIds.empty()) std::memcpy(wordIds, d->sigs.wordIds.data(), d->sigs.wordIds.size() * sizeof(int));
}
int hdb_get_ni(void* h, int nodeId) { return ((HDb*)h)->db.getNi(nodeId); }
int hdb_version_cmp(const char* a, const char* b) { return DbLoaderHip::versionCmp(a, b); }
// Memory::loadDataFromDb through the mirror (device: the dictionary's update() and ONE bulk registration)
int hmem_load_data_from_db(void* h, const char* path, int lastStateOnly) { return ((MemoryHip*)h)->loadDataFromDb(path ? path : "", lastStateOnly != 0); }
const char* hmem_load_error(void* h) { return ((MemoryHip*)h)->loadError().c_str(); }

// type 1: Memory::addLink of a global loop closure; type 0: a neighbour link as the database hands it to a replayed signature
int hmem_add_link(void* h, int from, int to, int type) {
    return ((MemoryHip*)h)->addLink(from, to, type == 0 ? MemoryHip::kNeighbor : MemoryHip::kGlobalClosure) ? 1 : 0;
}
int hmem_get_neighbors_id(void* h, int sigId, int maxGraphDepth, int* outIds, int* outMargins, int cap) {
    std::map<int, int> n = ((MemoryHip*)h)->getNeighborsId(sigId, maxGraphDepth);
    int k = 0;
    for (std::map<int, int>::iterator i = n.begin(); i != n.end(); ++i, ++k) if (k < cap) { outIds[k] = i->first; outMargins[k] = i->second; }
    return (int)n.size();
}
// which: 0 short-term memory, 1 working memory (the virtual place -1 included)
int hmem_ids(void* h, int which, int* out, int cap) {
    const std::set<int>& s = which == 0 ? ((MemoryHip*)h)->getStMem() : ((MemoryHip*)h)->getWorkingMem();
    int k = 0;
    for (std::set<int>::const_iterator i = s.begin(); i != s.end(); ++i, ++k) if (k < cap) out[k] = *i;
    return (int)s.size();
}

void* hbayes_create(const char* predictionLC, float virtualPlacePrior, int fullPredictionUpdate) {
    ParametersMap p;
    if (predictionLC && predictionLC[0]) p["Bayes/PredictionLC"] = predictionLC;
    p["Bayes/VirtualPlacePriorThr"] = std::to_string(virtualPlacePrior);
    p["Bayes/FullPredictionUpdate"] = fullPredictionUpdate ? "true" : "false";
    return new BayesFilterHip(p);
}
void hbayes_destroy(void* b) { delete (BayesFilterHip*)b; }
void hbayes_reset(void* b) { ((BayesFilterHip*)b)->reset(); }
void hbayes_set_prediction_lc(void* b, const char* s) { ((BayesFilterHip*)b)->setPredictionLC(s); }
int hbayes_get_prediction_lc(void* b, double* out, int cap) {
    const std::vector<double>& v = ((BayesFilterHip*)b)->getPredictionLC();
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}
// BayesFilter::computePosterior(memory, likelihood); returns the size of the posterior map, written to outIds / out; hyp = the
// highest hypothesis (id, value) as Rtabmap.cpp:2147-2158 reads it off
int hbayes_compute_posterior(void* b, void* mem, const int* ids, const float* values, int n, int* outIds, float* out, int cap, int* hypId,
                             float* hypValue) {
    std::map<int, float> L;
    for (int i = 0; i < n; ++i) L[ids[i]] = values[i];
    const std::map<int, float>& P = ((BayesFilterHip*)b)->computePosterior((MemoryHip*)mem, L);
    int k = 0;
    for (std::map<int, float>::const_iterator i = P.begin(); i != P.end(); ++i, ++k) if (k < cap) { outIds[k] = i->first; out[k] = i->second; }
    if (hypId) *hypId = ((BayesFilterHip*)b)->getHighestHypothesis().first;
    if (hypValue) *hypValue = ((BayesFilterHip*)b)->getHighestHypothesis().second;
    return (int)P.size();
}
const char* hbayes_last_error(void* b) { return ((BayesFilterHip*)b)->lastError().c_str(); }

void* hrtab_create(float loopThr, float loopRatio, int virtualPlaceLikelihoodRatio, int stmSize, const char* predictionLC, float virtualPlacePrior,
                   int device) {
    ParametersMap p = make_params(5, 1, 0.8f, 1, "");
    p["Rtabmap/LoopThr"] = std::to_string(loopThr);
    p["Rtabmap/LoopRatio"] = std::to_string(loopRatio);
    p["Rtabmap/VirtualPlaceLikelihoodRatio"] = std::to_string(virtualPlaceLikelihoodRatio);
    p["Mem/STMSize"] = std::to_string(stmSize);
    if (predictionLC && predictionLC[0]) p["Bayes/PredictionLC"] = predictionLC;
    p["Bayes/VirtualPlacePriorThr"] = std::to_string(virtualPlacePrior);
    return new RtabmapHip(p, device);
}
void hrtab_destroy(void* r) { delete (RtabmapHip*)r; }
void* hrtab_memory(void* r) { return ((RtabmapHip*)r)->getMemory(); }
// Rtabmap::process for one frame of descriptors; out4 = {last location id, highest hypothesis id, loop closure id, processed}
int hrtab_process(void* r, const void* desc, int rows, int cols, int type, int* out4, float* outValues2) {
    RtabmapHip* R = (RtabmapHip*)r;
    const bool ok = R->process(make_mat(desc, rows, cols, type));
    out4[0] = R->getLastLocationId(); out4[1] = R->getHighestHypothesisId(); out4[2] = R->getLoopClosureId(); out4[3] = ok ? 1 : 0;
    outValues2[0] = R->getHighestHypothesisValue(); outValues2[1] = R->getLoopClosureValue();
    return ok ? 1 : 0;
}
// which: 0 raw likelihood, 1 adjusted likelihood, 2 posterior of the last frame
int hrtab_vector(void* r, int which, int* outIds, float* out, int cap) {
    RtabmapHip* R = (RtabmapHip*)r;
    const std::map<int, float>& m = which == 0 ? R->getRawLikelihood() : which == 1 ? R->getLikelihood() : R->getPosterior();
    int k = 0;
    for (std::map<int, float>::const_iterator i = m.begin(); i != m.end(); ++i, ++k) if (k < cap) { outIds[k] = i->first; out[k] = i->second; }
    return (int)m.size();
}
int hrtab_last_word_ids(void* r, int* out, int cap) {
    const std::vector<int>& v = ((RtabmapHip*)r)->getLastWordIds();
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

}  // extern "C"
