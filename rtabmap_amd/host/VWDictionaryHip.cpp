// VWDictionaryHip.cpp -- see VWDictionaryHip.h.  Bookkeeping mirrors the reference line for line in behaviour
// (citations: /root/reference/corelib/src/VWDictionary.cpp); every search / scoring step is an lcd_* call.
#include "VWDictionaryHip.h"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rtabmap_amd {

const int VWDictionaryHip::ID_START = 1;
const int VWDictionaryHip::ID_INVALID = 0;

static void logError(const char* fmt, const char* a = "") { fprintf(stderr, "[ERROR] VWDictionaryHip: "); fprintf(stderr, fmt, a); fprintf(stderr, "\n"); }

Mat::Mat(int r, int c, int t, const void* src) : rows(r), cols(c), type_(t) {
    data.resize((size_t)r * c * (t == MAT_32F ? 4 : 1));
    if (src && !data.empty()) std::memcpy(data.data(), src, data.size());
}

// ---------------------------------------------------------------------------------------------- VisualWord
VisualWord::VisualWord(int id, const Mat& descriptor, int signatureId)
    : _id(id), _descriptor(descriptor), _saved(false), _totalReferences(0), _head(0), _mapValid(true) {
    if (signatureId) addRef(signatureId);
}
// the posting of a signature, or end(): the two ends first (the newest signature has the largest id, the one Memory forgets the smallest --
// where a stream's calls land), a binary search otherwise
std::vector<std::pair<int, int> >::iterator VisualWord::findRef(int signatureId) {
    if (_head == _refs.size()) return _refs.end();
    if (_refs.back().first == signatureId) return _refs.end() - 1;
    if (_refs[_head].first == signatureId) return _refs.begin() + _head;
    std::vector<std::pair<int, int> >::iterator it = std::lower_bound(_refs.begin() + _head, _refs.end(), std::pair<int, int>(signatureId, 0));
    return (it != _refs.end() && it->first == signatureId) ? it : _refs.end();
}
void VisualWord::addRef(int signatureId) {   // VisualWord.cpp:51-60: _references[signatureId] += 1 (inserted when missing), ++_totalReferences
    if (_head == _refs.size() || _refs.back().first < signatureId) _refs.push_back(std::pair<int, int>(signatureId, 1));
    else {
        std::vector<std::pair<int, int> >::iterator it = findRef(signatureId);
        if (it != _refs.end()) it->second += 1;
        else _refs.insert(std::lower_bound(_refs.begin() + _head, _refs.end(), std::pair<int, int>(signatureId, 0)), std::pair<int, int>(signatureId, 1));
    }
    ++_totalReferences;
    _mapValid = false;
}
int VisualWord::removeAllRef(int signatureId) {   // VisualWord.cpp:62-70
    int removed = 0;
    std::vector<std::pair<int, int> >::iterator it = findRef(signatureId);
    if (it != _refs.end()) {
        removed = it->second;
        if ((size_t)(it - _refs.begin()) == _head) {
            ++_head;
            if (_head == _refs.size()) { _refs.clear(); _head = 0; }
            else if (_head > 64 && _head > _refs.size() - _head) { _refs.erase(_refs.begin(), _refs.begin() + _head); _head = 0; }
        } else _refs.erase(it);
        _mapValid = false;
    }
    _totalReferences -= removed;
    return removed;
}
const std::map<int, int>& VisualWord::getReferences() const {
    if (!_mapValid) {
        _references.clear();
        for (size_t k = _head; k < _refs.size(); ++k) _references.insert(_references.end(), _refs[k]);
        _mapValid = true;
    }
    return _references;
}

// ---------------------------------------------------------------------------------------------- VWDictionaryHip
static bool parseBool(const std::string& s) { return !(s == "false" || s == "0" || s == "False" || s == "FALSE"); }

VWDictionaryHip::VWDictionaryHip(const ParametersMap& parameters, int device)
    : _totalActiveReferences(0), _incrementalDictionary(true), _nndrRatio(0.8f), _newWordsComparedTogether(true), _lastWordId(0),
      _strategy(kNNBruteForceHIP), _device(device), _engine(nullptr), _engineType(-1), _engineCols(0) {
    this->parseParameters(parameters);
}

VWDictionaryHip::~VWDictionaryHip() {
    this->clear(false);
    if (_engine) lcd_destroy(_engine);
}

void VWDictionaryHip::parseParameters(const ParametersMap& p) {   // VWDictionary.cpp:88-132
    ParametersMap::const_iterator it;
    if ((it = p.find("Kp/NndrRatio")) != p.end()) _nndrRatio = uStr2Float(it->second);
    if ((it = p.find("Kp/NewWordsComparedTogether")) != p.end()) _newWordsComparedTogether = parseBool(it->second);
    bool incremental = _incrementalDictionary;
    if ((it = p.find("Kp/IncrementalDictionary")) != p.end()) incremental = parseBool(it->second);
    if ((it = p.find("Kp/NNStrategy")) != p.end()) {
        const int s = atoi(it->second.c_str());
        // only the device strategy exists here; the reference's own values are accepted and run on the device too
        _strategy = (s >= 0 && s < kNNUndef) ? (NNStrategy)s : kNNBruteForceHIP;
    }
    std::string path = _dictionaryPath;
    if ((it = p.find("Kp/DictionaryPath")) != p.end()) path = it->second;
    if (incremental) this->setIncrementalDictionary();
    else this->setFixedDictionary(path);
}

void VWDictionaryHip::setIncrementalDictionary() {   // VWDictionary.cpp:134-157
    if (!_incrementalDictionary) { _incrementalDictionary = true; if (_visualWords.size()) fprintf(stderr, "[WARN] incremental dictionary set: already loaded words are kept\n"); }
    _dictionaryPath = "";
    _newDictionaryPath = "";
}

bool VWDictionaryHip::ensureEngine(int type, int cols) const {
    if (_engine) {
        if (_engineType == type && _engineCols == cols) return true;
        _lastError = "descriptor type/size differs from the one the device vocabulary was created with";
        return false;
    }
    lcd_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.device = _device;
    cfg.dtype = type == MAT_32F ? LCD_F32 : LCD_U8;
    cfg.dim = cols;
    const int rc = lcd_create(&cfg, &_engine);
    if (rc != LCD_OK) {
        _engine = nullptr;
        _lastError = "lcd_create failed: no gfx950 device (there is no CPU fallback in this class)";
        logError("%s", _lastError.c_str());
        return false;
    }
    _engineType = type;
    _engineCols = cols;
    return true;
}

// fixed dictionary, text format (VWDictionary.cpp:181-257): "WordID Descriptors...<dim>" then "id v0 .. v(dim-1)"
void VWDictionaryHip::setFixedDictionary(const std::string& dictionaryPath) {
    if (!dictionaryPath.empty() && (dictionaryPath != _dictionaryPath || _visualWords.empty())) {
        FILE* f = fopen(dictionaryPath.c_str(), "r");
        if (!f) { logError("Cannot open dictionary file \"%s\"", dictionaryPath.c_str()); }
        else {
            this->clear(false);
            std::vector<std::string> lines; std::string cur; int c;
            while ((c = fgetc(f)) != EOF) { if (c == '\n') { lines.push_back(cur); cur.clear(); } else cur.push_back((char)c); }
            if (!cur.empty()) lines.push_back(cur);
            fclose(f);
            int dim = 0;
            for (size_t li = 0; li < lines.size(); ++li) {
                const std::string& s = lines[li];
                if (li == 0) {   // uSplitNumChar: the first all-digit token is the dimension
                    size_t p = 0;
                    while (p < s.size() && !isdigit((unsigned char)s[p])) ++p;
                    size_t e = p;
                    while (e < s.size() && isdigit((unsigned char)s[e])) ++e;
                    dim = e > p ? atoi(s.substr(p, e - p).c_str()) : 0;
                    if (dim <= 0 || dim > 1000) { logError("Cannot parse the descriptor size from the header of \"%s\"", dictionaryPath.c_str()); break; }
                    continue;
                }
                std::vector<std::string> tok; std::string t;
                for (size_t p = 0; p <= s.size(); ++p) {
                    if (p == s.size() || s[p] == ' ') { if (!t.empty()) tok.push_back(t); t.clear(); } else t.push_back(s[p]);
                }
                if ((int)tok.size() != dim + 1) continue;   // "Cannot parse line" warning in the reference
                const int id = atoi(tok[0].c_str());
                std::vector<float> v(dim);
                for (int k = 0; k < dim; ++k) v[k] = uStr2Float(tok[k + 1]);   // (',' or '.', classic locale: UConversion.cpp:138)
                VisualWord* vw = new VisualWord(id, Mat(1, dim, MAT_32F, v.data()), 0);
                vw->setSaved(true);
                _visualWords.insert(_visualWords.end(), std::pair<int, VisualWord*>(id, vw));
                indexWord(vw);
                _notIndexedWords.insert(_notIndexedWords.end(), id);
                _unusedWords.insert(_unusedWords.end(), std::pair<int, VisualWord*>(id, vw));
                if (_lastWordId < id) _lastWordId = id;
            }
            _incrementalDictionary = false;
            this->update();
            _incrementalDictionary = false;
        }
    } else if (dictionaryPath.empty() && _visualWords.empty()) {
        // fixed dictionary without a path: words must come from addWord() (database) before use
    }
    if (_incrementalDictionary) _incrementalDictionary = false;
    _dictionaryPath = dictionaryPath;
    _newDictionaryPath = dictionaryPath;
}

int VWDictionaryHip::getLastIndexedWordId() const {
    return _mapIndexId.size() ? _mapIndexId.rbegin()->second : 0;
}

std::vector<int> VWDictionaryHip::getIndexedWordIds() const {
    std::vector<int> v;
    for (std::map<int, int>::const_iterator i = _mapIndexId.begin(); i != _mapIndexId.end(); ++i) v.push_back(i->second);
    return v;
}

// ---------------------------------------------------------------------------------------------- recovery
bool VWDictionaryHip::rebuildEngine() {
    if (_engine) { lcd_destroy(_engine); _engine = nullptr; }
    _deviceSigs.clear();
    _slotSig.clear();
    _sigSlot.clear();
    _deviceRows.clear();                                             // (their words are in _notIndexedWords: the next update() appends them)
    _dirtySigs.clear();
    for (std::map<int, std::vector<int> >::const_iterator s = _sigWords.begin(); s != _sigWords.end(); ++s) _dirtySigs.insert(s->first);
    if (_visualWords.empty()) return true;                           // nothing to replay: the engine is created by the next update()
    const Mat& d0 = _visualWords.begin()->second->getDescriptor();
    if (!ensureEngine(d0.type(), d0.cols)) return false;
    // the indexed words in ROW order (the distance tie-break): _mapIndexId is index -> word id
    std::vector<int32_t> ids;
    std::vector<unsigned char> rows;
    for (std::map<int, int>::const_iterator i = _mapIndexId.begin(); i != _mapIndexId.end(); ++i) {
        std::map<int, VisualWord*>::const_iterator w = _visualWords.find(i->second);
        if (w == _visualWords.end()) continue;
        const Mat& d = w->second->getDescriptor();
        rows.insert(rows.end(), d.data.begin(), d.data.end());
        ids.push_back(i->second);
    }
    if (!ids.empty() && lcd_vocab_append(_engine, rows.data(), (int)ids.size(), ids.data()) != LCD_OK) {
        _lastError = lcd_last_error(_engine);
        logError("%s", _lastError.c_str());
        return false;
    }
    // row indices are dense again (removed rows are gone); words waiting for update() stay in _notIndexedWords
    std::map<int, int> idx;
    _mapIdIndex.clear();
    int r = 0;
    for (size_t k = 0; k < ids.size(); ++k, ++r) { idx.insert(idx.end(), std::pair<int, int>(r, ids[k])); _mapIdIndex[ids[k]] = r; }
    _mapIndexId.swap(idx);
    _removedIndexedWords.clear();
    return true;                                                     // the references follow with the next flushReferences / computeLikelihood
}

// ---------------------------------------------------------------------------------------------- update()  :475-701
void VWDictionaryHip::update() {
    if (!_incrementalDictionary && !_notIndexedWords.size()) return;   // fixed dictionary already indexed (:482-489)
    if (_notIndexedWords.size() || _visualWords.size() == 0 || _removedIndexedWords.size()) {
        if (_visualWords.size() && !ensureEngine(_visualWords.begin()->second->getDescriptor().type(),
                                                 _visualWords.begin()->second->getDescriptor().cols)) {
            return;   // nothing is marked indexed: addNewWords/findNN will report the failure
        }
        // stage the not-indexed rows (ascending id, std::set order)
        std::vector<int32_t> newIds;
        std::vector<unsigned char> newRows;
        std::vector<int32_t> allNew(_notIndexedWords.begin(), _notIndexedWords.end());   // ascending id: the row order of the append branch
        for (std::set<int>::iterator it = _notIndexedWords.begin(); it != _notIndexedWords.end(); ++it) {
            if (_deviceRows.count(*it)) continue;                        // a row already (addNewWordsAndScore appended it on the device)
            std::map<int, VisualWord*>::iterator w = _visualWords.find(*it);
            if (w == _visualWords.end()) continue;
            const Mat& d = w->second->getDescriptor();
            newRows.insert(newRows.end(), d.data.begin(), d.data.end());
            newIds.push_back(*it);
        }
        // rows the device appended lie behind the indexed ones in creation (= ascending id) order; words that still have to be sent would
        // land behind THEM, which is the append branch's order only when their ids are higher -- otherwise the rebuild branch sorts
        const bool mixed = !_deviceRows.empty() && !newIds.empty() && newIds.front() < *_deviceRows.rbegin();
        if (_notIndexedWords.size() && _removedIndexedWords.size() == 0 && _visualWords.size() && !mixed) {
            // brute-force append branch (:571-609)
            if (!newIds.empty() && lcd_vocab_append(_engine, newRows.data(), (int)newIds.size(), newIds.data()) != LCD_OK) { _lastError = lcd_last_error(_engine); logError("%s", _lastError.c_str()); return; }
            int i = (int)_mapIndexId.size() ? _mapIndexId.rbegin()->first + 1 : 0;
            for (size_t k = 0; k < allNew.size(); ++k, ++i) {
                _mapIndexId.insert(_mapIndexId.end(), std::pair<int, int>(i, allNew[k]));
                _mapIdIndex.insert(std::pair<int, int>(allNew[k], i));
            }
        } else {
            // full rebuild in ascending word id (:610-690) -- done on the device: tombstone the removed rows, append the
            // new ones, then compact + order by id with one gather (the reference re-creates the whole matrix on the host)
            _mapIndexId.clear();
            _mapIdIndex.clear();
            if (_engine) {
                int rc = LCD_OK;
                if (_visualWords.empty()) rc = lcd_vocab_clear(_engine);
                else {
                    std::vector<int32_t> rem(_removedIndexedWords.begin(), _removedIndexedWords.end());
                    if (!rem.empty()) rc = lcd_vocab_remove(_engine, rem.data(), (int)rem.size());
                    if (rc == LCD_OK && !newIds.empty()) rc = lcd_vocab_append(_engine, newRows.data(), (int)newIds.size(), newIds.data());
                    if (rc == LCD_OK) rc = lcd_vocab_rebuild(_engine);
                }
                if (rc != LCD_OK) { _lastError = lcd_last_error(_engine); logError("%s", _lastError.c_str()); return; }
            }
            int i = 0;
            for (std::map<int, VisualWord*>::const_iterator it = _visualWords.begin(); it != _visualWords.end(); ++it, ++i) {
                _mapIndexId.insert(_mapIndexId.end(), std::pair<int, int>(i, it->first));
                _mapIdIndex.insert(_mapIdIndex.end(), std::pair<int, int>(it->first, i));
            }
        }
    }
    _notIndexedWords.clear();
    _removedIndexedWords.clear();
    _deviceRows.clear();
}

void VWDictionaryHip::clear(bool printWarningsIfNotEmpty) {   // :843-873
    if (printWarningsIfNotEmpty && (_visualWords.size() && _incrementalDictionary))
        fprintf(stderr, "[WARN] Visual dictionary would be already empty here (%d words still in dictionary).\n", (int)_visualWords.size());
    for (std::map<int, VisualWord*>::iterator i = _visualWords.begin(); i != _visualWords.end(); ++i) delete i->second;
    _visualWords.clear();
    _byId.clear();
    _notIndexedWords.clear();
    _removedIndexedWords.clear();
    _totalActiveReferences = 0;
    _lastWordId = 0;
    _mapIndexId.clear();
    _mapIdIndex.clear();
    _unusedWords.clear();
    if (_engine) {
        lcd_vocab_clear(_engine);
        for (std::set<int>::iterator s = _deviceSigs.begin(); s != _deviceSigs.end(); ++s) { lcd_sig_remove(_engine, *s); slotRetire(*s); }
    }
    _sigWords.clear();
    _dirtySigs.clear();
    _deviceSigs.clear();
    _deviceRows.clear();
}

bool VWDictionaryHip::addWordRef(int wordId, int signatureId) {   // :880-897
    std::map<int, VisualWord*>::iterator it = _visualWords.find(wordId);
    if (it != _visualWords.end()) {
        it->second->addRef(signatureId);
        _totalActiveReferences += 1;
        _unusedWords.erase(wordId);
        _sigWords[signatureId].push_back(wordId);
        _dirtySigs.insert(signatureId);
        return true;
    }
    fprintf(stderr, "[WARN] Not found word %d (dict size=%d)\n", wordId, (int)_visualWords.size());
    return false;
}

void VWDictionaryHip::removeAllWordRef(int wordId, int signatureId) {   // :899-911
    std::map<int, VisualWord*>::iterator it = _visualWords.find(wordId);
    if (it != _visualWords.end()) {
        _totalActiveReferences -= it->second->removeAllRef(signatureId);
        if (it->second->getReferencesCount() == 0) _unusedWords.insert(std::pair<int, VisualWord*>(wordId, it->second));
        std::map<int, std::vector<int> >::iterator s = _sigWords.find(signatureId);
        if (s != _sigWords.end()) {
            s->second.erase(std::remove(s->second.begin(), s->second.end(), wordId), s->second.end());
            _dirtySigs.insert(signatureId);
        }
    }
}

// Memory::disableWordsRef (Memory.cpp:6877-6897): removeAllWordRef(word, signature) for every unique word of a signature that leaves the
// memory, as ONE call -- the same bookkeeping per word, the signature's entry of the device mirror dropped once instead of searched and
// compacted per word (350 map finds + 350 linear passes over its 500 words per forgotten signature)
void VWDictionaryHip::removeAllWordRefs(const std::set<int>& wordIds, int signatureId) {
    bool any = false;
    for (std::set<int>::const_iterator k = wordIds.begin(); k != wordIds.end(); ++k) {
        VisualWord* vw = lookupWord(*k);
        if (!vw) continue;
        _totalActiveReferences -= vw->removeAllRef(signatureId);
        if (vw->getReferencesCount() == 0) _unusedWords.insert(std::pair<int, VisualWord*>(*k, vw));
        any = true;
    }
    std::map<int, std::vector<int> >::iterator s = _sigWords.find(signatureId);
    if (s != _sigWords.end() && any) {
        std::vector<int>& v = s->second;
        v.erase(std::remove_if(v.begin(), v.end(), [&](int w) { return wordIds.count(w) != 0 && lookupWord(w) != nullptr; }), v.end());
        _dirtySigs.insert(signatureId);
    }
}

void VWDictionaryHip::addWord(VisualWord* vw) {   // :1554-1573
    if (!vw) return;
    _visualWords.insert(_visualWords.end(), std::pair<int, VisualWord*>(vw->id(), vw));
    indexWord(vw);
    _notIndexedWords.insert(_notIndexedWords.end(), vw->id());
    if (vw->getReferencesCount()) {
        int s = 0;
        for (std::map<int, int>::const_iterator i = vw->getReferences().begin(); i != vw->getReferences().end(); ++i) {
            s += i->second;
            for (int k = 0; k < i->second; ++k) _sigWords[i->first].push_back(vw->id());
            _dirtySigs.insert(i->first);
        }
        _totalActiveReferences += s;
    } else {
        _unusedWords.insert(_unusedWords.end(), std::pair<int, VisualWord*>(vw->id(), vw));
    }
    if (_lastWordId < vw->id()) _lastWordId = vw->id();
}

const VisualWord* VWDictionaryHip::getWord(int id) const {
    std::map<int, VisualWord*>::const_iterator it = _visualWords.find(id);
    return it == _visualWords.end() ? 0 : it->second;
}
VisualWord* VWDictionaryHip::getUnusedWord(int id) const {
    std::map<int, VisualWord*>::const_iterator it = _unusedWords.find(id);
    return it == _unusedWords.end() ? 0 : it->second;
}
std::vector<VisualWord*> VWDictionaryHip::getUnusedWords() const {
    std::vector<VisualWord*> v;
    for (std::map<int, VisualWord*>::const_iterator i = _unusedWords.begin(); i != _unusedWords.end(); ++i) v.push_back(i->second);
    return v;
}
std::vector<int> VWDictionaryHip::getUnusedWordIds() const {
    std::vector<int> v;
    for (std::map<int, VisualWord*>::const_iterator i = _unusedWords.begin(); i != _unusedWords.end(); ++i) v.push_back(i->first);
    return v;
}
void VWDictionaryHip::removeWords(const std::vector<VisualWord*>& words) {   // :1595-1607
    for (unsigned int i = 0; i < words.size(); ++i) {
        _visualWords.erase(words[i]->id());
        if (words[i]->id() >= 0 && (size_t)words[i]->id() < _byId.size()) _byId[(size_t)words[i]->id()] = nullptr;
        _unusedWords.erase(words[i]->id());
        const bool onDevice = _deviceRows.erase(words[i]->id()) != 0;  // created by a device-resident frame: a vocabulary row although update() has not run
        if (_notIndexedWords.erase(words[i]->id()) == 0 || onDevice) _removedIndexedWords.insert(words[i]->id());
    }
}
void VWDictionaryHip::deleteUnusedWords() {   // :1609-1617
    std::vector<VisualWord*> unusedWords = getUnusedWords();
    removeWords(unusedWords);
    for (unsigned int i = 0; i < unusedWords.size(); ++i) delete unusedWords[i];
}

// ---------------------------------------------------------------------------------------------- addNewWords  :913-1229
std::list<int> VWDictionaryHip::addNewWords(const Mat& descriptorsIn, int signatureId) {
    std::list<int> wordIds;
    if (descriptorsIn.rows == 0 || descriptorsIn.cols == 0) { logError("Descriptors size is null!"); return wordIds; }                      // :920
    if (!_incrementalDictionary && _visualWords.empty()) { logError("Dictionary mode is set to fixed but no words are in it!"); return wordIds; }  // :926
    if (_visualWords.size()) {
        const Mat& first = _visualWords.begin()->second->getDescriptor();
        if (first.cols != descriptorsIn.cols) { logError("Descriptors are not the same size as already added words in dictionary"); return wordIds; }  // :948
        if (first.type() != descriptorsIn.type()) { logError("Descriptors are not the same type as already added words in dictionary"); return wordIds; }  // :953
    }
    if (descriptorsIn.type() != MAT_32F && descriptorsIn.type() != MAT_8U) { logError("Descriptors must be CV_32F or CV_8U"); return wordIds; }
    if (!ensureEngine(descriptorsIn.type(), descriptorsIn.cols)) { logError("%s", _lastError.c_str()); return wordIds; }                  // :986-996

    const int q = descriptorsIn.rows;
    std::vector<int32_t> out(q, 0);
    int32_t nNew = 0;
    const int flags = (_incrementalDictionary ? LCD_Q_INCREMENTAL : 0) | (_newWordsComparedTogether ? LCD_Q_NEW_WORDS_COMPARED : 0);
    if (lcd_quantize(_engine, descriptorsIn.data.data(), q, flags, _nndrRatio, out.data(), &nNew) != LCD_OK) {
        _lastError = lcd_last_error(_engine);
        logError("%s", _lastError.c_str());
        return wordIds;
    }
    // bookkeeping of the per-descriptor loop (:1162-1219), in descriptor order
    std::vector<int> created;   // ids of the words created by this call, in creation order
    for (int i = 0; i < q; ++i) {
        const int w = out[i];
        if (w < 0) {
            const int k = -w - 1;
            if (k == (int)created.size()) {
                // rejected by NNDR: new word from the ORIGINAL descriptor (:1185-1195)
                VisualWord* vw = new VisualWord(getNextId(), descriptorsIn.row(i), signatureId);
                _visualWords.insert(_visualWords.end(), std::pair<int, VisualWord*>(vw->id(), vw));
                indexWord(vw);
                _notIndexedWords.insert(_notIndexedWords.end(), vw->id());
                created.push_back(vw->id());
                wordIds.push_back(vw->id());
                _sigWords[signatureId].push_back(vw->id());
                _dirtySigs.insert(signatureId);
            } else if (k < (int)created.size()) {
                // matched a word created earlier in this call (:1140-1160, :1207)
                this->addWordRef(created[k], signatureId);
                wordIds.push_back(created[k]);
            } else {
                logError("inconsistent new-word index returned by the device");
                return std::list<int>();
            }
        } else if (w > 0) {
            this->addWordRef(w, signatureId);   // :1207 / :1215
            wordIds.push_back(w);
        }   // w == 0: fixed dictionary without candidate -> no entry (:1211-1218)
    }
    _totalActiveReferences += (int)_notIndexedWords.size();   // :1227 (sic)
    return wordIds;
}

// ---------------------------------------------------------------------------------------------- addNewWords + computeLikelihood, one device call
bool VWDictionaryHip::addNewWordsAndScore(const Mat& descriptorsIn, int signatureId, float N, const std::function<int(int)>& getNi,
                                          std::list<int>& wordIds, std::vector<float>& likelihoodBySlot) {
    wordIds.clear();
    if (!_incrementalDictionary || descriptorsIn.rows == 0 || descriptorsIn.cols == 0 || signatureId == 0) return false;
    if (_notIndexedWords.size() || _removedIndexedWords.size()) return false;       // update() first (Memory::preUpdate runs it)
    if (descriptorsIn.type() != MAT_32F && descriptorsIn.type() != MAT_8U) return false;
    if (_visualWords.size()) {
        const Mat& first = _visualWords.begin()->second->getDescriptor();
        if (first.cols != descriptorsIn.cols || first.type() != descriptorsIn.type()) return false;   // addNewWords reports it (:948-957)
    }
    if (descriptorsIn.rowBytes() % 4 != 0) return false;                            // rows the device pads: lcd_quantize takes them
    if (_deviceSigs.count(signatureId) || _sigWords.count(signatureId)) return false;
    if (!ensureEngine(descriptorsIn.type(), descriptorsIn.cols)) return false;
    if (!flushReferences(getNi)) return false;                                      // the device index is what the host maps say
    const int q = descriptorsIn.rows;
    std::vector<int32_t> out(q, 0);
    likelihoodBySlot.assign(_slotSig.size() + 1, 0.0f);
    int64_t nSlots = 0;
    lcd_frame_host_args a;
    std::memset(&a, 0, sizeof(a));
    a.struct_size = (int32_t)sizeof(a); a.q = q; a.descriptors = descriptorsIn.data.data();
    a.flags = LCD_Q_INCREMENTAL | (_newWordsComparedTogether ? LCD_Q_NEW_WORDS_COMPARED : 0); a.nndr_ratio = _nndrRatio;
    a.sig_id = signatureId; a.first_new_word_id = _lastWordId + 1; a.N = N; a.append_new_words = 1;
    a.word_ids = out.data(); a.likelihood = likelihoodBySlot.data(); a.likelihood_capacity = (int64_t)likelihoodBySlot.size(); a.n_slots = &nSlots;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    const int rc = lcd_frame_host(_engine, &a);
    _fastDeviceNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    _fastCalls += 1;
    if (rc != LCD_OK) {
        _lastError = lcd_last_error(_engine);
        if (rc != LCD_ERR_UNSUPPORTED) logError("%s", _lastError.c_str());
        return false;                                                               // nothing was registered: the caller takes the slow path
    }
    // Validate the device's answer BEFORE the host maps change (an incremental dictionary gives every descriptor a word: code 0 cannot
    // occur; the frame's k-th new word appears first as code -(k + 1), in order).  A result that fails this is not repaired by guessing:
    // the signature and the rows the device registered are dropped with the handle, the engine is replayed from the host maps (which are
    // untouched), and the caller takes the call-by-call path -- whose padding of missing ids (Memory.cpp:6029-6046) then applies.
    {
        int next = 0; bool ok = true;
        for (int i = 0; i < q && ok; ++i) {
            if (out[i] == 0) ok = false;
            else if (out[i] < 0) { const int k = -out[i] - 1; if (k == next) next += 1; else if (k > next) ok = false; }
        }
        if (!ok) {
            _lastError = "inconsistent word ids returned by the device (engine replayed from the host mirror)";
            logError("%s", _lastError.c_str());
            likelihoodBySlot.clear();
            this->rebuildEngine();
            return false;
        }
    }
    const bool slotsOk = nSlots == (int64_t)_slotSig.size() + 1;
    if (!slotsOk) {                                                                 // the mirror of the slot table is out of step: do not guess
        _lastError = "slot table of the device and its host mirror differ";
        logError("%s", _lastError.c_str());
        likelihoodBySlot.clear();
    }
    // bookkeeping of the per-descriptor loop (:1162-1219), in descriptor order; the references are on the device already
    std::vector<int> created;
    std::vector<int>& sw = _sigWords[signatureId];
    for (int i = 0; i < q; ++i) {
        const int w = out[i];
        int id = 0;
        if (w < 0) {
            const int k = -w - 1;
            if (k == (int)created.size()) {
                VisualWord* vw = new VisualWord(getNextId(), descriptorsIn.row(i), signatureId);   // :1185-1195
                _visualWords.insert(_visualWords.end(), std::pair<int, VisualWord*>(vw->id(), vw));
                indexWord(vw);
                _notIndexedWords.insert(_notIndexedWords.end(), vw->id());
                _deviceRows.insert(_deviceRows.end(), vw->id());
                created.push_back(vw->id());
                wordIds.push_back(vw->id());
                sw.push_back(vw->id());
                continue;
            }
            id = created[k];                                                        // a word created earlier in this call (:1140-1160, :1207; k validated above)
        } else if (w > 0) id = w;
        else continue;
        VisualWord* vwRef = lookupWord(id);                                         // addWordRef (:880-897) without the dirty mark
        if (!vwRef) { fprintf(stderr, "[WARN] Not found word %d (dict size=%d)\n", id, (int)_visualWords.size()); continue; }
        vwRef->addRef(signatureId);
        _totalActiveReferences += 1;
        _unusedWords.erase(id);
        sw.push_back(id);
        wordIds.push_back(id);
    }
    _totalActiveReferences += (int)_notIndexedWords.size();   // :1227 (sic)
    _deviceSigs.insert(signatureId);
    slotAdd(signatureId);
    // a slot table that was out of step stays out of step: the engine (a cache of these maps) is replayed from them -- slots, rows and
    // references are then the mirror's again; the frame's likelihood is not reported (the caller computes it the plain way)
    if (!slotsOk) this->rebuildEngine();
    return true;
}

// ---------------------------------------------------------------------------------------------- findNN  :1231-1552
std::vector<int> VWDictionaryHip::findNN(const std::list<VisualWord*>& vws) const {
    if (_visualWords.size() && vws.size()) {
        const int type = (*vws.begin())->getDescriptor().type();
        const int dim = (*vws.begin())->getDescriptor().cols;
        if (dim != _visualWords.begin()->second->getDescriptor().cols || type != _visualWords.begin()->second->getDescriptor().type()) {
            logError("Descriptors are not the same size/type as already added words in dictionary");
            return std::vector<int>(vws.size(), 0);
        }
        Mat query((int)vws.size(), dim, type);
        int index = 0;
        for (std::list<VisualWord*>::const_iterator iter = vws.begin(); iter != vws.end(); ++iter, ++index)
            std::memcpy(&query.data[(size_t)index * query.rowBytes()], (*iter)->getDescriptor().data.data(), query.rowBytes());
        return findNN(query);
    }
    return std::vector<int>(vws.size(), 0);
}

std::vector<int> VWDictionaryHip::findNN(const Mat& queryIn) const {
    std::vector<int> resultIds(queryIn.rows, 0);
    if (_visualWords.size() && queryIn.rows) {
        const Mat& first = _visualWords.begin()->second->getDescriptor();
        if (first.cols != queryIn.cols || first.type() != queryIn.type()) { logError("Descriptors are not the same size/type as already added words in dictionary"); return resultIds; }
        if (!ensureEngine(queryIn.type(), queryIn.cols)) { logError("%s", _lastError.c_str()); return resultIds; }
        // the words not yet indexed, ascending id (:1416-1451)
        std::vector<int32_t> extraIds;
        std::vector<unsigned char> extraRows;
        for (std::set<int>::const_iterator it = _notIndexedWords.begin(); it != _notIndexedWords.end(); ++it) {
            // a word a device-resident frame created is a row of the device vocabulary already (the indexed search covers it, behind every
            // indexed row: the same candidates, the same order on ties with indexed words): passed again as an extra it would be its own
            // second neighbour and fail the ratio test against itself
            if (_deviceRows.count(*it)) continue;
            const VisualWord* vw = _visualWords.at(*it);
            extraRows.insert(extraRows.end(), vw->getDescriptor().data.begin(), vw->getDescriptor().data.end());
            extraIds.push_back(*it);
        }
        std::vector<int32_t> out(queryIn.rows, 0);
        const int flags = _incrementalDictionary ? LCD_Q_INCREMENTAL : 0;
        if (lcd_find_nn(_engine, queryIn.data.data(), queryIn.rows, extraRows.data(), extraIds.data(), (int)extraIds.size(), flags, _nndrRatio,
                        out.data()) != LCD_OK) {
            _lastError = lcd_last_error(_engine);
            logError("%s", _lastError.c_str());
            return resultIds;
        }
        for (int i = 0; i < queryIn.rows; ++i) resultIds[i] = out[i];
    }
    return resultIds;
}

// ---------------------------------------------------------------------------------------------- references -> device
void VWDictionaryHip::markDirty(int signatureId) { _dirtySigs.insert(signatureId); }

// the device index is signature-granular: (re-)register every signature whose references changed since the last score
bool VWDictionaryHip::flushReferences(const std::function<int(int)>& getNi) {
    if (_dirtySigs.empty()) return true;
    if (!_engine) { _lastError = "no device engine"; return false; }
    for (std::set<int>::iterator s = _dirtySigs.begin(); s != _dirtySigs.end(); ++s) {
        if (_deviceSigs.count(*s)) {
            if (lcd_sig_remove(_engine, *s) != LCD_OK) { _lastError = lcd_last_error(_engine); return false; }
            _deviceSigs.erase(*s);
            slotRetire(*s);
        }
        std::map<int, std::vector<int> >::iterator w = _sigWords.find(*s);
        if (w != _sigWords.end() && !w->second.empty()) {
            const int ni = getNi ? getNi(*s) : (int)w->second.size();
            if (lcd_sig_add(_engine, *s, w->second.data(), (int)w->second.size(), ni) != LCD_OK) { _lastError = lcd_last_error(_engine); return false; }
            _deviceSigs.insert(*s);
            slotAdd(*s);
        } else if (w != _sigWords.end()) {
            _sigWords.erase(w);
        }
    }
    _dirtySigs.clear();
    return true;
}

bool VWDictionaryHip::flushReferencesBulk(const std::function<int(int)>& getNi) {
    if (_dirtySigs.empty()) return true;
    if (!_engine) { _lastError = "no device engine"; return false; }
    std::vector<int32_t> sigIds, words, ni;
    std::vector<int64_t> offsets(1, 0);
    for (std::set<int>::iterator s = _dirtySigs.begin(); s != _dirtySigs.end(); ++s) {
        if (_deviceSigs.count(*s)) continue;                              // registered already: its changes go through flushReferences
        std::map<int, std::vector<int> >::iterator w = _sigWords.find(*s);
        if (w == _sigWords.end() || w->second.empty()) continue;
        sigIds.push_back(*s);
        words.insert(words.end(), w->second.begin(), w->second.end());
        offsets.push_back((int64_t)words.size());
        ni.push_back(getNi ? getNi(*s) : (int)w->second.size());
    }
    if (!sigIds.empty()) {
        if (lcd_sig_add_bulk(_engine, (int)sigIds.size(), sigIds.data(), offsets.data(), words.data(), ni.data()) != LCD_OK) {
            _lastError = lcd_last_error(_engine);
            return false;
        }
        for (size_t k = 0; k < sigIds.size(); ++k) { _deviceSigs.insert(sigIds[k]); _dirtySigs.erase(sigIds[k]); slotAdd(sigIds[k]); }
    }
    return flushReferences(getNi);
}

// Memory::computeLikelihood, TF-IDF branch (Memory.cpp:2215-2291)
std::map<int, float> VWDictionaryHip::computeLikelihood(const std::list<int>& wordIds, const std::list<int>& ids, float N,
                                                        const std::function<int(int)>& getNi) {
    std::map<int, float> likelihood;
    if (ids.empty()) { fprintf(stderr, "[WARN] ids list is empty\n"); return likelihood; }   // :2227-2231
    for (std::list<int>::const_iterator i = ids.begin(); i != ids.end(); ++i) likelihood.insert(likelihood.end(), std::pair<int, float>(*i, 0.0f));
    if (!(N > 0.0f) || !_engine) return likelihood;
    if (!flushReferences(getNi)) { logError("%s", _lastError.c_str()); return likelihood; }
    // reference asserts that every word of the signature is in the dictionary (:2257); unknown ids are dropped here
    std::vector<int32_t> q;
    for (std::list<int>::const_iterator i = wordIds.begin(); i != wordIds.end(); ++i)
        if (*i > 0 && _visualWords.find(*i) != _visualWords.end()) q.push_back(*i);
    std::vector<int32_t> sig;
    for (std::map<int, float>::iterator i = likelihood.begin(); i != likelihood.end(); ++i) sig.push_back(i->first);
    std::vector<float> out(sig.size(), 0.0f);
    if (lcd_likelihood(_engine, q.data(), (int)q.size(), sig.data(), (int)sig.size(), N, out.data()) != LCD_OK) {
        _lastError = lcd_last_error(_engine);
        logError("%s", _lastError.c_str());
        return likelihood;
    }
    size_t k = 0;
    for (std::map<int, float>::iterator i = likelihood.begin(); i != likelihood.end(); ++i, ++k) i->second = out[k];
    return likelihood;
}

// ---------------------------------------------------------------------------------------------- exportDictionary :1619-1696
void VWDictionaryHip::exportDictionary(const char* fileNameReferences, const char* fileNameDescriptors) const {
    if (_visualWords.empty()) { fprintf(stderr, "[WARN] Dictionary is empty, cannot export it!\n"); return; }
    if (_visualWords.begin()->second->getDescriptor().type() != MAT_32F) { logError("Exporting binary descriptors is not implemented!"); return; }
    FILE* foutRef = fileNameReferences && fileNameReferences[0] ? fopen(fileNameReferences, "w") : 0;
    FILE* foutDesc = fileNameDescriptors && fileNameDescriptors[0] ? fopen(fileNameDescriptors, "w") : 0;
    if (foutRef) fprintf(foutRef, "WordID SignaturesID...\n");
    if (foutDesc) fprintf(foutDesc, "WordID Descriptors...%d\n", _visualWords.begin()->second->getDescriptor().cols);
    for (std::map<int, VisualWord*>::const_iterator iter = _visualWords.begin(); iter != _visualWords.end(); ++iter) {
        if (foutRef) {
            fprintf(foutRef, "%d ", iter->first);
            const std::map<int, int>& ref = iter->second->getReferences();
            for (std::map<int, int>::const_iterator jter = ref.begin(); jter != ref.end(); ++jter)
                for (int i = 0; i < jter->second; ++i) fprintf(foutRef, "%d ", jter->first);
            fprintf(foutRef, "\n");
        }
        if (foutDesc) {
            fprintf(foutDesc, "%d ", iter->first);
            const float* desc = (const float*)iter->second->getDescriptor().data.data();
            for (int i = 0; i < iter->second->getDescriptor().cols; i++) fprintf(foutDesc, "%f ", desc[i]);
            fprintf(foutDesc, "\n");
        }
    }
    if (foutRef) fclose(foutRef);
    if (foutDesc) fclose(foutDesc);
}

}  // namespace rtabmap_amd
