// DbLoaderHip.h -- the bulk-load boundary of SURVEY.md section 8 f3: the reference's database (SQLite, DatabaseSchema.sql.in) read
// straight into the flat arrays the engine's bulk entry points take, instead of one VWDictionary::addWord per word and one
// addWordRef per feature (Memory::loadDataFromDb, Memory.cpp:392-480).
//
// Mirrors (reference corelib/src/DBDriverSqlite3.cpp):
//   loadDictionary      <- loadQuery(VWDictionary&, lastStateOnly) :3541-3620  (Word: id, descriptor_size, descriptor; the blob is
//                          CV_8U when its byte count equals descriptor_size, CV_32F when a quarter of it does, an error otherwise)
//   loadSignatureWords  <- loadLastNodesQuery :3490-3540 (the nodes of the last saved state) or every node, and for each node the
//                          word ids of loadWordsQuery(signatures) :3847-3880 (Feature.word_id of that node, ORDER BY word_id)
//   getNi               <- getInvertedIndexNiQuery :2775-2810 (count(word_id) of a node: features without a word count too)
//   getLastWordId       <- getLastIdQuery("Word") :2736-2773
//   version             <- getDatabaseVersionQuery :285-320 ("SELECT version FROM Admin")
// Table names follow the version switches of those functions (Feature since 0.13.0, Map_Node_Word before; the last state is dated
// by Info since 0.11.11, by Statistics before).  Poses, sensor data, links and everything else in the database are out of scope.
//
// The SQLite library is the system's (libsqlite3.so.0, loaded at run time: this image has the library but not its header); the
// database is opened read-only.  No device call is made here: loadIntoEngine() hands the arrays to lcd_vocab_append and
// lcd_sig_add_bulk.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/lcd.h"

namespace rtabmap_amd {

struct DbDictionary {
    int type = -1;                          // MAT_32F (5) or MAT_8U (0), the type of the first word; -1: no words
    int cols = 0;                           // descriptor_size
    std::vector<int32_t> wordIds;           // ascending (ORDER BY id): the row order VWDictionary::update() gives a loaded dictionary
    std::vector<unsigned char> rows;        // wordIds.size() x cols elements, row-major
    int lastWordId = 0;                     // VWDictionary::setLastWordId
};

struct DbSignatures {
    std::vector<int32_t> sigIds;            // ascending node ids
    std::vector<int64_t> offsets;           // sigIds.size() + 1 offsets into wordIds
    std::vector<int32_t> wordIds;           // per node in ascending word id, duplicates = occurrences, ids <= 0 kept (they count in ni)
    std::vector<int32_t> ni;                // Memory::getNi of every LOADED node = Signature::getWords().size(): every feature row, a NULL
                                            // word_id included (sqlite3_column_int reads it as 0 and the reference inserts it,
                                            // DBDriverSqlite3.cpp:3912); DbLoaderHip::getNi(node) is the other reading -- the SQL
                                            // count(word_id) of a node that is NOT in memory, which skips NULLs (:2788-2794)
};

class DbLoaderHip {
public:
    DbLoaderHip();
    ~DbLoaderHip();
    bool open(const std::string& path);     // false + lastError(): no SQLite library, no such file, not a RTAB-Map database
    void close();
    bool isOpen() const { return _db != nullptr; }
    const std::string& version() const { return _version; }
    bool loadDictionary(DbDictionary& out, bool lastStateOnly = false);
    bool loadSignatureWords(DbSignatures& out, bool lastStateOnly = true);
    int getNi(int nodeId);
    int getLastWordId();
    const std::string& lastError() const { return _err; }

    // uStrNumCmp (UStl.h:717-790) on dotted version strings: < 0, 0, > 0
    static int versionCmp(const std::string& a, const std::string& b);
    struct Api;                             // the SQLite entry points, resolved at run time (DbLoaderHip.cpp)

private:
    bool fail(const std::string& what);
    void* _lib;
    void* _db;
    std::string _version, _err;
    Api* _api;
};

// Memory::loadDataFromDb's two loops as two calls: the dictionary in row (= id) order, then every signature's references.
// Returns an LCD_* status (LCD_OK on success); the engine must be empty of these ids.
int loadIntoEngine(lcd_engine* engine, const DbDictionary& dictionary, const DbSignatures& signatures);

}  // namespace rtabmap_amd
