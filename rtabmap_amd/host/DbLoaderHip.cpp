// DbLoaderHip.cpp -- see DbLoaderHip.h.
#include "DbLoaderHip.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "VWDictionaryHip.h"   // MAT_8U / MAT_32F

namespace rtabmap_amd {

// the handful of entry points of the SQLite C API this reader needs (sqlite3.h is not installed in this image; the ABI of these
// functions has not changed since 3.3.9)
struct DbLoaderHip::Api {
    int (*open_v2)(const char*, void**, int, const char*);
    int (*close)(void*);
    int (*prepare_v2)(void*, const char*, int, void**, const char**);
    int (*step)(void*);
    int (*finalize)(void*);
    int (*column_int)(void*, int);
    const void* (*column_blob)(void*, int);
    int (*column_bytes)(void*, int);
    const unsigned char* (*column_text)(void*, int);
    const char* (*errmsg)(void*);
};
namespace {
const int kSqliteOk = 0, kSqliteRow = 100, kSqliteDone = 101, kSqliteOpenReadOnly = 1;

// one prepared statement, finalised when it goes out of scope
struct Stmt {
    const DbLoaderHip::Api* api;
    void* st = nullptr;
    Stmt(const DbLoaderHip::Api* a, void* db, const std::string& sql, int* rc) : api(a) { *rc = api->prepare_v2(db, sql.c_str(), -1, &st, nullptr); }
    ~Stmt() { if (st) api->finalize(st); }
};
}  // namespace

DbLoaderHip::DbLoaderHip() : _lib(nullptr), _db(nullptr), _version("0.0.0"), _api(new Api()) {}
DbLoaderHip::~DbLoaderHip() {
    close();
    if (_lib) dlclose(_lib);
    delete _api;
}

bool DbLoaderHip::fail(const std::string& what) {
    _err = what;
    if (_db) { _err += ": "; _err += _api->errmsg(_db); }
    fprintf(stderr, "[ERROR] DbLoaderHip: %s\n", _err.c_str());
    return false;
}

int DbLoaderHip::versionCmp(const std::string& a, const std::string& b) {
    // uStrNumCmp on dotted versions: component by component as numbers (the reference compares digit runs by length, then as text --
    // the same order for components without leading zeros), over the components BOTH strings have: "0.21" equals "0.21.4", as there
    size_t i = 0, j = 0;
    while (i < a.size() && j < b.size()) {
        long x = 0, y = 0;
        while (i < a.size() && a[i] != '.') { if (a[i] >= '0' && a[i] <= '9') x = x * 10 + (a[i] - '0'); ++i; }
        while (j < b.size() && b[j] != '.') { if (b[j] >= '0' && b[j] <= '9') y = y * 10 + (b[j] - '0'); ++j; }
        if (x != y) return x < y ? -1 : 1;
        if (i < a.size()) ++i;
        if (j < b.size()) ++j;
    }
    return 0;
}

bool DbLoaderHip::open(const std::string& path) {
    close();
    _err.clear();
    if (!_lib) {
        const char* names[] = {"libsqlite3.so.0", "libsqlite3.so"};
        for (const char* n : names) { _lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (_lib) break; }
        if (!_lib) return fail("the SQLite library (libsqlite3.so.0) cannot be loaded");
        bool ok = true;
        auto sym = [&](const char* name) -> void* { void* p = dlsym(_lib, name); if (!p) ok = false; return p; };
        _api->open_v2 = (int (*)(const char*, void**, int, const char*))sym("sqlite3_open_v2");
        _api->close = (int (*)(void*))sym("sqlite3_close");
        _api->prepare_v2 = (int (*)(void*, const char*, int, void**, const char**))sym("sqlite3_prepare_v2");
        _api->step = (int (*)(void*))sym("sqlite3_step");
        _api->finalize = (int (*)(void*))sym("sqlite3_finalize");
        _api->column_int = (int (*)(void*, int))sym("sqlite3_column_int");
        _api->column_blob = (const void* (*)(void*, int))sym("sqlite3_column_blob");
        _api->column_bytes = (int (*)(void*, int))sym("sqlite3_column_bytes");
        _api->column_text = (const unsigned char* (*)(void*, int))sym("sqlite3_column_text");
        _api->errmsg = (const char* (*)(void*))sym("sqlite3_errmsg");
        if (!ok) { dlclose(_lib); _lib = nullptr; return fail("the SQLite library lacks an entry point"); }
    }
    void* db = nullptr;
    const int rc = _api->open_v2(path.c_str(), &db, kSqliteOpenReadOnly, nullptr);
    if (rc != kSqliteOk) {
        _db = db;                                                     // (sqlite3_open_v2 hands out a handle even on failure: its message, then closed)
        fail("cannot open \"" + path + "\"");
        close();
        return false;
    }
    _db = db;
    // getDatabaseVersionQuery: "0.0.0" when there is no Admin table (databases older than the table) -- not supported here
    _version = "0.0.0";
    {
        int prc = 0;
        Stmt q(_api, _db, "SELECT version FROM Admin;", &prc);
        if (prc == kSqliteOk && _api->step(q.st) == kSqliteRow) {
            const unsigned char* t = _api->column_text(q.st, 0);
            if (t) _version = reinterpret_cast<const char*>(t);
        }
    }
    if (versionCmp(_version, "0.11.2") < 0) {
        const std::string v = _version;
        fail("not a RTAB-Map database this reader knows (Admin.version " + v + ")");
        close();
        return false;
    }
    return true;
}

void DbLoaderHip::close() {
    if (_db) { _api->close(_db); _db = nullptr; }
}

bool DbLoaderHip::loadDictionary(DbDictionary& out, bool lastStateOnly) {
    out = DbDictionary();
    if (!_db) return fail("no database is open");
    std::string sql = "SELECT id, descriptor_size, descriptor FROM Word ";
    if (lastStateOnly)
        sql += versionCmp(_version, "0.11.11") >= 0 ? "WHERE time_enter >= (SELECT MAX(time_enter) FROM Info) "
                                                    : "WHERE time_enter >= (SELECT MAX(time_enter) FROM Statistics) ";
    sql += "ORDER BY id;";
    int rc = 0;
    Stmt q(_api, _db, sql, &rc);
    if (rc != kSqliteOk) return fail("DB error (" + _version + ")");
    while ((rc = _api->step(q.st)) == kSqliteRow) {
        const int id = _api->column_int(q.st, 0);
        const int size = _api->column_int(q.st, 1);
        const void* blob = _api->column_blob(q.st, 2);
        const int bytes = _api->column_bytes(q.st, 2);
        int type;
        if (bytes == size) type = MAT_8U;                             // CV_8U binary descriptors
        else if (bytes / (int)sizeof(float) == size) type = MAT_32F;  // CV_32F
        else {
            char msg[160];
            snprintf(msg, sizeof(msg), "Saved buffer size (%d bytes) is not the same as descriptor size (%d)", bytes, size);   // (UFATAL in the reference)
            out = DbDictionary();
            _err = msg;
            fprintf(stderr, "[ERROR] DbLoaderHip: %s\n", msg);
            return false;
        }
        if (out.type < 0) { out.type = type; out.cols = size; }
        if (type != out.type || size != out.cols || size <= 0 || !blob) {
            // VWDictionary::addNewWords refuses descriptors of another size or type than the dictionary's (VWDictionary.cpp:948-957); a
            // dictionary that mixes them cannot be searched
            out = DbDictionary();
            _err = "the Word table mixes descriptor sizes or types";
            fprintf(stderr, "[ERROR] DbLoaderHip: %s\n", _err.c_str());
            return false;
        }
        out.wordIds.push_back(id);
        const unsigned char* b = static_cast<const unsigned char*>(blob);
        // exactly one row: a CV_32F blob may carry up to three stray bytes (bytes / 4 == size) -- the reference copies them into a Mat
        // of its own per word, here the rows are one array with stride cols * elemSize
        out.rows.insert(out.rows.end(), b, b + (size_t)size * (type == MAT_32F ? sizeof(float) : 1));
    }
    if (rc != kSqliteDone) return fail("DB error (" + _version + ")");
    out.lastWordId = getLastWordId();
    return true;
}

int DbLoaderHip::getLastWordId() {
    if (!_db) return 0;
    int rc = 0, id = 0;
    Stmt q(_api, _db, "SELECT COALESCE(MAX(id), 0) FROM Word;", &rc);
    if (rc == kSqliteOk && _api->step(q.st) == kSqliteRow) id = _api->column_int(q.st, 0);
    return id;
}

int DbLoaderHip::getNi(int nodeId) {
    if (!_db) return 0;
    const std::string table = versionCmp(_version, "0.13.0") >= 0 ? "Feature" : "Map_Node_Word";
    int rc = 0, ni = 0;
    Stmt q(_api, _db, "SELECT count(word_id) FROM " + table + " WHERE node_id=" + std::to_string(nodeId) + ";", &rc);
    if (rc == kSqliteOk && _api->step(q.st) == kSqliteRow) ni = _api->column_int(q.st, 0);
    return ni;
}

bool DbLoaderHip::loadSignatureWords(DbSignatures& out, bool lastStateOnly) {
    out = DbSignatures();
    if (!_db) return fail("no database is open");
    const std::string table = versionCmp(_version, "0.13.0") >= 0 ? "Feature" : "Map_Node_Word";
    const std::string lastState = versionCmp(_version, "0.11.11") >= 0 ? "(SELECT MAX(time_enter) FROM Info)" : "(SELECT MAX(time_enter) FROM Statistics)";
    // the node ids first (loadLastNodesQuery, or every node): a node without features is a signature too (ni = 0)
    {
        std::string sql = "SELECT n.id FROM Node AS n ";
        if (lastStateOnly) sql += "WHERE n.time_enter >= " + lastState + " ";
        sql += "ORDER BY n.id;";
        int rc = 0;
        Stmt q(_api, _db, sql, &rc);
        if (rc != kSqliteOk) return fail("DB error (" + _version + ")");
        while ((rc = _api->step(q.st)) == kSqliteRow) out.sigIds.push_back(_api->column_int(q.st, 0));
        if (rc != kSqliteDone) return fail("DB error (" + _version + ")");
    }
    // then ONE pass over the feature table in (node, word) order -- the reference binds and runs a query per node
    out.offsets.assign(out.sigIds.size() + 1, 0);
    out.ni.assign(out.sigIds.size(), 0);
    {
        std::string sql = "SELECT node_id, word_id FROM " + table + " ";
        if (lastStateOnly) sql += "WHERE node_id IN (SELECT id FROM Node WHERE time_enter >= " + lastState + ") ";
        sql += "ORDER BY node_id, word_id;";
        int rc = 0;
        Stmt q(_api, _db, sql, &rc);
        if (rc != kSqliteOk) return fail("DB error (" + _version + ")");
        size_t s = 0;
        while ((rc = _api->step(q.st)) == kSqliteRow) {
            const int node = _api->column_int(q.st, 0);
            while (s < out.sigIds.size() && out.sigIds[s] < node) { ++s; out.offsets[s] = (int64_t)out.wordIds.size(); }
            if (s >= out.sigIds.size() || out.sigIds[s] != node) continue;   // a feature of a node that is not in the Node table
            out.wordIds.push_back(_api->column_int(q.st, 1));
            out.ni[s] += 1;
        }
        if (rc != kSqliteDone) return fail("DB error (" + _version + ")");
        while (s < out.sigIds.size()) { ++s; out.offsets[s] = (int64_t)out.wordIds.size(); }
    }
    return true;
}

int loadIntoEngine(lcd_engine* engine, const DbDictionary& dictionary, const DbSignatures& signatures) {
    if (!engine) return LCD_ERR_INVALID;
    if (!dictionary.wordIds.empty()) {
        const int rc = lcd_vocab_append(engine, dictionary.rows.data(), (int)dictionary.wordIds.size(), dictionary.wordIds.data());
        if (rc != LCD_OK) return rc;
    }
    if (signatures.sigIds.empty()) return LCD_OK;
    return lcd_sig_add_bulk(engine, (int)signatures.sigIds.size(), signatures.sigIds.data(), signatures.offsets.data(), signatures.wordIds.data(),
                            signatures.ni.data());
}

}  // namespace rtabmap_amd
