"""CPU: the database reader of the bulk-load boundary (rtabmap_amd/host/DbLoaderHip) -- no device call is made.

A database is made here with Python's sqlite3 from the reference's own table definitions (corelib/src/resources/DatabaseSchema.sql.in:
Node :16-29, Word :66-72, Feature :74-89, Info :101-109, Admin :119-...; only the columns the loaders read are filled) the way
DBDriverSqlite3 fills them: Word.descriptor is the raw row (descriptor_size bytes for CV_8U, 4 x descriptor_size for CV_32F,
DBDriverSqlite3.cpp:3541-3620), a Feature row per keypoint with its word id -- negative ids for features without a word included --
and time_enter stamps that date the last saved state (loadLastNodesQuery :3490-3540).  The reader must hand back exactly the arrays
the engine's bulk entry points take: words in ascending id, signatures in ascending node id with their word ids ascending
(loadWordsQuery's ORDER BY word_id :3880), ni = every feature of the node (getInvertedIndexNiQuery :2775-2810)."""
import ctypes as C
import os
import sqlite3

import numpy as np
import pytest

SCHEMA = """
CREATE TABLE Node (id INTEGER NOT NULL, map_id INTEGER NOT NULL, weight INTEGER, stamp FLOAT, pose BLOB, ground_truth_pose BLOB, velocity BLOB,
                   label TEXT, gps BLOB, env_sensors BLOB, time_enter DATE, PRIMARY KEY (id));
CREATE TABLE Word (id INTEGER NOT NULL, descriptor_size INTEGER NOT NULL, descriptor BLOB NOT NULL, time_enter DATE, PRIMARY KEY (id));
CREATE TABLE %(feature)s (node_id INTEGER NOT NULL, word_id INTEGER NOT NULL, pos_x FLOAT NOT NULL, pos_y FLOAT NOT NULL, size INTEGER NOT NULL,
                      dir FLOAT NOT NULL, response FLOAT NOT NULL, octave INTEGER NOT NULL, depth_x FLOAT, depth_y FLOAT, depth_z FLOAT,
                      descriptor_size INTEGER, descriptor BLOB, FOREIGN KEY (node_id) REFERENCES Node(id));
CREATE TABLE Info (STM_size INTEGER, last_sign_added INTEGER, process_mem_used INTEGER, database_mem_used INTEGER, dictionary_size INTEGER,
                   parameters TEXT, time_enter DATE);
CREATE TABLE Admin (version TEXT, preview_image BLOB);
"""


def _lib():
    from rtabmap_amd import vwdictionary as V
    L = V.lib()
    L.hdb_open.restype = C.c_void_p
    L.hdb_open.argtypes = [C.c_char_p]
    L.hdb_close.argtypes = [C.c_void_p]
    L.hdb_close.restype = None
    L.hdb_version.argtypes = [C.c_void_p]
    L.hdb_version.restype = C.c_char_p
    L.hdb_last_error.argtypes = [C.c_void_p]
    L.hdb_last_error.restype = C.c_char_p
    L.hdb_load_dictionary.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.hdb_dictionary_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hdb_dictionary_copy.restype = None
    L.hdb_load_signatures.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
    L.hdb_signatures_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hdb_signatures_copy.restype = None
    L.hdb_get_ni.argtypes = [C.c_void_p, C.c_int]
    L.hdb_version_cmp.argtypes = [C.c_char_p, C.c_char_p]
    return L


def _make_db(path, kind, version="0.21.4", n_words=300, n_nodes=40, seed=3, old_words=0, old_nodes=0, bad_word=None, reference_last_word=False,
             stray_word=None):
    """Returns (word ids, rows, {node: word ids in keypoint order}).  The first `old_words` words / `old_nodes` nodes carry a
    time_enter older than the last Info row: they are not part of the last saved state."""
    rng = np.random.default_rng(seed)
    feature = "Feature" if tuple(int(x) for x in version.split(".")) >= (0, 13, 0) else "Map_Node_Word"
    if os.path.exists(path):
        os.remove(path)
    db = sqlite3.connect(path)
    db.executescript(SCHEMA % {"feature": feature})
    db.execute("INSERT INTO Admin (version) VALUES (?)", (version,))
    db.execute("INSERT INTO Info (STM_size, time_enter) VALUES (10, '2020-01-01 00:00:00')")
    db.execute("INSERT INTO Info (STM_size, time_enter) VALUES (10, '2020-06-01 12:00:00')")          # the last saved state
    ids = np.sort(rng.choice(np.arange(1, 5 * n_words), n_words, replace=False)).astype(np.int32)     # gaps: words that were deleted
    rows = rng.standard_normal((n_words, 64)).astype(np.float32) if kind == "surf" else rng.integers(0, 256, (n_words, 32)).astype(np.uint8)
    order = rng.permutation(n_words)                                                                  # inserted out of order: ORDER BY id sorts
    for k in order:
        size = rows.shape[1]
        blob = rows[k].tobytes()
        if bad_word is not None and k == bad_word:
            blob = blob[:-3]
        if stray_word is not None and k == stray_word:
            blob = blob + b"\xAB\xCD"                                                                 # bytes / 4 == size still holds: accepted as CV_32F
        stamp = "2020-03-01 00:00:00" if k < old_words else "2020-06-01 12:00:00"
        db.execute("INSERT INTO Word (id, descriptor_size, descriptor, time_enter) VALUES (?,?,?,?)", (int(ids[k]), size, blob, stamp))
    sigs = {}
    for n in range(1, n_nodes + 1):
        stamp = "2020-03-01 00:00:00" if n <= old_nodes else "2020-06-01 12:00:01"
        db.execute("INSERT INTO Node (id, map_id, weight, time_enter) VALUES (?,0,0,?)", (3 * n, stamp))
        nf = 0 if n == 7 else int(rng.integers(20, 60))                                              # node 21: a featureless frame
        w = rng.choice(ids, nf).astype(np.int32)
        if nf:
            w[rng.integers(0, nf, 3)] = w[0]                                                         # a word seen several times
            neg = rng.integers(0, nf, 4)
            w[neg] = -np.arange(1, 5, dtype=np.int32)                                                # features without a word: -1, -2, ..
            if reference_last_word and n == 1:
                w[-1] = ids[-1]                                                                      # (the newest word is in use)
        sigs[3 * n] = w
        for x in w:
            db.execute("INSERT INTO %s (node_id, word_id, pos_x, pos_y, size, dir, response, octave) VALUES (?,?,0,0,1,0,0,0)" % feature, (3 * n, int(x)))
    db.execute("INSERT INTO %s (node_id, word_id, pos_x, pos_y, size, dir, response, octave) VALUES (9999, 5, 0,0,1,0,0,0)" % feature)   # an orphan feature
    db.commit()
    db.close()
    return ids, rows, sigs


def _read(L, path, last_dict, last_sigs):
    h = L.hdb_open(path.encode())
    assert h, "hdb_open failed"
    info = (C.c_int * 4)()
    n = L.hdb_load_dictionary(h, int(last_dict), info)
    assert n >= 0, L.hdb_last_error(h)
    wid = np.zeros(n, np.int32)
    rows = np.zeros(info[3], np.uint8)
    L.hdb_dictionary_copy(h, wid.ctypes.data, rows.ctypes.data)
    nw = C.c_longlong(0)
    ns = L.hdb_load_signatures(h, int(last_sigs), C.byref(nw))
    assert ns >= 0, L.hdb_last_error(h)
    sid, offs, words, ni = np.zeros(ns, np.int32), np.zeros(ns + 1, np.int64), np.zeros(nw.value, np.int32), np.zeros(ns, np.int32)
    L.hdb_signatures_copy(h, sid.ctypes.data, offs.ctypes.data, words.ctypes.data, ni.ctypes.data)
    out = dict(version=L.hdb_version(h).decode(), type=info[0], cols=info[1], last_word=info[2], word_ids=wid, rows=rows, sig_ids=sid, offsets=offs,
               words=words, ni=ni, ni_of=lambda node: L.hdb_get_ni(h, node))
    return h, out


@pytest.mark.parametrize("kind,version", [("surf", "0.21.4"), ("orb", "0.21.4"), ("surf", "0.12.0")])
def test_reader_hands_back_the_bulk_arrays(tmp_path, kind, version):
    L = _lib()
    path = str(tmp_path / "map.db")
    ids, rows, sigs = _make_db(path, kind, version=version)
    h, r = _read(L, path, False, False)
    assert r["version"] == version
    assert r["type"] == (5 if kind == "surf" else 0) and r["cols"] == rows.shape[1] and r["last_word"] == int(ids.max())
    assert r["word_ids"].tolist() == ids.tolist()                                   # ascending id, whatever the insertion order
    assert r["rows"].tobytes() == rows.tobytes()                                    # bit for bit
    assert r["sig_ids"].tolist() == sorted(sigs)
    assert r["offsets"][0] == 0 and r["offsets"][-1] == r["words"].size == sum(len(w) for w in sigs.values())
    for k, node in enumerate(sorted(sigs)):
        got = r["words"][r["offsets"][k]: r["offsets"][k + 1]]
        assert got.tolist() == sorted(sigs[node].tolist()), "node %d" % node         # ORDER BY word_id; duplicates and negative ids kept
        assert r["ni"][k] == len(sigs[node]) == r["ni_of"](node)
    assert r["ni_of"](424242) == 0
    L.hdb_close(h)


def test_a_blob_with_stray_bytes_does_not_shift_the_rows_behind_it(tmp_path):
    """DBDriverSqlite3.cpp:3593 accepts a CV_32F blob of 4 x size + 1..3 bytes (integer division) and builds ONE Mat per word; the reader's
    rows are one array with stride cols x 4, so it must take exactly one row from such a blob (the advisor's round-4 finding)"""
    L = _lib()
    path = str(tmp_path / "stray.db")
    ids, rows, sigs = _make_db(path, "surf", n_words=40, n_nodes=4, stray_word=11)
    h, r = _read(L, path, False, False)
    assert r["word_ids"].tolist() == ids.tolist()
    assert r["rows"].size == rows.nbytes and r["rows"].tobytes() == rows.tobytes()
    L.hdb_close(h)


def test_last_state_only(tmp_path):
    """loadLastNodesQuery / load(dictionary, lastStateOnly = true): only what was saved with the last Info row"""
    L = _lib()
    path = str(tmp_path / "map.db")
    ids, rows, sigs = _make_db(path, "orb", n_words=120, n_nodes=30, old_words=50, old_nodes=12)
    h, r = _read(L, path, True, True)
    assert r["word_ids"].tolist() == ids[50:].tolist() and r["rows"].tobytes() == rows[50:].tobytes()
    assert r["last_word"] == int(ids.max())                                          # getLastWordId looks at the whole table
    recent = [n for n in sorted(sigs) if n > 3 * 12]
    assert r["sig_ids"].tolist() == recent
    assert r["ni"].tolist() == [len(sigs[n]) for n in recent]
    L.hdb_close(h)
    h, r = _read(L, path, False, False)                                              # and everything when not asked for the last state
    assert r["word_ids"].size == 120 and r["sig_ids"].size == 30
    L.hdb_close(h)


def test_errors_are_reported_not_thrown(tmp_path, capfd):
    L = _lib()
    assert not L.hdb_open(str(tmp_path / "missing.db").encode())                    # read-only open: nothing is created
    assert not os.path.exists(str(tmp_path / "missing.db"))
    junk = tmp_path / "junk.db"
    junk.write_bytes(b"this is not a database" * 100)
    assert not L.hdb_open(str(junk).encode())
    plain = str(tmp_path / "plain.db")                                               # a SQLite file without an Admin table
    sqlite3.connect(plain).execute("CREATE TABLE t (x INTEGER)").connection.commit()
    assert not L.hdb_open(plain.encode())
    # a blob whose size fits neither CV_8U nor CV_32F (UFATAL in the reference): the load fails, nothing half-loaded is handed out
    path = str(tmp_path / "bad.db")
    _make_db(path, "surf", n_words=20, n_nodes=3, bad_word=5)
    h = L.hdb_open(path.encode())
    assert h
    info = (C.c_int * 4)()
    assert L.hdb_load_dictionary(h, 0, info) == -1
    assert b"is not the same as descriptor size" in L.hdb_last_error(h)
    L.hdb_close(h)
    capfd.readouterr()


def test_version_order():
    L = _lib()
    cmp = lambda a, b: L.hdb_version_cmp(a.encode(), b.encode())
    assert cmp("0.13.0", "0.13.0") == 0 and cmp("0.12.9", "0.13.0") < 0 and cmp("0.21.4", "0.13.0") > 0
    assert cmp("0.9.0", "0.11.11") < 0 and cmp("0.11.11", "0.11.2") > 0 and cmp("1.0", "0.99.99") > 0
