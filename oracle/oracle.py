"""ctypes bindings of the CPU oracle (oracle/liblcd_oracle.so) and of the reference's own rtflann compiled in place
(oracle/_ref/librtflann_ref.so).  TEST INFRASTRUCTURE ONLY -- the product never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
T_F32, T_U8 = 0, 1
METRIC_L2, METRIC_HAMMING, METRIC_L1, METRIC_HAMMING_CV = 0, 1, 2, 3
# METRIC_HAMMING = rtflann::Hamming (ignores the bytes beyond a multiple of 8, like the reference's FLANN strategies);
# METRIC_HAMMING_CV = cv::NORM_HAMMING (every byte: the reference's brute-force strategies and same-frame comparison)
ALGO_LINEAR, ALGO_KDTREE = 0, 1
# Kp/NNStrategy (reference VWDictionary.h:49-55)
kNNFlannNaive, kNNFlannKdTree, kNNFlannLSH, kNNBruteForce, kNNBruteForceGPU = 0, 1, 2, 3, 4

_lib = None
_ref = None


def build(ref=True):
    """Compile the checker(s).  `make ref` is a no-op message when /root/reference is absent."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liblcd_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.orc_dist_l2.restype = C.c_float
        L.orc_dist_l1.restype = C.c_float
        L.orc_dist_hamming.restype = C.c_uint
        for f in ("orc_dist_l2", "orc_dist_l1", "orc_dist_hamming"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_knn2_linear.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_long,
                                      C.c_void_p, C.c_void_p, C.c_int]
        L.orc_knn2_linear.restype = None
        L.orc_dist_matrix.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
        L.orc_dist_matrix.restype = None
        L.orc_vwd_create.restype = C.c_void_p
        L.orc_vwd_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
        L.orc_vwd_destroy.argtypes = [C.c_void_p]
        L.orc_vwd_last_error.restype = C.c_char_p
        L.orc_vwd_last_error.argtypes = [C.c_void_p]
        L.orc_vwd_add_new_words.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_vwd_find_nn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_vwd_update.argtypes = [C.c_void_p]
        L.orc_vwd_add_word.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_vwd_add_word_ref.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_vwd_remove_all_word_ref.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_vwd_get_unused_word_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_vwd_remove_words.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_vwd_delete_unused_words.argtypes = [C.c_void_p]
        L.orc_vwd_clear.argtypes = [C.c_void_p]
        L.orc_vwd_stat.argtypes = [C.c_void_p, C.c_int]
        L.orc_vwd_stat.restype = C.c_long
        L.orc_vwd_get_word_refs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_vwd_word_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_vwd_index_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_vwd_load_fixed_text.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_vwd_export_text.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_mem_create.restype = C.c_void_p
        L.orc_mem_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
        L.orc_mem_destroy.argtypes = [C.c_void_p]
        L.orc_mem_vwd.restype = C.c_void_p
        L.orc_mem_vwd.argtypes = [C.c_void_p]
        L.orc_mem_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_mem_add_signature.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_mem_add_signature_with_id.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_mem_add_signatures_bulk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_mem_forget.argtypes = [C.c_void_p, C.c_int]
        L.orc_mem_get_ni.argtypes = [C.c_void_p, C.c_int]
        L.orc_mem_num_signatures.argtypes = [C.c_void_p]
        L.orc_mem_num_signatures.restype = C.c_long
        L.orc_mem_signature_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_mem_compute_likelihood.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_adjust_likelihood.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.orc_bayes_create.restype = C.c_void_p
        L.orc_bayes_create.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.orc_bayes_destroy.argtypes = [C.c_void_p]
        L.orc_bayes_reset.argtypes = [C.c_void_p]
        L.orc_bayes_set_neighbors.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_bayes_set_stm.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_bayes_compute_posterior.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_bayes_hypothesis.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "librtflann_ref.so"))


def ref():
    """The reference's own rtflann (compiled in place from /root/reference by oracle/Makefile)."""
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(_HERE, "_ref", "librtflann_ref.so"))
        R.ref_index_create.restype = C.c_void_p
        R.ref_index_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]
        R.ref_index_destroy.argtypes = [C.c_void_p]
        R.ref_index_add.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]
        R.ref_index_remove.argtypes = [C.c_void_p, C.c_size_t]
        R.ref_index_size.argtypes = [C.c_void_p]
        R.ref_index_size.restype = C.c_size_t
        R.ref_index_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        R.ref_dist_l2.restype = C.c_float
        R.ref_dist_l1.restype = C.c_float
        R.ref_dist_hamming.restype = C.c_uint
        for f in ("ref_dist_l2", "ref_dist_l1", "ref_dist_hamming"):
            getattr(R, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        if hasattr(R, "ref_umean_list"):                          # (a library built before these entries existed lacks them: rebuild with make ref)
            R.ref_umean_list.restype = C.c_float
            R.ref_umean_list.argtypes = [C.c_void_p, C.c_size_t]
            R.ref_uvariance_list.restype = C.c_float
            R.ref_uvariance_list.argtypes = [C.c_void_p, C.c_size_t, C.c_float]
            R.ref_ustr2float.restype = C.c_float
            R.ref_ustr2float.argtypes = [C.c_char_p]
        if hasattr(R, "ref_ustrnumcmp"):
            R.ref_ustrnumcmp.argtypes = [C.c_char_p, C.c_char_p]
        if hasattr(R, "ref_dictionary_text_roundtrip"):
            R.ref_dictionary_text_roundtrip.argtypes = [C.c_char_p, C.c_char_p]
        _ref = R
    return _ref


def _type_of(a):
    if a.dtype == np.float32:
        return T_F32
    if a.dtype == np.uint8:
        return T_U8
    raise TypeError("descriptors must be float32 or uint8")


def metric_of(a, l1=False):
    return METRIC_HAMMING if a.dtype == np.uint8 else (METRIC_L1 if l1 else METRIC_L2)


# --------------------------------------------------------------------------------------------- plain functions
def knn2_linear(train, queries, removed=None, metric=None, threads=1):
    """Exact 2-NN, reference tie-break (lower row wins).  Returns (idx int64 [q,2], dist float32 [q,2]); -1 = none."""
    train = np.ascontiguousarray(train)
    queries = np.ascontiguousarray(queries)
    m = metric_of(train) if metric is None else metric
    nq = queries.shape[0]
    idx = np.empty((nq, 2), np.int64)
    dist = np.empty((nq, 2), np.float32)
    rm = None if removed is None else np.ascontiguousarray(removed, dtype=np.uint8)
    lib().orc_knn2_linear(m, _ptr(train), train.shape[0], train.shape[1], None if rm is None else _ptr(rm),
                          _ptr(queries), nq, _ptr(idx), _ptr(dist), threads)
    return idx, dist


def dist_matrix(a, b, metric=None):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    m = metric_of(a) if metric is None else metric
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().orc_dist_matrix(m, _ptr(a), a.shape[0], _ptr(b), b.shape[0], a.shape[1], _ptr(out))
    return out


def flat_tfidf(word_off, post_sig, post_cnt, ni, N, threads=1):
    """bench.py's generous CPU baseline (NOT a parity reference): TF-IDF over flat word-major postings, OpenMP over the words"""
    word_off = np.ascontiguousarray(word_off, np.int64)
    post_sig = np.ascontiguousarray(post_sig, np.int32)
    post_cnt = np.ascontiguousarray(post_cnt, np.int32)
    ni = np.ascontiguousarray(ni, np.int32)
    out = np.zeros(ni.shape[0], np.float32)
    L = lib()
    L.orc_flat_tfidf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
    L.orc_flat_tfidf.restype = None
    L.orc_flat_tfidf(_ptr(word_off), _ptr(post_sig), _ptr(post_cnt), _ptr(ni), ni.shape[0], word_off.shape[0] - 1, float(N), int(threads), _ptr(out))
    return out


def set_log10_double(on):
    """Memory.cpp:2266 `log10(N/nw)` read as the double overload (older standard libraries) instead of log10f -- see lcd_oracle.cpp"""
    lib().orc_set_log10_double(1 if on else 0)


def adjust_likelihood(L, ratio=0.0):
    L = np.ascontiguousarray(L, dtype=np.float32).copy()
    lib().orc_adjust_likelihood(_ptr(L), L.shape[0], ratio)
    return L


class RefIndex:
    """rtflann::Index<L2|L1|Hamming> of the reference (LINEAR = exact, KDTREE = the reference default)."""

    def __init__(self, rows, metric=None, algo=ALGO_LINEAR, trees=4):
        rows = np.ascontiguousarray(rows)
        self.metric = metric_of(rows) if metric is None else metric
        self.dtype = rows.dtype
        self.h = ref().ref_index_create(self.metric, algo, trees, _ptr(rows), rows.shape[0], rows.shape[1])
        if not self.h:
            raise RuntimeError("ref_index_create failed")

    def add(self, rows, rebuild=2.0):
        rows = np.ascontiguousarray(rows, dtype=self.dtype)
        ref().ref_index_add(self.h, _ptr(rows), rows.shape[0], rebuild)

    def remove(self, idx):
        ref().ref_index_remove(self.h, int(idx))

    def knn(self, queries, k=2, checks=32, cores=1):
        queries = np.ascontiguousarray(queries, dtype=self.dtype)
        nq = queries.shape[0]
        idx = np.empty((nq, k), np.uint64)
        dist = np.empty((nq, k), np.float32)
        ref().ref_index_knn(self.h, _ptr(queries), nq, k, checks, cores, _ptr(idx), _ptr(dist))
        return idx.astype(np.int64), dist

    def close(self):
        if self.h:
            ref().ref_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------------------------- VWDictionary / Memory
class OracleVWDictionary:
    """Restated rtabmap::VWDictionary (see lcd_oracle.cpp)."""

    def __init__(self, strategy=kNNBruteForce, incremental=True, nndr=0.8, new_words_compared_together=True,
                 incremental_flann=True, _handle=None, _owner=None):
        self._owner = _owner
        self.h = _handle if _handle is not None else lib().orc_vwd_create(
            strategy, int(incremental), float(nndr), int(new_words_compared_together), int(incremental_flann))

    def close(self):
        if self.h and self._owner is None:
            lib().orc_vwd_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return lib().orc_vwd_last_error(self.h).decode()

    def add_new_words(self, desc, sig_id):
        desc = np.ascontiguousarray(desc)
        out = np.zeros(max(desc.shape[0], 1), np.int32)
        n = lib().orc_vwd_add_new_words(self.h, _ptr(desc), desc.shape[0], desc.shape[1] if desc.ndim == 2 else 0,
                                        _type_of(desc), sig_id, _ptr(out), out.shape[0])
        return [] if n < 0 else out[:n].tolist()

    def find_nn(self, desc):
        desc = np.ascontiguousarray(desc)
        out = np.zeros(desc.shape[0], np.int32)
        lib().orc_vwd_find_nn(self.h, _ptr(desc), desc.shape[0], desc.shape[1], _type_of(desc), _ptr(out))
        return out.tolist()

    def update(self):
        lib().orc_vwd_update(self.h)

    def add_word(self, word_id, desc):
        desc = np.ascontiguousarray(desc).reshape(-1)
        lib().orc_vwd_add_word(self.h, word_id, _ptr(desc), desc.shape[0], _type_of(desc))

    def add_word_ref(self, word_id, sig_id):
        return bool(lib().orc_vwd_add_word_ref(self.h, word_id, sig_id))

    def remove_all_word_ref(self, word_id, sig_id):
        lib().orc_vwd_remove_all_word_ref(self.h, word_id, sig_id)

    def get_unused_word_ids(self):
        n = lib().orc_vwd_get_unused_word_ids(self.h, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        lib().orc_vwd_get_unused_word_ids(self.h, _ptr(out), n)
        return out[:n].tolist()

    def remove_words(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        lib().orc_vwd_remove_words(self.h, _ptr(a), a.shape[0])

    def delete_unused_words(self):
        lib().orc_vwd_delete_unused_words(self.h)

    def clear(self):
        lib().orc_vwd_clear(self.h)

    def stat(self, which):
        return int(lib().orc_vwd_stat(self.h, which))

    visual_words = property(lambda s: s.stat(0))
    not_indexed_words = property(lambda s: s.stat(1))
    indexed_words = property(lambda s: s.stat(2))
    total_active_references = property(lambda s: s.stat(3))
    last_word_id = property(lambda s: s.stat(4))
    unused_words = property(lambda s: s.stat(5))

    def word_refs(self, word_id):
        n = lib().orc_vwd_get_word_refs(self.h, word_id, None, None, 0)
        if n < 0:
            return None
        s = np.zeros(max(n, 1), np.int32)
        c = np.zeros(max(n, 1), np.int32)
        lib().orc_vwd_get_word_refs(self.h, word_id, _ptr(s), _ptr(c), n)
        return dict(zip(s[:n].tolist(), c[:n].tolist()))

    def word_ids(self):
        n = lib().orc_vwd_word_ids(self.h, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        lib().orc_vwd_word_ids(self.h, _ptr(out), n)
        return out[:n].tolist()

    def index_ids(self):
        n = lib().orc_vwd_index_ids(self.h, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        lib().orc_vwd_index_ids(self.h, _ptr(out), n)
        return out[:n].tolist()

    def load_fixed_text(self, path):
        return lib().orc_vwd_load_fixed_text(self.h, path.encode())

    def export_text(self, refs_path, desc_path):
        return lib().orc_vwd_export_text(self.h, (refs_path or "").encode(), (desc_path or "").encode())


class OracleMemory:
    """Restated hot-path subset of rtabmap::Memory (update -> addNewWords, computeLikelihood, forget)."""

    def __init__(self, strategy=kNNBruteForce, incremental=True, nndr=0.8, new_words_compared_together=True,
                 incremental_flann=True):
        self.h = lib().orc_mem_create(strategy, int(incremental), float(nndr), int(new_words_compared_together),
                                      int(incremental_flann))
        self.vwd = OracleVWDictionary(_handle=lib().orc_mem_vwd(self.h), _owner=self)

    def close(self):
        if self.h:
            lib().orc_mem_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, desc, nq=None):
        desc = np.ascontiguousarray(desc)
        rows = desc.shape[0]
        out = np.zeros(max(rows, 1), np.int32)
        sid = lib().orc_mem_update(self.h, _ptr(desc), rows, desc.shape[1], _type_of(desc),
                                   rows if nq is None else nq, _ptr(out))
        return sid, out[:rows].tolist()

    def add_signature(self, word_ids):
        a = np.ascontiguousarray(word_ids, dtype=np.int32)
        return lib().orc_mem_add_signature(self.h, _ptr(a), a.shape[0])

    def add_signatures_bulk(self, words):
        """words: [n_sigs, q] int32.  Same state as add_signature row by row (see lcd_oracle.cpp); returns the first new id."""
        a = np.ascontiguousarray(words, dtype=np.int32)
        return lib().orc_mem_add_signatures_bulk(self.h, _ptr(a), a.shape[0], a.shape[1])

    def add_signature_with_id(self, sig_id, word_ids):
        a = np.ascontiguousarray(word_ids, dtype=np.int32)
        return lib().orc_mem_add_signature_with_id(self.h, sig_id, _ptr(a), a.shape[0])

    def forget(self, sig_id):
        lib().orc_mem_forget(self.h, sig_id)

    def get_ni(self, sig_id):
        return lib().orc_mem_get_ni(self.h, sig_id)

    def num_signatures(self):
        return int(lib().orc_mem_num_signatures(self.h))

    def signature_ids(self):
        n = lib().orc_mem_signature_ids(self.h, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        lib().orc_mem_signature_ids(self.h, _ptr(out), n)
        return out[:n].tolist()

    def compute_likelihood(self, words, ids):
        """Returns (ids ascending, float32 likelihood) as the std::map<int,float> of the reference iterates."""
        w = np.ascontiguousarray(words, dtype=np.int32)
        i = np.ascontiguousarray(ids, dtype=np.int32)
        oid = np.zeros(max(i.shape[0], 1), np.int32)
        out = np.zeros(max(i.shape[0], 1), np.float32)
        n = lib().orc_mem_compute_likelihood(self.h, _ptr(w), w.shape[0], _ptr(i), i.shape[0], _ptr(oid), _ptr(out))
        if n < 0:
            return np.zeros(0, np.int32), np.zeros(0, np.float32)
        return oid[:n], out[:n]


# Bayes/PredictionLC default (reference Parameters.h:363) and Bayes/VirtualPlacePriorThr (:361)
DEFAULT_PREDICTION_LC = [0.1, 0.36, 0.30, 0.16, 0.062, 0.0151, 0.00255, 0.000324, 2.5e-05, 1.3e-06, 4.8e-08, 1.2e-09, 1.9e-11, 2.2e-13,
                         1.7e-15, 8.5e-18, 2.9e-20, 6.9e-23]
DEFAULT_VIRTUAL_PLACE_PRIOR = 0.9


class OracleBayesFilter:
    """BayesFilter (reference BayesFilter.cpp); Memory::getNeighborsId / isInSTM are answered from set_neighbors / set_stm."""

    def __init__(self, prediction_lc=None, virtual_place_prior=DEFAULT_VIRTUAL_PLACE_PRIOR):
        lc = np.ascontiguousarray(DEFAULT_PREDICTION_LC if prediction_lc is None else prediction_lc, dtype=np.float64)
        self.h = lib().orc_bayes_create(_ptr(lc), lc.shape[0], virtual_place_prior)

    def close(self):
        if self.h:
            lib().orc_bayes_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        lib().orc_bayes_reset(self.h)

    def set_neighbors(self, sig_id, nbr_ids, margins):
        a = np.ascontiguousarray(nbr_ids, dtype=np.int32)
        b = np.ascontiguousarray(margins, dtype=np.int32)
        lib().orc_bayes_set_neighbors(self.h, int(sig_id), _ptr(a), _ptr(b), a.shape[0])

    def set_stm(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        lib().orc_bayes_set_stm(self.h, _ptr(a), a.shape[0])

    def compute_posterior(self, ids, likelihood, dense=False, incremental=False):
        """ids ascending (ids[0] may be -1, the virtual place); returns the posterior in the same order (float32).
        dense: the statement-by-statement m x m evaluation; incremental (implies dense): Bayes/FullPredictionUpdate = false, the
        reference's default -- the matrix of the last call patched by updatePrediction (BayesFilter.cpp:502-706)."""
        i = np.ascontiguousarray(ids, dtype=np.int32)
        l = np.ascontiguousarray(likelihood, dtype=np.float32)
        out = np.zeros(i.shape[0], np.float32)
        mode = 2 if incremental else (1 if dense else 0)
        rc = lib().orc_bayes_compute_posterior(self.h, _ptr(i), _ptr(l), i.shape[0], mode, _ptr(out))
        if rc:
            raise RuntimeError("orc_bayes_compute_posterior: %d" % rc)
        return out

    @staticmethod
    def hypothesis(ids, posterior):
        i = np.ascontiguousarray(ids, dtype=np.int32)
        p = np.ascontiguousarray(posterior, dtype=np.float32)
        oid = C.c_int(0)
        val = C.c_float(0)
        lib().orc_bayes_hypothesis(_ptr(i), _ptr(p), i.shape[0], C.byref(oid), C.byref(val))
        return oid.value, val.value
