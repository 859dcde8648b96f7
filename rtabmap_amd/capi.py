"""ctypes binding of include/lcd.h (rtabmap_amd/liblcd_hip.so).

This is plumbing: the product is the shared library.  There is no Python/CPU fallback -- if the library is missing it
is built with hipcc, and if that is impossible the import fails loudly.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import build as _build

LCD_OK = 0
LCD_F32, LCD_U8 = 0, 1
LCD_Q_INCREMENTAL, LCD_Q_NEW_WORDS_COMPARED = 1, 2
LCD_NEW_WORD_IDS_AUTO = -1      # lcd_frame_args.first_new_word_id: the device numbers the frame's new words (include/lcd.h)
STATUS = {0: "LCD_OK", 1: "LCD_ERR_INVALID", 2: "LCD_ERR_HIP", 3: "LCD_ERR_NOMEM", 4: "LCD_ERR_STATE", 5: "LCD_ERR_UNSUPPORTED"}

# every symbol include/lcd.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "lcd_abi_version", "lcd_create", "lcd_destroy", "lcd_last_error", "lcd_synchronize", "lcd_pipeline_depth",
    "lcd_vocab_clear", "lcd_vocab_append", "lcd_vocab_remove", "lcd_vocab_remove_unused", "lcd_vocab_remove_unused_async", "lcd_vocab_rebuild", "lcd_vocab_count", "lcd_vocab_read",
    "lcd_knn2", "lcd_selfdist", "lcd_quantize", "lcd_find_nn",
    "lcd_sig_add", "lcd_sig_remove", "lcd_sig_add_bulk", "lcd_sig_count", "lcd_word_nrefs",
    "lcd_likelihood", "lcd_adjust_likelihood", "lcd_adjust_likelihood_dev", "lcd_frame_dev", "lcd_frame_host", "lcd_slot_count", "lcd_knn2_dev", "lcd_shard_knn2_dev", "lcd_shard_frame_dev", "lcd_finalize_dev", "lcd_slots_dev", "lcd_stream", "lcd_get_stats", "lcd_profile_begin", "lcd_profile_read", "lcd_profile_read_likelihood", "lcd_profile_score_work", "lcd_set_option", "lcd_record_event", "lcd_trace_push", "lcd_trace_pop",
    "lcd_bayes_configure", "lcd_bayes_reset", "lcd_bayes_set_neighbors", "lcd_bayes_update_dev", "lcd_bayes_update", "lcd_bayes_posterior",
]


LCD_KNN_DEFAULT, LCD_KNN_EXACT_VALU, LCD_KNN_F32_MFMA, LCD_KNN_BF16X3, LCD_KNN_F16 = 0, 1, 2, 3, 4
KNN_MODES = {None: 0, "default": 0, "valu": 1, "exact": 1, "mfma32": 2, "f32": 2, "bf16": 3, "bf16x3": 3, "f16": 4, "fp16": 4}


class LcdConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("dtype", C.c_int32), ("dim", C.c_int32),
                ("vocab_capacity", C.c_int64), ("sig_capacity", C.c_int64), ("max_queries", C.c_int32),
                ("knn_mode", C.c_int32), ("stream", C.c_void_p), ("pipeline", C.c_int32), ("reserved1", C.c_int32)]


class LcdHypothesis(C.Structure):
    _fields_ = [("sig_id", C.c_int32), ("slot", C.c_int32), ("likelihood", C.c_float), ("adjusted", C.c_float),
                ("virtual_place", C.c_float), ("mean", C.c_float), ("stddev", C.c_float), ("n_positive", C.c_int32)]


class LcdFrameArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("q", C.c_int32), ("d_descriptors", C.c_void_p), ("flags", C.c_int32),
                ("nndr_ratio", C.c_float), ("sig_id", C.c_int32), ("first_new_word_id", C.c_int32), ("N", C.c_float),
                ("exclude_recent", C.c_int32), ("d_word_ids", C.c_void_p), ("d_likelihood", C.c_void_p),
                ("likelihood_capacity", C.c_int64), ("d_hypothesis", C.c_void_p), ("d_adjusted", C.c_void_p),
                ("virtual_place_ratio", C.c_float), ("append_new_words", C.c_int32), ("d_first_new_word_id", C.c_void_p),
                ("d_posterior", C.c_void_p), ("d_bayes", C.c_void_p)]


class LcdFrameHostArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("q", C.c_int32), ("descriptors", C.c_void_p), ("flags", C.c_int32), ("nndr_ratio", C.c_float),
                ("sig_id", C.c_int32), ("first_new_word_id", C.c_int32), ("N", C.c_float), ("append_new_words", C.c_int32),
                ("word_ids", C.c_void_p), ("likelihood", C.c_void_p), ("likelihood_capacity", C.c_int64), ("n_slots", C.c_void_p)]


class LcdBayesResult(C.Structure):
    _fields_ = [("sig_id", C.c_int32), ("slot", C.c_int32), ("posterior", C.c_float), ("value", C.c_float),
                ("virtual_place", C.c_float), ("n_considered", C.c_int32), ("sum", C.c_float), ("reserved", C.c_int32)]


class LcdStats(C.Structure):
    _fields_ = [("vocab_rows", C.c_int64), ("vocab_live", C.c_int64), ("signatures", C.c_int64), ("postings", C.c_int64),
                ("knn_launches", C.c_int64), ("likelihood_launches", C.c_int64), ("rebuilds", C.c_int64),
                ("buckets_sealed", C.c_int64), ("word_slots", C.c_int64), ("dense_words", C.c_int64),
                ("frame_calls", C.c_int64), ("frame_host_ns", C.c_int64),
                ("bytes_device", C.c_int64), ("knn_last_fallback_queries", C.c_int64), ("knn_max_err_ratio", C.c_double),
                ("clean_divergent_refs", C.c_int64)]


class LcdError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s: %s" % (STATUS.get(status, status), msg))
        self.status = status


_lib = None


def library_path():
    return _build.OUT


def load():
    """Load (building if needed) liblcd_hip.so.  Raises if it cannot be produced: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("LCD_LIB_PATH") or _build.build()      # LCD_LIB_PATH: timing experiments with variant builds
    # PyTorch-ROCm ships its own libamdhip64: a process that loads the system runtime first (through this library) and torch
    # later ends up with two HIP runtimes, and the second one finds no GPU.  When torch is installed, let it load first.
    if "torch" not in sys.modules and not os.environ.get("LCD_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(path)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.lcd_abi_version.restype = C.c_int
    L.lcd_create.argtypes = [C.POINTER(LcdConfig), C.POINTER(vp)]
    L.lcd_destroy.argtypes = [vp]
    L.lcd_destroy.restype = None
    L.lcd_last_error.argtypes = [vp]
    L.lcd_last_error.restype = C.c_char_p
    L.lcd_synchronize.argtypes = [vp]
    L.lcd_pipeline_depth.argtypes = [vp]
    L.lcd_pipeline_depth.restype = C.c_int
    L.lcd_vocab_clear.argtypes = [vp]
    L.lcd_vocab_append.argtypes = [vp, vp, C.c_int, vp]
    L.lcd_vocab_remove.argtypes = [vp, vp, C.c_int]
    L.lcd_vocab_remove_unused.argtypes = [vp, vp, C.c_int, vp]
    L.lcd_vocab_remove_unused_async.argtypes = [vp]
    L.lcd_vocab_rebuild.argtypes = [vp]
    L.lcd_vocab_count.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.lcd_vocab_read.argtypes = [vp, i64, C.c_int, vp, vp]
    L.lcd_knn2.argtypes = [vp, vp, C.c_int, vp, vp]
    L.lcd_selfdist.argtypes = [vp, vp, C.c_int, vp]
    L.lcd_quantize.argtypes = [vp, vp, C.c_int, C.c_int, f32, vp, C.POINTER(i32)]
    L.lcd_find_nn.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, f32, vp]
    L.lcd_sig_add.argtypes = [vp, i32, vp, C.c_int, i32]
    L.lcd_sig_remove.argtypes = [vp, i32]
    L.lcd_sig_add_bulk.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.lcd_sig_count.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.lcd_word_nrefs.argtypes = [vp, i32, C.POINTER(i32)]
    L.lcd_likelihood.argtypes = [vp, vp, C.c_int, vp, C.c_int, f32, vp]
    L.lcd_adjust_likelihood.argtypes = [vp, vp, C.c_int, f32]
    L.lcd_adjust_likelihood_dev.argtypes = [vp, vp, C.c_int, f32]
    L.lcd_frame_dev.argtypes = [vp, C.POINTER(LcdFrameArgs)]
    L.lcd_frame_host.argtypes = [vp, C.POINTER(LcdFrameHostArgs)]
    L.lcd_slot_count.argtypes = [vp, C.POINTER(C.c_int64)]
    L.lcd_knn2_dev.argtypes = [vp, vp, C.c_int, vp, vp]
    L.lcd_shard_knn2_dev.argtypes = [vp, vp, C.c_int, vp]
    L.lcd_shard_frame_dev.argtypes = [vp, vp, C.c_int, C.c_int, f32, i32, i32, f32, C.c_int, C.c_int, vp, i64, vp, vp, i64]
    L.lcd_finalize_dev.argtypes = [vp, vp, i64, vp]
    L.lcd_slots_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.lcd_stream.argtypes = [vp]
    L.lcd_stream.restype = vp
    L.lcd_profile_begin.argtypes = [vp, C.c_int]
    L.lcd_profile_read.argtypes = [vp, C.POINTER(f32), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
    L.lcd_profile_read_likelihood.argtypes = [vp, C.POINTER(f32), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
    L.lcd_get_stats.argtypes = [vp, C.POINTER(LcdStats)]
    L.lcd_profile_score_work.argtypes = [vp, C.POINTER(i64)]
    L.lcd_set_option.argtypes = [vp, C.c_char_p, i64]
    L.lcd_record_event.argtypes = [vp, vp]
    L.lcd_bayes_configure.argtypes = [vp, vp, C.c_int, f32]
    L.lcd_bayes_reset.argtypes = [vp]
    L.lcd_bayes_set_neighbors.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.lcd_bayes_update_dev.argtypes = [vp, vp, C.c_int, vp, vp]
    L.lcd_bayes_update.argtypes = [vp, vp, vp, C.c_int, vp]
    L.lcd_bayes_posterior.argtypes = [vp, vp, C.c_int, vp]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One lcd_engine handle.  numpy in / numpy out through the C-ABI; *_dev methods take raw device pointers."""

    def __init__(self, dtype, dim, device=0, vocab_capacity=0, sig_capacity=0, stream=None, knn_mode=None, pipeline=False):
        self.L = load()
        self.dtype = LCD_F32 if dtype in (LCD_F32, np.float32, "f32") else LCD_U8
        self.np_dtype = np.float32 if self.dtype == LCD_F32 else np.uint8
        self.dim = int(dim)
        if knn_mode is None:                              # test runs of the whole suite on another filter (this glue only: the library reads no environment)
            knn_mode = os.environ.get("LCD_PY_KNN_MODE") or None
        mode = KNN_MODES[knn_mode] if (knn_mode is None or isinstance(knn_mode, str)) else int(knn_mode)
        cfg = LcdConfig(C.sizeof(LcdConfig), device, self.dtype, self.dim, vocab_capacity, sig_capacity, 0, mode, stream,
                        1 if pipeline else 0, 0)
        h = C.c_void_p()
        rc = self.L.lcd_create(C.byref(cfg), C.byref(h))
        if rc != LCD_OK:
            raise LcdError(rc, "lcd_create failed (no gfx950 device / HIP runtime?)")
        self.h = h
        # experiments: LCD_PY_OPTS="key=value,..." sets engine options on every handle this glue creates (the library reads no environment)
        for kv in filter(None, os.environ.get("LCD_PY_OPTS", "").split(",")):
            self._ck(self.L.lcd_set_option(self.h, kv.split("=")[0].encode(), int(kv.split("=")[1])))

    def close(self):
        if getattr(self, "h", None):
            self.L.lcd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != LCD_OK:
            raise LcdError(rc, self.L.lcd_last_error(self.h).decode())

    def _rows(self, a):
        a = np.ascontiguousarray(a, dtype=self.np_dtype)
        if a.ndim != 2 or a.shape[1] != self.dim:
            raise ValueError("expected [n, %d] %s" % (self.dim, self.np_dtype.__name__))
        return a

    def synchronize(self):
        self._ck(self.L.lcd_synchronize(self.h))

    def pipeline_depth(self):
        """later frame_dev calls that still enqueue work of a frame (0: plain handle): keep pipeline_depth() + 1 buffer sets"""
        return int(self.L.lcd_pipeline_depth(self.h))

    # ---- vocabulary
    def vocab_clear(self):
        self._ck(self.L.lcd_vocab_clear(self.h))

    def vocab_append(self, rows, word_ids):
        rows = self._rows(rows)
        ids = np.ascontiguousarray(word_ids, dtype=np.int32)
        assert ids.shape[0] == rows.shape[0]
        self._ck(self.L.lcd_vocab_append(self.h, _p(rows), rows.shape[0], _p(ids)))

    def vocab_remove(self, word_ids):
        ids = np.ascontiguousarray(word_ids, dtype=np.int32)
        self._ck(self.L.lcd_vocab_remove(self.h, _p(ids), ids.shape[0]))

    def vocab_remove_unused(self, capacity=0):
        """Memory::cleanUnusedWords from the device's reference counts -> (number removed, the first `capacity` removed ids)"""
        out = np.zeros(max(capacity, 1), np.int32)
        n = C.c_int32()
        self._ck(self.L.lcd_vocab_remove_unused(self.h, _p(out) if capacity else None, capacity, C.byref(n)))
        return n.value, out[: min(n.value, capacity)]

    def vocab_remove_unused_async(self):
        """cleanUnusedWords enqueued behind the frames in flight: nothing is completed, nothing comes back"""
        self._ck(self.L.lcd_vocab_remove_unused_async(self.h))

    def vocab_rebuild(self):
        self._ck(self.L.lcd_vocab_rebuild(self.h))

    def vocab_count(self):
        r, l = C.c_int64(), C.c_int64()
        self._ck(self.L.lcd_vocab_count(self.h, C.byref(r), C.byref(l)))
        return r.value, l.value

    def vocab_read(self, first, n):
        rows = np.empty((n, self.dim), self.np_dtype)
        ids = np.empty(n, np.int32)
        self._ck(self.L.lcd_vocab_read(self.h, first, n, _p(rows), _p(ids)))
        return rows, ids

    # ---- search
    def knn2(self, queries):
        q = self._rows(queries)
        ids = np.zeros((q.shape[0], 2), np.int32)
        dist = np.zeros((q.shape[0], 2), np.float32)
        self._ck(self.L.lcd_knn2(self.h, _p(q), q.shape[0], _p(ids), _p(dist)))
        return ids, dist

    def selfdist(self, queries):
        q = self._rows(queries)
        out = np.zeros((q.shape[0], q.shape[0]), np.float32)
        self._ck(self.L.lcd_selfdist(self.h, _p(q), q.shape[0], _p(out)))
        return out

    def quantize(self, descriptors, incremental=True, new_words_compared=True, nndr=0.8):
        q = self._rows(descriptors)
        out = np.zeros(q.shape[0], np.int32)
        nn = C.c_int32()
        flags = (LCD_Q_INCREMENTAL if incremental else 0) | (LCD_Q_NEW_WORDS_COMPARED if new_words_compared else 0)
        self._ck(self.L.lcd_quantize(self.h, _p(q), q.shape[0], flags, nndr, _p(out), C.byref(nn)))
        return out, nn.value

    def find_nn(self, queries, extra_rows=None, extra_word_ids=None, incremental=True, nndr=0.8):
        q = self._rows(queries)
        out = np.zeros(q.shape[0], np.int32)
        ne = 0
        er = ei = None
        if extra_rows is not None and len(extra_rows):
            er = self._rows(extra_rows)
            ei = np.ascontiguousarray(extra_word_ids, dtype=np.int32)
            ne = er.shape[0]
        flags = LCD_Q_INCREMENTAL if incremental else 0
        self._ck(self.L.lcd_find_nn(self.h, _p(q), q.shape[0], _p(er), _p(ei), ne, flags, nndr, _p(out)))
        return out

    # ---- inverted index
    def sig_add(self, sig_id, word_ids, ni=None):
        w = np.ascontiguousarray(word_ids, dtype=np.int32)
        self._ck(self.L.lcd_sig_add(self.h, sig_id, _p(w), w.shape[0], w.shape[0] if ni is None else ni))

    def sig_add_bulk(self, sig_ids, offsets, word_ids, ni=None):
        s = np.ascontiguousarray(sig_ids, dtype=np.int32)
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        w = np.ascontiguousarray(word_ids, dtype=np.int32)
        n = None if ni is None else np.ascontiguousarray(ni, dtype=np.int32)
        self._ck(self.L.lcd_sig_add_bulk(self.h, s.shape[0], _p(s), _p(o), _p(w), _p(n)))

    def sig_remove(self, sig_id):
        self._ck(self.L.lcd_sig_remove(self.h, sig_id))

    def sig_count(self):
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.L.lcd_sig_count(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def word_nrefs(self, word_id):
        v = C.c_int32()
        self._ck(self.L.lcd_word_nrefs(self.h, word_id, C.byref(v)))
        return v.value

    def likelihood(self, query_word_ids, sig_ids, N):
        w = np.ascontiguousarray(query_word_ids, dtype=np.int32)
        s = np.ascontiguousarray(sig_ids, dtype=np.int32)
        out = np.zeros(s.shape[0], np.float32)
        self._ck(self.L.lcd_likelihood(self.h, _p(w), w.shape[0], _p(s), s.shape[0], float(N), _p(out)))
        return out

    def adjust_likelihood_dev(self, d_ptr, n, ratio=0.0):
        self._ck(self.L.lcd_adjust_likelihood_dev(self.h, d_ptr, n, ratio))

    def adjust_likelihood(self, L, ratio=0.0):
        a = np.ascontiguousarray(L, dtype=np.float32).copy()
        self._ck(self.L.lcd_adjust_likelihood(self.h, _p(a), a.shape[0], ratio))
        return a

    # ---- device-resident frame path
    def frame_dev(self, d_desc_ptr, q, sig_id, N, d_word_ids_ptr, d_like_ptr, like_capacity, incremental=True,
                  new_words_compared=True, nndr=0.8, first_new_word_id=0, d_hypothesis_ptr=None, d_adjusted_ptr=None,
                  exclude_recent=0, virtual_place_ratio=0.0, d_first_new_word_id_ptr=None, d_posterior_ptr=None, d_bayes_ptr=None, append_new_words=False):
        flags = (LCD_Q_INCREMENTAL if incremental else 0) | (LCD_Q_NEW_WORDS_COMPARED if new_words_compared else 0)
        a = LcdFrameArgs(C.sizeof(LcdFrameArgs), q, d_desc_ptr, flags, nndr, sig_id, first_new_word_id, float(N), exclude_recent,
                         d_word_ids_ptr, d_like_ptr, like_capacity, d_hypothesis_ptr, d_adjusted_ptr, virtual_place_ratio,
                         1 if append_new_words else 0, d_first_new_word_id_ptr, d_posterior_ptr, d_bayes_ptr)
        self._ck(self.L.lcd_frame_dev(self.h, C.byref(a)))

    def frame_host(self, desc, sig_id, N, incremental=True, new_words_compared=True, nndr=0.8, first_new_word_id=0, append_new_words=False,
                   want_likelihood=True):
        """lcd_frame_host: host descriptors in, (word ids, dense likelihood over the signature slots) out, one synchronisation."""
        d = np.ascontiguousarray(desc)
        q = d.shape[0]
        flags = (LCD_Q_INCREMENTAL if incremental else 0) | (LCD_Q_NEW_WORDS_COMPARED if new_words_compared else 0)
        n = C.c_int64(0)
        self._ck(self.L.lcd_slot_count(self.h, C.byref(n)))
        words = np.zeros(q, np.int32)
        like = np.zeros(n.value + 1, np.float32) if want_likelihood else None
        ns = C.c_int64(0)
        a = LcdFrameHostArgs(C.sizeof(LcdFrameHostArgs), q, _p(d), flags, nndr, sig_id, first_new_word_id, float(N), 1 if append_new_words else 0,
                             _p(words), _p(like) if like is not None else None, like.shape[0] if like is not None else 0,
                             C.cast(C.byref(ns), C.c_void_p))
        self._ck(self.L.lcd_frame_host(self.h, C.byref(a)))
        return words, (like[: ns.value] if like is not None else None)

    def frame_args(self, **kw):
        """A reusable argument block for frame_dev_args (callers in a tight loop change a few fields per frame)."""
        a = LcdFrameArgs()
        a.struct_size = C.sizeof(LcdFrameArgs)
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    def frame_dev_args(self, a):
        rc = self.L.lcd_frame_dev(self.h, C.byref(a))
        if rc != LCD_OK:
            self._ck(rc)

    # ---- Bayes filter (BayesFilter.cpp)
    def bayes_configure(self, prediction_lc, virtual_place_prior=0.9):
        lc = np.ascontiguousarray(prediction_lc, dtype=np.float64)
        self._ck(self.L.lcd_bayes_configure(self.h, _p(lc), lc.shape[0], virtual_place_prior))

    def bayes_reset(self):
        self._ck(self.L.lcd_bayes_reset(self.h))

    def bayes_set_neighbors(self, sig_ids, offsets, nbr_sig_ids, nbr_margins):
        s = np.ascontiguousarray(sig_ids, dtype=np.int32)
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        n = np.ascontiguousarray(nbr_sig_ids, dtype=np.int32)
        m = np.ascontiguousarray(nbr_margins, dtype=np.int32)
        assert o.shape[0] == s.shape[0] + 1 and n.shape[0] == m.shape[0]
        self._ck(self.L.lcd_bayes_set_neighbors(self.h, s.shape[0], _p(s), _p(o), _p(n), _p(m)))

    def bayes_neighbors_prepared(self, sig_ids, offsets, nbr_sig_ids, nbr_margins):
        """The argument pointers of bayes_set_neighbors for caller-owned arrays (int32 / int64 / int32 / int32, C-contiguous), built
        once for a tight loop: the arrays' CONTENTS may change between calls, their shapes may not."""
        assert sig_ids.dtype == np.int32 and offsets.dtype == np.int64 and nbr_sig_ids.dtype == np.int32 and nbr_margins.dtype == np.int32
        assert offsets.shape[0] == sig_ids.shape[0] + 1 and nbr_sig_ids.shape[0] == nbr_margins.shape[0]
        return (sig_ids.shape[0], _p(sig_ids), _p(offsets), _p(nbr_sig_ids), _p(nbr_margins), (sig_ids, offsets, nbr_sig_ids, nbr_margins))

    def bayes_set_neighbors_prepared(self, prep):
        rc = self.L.lcd_bayes_set_neighbors(self.h, prep[0], prep[1], prep[2], prep[3], prep[4])
        if rc != LCD_OK:
            self._ck(rc)

    def bayes_update_dev(self, d_adjusted_ptr, exclude_recent=0, d_posterior_ptr=None, d_result_ptr=None):
        self._ck(self.L.lcd_bayes_update_dev(self.h, d_adjusted_ptr, exclude_recent, d_posterior_ptr, d_result_ptr))

    def bayes_update(self, sig_ids, adjusted):
        """lcd_bayes_update: the likelihood as parallel host arrays in std::map order (-1 first); returns the highest hypothesis."""
        s = np.ascontiguousarray(sig_ids, dtype=np.int32)
        a = np.ascontiguousarray(adjusted, dtype=np.float32)
        assert s.shape[0] == a.shape[0]
        r = LcdBayesResult()
        self._ck(self.L.lcd_bayes_update(self.h, _p(s), _p(a), s.shape[0], C.byref(r)))
        return r

    def bayes_posterior(self, sig_ids):
        s = np.ascontiguousarray(sig_ids, dtype=np.int32)
        out = np.zeros(s.shape[0], np.float32)
        self._ck(self.L.lcd_bayes_posterior(self.h, _p(s), s.shape[0], _p(out)))
        return out

    def knn2_dev(self, d_queries_ptr, q, d_word_ids_ptr, d_dist_ptr):
        self._ck(self.L.lcd_knn2_dev(self.h, d_queries_ptr, q, d_word_ids_ptr, d_dist_ptr))

    def shard_knn2_dev(self, d_desc_ptr, q, d_cand_ptr):
        self._ck(self.L.lcd_shard_knn2_dev(self.h, d_desc_ptr, q, d_cand_ptr))

    def shard_frame_dev(self, d_desc_ptr, q, sig_id, N, rank, world, d_all_cand_ptr, total_live_rows, d_word_ids_ptr, d_lfix_ptr,
                        lfix_capacity, incremental=True, new_words_compared=True, nndr=0.8, first_new_word_id=0):
        flags = (LCD_Q_INCREMENTAL if incremental else 0) | (LCD_Q_NEW_WORDS_COMPARED if new_words_compared else 0)
        self._ck(self.L.lcd_shard_frame_dev(self.h, d_desc_ptr, q, flags, nndr, sig_id, first_new_word_id, float(N), rank, world,
                                            d_all_cand_ptr, total_live_rows, d_word_ids_ptr, d_lfix_ptr, lfix_capacity))

    def finalize_dev(self, d_lfix_ptr, n, d_like_ptr):
        self._ck(self.L.lcd_finalize_dev(self.h, d_lfix_ptr, n, d_like_ptr))

    def slots_dev(self):
        p, n = C.c_void_p(), C.c_int64()
        self._ck(self.L.lcd_slots_dev(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def stream(self):
        return self.L.lcd_stream(self.h)

    def profile_begin(self, max_samples):
        self._ck(self.L.lcd_profile_begin(self.h, max_samples))

    def profile_read(self):
        ms, n, name = C.c_float(), C.c_int(), C.c_char_p()
        self._ck(self.L.lcd_profile_read(self.h, C.byref(ms), C.byref(n), C.byref(name)))
        return ms.value, n.value, (name.value or b"").decode()

    def profile_read_likelihood(self):
        ms, n, name = C.c_float(), C.c_int(), C.c_char_p()
        self._ck(self.L.lcd_profile_read_likelihood(self.h, C.byref(ms), C.byref(n), C.byref(name)))
        return ms.value, n.value, (name.value or b"").decode()

    def record_event(self, event_handle):
        self._ck(self.L.lcd_record_event(self.h, event_handle))

    def set_option(self, key, value):
        self._ck(self.L.lcd_set_option(self.h, key.encode(), int(value)))

    def profile_score_work(self):
        out = (C.c_int64 * 8)()
        self._ck(self.L.lcd_profile_score_work(self.h, out))
        keys = ["dense_row_bytes", "sparse_postings", "directory_lookups", "directory_hits", "open_log_entries", "postings",
                "unique_words", "dense_words"]
        return {k: int(v) for k, v in zip(keys, out)}

    def stats(self):
        s = LcdStats()
        self._ck(self.L.lcd_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in LcdStats._fields_}
