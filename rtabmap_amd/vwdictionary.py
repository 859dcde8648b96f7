"""ctypes view of the C++ host mirror (rtabmap_amd/host: VWDictionaryHip / MemoryHip over the C-ABI).

Same method names as the reference classes (snake_case): update / add_new_words / find_nn / add_word_ref /
remove_all_word_ref / compute_likelihood ...  All search and scoring runs on the device through liblcd_hip.so.
"""
import ctypes as C

import numpy as np

from . import build as _build

kNNFlannNaive, kNNFlannKdTree, kNNFlannLSH, kNNBruteForce, kNNBruteForceGPU, kNNBruteForceHIP = 0, 1, 2, 3, 4, 5
_lib = None


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _type_of(a):
    if a.dtype == np.float32:
        return 0
    if a.dtype == np.uint8:
        return 1
    raise TypeError("descriptors must be float32 or uint8")


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_build.build_host())
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.hvwd_create.restype = vp
        L.hvwd_create.argtypes = [ci, ci, cf, ci, C.c_char_p, ci]
        L.hvwd_destroy.argtypes = [vp]
        L.hvwd_available.argtypes = [vp]
        L.hvwd_last_error.argtypes = [vp]
        L.hvwd_last_error.restype = C.c_char_p
        L.hvwd_add_new_words.argtypes = [vp, vp, ci, ci, ci, ci, vp, ci]
        L.hvwd_find_nn.argtypes = [vp, vp, ci, ci, ci, vp]
        L.hvwd_update.argtypes = [vp]
        L.hvwd_add_word.argtypes = [vp, ci, vp, ci, ci]
        L.hvwd_add_word_ref.argtypes = [vp, ci, ci]
        L.hvwd_remove_all_word_ref.argtypes = [vp, ci, ci]
        L.hvwd_get_unused_word_ids.argtypes = [vp, vp, ci]
        L.hvwd_delete_unused_words.argtypes = [vp]
        L.hvwd_clear.argtypes = [vp]
        L.hvwd_rebuild_engine.argtypes = [vp]
        L.hvwd_rebuild_engine.restype = ci
        L.hvwd_stat.argtypes = [vp, ci]
        L.hvwd_stat.restype = C.c_long
        L.hvwd_get_word_refs.argtypes = [vp, ci, vp, vp, ci]
        L.hvwd_index_ids.argtypes = [vp, vp, ci]
        L.hvwd_export_text.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.hmem_create.restype = vp
        L.hmem_create.argtypes = [ci, ci, cf, ci, C.c_char_p, ci]
        L.hmem_create_stm.restype = vp
        L.hmem_create_stm.argtypes = [ci, ci, cf, ci, C.c_char_p, ci, ci]
        L.hmem_time_loop.argtypes = [vp, vp, ci, ci, ci, ci, ci]
        L.hmem_time_loop.restype = C.c_double
        L.hmem_time_loop_modes.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.hmem_set_device_frames.argtypes = [vp, ci]
        L.hmem_set_device_frames.restype = None
        L.hmem_fast_frame_device_ms.argtypes = [vp]
        L.hmem_fast_frame_device_ms.restype = C.c_double
        L.hmem_add_signatures_bulk.argtypes = [vp, vp, ci, ci, ci]
        L.hmem_compute_likelihood_flat.argtypes = [vp, ci, vp, vp, ci]
        L.hmem_compute_likelihood_of.argtypes = [vp, ci, vp, ci, vp, vp]
        L.hmem_add_link.argtypes = [vp, ci, ci, ci]
        L.hmem_get_neighbors_id.argtypes = [vp, ci, ci, vp, vp, ci]
        L.hmem_ids.argtypes = [vp, ci, vp, ci]
        L.hbayes_create.restype = vp
        L.hbayes_create.argtypes = [C.c_char_p, cf, ci]
        L.hbayes_destroy.argtypes = [vp]
        L.hbayes_reset.argtypes = [vp]
        L.hbayes_set_prediction_lc.argtypes = [vp, C.c_char_p]
        L.hbayes_get_prediction_lc.argtypes = [vp, vp, ci]
        L.hbayes_compute_posterior.argtypes = [vp, vp, vp, vp, ci, vp, vp, ci, vp, vp]
        L.hbayes_last_error.argtypes = [vp]
        L.hbayes_last_error.restype = C.c_char_p
        L.hrtab_create.restype = vp
        L.hrtab_create.argtypes = [cf, cf, ci, ci, C.c_char_p, cf, ci]
        L.hrtab_destroy.argtypes = [vp]
        L.hrtab_memory.restype = vp
        L.hrtab_memory.argtypes = [vp]
        L.hrtab_process.argtypes = [vp, vp, ci, ci, ci, vp, vp]
        L.hrtab_vector.argtypes = [vp, ci, vp, vp, ci]
        L.hrtab_last_word_ids.argtypes = [vp, vp, ci]
        L.hmem_destroy.argtypes = [vp]
        L.hmem_vwd.restype = vp
        L.hmem_vwd.argtypes = [vp]
        L.hmem_update.argtypes = [vp, vp, ci, ci, ci, ci, vp]
        L.hmem_add_signature.argtypes = [vp, ci, vp, ci]
        L.hmem_forget.argtypes = [vp, ci]
        L.hmem_statistic.argtypes = [vp, C.c_char_p, ci]
        L.hmem_statistic.restype = C.c_float
        L.hmem_set_engine_option.argtypes = [vp, C.c_char_p, C.c_long]
        L.hmem_get_ni.argtypes = [vp, ci]
        L.hmem_num_signatures.argtypes = [vp]
        L.hmem_num_signatures.restype = C.c_long
        L.hmem_compute_likelihood.argtypes = [vp, vp, ci, vp, ci, vp, vp]
        L.hmem_load_data_from_db.argtypes = [vp, C.c_char_p, ci]
        L.hmem_load_error.argtypes = [vp]
        L.hmem_load_error.restype = C.c_char_p
        _lib = L
    return _lib


class VWDictionaryHip:
    def __init__(self, strategy=kNNBruteForceHIP, incremental=True, nndr=0.8, new_words_compared_together=True,
                 dictionary_path="", device=0, _handle=None, _owner=None):
        self._owner = _owner
        self.h = _handle if _handle is not None else lib().hvwd_create(
            strategy, int(incremental), float(nndr), int(new_words_compared_together), dictionary_path.encode(), device)

    def close(self):
        if self.h and self._owner is None:
            lib().hvwd_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return lib().hvwd_last_error(self.h).decode()

    def add_new_words(self, desc, sig_id):
        desc = np.ascontiguousarray(desc)
        out = np.zeros(max(desc.shape[0], 1), np.int32)
        n = lib().hvwd_add_new_words(self.h, _p(desc), desc.shape[0], desc.shape[1] if desc.ndim == 2 else 0, _type_of(desc),
                                     sig_id, _p(out), out.shape[0])
        return out[:n].tolist()

    def find_nn(self, desc):
        desc = np.ascontiguousarray(desc)
        out = np.zeros(desc.shape[0], np.int32)
        lib().hvwd_find_nn(self.h, _p(desc), desc.shape[0], desc.shape[1], _type_of(desc), _p(out))
        return out.tolist()

    def update(self):
        lib().hvwd_update(self.h)

    def add_word(self, word_id, desc):
        desc = np.ascontiguousarray(desc).reshape(-1)
        lib().hvwd_add_word(self.h, word_id, _p(desc), desc.shape[0], _type_of(desc))

    def add_word_ref(self, word_id, sig_id):
        return bool(lib().hvwd_add_word_ref(self.h, word_id, sig_id))

    def remove_all_word_ref(self, word_id, sig_id):
        lib().hvwd_remove_all_word_ref(self.h, word_id, sig_id)

    def get_unused_word_ids(self):
        n = lib().hvwd_get_unused_word_ids(self.h, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        lib().hvwd_get_unused_word_ids(self.h, _p(out), n)
        return out[:n].tolist()

    def delete_unused_words(self):
        lib().hvwd_delete_unused_words(self.h)

    def clear(self):
        lib().hvwd_clear(self.h)

    def rebuild_engine(self):
        """VWDictionaryHip::rebuildEngine: a fresh device handle, replayed from the host maps (recovery after a device fault)"""
        return bool(lib().hvwd_rebuild_engine(self.h))

    def stat(self, which):
        return int(lib().hvwd_stat(self.h, which))

    visual_words = property(lambda s: s.stat(0))
    not_indexed_words = property(lambda s: s.stat(1))
    indexed_words = property(lambda s: s.stat(2))
    total_active_references = property(lambda s: s.stat(3))
    unused_words = property(lambda s: s.stat(5))

    def word_refs(self, word_id):
        n = lib().hvwd_get_word_refs(self.h, word_id, None, None, 0)
        if n < 0:
            return None
        s = np.zeros(max(n, 1), np.int32)
        c = np.zeros(max(n, 1), np.int32)
        lib().hvwd_get_word_refs(self.h, word_id, _p(s), _p(c), n)
        return dict(zip(s[:n].tolist(), c[:n].tolist()))

    def index_ids(self):
        n = lib().hvwd_index_ids(self.h, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        lib().hvwd_index_ids(self.h, _p(out), n)
        return out[:n].tolist()

    def export_text(self, refs_path, desc_path):
        return lib().hvwd_export_text(self.h, (refs_path or "").encode(), (desc_path or "").encode())


class MemoryHip:
    def __init__(self, strategy=kNNBruteForceHIP, incremental=True, nndr=0.8, new_words_compared_together=True,
                 dictionary_path="", device=0, stm_size=10, _handle=None, _owner=None):
        self._owner = _owner                    # a RtabmapHip owns its memory
        self.h = _handle if _handle is not None else lib().hmem_create_stm(
            strategy, int(incremental), float(nndr), int(new_words_compared_together), dictionary_path.encode(), device, int(stm_size))
        self.vwd = VWDictionaryHip(_handle=lib().hmem_vwd(self.h), _owner=self)

    def close(self):
        if self.h and self._owner is None:
            lib().hmem_destroy(self.h)
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
        sid = lib().hmem_update(self.h, _p(desc), rows, desc.shape[1], _type_of(desc), -1 if nq is None else nq, _p(out))
        return sid, out[:rows].tolist()

    def add_signature(self, word_ids, sig_id=0):
        a = np.ascontiguousarray(word_ids, dtype=np.int32)
        return lib().hmem_add_signature(self.h, sig_id, _p(a), a.shape[0])

    def forget(self, sig_id):
        lib().hmem_forget(self.h, sig_id)

    STAT_NAMES = ("TimingMem/Pre_update/ms", "TimingMem/Joining_dictionary_update/ms", "TimingMem/Add_new_words/ms",
                  "Timing/Likelihood_computation/ms", "Timing/Forgetting/ms", "Keypoint/Dictionary_size/words",
                  "Keypoint/Current_frame/words", "Keypoint/Indexed_words/words", "Keypoint/Index_memory_usage/KB")

    def statistics(self, refresh=False):
        """MemoryHip::getStatistics(): the reference's statistic names (Statistics.h:178-212) -> value of the last frame."""
        out = {}
        for k, name in enumerate(self.STAT_NAMES):
            out[name] = float(lib().hmem_statistic(self.h, name.encode(), 1 if (refresh and k == 0) else 0))
        return out

    def set_engine_option(self, key, value):
        return int(lib().hmem_set_engine_option(self.h, key.encode(), int(value)))

    def load_data_from_db(self, path, last_state_only=True):
        """Memory::loadDataFromDb from a RTAB-Map database file (MemoryHip::loadDataFromDb, rtabmap_amd/host/DbLoaderHip.h): the
        dictionary indexed by one update(), every signature's references registered on the device in one bulk call.
        Returns the number of signatures loaded; raises with the loader's message on failure."""
        n = lib().hmem_load_data_from_db(self.h, str(path).encode(), int(last_state_only))
        if n < 0:
            raise RuntimeError("loadDataFromDb: " + lib().hmem_load_error(self.h).decode())
        return n

    def time_loop(self, frames, steps):
        """update + computeLikelihood against every signature + forget(oldest) per frame, looped and timed in C++ -> ms per frame"""
        f = np.ascontiguousarray(frames)
        return float(lib().hmem_time_loop(self.h, _p(f), f.shape[0], f.shape[1], f.shape[2], _type_of(f), int(steps)))

    def time_loop_modes(self, frames, steps, mode):
        """The same loop with the caller's list of ids kept from frame to frame; mode 0: std::map by value (the reference's signature),
        1: into a caller-owned std::map updated in place, 2: flat vectors.  -> ms per frame {step, update, likelihood, forget}"""
        f = np.ascontiguousarray(frames)
        out = np.zeros(4, np.float64)
        if lib().hmem_time_loop_modes(self.h, _p(f), f.shape[0], f.shape[1], f.shape[2], _type_of(f), int(steps), int(mode), _p(out)) != 0:
            raise RuntimeError("hmem_time_loop_modes failed: " + self.vwd.last_error())
        return dict(zip(("step", "update", "likelihood", "forget"), out.tolist()))

    def fast_frame_device_ms(self):
        """mean ms a device-resident update() spent inside lcd_frame_host (the rest of update() is the mirror's std::map bookkeeping)"""
        return float(lib().hmem_fast_frame_device_ms(self.h))

    def set_device_frames(self, on):
        """MemoryHip::setDeviceFrames: update() as ONE device call (lcd_frame_host) that also brings the likelihood back (default), or the
        call-by-call path (lcd_quantize / lcd_sig_add / lcd_likelihood)."""
        lib().hmem_set_device_frames(self.h, int(bool(on)))

    def add_signatures_bulk(self, words, first_id=1):
        """n signatures (rows of `words`) through Memory::addSignature in C++, then ONE bulk registration on the device"""
        w = np.ascontiguousarray(words, dtype=np.int32)
        n = lib().hmem_add_signatures_bulk(self.h, _p(w), w.shape[0], w.shape[1], int(first_id))
        if n != w.shape[0]:
            raise RuntimeError("add_signatures_bulk failed: " + self.vwd.last_error())
        return n

    def compute_likelihood_of(self, sig_id, ids):
        """Memory::computeLikelihood(signature id, ids) -> (ids ascending, values)"""
        a = np.ascontiguousarray(ids, dtype=np.int32)
        oi, ov = np.zeros(max(a.shape[0], 1), np.int32), np.zeros(max(a.shape[0], 1), np.float32)
        n = lib().hmem_compute_likelihood_of(self.h, int(sig_id), _p(a), a.shape[0], _p(oi), _p(ov))
        return oi[:n], ov[:n]

    def compute_likelihood_flat(self, sig_id, cap=1 << 21):
        oi, ov = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        n = lib().hmem_compute_likelihood_flat(self.h, int(sig_id), _p(oi), _p(ov), cap)
        if n < 0:
            return None
        return oi[:n], ov[:n]

    def get_ni(self, sig_id):
        return lib().hmem_get_ni(self.h, sig_id)

    def num_signatures(self):
        return int(lib().hmem_num_signatures(self.h))

    def add_link(self, a, b, neighbor=False):
        """Memory::addLink for a global loop closure between two signatures in memory (neighbor=True: an odometry link of a
        signature replayed from the database)."""
        return bool(lib().hmem_add_link(self.h, a, b, 0 if neighbor else 1))

    def get_neighbors_id(self, sig_id, max_graph_depth):
        cap = 4096
        ids, mg = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = lib().hmem_get_neighbors_id(self.h, sig_id, max_graph_depth, _p(ids), _p(mg), cap)
        assert n <= cap
        return dict(zip(ids[:n].tolist(), mg[:n].tolist()))

    def _ids(self, which):
        n = lib().hmem_ids(self.h, which, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        lib().hmem_ids(self.h, which, _p(out), n)
        return out[:n].tolist()

    def st_mem(self):
        return self._ids(0)

    def working_mem(self):
        """Ids of the working memory, the virtual place (-1) first, as Memory::getWorkingMem()."""
        return self._ids(1)

    def compute_likelihood(self, words, ids):
        w = np.ascontiguousarray(words, dtype=np.int32)
        i = np.ascontiguousarray(ids, dtype=np.int32)
        oid = np.zeros(max(i.shape[0], 1), np.int32)
        out = np.zeros(max(i.shape[0], 1), np.float32)
        n = lib().hmem_compute_likelihood(self.h, _p(w), w.shape[0], _p(i), i.shape[0], _p(oid), _p(out))
        return oid[:n], out[:n]


class BayesFilterHip:
    """rtabmap::BayesFilter's interface over the device filter (rtabmap_amd/host/BayesFilterHip.h)."""

    def __init__(self, prediction_lc="", virtual_place_prior=0.9, full_prediction_update=False):
        self.h = lib().hbayes_create(prediction_lc.encode(), float(virtual_place_prior), int(full_prediction_update))
        self.highest_hypothesis = (0, 0.0)

    def close(self):
        if self.h:
            lib().hbayes_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        lib().hbayes_reset(self.h)

    def set_prediction_lc(self, s):
        lib().hbayes_set_prediction_lc(self.h, s.encode())

    def get_prediction_lc(self):
        out = np.zeros(64, np.float64)
        n = lib().hbayes_get_prediction_lc(self.h, _p(out), 64)
        return out[:n].copy()

    def compute_posterior(self, memory, ids, values):
        """likelihood = {ids[i]: values[i]}; returns (ids, posterior) in std::map order."""
        i = np.ascontiguousarray(ids, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=np.float32)
        cap = max(i.shape[0], 1) + 8
        oid, out = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        hid, hv = C.c_int(0), C.c_float(0.0)
        n = lib().hbayes_compute_posterior(self.h, memory.h, _p(i), _p(v), i.shape[0], _p(oid), _p(out), cap, C.byref(hid), C.byref(hv))
        self.highest_hypothesis = (hid.value, hv.value)
        return oid[:n].copy(), out[:n].copy()

    def last_error(self):
        return lib().hbayes_last_error(self.h).decode()


class RtabmapHip:
    """The loop-closure detection block of Rtabmap::process (rtabmap_amd/host/RtabmapHip.h): one frame of descriptors in, the highest
    hypothesis and the accepted loop closure out."""

    def __init__(self, loop_thr=0.11, loop_ratio=0.0, virtual_place_likelihood_ratio=0, stm_size=10, prediction_lc="",
                 virtual_place_prior=0.9, device=0):
        self.h = lib().hrtab_create(float(loop_thr), float(loop_ratio), int(virtual_place_likelihood_ratio), int(stm_size),
                                    prediction_lc.encode(), float(virtual_place_prior), device)
        self.memory = MemoryHip(_handle=lib().hrtab_memory(self.h), _owner=self)

    def close(self):
        if self.h:
            self.memory.h = None
            lib().hrtab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process(self, desc):
        """-> dict(id, highest=(id, value), loop=(id, value))"""
        desc = np.ascontiguousarray(desc)
        out4 = np.zeros(4, np.int32)
        val2 = np.zeros(2, np.float32)
        lib().hrtab_process(self.h, _p(desc), desc.shape[0], desc.shape[1], _type_of(desc), _p(out4), _p(val2))
        return {"id": int(out4[0]), "highest": (int(out4[1]), float(val2[0])), "loop": (int(out4[2]), float(val2[1])), "ok": bool(out4[3])}

    def vector(self, which):
        """which: "raw" | "likelihood" | "posterior" of the last frame -> (ids, values) in std::map order"""
        w = {"raw": 0, "likelihood": 1, "posterior": 2}[which]
        n = lib().hrtab_vector(self.h, w, None, None, 0)
        ids, out = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float32)
        lib().hrtab_vector(self.h, w, _p(ids), _p(out), n)
        return ids[:n].copy(), out[:n].copy()

    def last_word_ids(self):
        out = np.zeros(16384, np.int32)
        n = lib().hrtab_last_word_ids(self.h, _p(out), 16384)
        return out[:n].tolist()
