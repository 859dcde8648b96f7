"""The interval set of recycled postings keys (host code of liblcd_hip.so, no GPU): freeing a batch of verdicts run by run
(Tfidf::free_wslot_run, what the engine does since round 4) leaves exactly the intervals that freeing key by key leaves."""
import ctypes as C

import numpy as np

import rtabmap_amd


def _intervals(lib, keys, ok, by_runs, take=0):
    keys = np.ascontiguousarray(keys, np.int32)
    ok = np.ascontiguousarray(ok, np.uint8)
    out = np.zeros(2 * (len(keys) + 4), np.int32)
    cnt = C.c_longlong(0)
    lib.lcd_debug_key_intervals.restype = C.c_int
    n = lib.lcd_debug_key_intervals(keys.ctypes.data_as(C.c_void_p), ok.ctypes.data_as(C.c_void_p), len(keys), int(by_runs), int(take),
                                    out.ctypes.data_as(C.c_void_p), len(keys) + 4, C.byref(cnt))
    assert n >= 0
    return out[: 2 * n].reshape(-1, 2).tolist(), cnt.value


def test_runs_and_single_keys_leave_the_same_intervals():
    lib = rtabmap_amd.load()
    rng = np.random.default_rng(5)
    for trial in range(40):
        # what a batch looks like: the keys frames reserved (runs of up to 500 consecutive keys, in any order of the runs), most of them
        # unused (verdict "free"), some used or still referenced in the middle of a run
        starts = rng.permutation(np.arange(0, 64 * 600, 600))[: rng.integers(1, 40)]
        keys = np.concatenate([np.arange(s, s + rng.integers(1, 500)) for s in starts])
        ok = (rng.random(len(keys)) < (0.95 if trial % 2 else 0.6)).astype(np.uint8)
        a, ca = _intervals(lib, keys, ok, by_runs=False)
        b, cb = _intervals(lib, keys, ok, by_runs=True)
        assert a == b and ca == cb == int(ok.sum())
        free = np.unique(keys[ok == 1])                       # ... and they are the maximal runs of the SET of freed keys
        cuts = np.flatnonzero(np.diff(free) != 1)
        first = np.concatenate([[0], cuts + 1])
        last = np.concatenate([cuts, [free.size - 1]])
        assert b == [[int(free[i]), int(free[j] - free[i] + 1)] for i, j in zip(first, last)] if free.size else b == []
        # the set is a partition into maximal intervals: sorted, disjoint, not adjacent
        for (s0, l0), (s1, _) in zip(b, b[1:]):
            assert s0 + l0 < s1
        # and taking keys back (the highest first) agrees as well
        take = int(rng.integers(0, max(1, int(ok.sum()))))
        assert _intervals(lib, keys, ok, False, take) == _intervals(lib, keys, ok, True, take)


def test_a_key_freed_twice_is_ignored_either_way():
    lib = rtabmap_amd.load()
    keys = np.array([10, 11, 12, 11, 12, 13, 40, 41, 13], np.int32)
    ok = np.ones(len(keys), np.uint8)
    a, _ = _intervals(lib, keys, ok, by_runs=False)
    b, _ = _intervals(lib, keys, ok, by_runs=True)
    assert a == b == [[10, 4], [40, 2]]
