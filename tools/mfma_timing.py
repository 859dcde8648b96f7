#!/usr/bin/env python
"""Timing experiment: per-wave timestamps inside knn_mfma_filter_kernel (build with LCD_EXTRA_HIPCC_FLAGS=-DLCD_MFMA_TIMING).

Prints, for the last of a few 500 x 49k searches, when (in us after the first wave started) waves entered the kernel, entered
the tile loop, left it and left the kernel.
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtabmap_amd  # noqa: E402
from rtabmap_amd import capi, synth  # noqa: E402


def main():
    n_rows, q = 49000, 500
    v = synth.vocab_surf(n_rows, seed=1)
    qs = synth.queries_surf(v, q, seed=2)
    eng = rtabmap_amd.Engine("f32", 64)
    eng.vocab_append(v, np.arange(1, n_rows + 1, dtype=np.int32))
    for _ in range(5):
        eng.knn2(qs)
    lib = capi.load()
    n_waves = 256 * 4
    buf = (ctypes.c_ulonglong * (4 * 4096))()
    lib.lcd_debug_mfma_timing.restype = ctypes.c_int
    rc = lib.lcd_debug_mfma_timing(buf, 4 * 4096)
    assert rc == 0, rc
    t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)[:n_waves].astype(np.float64)
    t = (t - t[:, 0].min()) / 100.0          # 100 MHz -> us
    names = ["kernel entry", "loop entry", "loop exit", "kernel exit"]
    for i, nme in enumerate(names):
        c = t[:, i]
        print("%-13s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (nme, c.min(), np.percentile(c, 10), np.median(c),
                                                                                   np.percentile(c, 90), c.max()))
    for a, b in ((0, 1), (1, 2), (2, 3)):
        d = t[:, b] - t[:, a]
        print("%-13s -> %-13s min %6.2f  median %6.2f  max %6.2f us" % (names[a], names[b], d.min(), np.median(d), d.max()))
    if hasattr(lib, "lcd_debug_mfma_timing2"):
        buf2 = (ctypes.c_ulonglong * (8 * 4096))()
        lib.lcd_debug_mfma_timing2.restype = ctypes.c_int
        assert lib.lcd_debug_mfma_timing2(buf2, 8 * 4096) == 0
        u = np.frombuffer(buf2, dtype=np.uint64).reshape(-1, 8)[:n_waves, :6].astype(np.float64) / 100.0
        ok = u[:, 0] > 0
        d = np.diff(u[ok], axis=1)
        print("loop trips (start to start), us, median / p90 over %d waves:" % ok.sum())
        for i in range(d.shape[1]):
            print("  trip %d  median %5.2f  p90 %5.2f" % (i, np.median(d[:, i]), np.percentile(d[:, i], 90)))
    eng.close()


if __name__ == "__main__":
    main()
