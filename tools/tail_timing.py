#!/usr/bin/env python
"""Timing experiment: phases of frame_tail_kernel (build with LCD_EXTRA_HIPCC_FLAGS=-DLCD_TAIL_TIMING or a variant lib via LCD_LIB_PATH)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtabmap_amd  # noqa: E402
from rtabmap_amd import capi, synth  # noqa: E402


def score_timing(lib, eng, vocab, words, n_sig, q, d_words, d_like, cap):
    """phases of score_sealed_body per workgroup (LCD_SCORE_TIMING build)"""
    lib.lcd_debug_score_timing.restype = ctypes.c_int
    for i in range(6):
        f = torch.from_numpy(synth.frame_from_signature(vocab, words[i * 11], seed=i)).cuda()
        eng.frame_dev(f.data_ptr(), q, n_sig + 1 + i, float(n_sig + 1), d_words.data_ptr(), d_like.data_ptr(), cap, incremental=True,
                      new_words_compared=True, nndr=0.8)
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    assert lib.lcd_debug_score_timing(buf, 1024 * 8) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)[:, :4].astype(np.float64) / 100.0
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print("%d sealed-bucket workgroups; start spread %.2f us" % (len(t), t[:, 0].max() - t0))
    d = np.diff(t, axis=1)
    for i, nme in enumerate(["lists + directory + dense rows", "dense flush (LDS)", "sparse postings"]):
        print("  %-28s median %5.2f  p90 %5.2f us" % (nme, np.median(d[:, i]), np.percentile(d[:, i], 90)))
    print("  last workgroup leaves %.2f us after the first started" % (t[:, 3].max() - t0))
    eng.close()


def main():
    n_words, q, n_sig = 49000, 500, int(os.environ.get("N_SIG", "100000"))
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(n_sig, q, n_words, seed=100000)
    eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n_words + 1024, sig_capacity=n_sig + 4096, pipeline=bool(os.environ.get("PIPE")))
    if os.environ.get("SCORE_BLOCK"):
        eng.set_option("score_block", int(os.environ["SCORE_BLOCK"]))
    eng.vocab_append(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    offsets = np.arange(0, (n_sig + 1) * q, q, dtype=np.int64)
    eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), offsets, words.reshape(-1))
    d_words = torch.zeros(q, dtype=torch.int32, device="cuda")
    cap = n_sig + 4096
    d_like = torch.zeros(cap, dtype=torch.float32, device="cuda")
    lib = capi.load()
    if hasattr(lib, "lcd_debug_score_timing"):
        return score_timing(lib, eng, vocab, words, n_sig, q, d_words, d_like, cap)
    lib.lcd_debug_tail_timing.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 64)()
    rows, rrows = [], []
    for i in range(12):
        f = torch.from_numpy(synth.frame_from_signature(vocab, words[i * 11], seed=i)).cuda()
        eng.frame_dev(f.data_ptr(), q, n_sig + 1 + i, float(n_sig + 1), d_words.data_ptr(), d_like.data_ptr(), cap, incremental=True,
                      new_words_compared=True, nndr=0.8)
        if os.environ.get("PIPE"):      # the tail of this frame runs inside the next frame's filter launch (256-thread workgroup)
            eng.frame_dev(f.data_ptr(), q, 0, float(n_sig + 1), d_words.data_ptr(), 0, 0)
            torch.cuda.synchronize()
        assert (lib.lcd_debug_tail_timing_pipe if os.environ.get("PIPE") else lib.lcd_debug_tail_timing)(buf) == 0
        t = np.array(buf[:4], dtype=np.float64) / 100.0
        rows.append(np.diff(t))
        rrows.append(np.diff(np.array(buf[8:14], dtype=np.float64) / 100.0))
    rows = np.array(rows[2:])
    print("frame tail phases (us, median over %d frames): resolve %.2f  retire %.2f  frame_words %.2f" % (len(rows), *np.median(rows, axis=0)))
    if os.environ.get("PIPE"):
        sw = np.array(buf[16:48], dtype=np.float64).reshape(4, 8) / 100.0
        print("  first sweep, per wave (us from the wave's entry): after barrier | descriptor 0..3 done | before 2nd barrier | after")
        for w in range(4):
            print("    wave %d: %s" % (w, "  ".join("%.2f" % (x - sw[w, 0]) for x in sw[w, 1:])))
    rr = np.median(np.array(rrows[2:]), axis=0)
    print("  inside the decision loop (us): loads + bit rows %.2f  sweep 0 %.2f  sweeps %.2f  prefix %.2f  output %.2f" % tuple(rr))
    eng.close()


if __name__ == "__main__":
    main()
