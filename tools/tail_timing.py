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


def main():
    n_words, q, n_sig = 49000, 500, 3000
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(n_sig, q, n_words, seed=100000)
    eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n_words + 1024, sig_capacity=n_sig + 4096)
    eng.vocab_append(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    offsets = np.arange(0, (n_sig + 1) * q, q, dtype=np.int64)
    eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), offsets, words.reshape(-1))
    d_words = torch.zeros(q, dtype=torch.int32, device="cuda")
    cap = n_sig + 4096
    d_like = torch.zeros(cap, dtype=torch.float32, device="cuda")
    lib = capi.load()
    lib.lcd_debug_tail_timing.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 8)()
    rows = []
    for i in range(12):
        f = torch.from_numpy(synth.frame_from_signature(vocab, words[i * 11], seed=i)).cuda()
        eng.frame_dev(f.data_ptr(), q, n_sig + 1 + i, float(n_sig + 1), d_words.data_ptr(), d_like.data_ptr(), cap, incremental=True,
                      new_words_compared=True, nndr=0.8)
        assert lib.lcd_debug_tail_timing(buf) == 0
        t = np.array(buf[:4], dtype=np.float64) / 100.0
        rows.append(np.diff(t))
    rows = np.array(rows[2:])
    print("frame tail phases (us, median over %d frames): resolve %.2f  retire %.2f  frame_words %.2f" % (len(rows), *np.median(rows, axis=0)))
    eng.close()


if __name__ == "__main__":
    main()
