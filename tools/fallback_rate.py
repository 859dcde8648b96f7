#!/usr/bin/env python
"""Diagnostic: how many queries per bench-like frame fail the filter certificate (and go to the exact row-parallel scan)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtabmap_amd  # noqa: E402
from rtabmap_amd import synth  # noqa: E402


def main():
    n_words, q, n_sig = 49000, 500, 2000
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(n_sig, q, n_words, seed=100000)
    eng = rtabmap_amd.Engine("f32", 64)
    eng.vocab_append(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    fbs = []
    for i in range(16):
        f = synth.frame_from_signature(vocab, words[i * 7], seed=i)
        ids, d = eng.knn2(f)
        st = eng.stats()
        fbs.append(st["knn_last_fallback_queries"])
        if i == 0:
            print("frame 0: d1 quantiles", np.quantile(d[:, 0], [0, 0.1, 0.5, 0.9, 1.0]), "exact-zero d1:", int((d[:, 0] == 0).sum()))
    print("fallback queries per frame:", fbs, " max err/eps:", eng.stats()["knn_max_err_ratio"])
    eng.close()


if __name__ == "__main__":
    main()
