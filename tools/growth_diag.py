#!/usr/bin/env python
"""Diagnostic: certificate failures and filter error per frame while an incremental dictionary grows (append_new_words, pipelined handle)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtabmap_amd  # noqa: E402
from rtabmap_amd import synth  # noqa: E402


def main():
    n_words, q, n_sig = 49000, 500, 20000
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(n_sig, q, n_words, seed=100000)
    eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n_words + 65536, sig_capacity=n_sig + 4096, pipeline=True, knn_mode=os.environ.get("KNN_MODE") or None)
    eng.vocab_append(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
    cap = n_sig + 4096
    d_words = torch.zeros((64, q), dtype=torch.int32, device="cuda")
    d_like = torch.zeros((4, cap), dtype=torch.float32, device="cuda")
    frames = [torch.from_numpy(synth.frame_from_signature(vocab, words[i * 11], seed=i)).cuda() for i in range(48)]
    out = []
    burst = int(os.environ.get("BURST", "1"))
    for i in range(48):
        eng.frame_dev(frames[i].data_ptr(), q, n_sig + 1 + i, float(n_sig + 1), d_words[i].data_ptr(), d_like[i % 4].data_ptr(), cap,
                      first_new_word_id=n_words + 1 + i * q, append_new_words=True)
        if i % burst == burst - 1:
            st = eng.stats()
            out.append((i, st["vocab_rows"], st["knn_last_fallback_queries"], round(st["knn_max_err_ratio"], 4)))
    print("frame, rows, fallback queries of the last 2-NN, max |score - distance| / eps so far")
    for o in out:
        print(o)
    eng.close()


if __name__ == "__main__":
    main()
