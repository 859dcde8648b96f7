#!/usr/bin/env python
"""Where the HOST time of a pipelined lcd_frame_dev call goes (lcd_debug_host_profile: section timers inside the library): the headline
stream, 400 frames back to back.  python tools/host_profile.py  ->  us per call and section."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtabmap_amd  # noqa: E402
from rtabmap_amd import capi, synth  # noqa: E402

NAMES = ["checks + throttle + capacity", "ring reservations", "reserve_frame_words + decision-loop args", "registration + scoring args", "filter plan (build_knn)",
         "launch A", "launch B", "rest of pipeline_launch"]


def main():
    n_words, q, n_sig = 49000, 500, 100000
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(n_sig, q, n_words, seed=100000)
    eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n_words + 65536, sig_capacity=n_sig + 8192, pipeline=True, knn_mode="f16")
    eng.vocab_append(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
    frames = [torch.from_numpy(synth.frame_from_signature(vocab, words[i * 13], seed=i)).cuda() for i in range(64)]
    cap = n_sig + 8192
    d_w = torch.zeros((4, q), dtype=torch.int32, device="cuda")
    d_l = torch.zeros((4, cap), dtype=torch.float32, device="cuda")
    lib = capi.load()
    lib.lcd_debug_host_profile.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    a = eng.frame_args(q=q, flags=3, nndr_ratio=0.8, append_new_words=1)
    first_new = n_words + 1

    def run(n, base):
        nonlocal first_new
        for i in range(n):
            a.d_descriptors = frames[(base + i) % 64].data_ptr()
            a.sig_id = n_sig + 1 + base + i
            a.N = float(n_sig + 1)
            a.first_new_word_id = first_new
            first_new += q
            a.d_word_ids = d_w[i % 4].data_ptr()
            a.d_likelihood = d_l[i % 4].data_ptr()
            a.likelihood_capacity = cap
            eng.frame_dev_args(a)
            eng.sig_remove(base + i + 1)
    run(100, 0)
    eng.synchronize()
    before = (C.c_int64 * 9)()
    lib.lcd_debug_host_profile(eng.h, before)
    t0 = time.perf_counter()
    run(400, 100)
    t_enq = time.perf_counter() - t0
    eng.synchronize()
    t_all = time.perf_counter() - t0
    after = (C.c_int64 * 9)()
    lib.lcd_debug_host_profile(eng.h, after)
    calls = after[8] - before[8]
    print("pipelined lcd_frame_dev, %d calls: enqueue loop %.2f us per frame (python included), until the device was done %.2f us per frame" % (calls, 1e6 * t_enq / 400, 1e6 * t_all / 400))
    tot = 0.0
    for i, nm in enumerate(NAMES):
        us = 1e-3 * (after[i] - before[i]) / max(calls, 1)
        tot += us
        print("  %-45s %6.2f us" % (nm, us))
    print("  %-45s %6.2f us" % ("sum of the sections", tot))
    eng.close()


if __name__ == "__main__":
    main()
