#!/usr/bin/env python
"""Cost of VWDictionary::update() on the device-resident vocabulary (not part of bench.py's step): per frame the reference
removes the words that lost their last reference and appends the frame's new words (VWDictionary.cpp:475-701).
Prints one JSON line with the wall time of append / remove+append+rebuild for 150 words on a 49k-word SURF vocabulary."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import rtabmap_amd
    from rtabmap_amd import synth
    n, k = 49000, 150
    vocab = synth.vocab_surf(n)
    eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n + 100000)
    eng.vocab_append(vocab, np.arange(1, n + 1, dtype=np.int32))
    new = synth.vocab_surf(k * 64, seed=7)
    next_id = n + 1
    t_app, t_reb = [], []
    for it in range(40):
        rows = new[(it % 64) * k:(it % 64 + 1) * k]
        ids = np.arange(next_id, next_id + k, dtype=np.int32)
        next_id += k
        t0 = time.perf_counter()
        eng.vocab_append(rows, ids)                         # append branch
        eng.synchronize()
        t1 = time.perf_counter()
        dead = ids - k * 1 if it else np.arange(1, k + 1, dtype=np.int32)
        eng.vocab_remove(dead)                              # rebuild branch: tombstone + compact/order
        eng.vocab_rebuild()
        eng.synchronize()
        t2 = time.perf_counter()
        t_app.append(t1 - t0)
        t_reb.append(t2 - t1)
    print(json.dumps({"vocab_rows": n, "words_per_frame": k, "append_ms_median": 1e3 * float(np.median(t_app[5:])),
                      "remove_rebuild_ms_median": 1e3 * float(np.median(t_reb[5:]))}))
    eng.close()


if __name__ == "__main__":
    main()
