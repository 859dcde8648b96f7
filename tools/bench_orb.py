#!/usr/bin/env python
"""Config 3 of BASELINE.json (ORB 256-bit Hamming 2-NN, 200k-word vocabulary, 500 descriptors/frame, 1 GPU): kernel time of
the exact Hamming scan (events around knn2_hamming_kernel) + the device frame path.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import rtabmap_amd
    from rtabmap_amd import synth
    n_words, q, n_sig = 200000, 500, int(os.environ.get("ORB_SIGS", "20000"))
    vocab = synth.vocab_orb(n_words)
    stream = torch.cuda.Stream()
    eng = rtabmap_amd.Engine("u8", 32, vocab_capacity=n_words + 1024, sig_capacity=n_sig + 4096, stream=stream.cuda_stream)
    eng.vocab_append(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    words = synth.zipf_words(n_sig, q, n_words, seed=5)
    eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
    frames = [torch.from_numpy(synth.frame_from_signature(vocab, words[i * 7], seed=i, sigma=0.05)).cuda() for i in range(16)]
    d_w = torch.zeros(q * 2, dtype=torch.int32, device="cuda")
    d_d = torch.zeros(q * 2, dtype=torch.float32, device="cuda")
    for i in range(5):
        eng.knn2_dev(frames[i].data_ptr(), q, d_w.data_ptr(), d_d.data_ptr())
    torch.cuda.synchronize()
    reps = 100
    eng.profile_begin(reps)
    for i in range(reps):
        eng.knn2_dev(frames[i % 16].data_ptr(), q, d_w.data_ptr(), d_d.data_ptr())
    ms, n, name = eng.profile_read()
    # whole frames
    d_words = torch.zeros(q, dtype=torch.int32, device="cuda")
    cap = n_sig + 4096
    d_like = torch.zeros(cap, dtype=torch.float32, device="cuda")
    for i in range(10):
        eng.frame_dev(frames[i % 16].data_ptr(), q, n_sig + 1 + i, float(n_sig + 1), d_words.data_ptr(), d_like.data_ptr(), cap)
        eng.sig_remove(1 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 200
    for i in range(steps):
        eng.frame_dev(frames[i % 16].data_ptr(), q, n_sig + 11 + i, float(n_sig + 1), d_words.data_ptr(), d_like.data_ptr(), cap)
        eng.sig_remove(11 + i)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    lane_ops = q * n_words * 19.0          # 8 xor + 8 bcnt + 3 key/min/max per (query, row) pair
    print(json.dumps({"config": "ORB 256-bit Hamming 2-NN, 200k words, 500 desc/frame, %d signatures" % n_sig, "kernel": name,
                      "kernel_ms": ms, "bit_compares_per_s": q * n_words * 256 / (ms * 1e-3),
                      "valu_lane_ops_per_s": lane_ops / (ms * 1e-3), "valu_peak_lane_ops_per_s": 256 * 4 * 32 * 2.4e9,
                      "algorithmic_gbps": (n_words * 32 + q * 48) / (ms * 1e-3) / 1e9, "frame_ms": 1e3 * wall / steps,
                      "frames_per_s": steps / wall}))
    eng.close()


if __name__ == "__main__":
    main()
