"""Experiment: do two frame streams on two HIP streams share one MI355X?  The launches of a pipelined frame are latency chains, not
throughput: if two independent handles on two streams run at ~2x the aggregate rate, the chip has room to run launch A and launch B
of ONE stream side by side once they stop depending on each other.  Prints ms/frame of one handle alone and of two at once.

    python tools/two_streams.py [frames]
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B          # noqa: E402


def main():
    import torch
    import rtabmap_amd
    from rtabmap_amd import synth
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    n_sig = B.N_SIG
    vocab, words = B.make_state(n_sig)
    cap = n_sig + 4 * steps + 4096
    engines = []
    for s in range(2):
        rng = np.random.default_rng(7 + s)
        src = rng.integers(0, n_sig, 64)
        fr = [torch.from_numpy(synth.frame_from_signature(vocab, words[x], seed=1000 * s + i)).cuda() for i, x in enumerate(src)]
        stream = torch.cuda.Stream()
        eng = rtabmap_amd.Engine("f32", B.DIM, device=0, vocab_capacity=B.N_WORDS + 1024 + B.Q * (4 * steps + 64), sig_capacity=n_sig + 8192,
                                 stream=stream.cuda_stream, pipeline=1, knn_mode=B.KNN_MODE)
        B.load_engine(eng, vocab, words)
        engines.append((eng, B.Stepper(eng, torch, fr, n_sig, cap), stream))

    def run(idx, n, base):
        eng, st, _ = engines[idx]
        for i in range(n):
            st(base + i)

    def timed(which, n, base):
        for e, _, _ in engines:
            e.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(i, n, base)) for i in which]
        for t in th:
            t.start()
        for t in th:
            t.join()
        t_enq = time.perf_counter() - t0
        for i in which:
            engines[i][0].synchronize()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, t_enq

    timed([0, 1], 40, 0)
    base = 40
    for which in ([0], [1], [0, 1], [0], [0, 1]):
        w, enq = timed(which, steps, base)
        base += steps
        print("handles %s: %d frames each, %.4f ms per frame-slot (aggregate %.4f ms/frame), host enqueue %.4f ms per slot"
              % (which, steps, 1e3 * w / steps, 1e3 * w / (steps * len(which)), 1e3 * enq / steps), flush=True)
    for e, _, _ in engines:
        e.close()


if __name__ == "__main__":
    main()
