#!/usr/bin/env python
"""The bench step with the Bayes filter behind it (bench.py's BayesStepper), on its own: ms per step, host time per call.
    python tools/bench_bayes.py [steps]          (rocprofv3 --kernel-trace --stats -- python tools/bench_bayes.py for the kernel times)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    import torch
    import rtabmap_amd
    from rtabmap_amd import synth
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_sig = int(os.environ.get("N_SIG", B.N_SIG))
    pipeline = int(os.environ.get("PIPE", "1"))
    vocab, words = B.make_state(n_sig)
    rng = np.random.default_rng(7)
    frames_np = [synth.frame_from_signature(vocab, words[s], seed=i) for i, s in enumerate(rng.integers(0, n_sig, 32))]
    stream = torch.cuda.Stream()
    d_frames = [torch.from_numpy(f).cuda() for f in frames_np]
    eng = rtabmap_amd.Engine("f32", B.DIM, vocab_capacity=B.N_WORDS + 4096, sig_capacity=n_sig + 8192, stream=stream.cuda_stream, pipeline=pipeline)
    B.load_engine(eng, vocab, words)
    cap = n_sig + 8192
    out = {}
    for name, cls in (("plain", B.Stepper), ("bayes", B.BayesStepper)):
        st = cls(eng, torch, d_frames, n_sig, cap) if name == "plain" else None
        if name == "bayes":
            eng.close()
            eng = rtabmap_amd.Engine("f32", B.DIM, vocab_capacity=B.N_WORDS + 4096, sig_capacity=n_sig + 8192, stream=stream.cuda_stream, pipeline=pipeline)
            B.load_engine(eng, vocab, words)
            st = cls(eng, torch, d_frames, n_sig, cap)
        for i in range(20):
            st(i)
        eng.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            st(20 + i)
        t_host = time.perf_counter() - t0
        eng.synchronize()
        torch.cuda.synchronize()
        out[name] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / steps, "host_ms_per_step": 1e3 * t_host / steps}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
