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
        if name == "bayes" and os.environ.get("INLINE_FIRST", "0") == "1":
            acc = {"frame_dev": 0.0, "sig_remove": 0.0, "set_neighbors": 0.0}
            n = 300
            tl0 = time.perf_counter()
            for i in range(n):
                a = st.args
                a.d_descriptors = st.ptrs[i % len(st.ptrs)]
                a.sig_id = st.next_sig
                a.first_new_word_id = st.first_new
                t1 = time.perf_counter(); eng.frame_dev_args(a); t2 = time.perf_counter()
                eng.sig_remove(st.oldest); t4 = time.perf_counter()
                st.one_nbr[:] = st.one_base + st.next_sig
                st.one_id[0] = st.next_sig
                t5 = time.perf_counter(); eng.bayes_set_neighbors_prepared(st.one_prep); t6 = time.perf_counter()
                st.next_sig += 1; st.oldest += 1; st.first_new += B.Q
                acc["frame_dev"] += t2 - t1; acc["sig_remove"] += t4 - t2; acc["set_neighbors"] += t6 - t5
            tl1 = time.perf_counter()
            eng.synchronize()
            tl2 = time.perf_counter()
            st.one_id[0] = st.next_sig
            out["bayes_first_loop_us"] = {k: 1e6 * v / n for k, v in acc.items()}
            out["bayes_first_loop_us"].update({"host_loop": 1e6 * (tl1 - tl0) / n, "until_drained": 1e6 * (tl2 - tl0) / n})
        t0 = time.perf_counter()
        for i in range(steps):
            st(20 + i)
        t_host = time.perf_counter() - t0
        eng.synchronize()
        torch.cuda.synchronize()
        out[name] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / steps, "host_ms_per_step": 1e3 * t_host / steps}
        if name == "bayes":
            # host time of the three calls of a step, one by one (device idle in between: every call is followed by a synchronize)
            acc = {"frame_dev": 0.0, "sig_remove": 0.0, "set_neighbors": 0.0}
            for i in range(50):
                a = st.args
                a.d_descriptors = st.ptrs[i % len(st.ptrs)]
                a.sig_id = st.next_sig
                a.first_new_word_id = st.first_new
                torch.cuda.synchronize()
                t1 = time.perf_counter(); eng.frame_dev_args(a); t2 = time.perf_counter()
                torch.cuda.synchronize()
                t3 = time.perf_counter(); eng.sig_remove(st.oldest); t4 = time.perf_counter()
                st.one_nbr[:] = st.one_base + st.next_sig
                st.one_id[0] = st.next_sig
                torch.cuda.synchronize()
                t5 = time.perf_counter(); eng.bayes_set_neighbors_prepared(st.one_prep); t6 = time.perf_counter()
                st.next_sig += 1; st.oldest += 1; st.first_new += B.Q
                acc["frame_dev"] += t2 - t1; acc["sig_remove"] += t4 - t3; acc["set_neighbors"] += t6 - t5
            out["bayes_host_us_per_call"] = {k: 1e6 * v / 50 for k, v in acc.items()}
            out["stats"] = {k: eng.stats()[k] for k in ("frame_calls", "frame_host_ns")}
            # the same three calls back to back (the device queue never drains): where the host's time goes while streaming
            acc = {"frame_dev": 0.0, "sig_remove": 0.0, "set_neighbors": 0.0, "python": 0.0}
            eng.synchronize()
            n = 200
            tl0 = time.perf_counter()
            for i in range(n):
                a = st.args
                a.d_descriptors = st.ptrs[i % len(st.ptrs)]
                a.sig_id = st.next_sig
                a.first_new_word_id = st.first_new
                t1 = time.perf_counter(); eng.frame_dev_args(a); t2 = time.perf_counter()
                eng.sig_remove(st.oldest); t4 = time.perf_counter()
                st.one_nbr[:] = st.one_base + st.next_sig
                st.one_id[0] = st.next_sig
                t5 = time.perf_counter(); eng.bayes_set_neighbors_prepared(st.one_prep); t6 = time.perf_counter()
                st.next_sig += 1; st.oldest += 1; st.first_new += B.Q
                acc["frame_dev"] += t2 - t1; acc["sig_remove"] += t4 - t2; acc["set_neighbors"] += t6 - t5
            tl1 = time.perf_counter()
            eng.synchronize()
            tl2 = time.perf_counter()
            acc["python"] = (tl1 - tl0) - sum(acc.values())
            out["bayes_streaming_us_per_call"] = {k: 1e6 * v / n for k, v in acc.items()}
            out["bayes_streaming_us_per_step"] = {"host_loop": 1e6 * (tl1 - tl0) / n, "until_drained": 1e6 * (tl2 - tl0) / n}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
