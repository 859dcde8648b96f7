#!/usr/bin/env python
"""bench.py -- loop-closure candidates/sec of the MI355X engine on BASELINE.json's headline configuration.

One STEP = one query frame through the hot path, inputs already resident in HBM:
    500 SURF-64 descriptors -> exact 2-NN against the 49k-word vocabulary + NNDR + same-frame resolution
    (VWDictionary::addNewWords) -> the frame's references registered in the inverted index and the oldest signature
    retired (memory stays at 100k signatures, as Rtabmap's WM->LTM transfer keeps it) -> TF-IDF likelihood of the frame
    against every signature (Memory::computeLikelihood).
    candidates/sec = frames/sec x N_signatures.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the step -- the longer of the 2-NN filter and the fused
TF-IDF scoring kernel, both bracketed by HIP events inside the timed region (the other one is reported next to it as
`roofline_other`); `cpu_baseline` times the
reference-style CPU path (the reference's own rtflann kd-tree when oracle/_ref is present + the restated std::map
computeLikelihood) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_WORDS, N_SIG, Q, DIM = 49000, 100000, 500, 64
NNDR = 0.8
# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md): fp32 matrix/vector 157.3 TFLOP/s, bf16 MFMA 2.5 PFLOP/s dense, HBM3E 8 TB/s
PEAK_F32_TFLOPS = 157.3
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBPS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_state(eng, rank, world, n_sig, seed=100000):
    from rtabmap_amd import synth
    t0 = time.time()
    vocab = synth.vocab_surf(N_WORDS)
    words = synth.zipf_words(n_sig, Q, N_WORDS, seed=seed)
    eng.vocab_append(vocab, np.arange(1, N_WORDS + 1, dtype=np.int32))
    offsets = np.arange(0, (n_sig + 1) * Q, Q, dtype=np.int64)
    chunk = 10000
    for a in range(0, n_sig, chunk):
        b = min(a + chunk, n_sig)
        eng.sig_add_bulk(np.arange(a + 1, b + 1, dtype=np.int32), offsets[a:b + 1] - offsets[a], words[a:b].reshape(-1))
    log("[bench] state built in %.1fs: %d words, %d signatures" % (time.time() - t0, N_WORDS, n_sig))
    return vocab, words


def cpu_baseline(vocab, words, frames, sample_sigs, n_frames):
    """Reference-style CPU path on a bounded sample: kd-tree 2-NN (the reference default, Kp/NNStrategy=1: 4 trees,
    32 checks, 1 thread -- the REAL rtflann when oracle/_ref is there, else the exact linear port) + NNDR, then the
    restated std::map Memory::computeLikelihood over `sample_sigs` signatures.  candidates/s = frames/s x sample_sigs."""
    import oracle as O
    t0 = time.time()
    m = O.OracleMemory(strategy=O.kNNBruteForce)
    for w in range(1, N_WORDS + 1):
        m.vwd.add_word(w, np.zeros(1, np.float32))
    for s in range(sample_sigs):
        m.add_signature(words[s])
    ids = np.arange(1, sample_sigs + 1, dtype=np.int32)
    kind = "port"
    index = None
    if O.have_ref():
        index = O.RefIndex(vocab, algo=O.ALGO_KDTREE, trees=4)
        kind = "reference"
    log("[bench] cpu baseline state (%d signatures) built in %.1fs" % (sample_sigs, time.time() - t0))
    t_knn = t_lik = 0.0
    for f in range(n_frames):
        desc = frames[f]
        t1 = time.perf_counter()
        if index is not None:
            idx, dist = index.knn(desc, k=2, checks=32, cores=1)
        else:
            idx, dist = O.knn2_linear(vocab, desc)
        accept = ~(dist[:, 0] > np.float32(NNDR) * dist[:, 1])
        qwords = (idx[accept, 0] + 1).astype(np.int32)
        t2 = time.perf_counter()
        m.compute_likelihood(qwords, ids)
        t3 = time.perf_counter()
        t_knn += t2 - t1
        t_lik += t3 - t2
    per_frame = (t_knn + t_lik) / n_frames
    return {"value": sample_sigs / per_frame, "unit": "candidates/s", "cores": 1, "kind": kind,
            "sample": "%d frames x %d descriptors; 2-NN = %s over the full 49k vocabulary (%.1f ms/frame), TF-IDF = restated "
                      "std::map computeLikelihood over a %d-signature memory (%.1f ms/frame); rate scaled by the sample's "
                      "signature count" % (n_frames, Q, "reference rtflann kd-tree (4 trees, 32 checks)" if index is not None
                                           else "exact linear port", 1e3 * t_knn / n_frames, sample_sigs, 1e3 * t_lik / n_frames)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--signatures", type=int, default=N_SIG)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallelism", choices=["replicas", "shard"], default="replicas",
                    help="N > 1: independent frame streams per GPU (weak scaling, no data-path collective) or ONE stream with the "
                         "vocabulary sharded by word-id range + all-gather / all-reduce per frame (strong scaling)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LCD_BENCH_BACKEND", "nccl")      # "gloo" only for single-GPU sanity runs of the N > 1 code path
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import rtabmap_amd
    from rtabmap_amd import synth
    stream = torch.cuda.Stream()
    n_sig = args.signatures
    shard = world > 1 and args.parallelism == "shard"
    if shard:
        from rtabmap_amd.sharded import ShardedLoopClosure
        sh = ShardedLoopClosure("f32", DIM, rank=rank, world=world, device=local, stream=stream, vocab_capacity=N_WORDS + 1024,
                                sig_capacity=n_sig + 8192)
        eng = sh.eng
        t0 = time.time()
        vocab = synth.vocab_surf(N_WORDS)
        words = synth.zipf_words(n_sig, Q, N_WORDS, seed=100000)
        sh.load_vocabulary(vocab, np.arange(1, N_WORDS + 1, dtype=np.int32))
        offsets = np.arange(0, (n_sig + 1) * Q, Q, dtype=np.int64)
        for a in range(0, n_sig, 10000):
            b = min(a + 10000, n_sig)
            w = words[a:b].reshape(-1)
            sh.add_signatures_bulk(np.arange(a + 1, b + 1, dtype=np.int32), offsets[a:b + 1] - offsets[a], w,
                                   owned_mask=(w > sh.lo) & (w <= sh.hi))
        log("[bench] rank %d: sharded state built in %.1fs (rows %d..%d)" % (rank, time.time() - t0, sh.lo, sh.hi))
    else:
        eng = rtabmap_amd.Engine("f32", DIM, device=local, vocab_capacity=N_WORDS + 1024, sig_capacity=n_sig + 8192,
                                 stream=stream.cuda_stream)
        vocab, words = build_state(eng, rank, world, n_sig)

    # frames resident in HBM: revisits of earlier places (70 % of the descriptors quantise back to that place's words).
    # replicas: every rank has its own stream of frames; shard: all ranks see the same frames.
    n_frames = min(64, max(8, args.steps))
    rng = np.random.default_rng(7 + (0 if shard else rank))
    src = rng.integers(0, n_sig, n_frames)
    frames = [synth.frame_from_signature(vocab, words[s], seed=1000 * (0 if shard else rank) + i) for i, s in enumerate(src)]
    d_frames = [torch.from_numpy(f).cuda() for f in frames]
    d_words = torch.zeros(Q, dtype=torch.int32, device="cuda")
    cap = n_sig + args.steps + args.warmup + 4096
    d_like = torch.zeros(cap, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    next_sig = n_sig + 1
    oldest = 1
    last_like = [None]

    def step(i):
        nonlocal next_sig, oldest
        if shard:
            _, last_like[0] = sh.frame(d_frames[i % n_frames], next_sig, float(n_sig + 1), incremental=True, new_words_compared=True,
                                       nndr=NNDR)
            sh.retire(oldest)
        else:
            eng.frame_dev(d_frames[i % n_frames].data_ptr(), Q, next_sig, float(n_sig + 1), d_words.data_ptr(), d_like.data_ptr(), cap,
                          incremental=True, new_words_compared=True, nndr=NNDR)
            eng.sig_remove(oldest)
        next_sig += 1
        oldest += 1

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.profile_begin(max(10, args.steps // 5))   # HIP events around the dominant kernel of the first 20 % of the timed steps (engine stream)
    t0 = time.perf_counter()
    e0.record(stream)
    for i in range(args.steps):
        step(args.warmup + i)
    e1.record(stream)
    host_enqueue = time.perf_counter() - t0          # host time to enqueue the timed steps (diagnostic: launch-bound if ~ wall)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    sc_ms, sc_n, sc_name = eng.profile_read_likelihood()
    kern_ms, kern_n, kern_name = eng.profile_read()
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # sanity of the last frame: the revisited place must be the arg-max (excluding the frame itself)
    like = (last_like[0] if shard else d_like[: n_sig + args.steps + args.warmup]).cpu().numpy()
    last = (args.warmup + args.steps - 1) % n_frames

    # ---- rooflines, from the events recorded inside the timed region (first 20 % of the steps).
    # (1) 2-NN filter.  ALGORITHMIC work per launch: 2*Q*N*D = 3.136 GFLOP GEMM-equivalent over this rank's rows.
    #     The bf16x3 filter executes three bf16 products per algorithmic product (+ the f32 augmentation step): `executed` counts
    #     those, `achieved` only the algorithmic ones, both against the dense MFMA peak of the type the kernel multiplies in.
    n_rows_rank = (sh.hi - sh.lo) if shard else N_WORDS
    flops = 2.0 * Q * n_rows_rank * DIM
    achieved = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
    bf16 = "bf16" in kern_name
    peak = PEAK_BF16_TFLOPS if bf16 else PEAK_F32_TFLOPS

    def pmc_traffic(name):
        # HBM traffic per launch: from the committed rocprofv3 --pmc summary (FETCH_SIZE / WRITE_SIZE cannot be read from inside
        # the process); null when the profile is not there or was taken for another vocabulary split
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
            return pmc[name]["hbm_bytes_per_launch"] / 1e9 if (not shard and name in pmc) else None
        except Exception:
            return None

    roof_knn = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": pmc_traffic(kern_name), "traffic_unit": "GB per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE)", "kernel": kern_name,
                "ms": kern_ms, "samples": kern_n, "mfma_dtype": "bf16 (3 products per fp32 product, fp32 accumulate)" if bf16 else "f32",
                "executed_tflops": (3.0 if bf16 else 1.0) * achieved,
                "frac_of_f32_mfma_peak": achieved / PEAK_F32_TFLOPS,
                "algorithmic_gbps": (n_rows_rank * DIM * 4 + Q * DIM * 4 + Q * 16) / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0}
    # (2) fused TF-IDF scoring kernel (single-GPU path).  ALGORITHMIC bytes per launch: 4 B per posting of the frame's words (this
    #     engine packs a posting in 4 B; SURVEY.md 8d budgets 8) + ni read + likelihood write (4 B per signature each).
    roof_score = None
    if not shard and sc_ms > 0:
        t0h = time.time()
        srt = np.sort(words, axis=1)
        first = np.ones_like(srt, dtype=bool)
        first[:, 1:] = srt[:, 1:] != srt[:, :-1]
        npost = np.bincount(srt[first], minlength=N_WORDS + 1)              # signatures that contain each word (initial memory)
        fw = np.unique(d_words.cpu().numpy())
        fw = fw[(fw > 0) & (fw <= N_WORDS)]
        p_frame = int(npost[fw].sum())
        sc_bytes = 4.0 * p_frame + 8.0 * n_sig
        gbps = sc_bytes / (sc_ms * 1e-3) / 1e9
        roof_score = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                      "traffic": pmc_traffic(sc_name), "traffic_unit": "GB per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE)", "kernel": sc_name,
                      "ms": sc_ms, "samples": sc_n, "postings_per_launch": p_frame, "algorithmic_bytes_per_launch": sc_bytes}
        log("[bench] postings of the last frame's words: %d (%.1fs)" % (p_frame, time.time() - t0h))
    if roof_score is not None and roof_score["ms"] > roof_knn["ms"]:
        roofline, roofline_other = roof_score, roof_knn
    else:
        roofline, roofline_other = roof_knn, roof_score

    out = {
        "metric": "loop-closure candidates/sec (49k vocab, 100k signatures, 500 desc/frame)",
        "value": (1 if shard else world) * args.steps * n_sig / wall,
        "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SURF-64 fp32 brute-force 2-NN (49k words) + NNDR + TF-IDF likelihood (%d signatures x 500 words, "
                               "Zipf), 500 desc/frame, 1 frame/step" % n_sig,
                   "frames_per_s": (1 if shard else world) * args.steps / wall, "device_ms_per_step": dev_ms / args.steps,
                   "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
                   "parallelism": ("vocabulary sharded by word-id range over %d GPUs (all-gather top-2 + int64 all-reduce)" % world) if shard
                   else ("%d independent replicas (one frame stream per GPU, no data-path collective)" % world if world > 1 else "1 GPU")},
        "roofline": roofline,
        "roofline_other": roofline_other,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU leg runs at N = 1 only
        out["cpu_baseline"] = cpu_baseline(vocab, words, frames, sample_sigs=min(10000, n_sig), n_frames=3)
    if rank == 0:
        exp_top = int(src[last]) + 1
        got_top = int(np.argmax(like[:n_sig])) + 1
        out["config"]["last_frame_top_candidate_ok"] = bool(got_top == exp_top or exp_top < oldest)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
