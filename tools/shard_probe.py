"""Where the HOST time of a sharded frame goes with two ranks (sharing one GPU): wraps the engine entries and the exchanges of
ShardedLoopClosure.frame with wall-clock timers.  python tools/shard_probe.py [--frames 100] [--exchange p2p|staged]"""
import argparse
import collections
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, a, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from rtabmap_amd import synth
    from rtabmap_amd.sharded import ShardedLoopClosure
    n_words, n_sig, q = a.words, a.signatures, 500
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(n_sig, q, n_words, seed=3)
    sh = ShardedLoopClosure("f32", 64, rank=rank, world=world, device=0, stream=torch.cuda.Stream(), vocab_capacity=n_words + 65536,
                            sig_capacity=n_sig + 8192, knn_mode="f16")
    if a.exchange == "p2p" and world > 1:
        sh.enable_p2p(q, n_sig + 8192)
    sh.force_sharded_path = True
    sh.load_vocabulary(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    w = words.reshape(-1)
    sh.add_signatures_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), w, owned_mask=(w > sh.lo) & (w <= sh.hi))
    if a.append:
        sh.enable_device_append(n_words + 1, 16)
    acc = collections.defaultdict(float)
    if a.driver == "native":
        # the C++ driver of include/lcd_shard.h around the same engine: lcd_shard_frame_deferred per frame, nothing of Python in between
        from rtabmap_amd.sharded import NativeShardComm, P2PTransport
        tr = P2PTransport(rank, world, q * 2 * 16, n_sig + 8192 + 2) if world > 1 else None
        comm = NativeShardComm(sh.eng, rank, world, transport=tr)
        if a.append:
            comm.set_growth(n_words + 1, 16)
            comm.set_append(True)
        cap = n_sig + 8192
        d_w = torch.zeros(q, dtype=torch.int32, device="cuda")
        d_l = [torch.zeros(cap, dtype=torch.float32, device="cuda") for _ in range(2)]
        frames = [torch.from_numpy(synth.frame_from_signature(vocab, words[(37 * t + 5) % n_sig], seed=t)).cuda() for t in range(64)]
        first_new = n_words + 1
        torch.cuda.synchronize()
        dist.barrier()
        for t in range(a.frames):
            if t == a.frames // 2:
                torch.cuda.synchronize(); t_all = time.perf_counter(); n0 = t
            comm.frame(frames[t % 64].data_ptr(), q, n_sig + 1 + t, float(n_sig + 1 + t), n_words, d_w.data_ptr(), d_l[t & 1].data_ptr(), cap,
                       first_new_word_id=first_new, defer=True)
            first_new += q
        t_host = time.perf_counter() - t_all
        torch.cuda.synchronize()
        t_dev = time.perf_counter() - t_all
        n = a.frames - n0
        out.put({"rank": rank, "world": world, "driver": "liblcd_shard.so", "frames_timed": n, "host_ms_per_frame": t_host / n * 1e3,
                 "wall_ms_per_frame": t_dev / n * 1e3, "p2p_status": tr.status() if tr else 0})
        dist.barrier()
        comm.close()
        if tr:
            tr.close()
        sh.close()
        dist.destroy_process_group()
        return

    def wrap(obj, name):
        f = getattr(obj, name)

        def g(*x, **k):
            t0 = time.perf_counter()
            r = f(*x, **k)
            acc[name] += time.perf_counter() - t0
            return r
        setattr(obj, name, g)
    for n in ("shard_knn2_dev", "shard_frame_dev", "finalize_dev", "slots_dev", "sig_remove"):
        wrap(sh.eng, n)
    for n in ("_all_gather", "_all_reduce_sum", "_complete_pending"):
        wrap(sh, n)
    frames = [torch.from_numpy(synth.frame_from_signature(vocab, words[(37 * t + 5) % n_sig], seed=t)).cuda() for t in range(64)]
    first_new = n_words + 1
    torch.cuda.synchronize()
    dist.barrier()
    t_all = time.perf_counter()
    for t in range(a.frames):
        sh.frame(frames[t % 64], n_sig + 1 + t, float(n_sig + 1 + t), defer=True, first_new_word_id=first_new)
        first_new += q
        if t == a.frames // 2:
            torch.cuda.synchronize(); acc.clear(); t_all = time.perf_counter(); n0 = t
    t_host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t_all
    n = a.frames - n0 - 1
    res = {"rank": rank, "world": world, "frames_timed": n, "host_ms_per_frame": t_host / n * 1e3, "wall_ms_per_frame": t_dev / n * 1e3,
           "host_ms_by_call": {k: round(v / n * 1e3, 4) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}}
    out.put(res)
    dist.barrier()
    sh.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--words", type=int, default=49000)
    ap.add_argument("--signatures", type=int, default=20000)
    ap.add_argument("--exchange", default="p2p")
    ap.add_argument("--append", type=int, default=1)
    ap.add_argument("--driver", default="python", choices=["python", "native"])
    a = ap.parse_args()
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, a.world, port, a, out)) for r in range(a.world)]
    for p in procs:
        p.start()
    for _ in range(a.world):
        print(json.dumps(out.get(timeout=600)))
    for p in procs:
        p.join(timeout=60)
