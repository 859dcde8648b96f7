"""Latency of liblcd_p2p.so's exchanges between two processes on ONE GPU (the only multi-rank configuration a 1-GPU box offers: the
arenas are mapped through hipIpc exactly as between two GPUs, the 'wire' is the local HBM instead of xGMI).  Prints one JSON line:
microseconds per exchange, enqueued back to back on one stream, max over the two ranks.  python tools/p2p_bench.py [--iters 300]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, iters, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from rtabmap_amd.sharded import P2PTransport
    max_count = 1000001
    tr = P2PTransport(rank, world, 65536, max_count, timeout_ms=20000)
    s = torch.cuda.Stream()
    send = torch.zeros(16000, dtype=torch.uint8, device="cuda")
    recv = torch.zeros(16000 * world, dtype=torch.uint8, device="cuda")
    buf = torch.ones(max_count, dtype=torch.int64, device="cuda")
    res = {}

    def timed(name, fn):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = torch.tensor([(time.perf_counter() - t0) / iters * 1e6], dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        res[name] = round(float(dt.item()), 2)

    timed("all_gather_16000B_us", lambda: tr.all_gather(send.data_ptr(), recv.data_ptr(), 16000, s.cuda_stream))
    for wire in ("i64", "f32"):
        tr.set_wire(wire)
        for count in (100001, 1000001):
            buf.fill_(1)
            timed("all_reduce_%s_%d_us" % (wire, count), lambda: tr.all_reduce_sum_i64(buf.data_ptr(), count, s.cuda_stream))
    res["status"] = tr.status()
    # what the test transport of earlier rounds costs for the same two exchanges (stream synchronisation + host staging + gloo)
    h = torch.zeros(16000, dtype=torch.uint8)
    parts = [torch.empty_like(h) for _ in range(world)]
    hb = torch.zeros(100001, dtype=torch.int64)

    def staged():
        s.synchronize()
        h.copy_(send)
        dist.all_gather(parts, h)
        recv.copy_(torch.cat(parts))
        hb.copy_(buf[:100001])
        dist.all_reduce(hb)
        buf[:100001].copy_(hb)
    timed("host_staged_gloo_pair_16000B_100001_us", staged)
    torch.cuda.synchronize()
    dist.barrier()
    tr.close()
    if rank == 0:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--world", type=int, default=2, help="ranks (processes) sharing the GPU")
    a = ap.parse_args()
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, a.world, port, a.iters, out)) for r in range(a.world)]
    for p in procs:
        p.start()
    res = out.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
    res.update({"world": a.world, "gpus": 1, "iters": a.iters, "note": "%d processes sharing one MI355X;" % a.world + " arenas mapped with hipIpc; no xGMI link involved"})
    print(json.dumps(res))
