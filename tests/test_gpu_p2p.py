"""GPU (-m gpu): liblcd_p2p.so's exchanges between TWO processes sharing the test box's one GPU -- every byte a rank receives was written
into its hipIpc-mapped arena by a kernel of the other process.  The expected values are recomputed from the seeds on each rank (integer
sums exactly; the 32-bit float wire within its stated bound), over sizes from one vector to the 10^6-signature likelihood, repeated so the
epochs and the double-buffered mailbox wrap, with an all-gather and an all-reduce in flight on two streams at once (what the deferred
driver of lcd_shard.h does), and with one peer arriving late on purpose: the waiting kernel must give up with a status bit, not hang."""
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_sharded import _free_port, _init


def _block(rank, it, nbytes):
    return np.random.default_rng(1000 * it + rank).integers(0, 256, nbytes, dtype=np.uint8)


def _operand(rank, it, count):
    # partial likelihoods are non-negative fixed-point sums below 2^40; the integer wire must also carry negatives and wrap like RCCL's sum
    return np.random.default_rng(77 * it + rank).integers(0, 1 << 40, count, dtype=np.int64)


def _p2p_worker(rank, world, port, out):
    _init(rank, world, port)
    torch.cuda.set_device(0)
    from rtabmap_amd.sharded import P2PTransport
    msgs = []
    max_count = (1 << 20) + 5
    tr = P2PTransport(rank, world, 65536, max_count, timeout_ms=20000)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    it = 0
    # (1) all-gather: one vector, the headline's 16 000 bytes (500 x 2 records of 16), the capacity; thrice each (mailbox halves alternate)
    for nbytes in (16, 16000, 65536):
        for _ in range(3):
            it += 1
            send = torch.from_numpy(_block(rank, it, nbytes)).cuda()
            recv = torch.zeros(world * nbytes, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            tr.all_gather(send.data_ptr(), recv.data_ptr(), nbytes, s1.cuda_stream)
            s1.synchronize()
            want = np.concatenate([_block(r, it, nbytes) for r in range(world)])
            if not np.array_equal(recv.cpu().numpy(), want):
                msgs.append("all-gather of %d bytes, exchange %d" % (nbytes, it))
    # (2) all-reduce, integer wire: exact, including negative operands; float wire: within world * 2^-24 relative of the exact sum;
    #     the integer wire once more with the conservative fences (lcd_p2p_set_conservative_fences: same results, slower)
    for wire in ("i64", "f32", "i64 conservative"):
        tr.set_conservative_fences(wire.endswith("conservative"))
        wire = wire.split()[0]
        tr.set_wire(wire)
        for count in (1, 5, 1000, 100001, max_count):
            for rep in range(2):
                it += 1
                ops = [_operand(r, it, count) for r in range(world)]
                if wire == "i64" and rep == 1:
                    ops = [o - (1 << 39) for o in ops]
                buf = torch.from_numpy(ops[rank].copy()).cuda()
                torch.cuda.synchronize()
                tr.all_reduce_sum_i64(buf.data_ptr(), count, s1.cuda_stream)
                s1.synchronize()
                got, want = buf.cpu().numpy(), np.sum(ops, axis=0)
                if wire == "i64":
                    if not np.array_equal(got, want):
                        msgs.append("integer all-reduce of %d, exchange %d" % (count, it))
                else:
                    # what the kernels compute, restated: float(x_r) summed in rank order in float, rounded to the nearest integer
                    acc = ops[0].astype(np.float32)
                    for o in ops[1:]:
                        acc = acc + o.astype(np.float32)
                    if not np.array_equal(got, np.rint(acc.astype(np.float64)).astype(np.int64)):
                        msgs.append("float-wire all-reduce of %d is not the rank-ordered float sum, exchange %d" % (count, it))
                    if np.abs(got - want).max() > want.max() * world * 2.0 ** -24:
                        msgs.append("float-wire all-reduce of %d beyond its bound, exchange %d" % (count, it))
    tr.set_wire("i64")
    tr.set_conservative_fences(False)
    # (3) an all-gather (stream 1) beside an all-reduce (stream 2), 40 pairs enqueued back to back without a host synchronisation in between
    n_pairs, nbytes, count = 40, 16000, 100001
    sends = [torch.from_numpy(_block(rank, 5000 + k, nbytes)).cuda() for k in range(n_pairs)]
    recvs = [torch.zeros(world * nbytes, dtype=torch.uint8, device="cuda") for _ in range(n_pairs)]
    bufs = [torch.from_numpy(_operand(rank, 5000 + k, count)).cuda() for k in range(n_pairs)]
    torch.cuda.synchronize()
    dist.barrier()
    for k in range(n_pairs):
        tr.all_gather(sends[k].data_ptr(), recvs[k].data_ptr(), nbytes, s1.cuda_stream)
        tr.all_reduce_sum_i64(bufs[k].data_ptr(), count, s2.cuda_stream)
    torch.cuda.synchronize()
    for k in range(n_pairs):
        if not np.array_equal(recvs[k].cpu().numpy(), np.concatenate([_block(r, 5000 + k, nbytes) for r in range(world)])):
            msgs.append("concurrent all-gather %d" % k)
        if not np.array_equal(bufs[k].cpu().numpy(), np.sum([_operand(r, 5000 + k, count) for r in range(world)], axis=0)):
            msgs.append("concurrent all-reduce %d" % k)
    if tr.status() != 0:
        msgs.append("status %d after the exchanges that must all arrive" % tr.status())
    # (4) the last rank arrives 1.5 s late to an all-gather the others wait 300 ms for: their kernels end with LCD_P2P_TIMEOUT_GATHER (and
    #     whatever lay in the mailbox), the late rank finds every block already there; the next exchange is in step again
    dist.barrier()
    tr._ck(tr.L.lcd_p2p_set_timeout_ms(tr.h, 300), "lcd_p2p_set_timeout_ms")
    send = torch.from_numpy(_block(rank, 9001, 16000)).cuda()
    recv = torch.zeros(world * 16000, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    if rank == world - 1:
        time.sleep(1.5)
    t0 = time.time()
    tr.all_gather(send.data_ptr(), recv.data_ptr(), 16000, s1.cuda_stream)
    s1.synchronize()
    waited = time.time() - t0
    if rank != world - 1:
        if tr.status() != 1:
            msgs.append("rank %d waited for a late peer and reports status %d, not LCD_P2P_TIMEOUT_GATHER" % (rank, tr.status()))
        if not 0.25 < waited < 1.4:
            msgs.append("rank %d's bounded wait took %.2f s" % (rank, waited))
        tr.L.lcd_p2p_clear_status(tr.h)
    else:
        if tr.status() != 0 or not np.array_equal(recv.cpu().numpy(), np.concatenate([_block(r, 9001, 16000) for r in range(world)])):
            msgs.append("the late rank did not find its peers' blocks")
    dist.barrier()
    tr._ck(tr.L.lcd_p2p_set_timeout_ms(tr.h, 20000), "lcd_p2p_set_timeout_ms")
    send = torch.from_numpy(_block(rank, 9002, 16000)).cuda()
    torch.cuda.synchronize()
    tr.all_gather(send.data_ptr(), recv.data_ptr(), 16000, s1.cuda_stream)
    s1.synchronize()
    if tr.status() != 0 or not np.array_equal(recv.cpu().numpy(), np.concatenate([_block(r, 9002, 16000) for r in range(world)])):
        msgs.append("the exchange after the timed-out one")
    # (5) argument checks: nothing larger than the capacities both sides mapped, nothing misaligned
    for bad in (lambda: tr.all_gather(send.data_ptr(), recv.data_ptr(), 65536 + 16, s1.cuda_stream),
                lambda: tr.all_gather(send.data_ptr(), recv.data_ptr(), 24, s1.cuda_stream),
                lambda: tr.all_reduce_sum_i64(bufs[0].data_ptr(), max_count + 1, s1.cuda_stream)):
        try:
            bad()
            msgs.append("an out-of-contract call was accepted")
        except RuntimeError:
            pass
    torch.cuda.synchronize()
    dist.barrier()
    tr.close()
    out.put((rank, msgs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_p2p_exchanges_between_processes(world):
    """world 2 and 4 (four processes on the one GPU: three peers per rank, slices of a quarter, counts that do not divide by the world)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = [out.get(timeout=600) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    for rank, msgs in got:
        assert not msgs, (rank, msgs)


@pytest.mark.gpu
def test_p2p_world_of_one_is_a_copy():
    """world = 1 (what bench.py's shard_stages_world1 leg uses): the all-gather is a device copy, the all-reduce nothing; no peer, no flag"""
    from rtabmap_amd.sharded import P2PTransport
    tr = P2PTransport(0, 1, 4096, 1024)
    send = torch.arange(4096, dtype=torch.uint8, device="cuda")
    recv = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    buf = torch.arange(1000, dtype=torch.int64, device="cuda")
    tr.all_gather(send.data_ptr(), recv.data_ptr(), 4096, torch.cuda.current_stream().cuda_stream)
    tr.all_reduce_sum_i64(buf.data_ptr(), 1000, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(send, recv) and torch.equal(buf, torch.arange(1000, dtype=torch.int64, device="cuda")) and tr.status() == 0
    tr.close()
