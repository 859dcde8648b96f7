"""Word-ID-range sharding of the loop-closure engine over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank owns a consecutive range of
the vocabulary rows and the references of those words; per frame there are exactly two exchanges:

    all-gather   q x 2 candidate records of 16 B per rank   (local exact 2-NN -> global 2-NN)
    all-reduce   n_slots int64 partial likelihood sums      (integer sum: order-free, equals the 1-GPU result bit for bit)

Everything else (same-frame resolution, registration of the frame, scoring of the owned words) runs replicated / locally
in the engine (lcd_shard_knn2_dev / lcd_shard_frame_dev / lcd_finalize_dev).  This module is plumbing: torch tensors hold
the exchange buffers, torch.distributed moves them.

frame(..., defer=True) overlaps the all-reduce of frame t with the nearest-neighbour search of frame t + 1 (SURVEY.md hard
part 9): the reduction runs on a second stream, the likelihood of frame t is finalised -- and handed back -- inside the call
for frame t + 1, before that frame touches the index (the same one-frame deferral as the single-GPU pipelined handle);
retirements asked for in between wait in a queue until then, so every frame sees exactly the state the undeferred order gives.
"""
import numpy as np
import torch
import torch.distributed as dist

from .capi import Engine


def shard_bounds(n_rows, world):
    """Consecutive, near-equal row ranges: rank r owns rows [b[r], b[r+1])."""
    base, rem = divmod(n_rows, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + base + (1 if r < rem else 0))
    return b


class ShardedLoopClosure:
    def __init__(self, dtype, dim, rank=None, world=None, device=0, group=None, stream=None, vocab_capacity=0, sig_capacity=0, knn_mode=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.device = torch.device("cuda", device)
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        # one rank owns everything: the sharded frame IS the single-GPU frame, software-pipelined like it (frame(defer=True) hands the
        # likelihood back one call late either way); several ranks: plain handles, the exchanges sit between the stages
        self.eng = Engine(dtype, dim, device=device, vocab_capacity=vocab_capacity, sig_capacity=sig_capacity,
                          stream=self.stream.cuda_stream, knn_mode=knn_mode)
        self._append = False
        self.force_sharded_path = False  # world 1 only: run the sharded stages (local search -> records -> merge -> registration -> integer
                                         # scoring -> conversion) instead of the fused single-GPU frame: what one rank of N pays without the wire
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.total_rows = 0
        self.lo = self.hi = 0            # owned word-id range (ids lo+1 .. hi when ids are 1..n in row order)
        self._bufs = {}
        self.comm = None                 # second stream: the deferred all-reduce (created on first use)
        self._pending = None             # (lfix, like, n_slots, event: all-reduce done) of the frame whose likelihood is owed
        self._retire_q = []              # retirements asked for while a likelihood is owed
        self._n_frames = 0
        self._owed_single = None
        self.p2p = None                  # P2PTransport: the exchanges as liblcd_p2p.so's kernels instead of the process group's calls

    def enable_p2p(self, q_max, slots_max, wire="i64", timeout_ms=10000):
        """The two per-frame exchanges through liblcd_p2p.so (include/lcd_p2p.h): kernels that write the peers' hipIpc-mapped arenas, enqueued
        on the engine's stream (the deferred all-reduce on the second one) -- no RCCL call, no host staging.  Every rank calls it, with the
        same capacities (frames of at most q_max descriptors, at most slots_max signature slots), before the first frame; the process
        group carries the 128-byte exports once.  wire "f32": the all-reduce moves 32-bit floats (half the bytes, rank-ordered sums)."""
        if self.world > 1:
            self.p2p = P2PTransport(self.rank, self.world, int(q_max) * 2 * 16, int(slots_max) + 2, group=self.group, wire=wire, timeout_ms=timeout_ms)

    # ---- state
    def load_vocabulary(self, rows, word_ids):
        """rows/word_ids: the FULL vocabulary in its (tie-break) row order; this rank keeps its consecutive slice."""
        b = shard_bounds(rows.shape[0], self.world)
        self.lo, self.hi = b[self.rank], b[self.rank + 1]
        self.owned_ids = set(np.asarray(word_ids[self.lo:self.hi]).tolist())
        self.total_rows = rows.shape[0]
        if self.hi > self.lo:
            self.eng.vocab_append(rows[self.lo:self.hi], word_ids[self.lo:self.hi])

    def add_signatures_bulk(self, sig_ids, offsets, word_ids, owned_mask=None):
        """Every rank registers every signature (identical slot numbering) with the words it owns; ni = all features."""
        w = np.asarray(word_ids, dtype=np.int32)
        if owned_mask is None:
            owned_mask = np.isin(w, np.fromiter(self.owned_ids, dtype=np.int32, count=len(self.owned_ids)))
        mine = np.where(owned_mask, w, -1).astype(np.int32)
        ni = np.diff(np.asarray(offsets, dtype=np.int64)).astype(np.int32)
        self.eng.sig_add_bulk(sig_ids, offsets, mine, ni)

    def enable_device_append(self, first_incremental_id, block=16):
        """VWDictionary::update()'s append on the device for every frame from here on: the words frames create (ids from
        first_incremental_id on) become rows of the rank that owns them -- block-cyclically over the ranks -- inside the frame call
        (lcd_set_option "shard_growth_first" / "shard_growth_block" / "shard_append"); world 1: lcd_frame_args.append_new_words."""
        self._append = True
        if self.world > 1 or self.force_sharded_path:
            self.eng.set_option("shard_growth_first", int(first_incremental_id))
            self.eng.set_option("shard_growth_block", int(block))
            self.eng.set_option("shard_append", 1)

    def retire(self, sig_id):
        if self._pending is not None:    # the owed likelihood is finalised against the signature table as its frame left it
            self._retire_q.append(int(sig_id))
        else:
            self.eng.sig_remove(sig_id)

    def _complete_pending(self):
        """Finalise the owed likelihood (engine stream waits for its all-reduce), then apply the queued retirements."""
        like, self._owed_single = self._owed_single, None       # (world 1: the fused frame has finalised it already)
        if self._pending is not None:
            lfix, like_buf, n_slots, done = self._pending
            self._pending = None
            self.stream.wait_event(done)
            with torch.cuda.stream(self.stream):
                self.eng.finalize_dev(lfix.data_ptr(), n_slots, like_buf.data_ptr())
            like = like_buf[:n_slots]
        for s in self._retire_q:
            self.eng.sig_remove(s)
        self._retire_q = []
        return like

    def flush(self, synchronize=True):
        """The likelihood still owed after the last frame(defer=True) (None if there is none); queued retirements are applied.
        Like everything frame() returns, the tensor is written on the engine stream: synchronize=True waits for it."""
        like = self._complete_pending()
        if synchronize:
            self.stream.synchronize()
        return like

    # ---- collectives (RCCL directly on device tensors; other backends are staged through the host: tests only)
    def _all_gather(self, out, inp):
        if self.world == 1:
            out.copy_(inp.reshape(out.shape))
        elif getattr(self, "p2p", None) is not None:
            self.p2p.all_gather(inp.data_ptr(), out.data_ptr(), inp.numel() * inp.element_size(), torch.cuda.current_stream(self.device).cuda_stream)
        elif self.backend == "nccl":
            dist.all_gather_into_tensor(out, inp, group=self.group)
        else:
            self.stream.synchronize()
            parts = [torch.empty_like(inp, device="cpu") for _ in range(self.world)]
            dist.all_gather(parts, inp.cpu(), group=self.group)
            out.copy_(torch.cat(parts).reshape(out.shape).to(out.device))

    def _all_reduce_sum(self, t, stream=None):
        if self.world == 1:
            return
        if getattr(self, "p2p", None) is not None:
            self.p2p.all_reduce_sum_i64(t.data_ptr(), t.numel(), torch.cuda.current_stream(self.device).cuda_stream)
        elif self.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            (stream or self.stream).synchronize()
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(c.to(t.device))

    def _buf(self, name, shape, dtype):
        """Exchange buffer of at least `shape[0]` elements (grows geometrically, so per-frame growth does not reallocate)."""
        n = int(shape[0])
        t = self._bufs.get(name)
        if t is None or t.shape[0] < n or t.dtype != dtype:
            cap = 1024
            while cap < n:
                cap *= 2
            t = torch.zeros((cap,), dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t

    # ---- one frame
    def frame(self, d_desc, sig_id, N, incremental=True, new_words_compared=True, nndr=0.8, want_likelihood=True, first_new_word_id=0,
              defer=False):
        """d_desc: [q, dim] device tensor.  Returns (word ids int32 [q], likelihood float32 [n_slots]) device tensors.
        defer=True: the likelihood returned is the PREVIOUS deferred frame's (None if there is none; flush() hands out the last) --
        this frame's all-reduce is left running on the second stream, under the next frame's nearest-neighbour search.  The
        buffers alternate between two sets: a returned tensor stays valid until the second frame after it.  Returned tensors are
        written on the engine stream (self.stream): read them there, or synchronise it first."""
        q = d_desc.shape[0]
        par = self._n_frames & 1
        self._n_frames += 1
        if self.world == 1 and not self.force_sharded_path:
            # one rank owns everything: the sharded frame IS the single-GPU frame (fused launches, no exchange)
            prev = self._complete_pending()
            with torch.cuda.stream(self.stream):
                words = self._buf("words%d" % par, (q,), torch.int32)
                _, n_slots = self.eng.slots_dev()
                like = self._buf("like%d" % par, (n_slots + 2,), torch.float32)
                self.eng.frame_dev(d_desc.data_ptr(), q, sig_id, N, words.data_ptr(), like.data_ptr() if want_likelihood else None,
                                   like.shape[0], incremental=incremental, new_words_compared=new_words_compared, nndr=nndr,
                                   first_new_word_id=first_new_word_id, append_new_words=self._append)
                _, n_slots = self.eng.slots_dev()
            if defer and want_likelihood:                 # nothing to overlap; the same contract: one frame late
                self._owed_single = like[:n_slots]
                return words[:q], prev
            return words[:q], like[:n_slots]
        with torch.cuda.stream(self.stream):
            cand = self._buf("cand", (q * 2 * 2,), torch.int64)                  # 16-byte records as 2 x int64
            allc = self._buf("allc", (self.world * q * 2 * 2,), torch.int64)
            words = self._buf("words%d" % par, (q,), torch.int32)
            cand, allc = cand[: q * 4], allc[: self.world * q * 4]
            self.eng.shard_knn2_dev(d_desc.data_ptr(), q, cand.data_ptr())      # (the owed all-reduce runs under this)
            self._all_gather(allc, cand)
        prev = self._complete_pending()                   # before this frame's registration and scoring touch the index
        with torch.cuda.stream(self.stream):
            _, n_slots = self.eng.slots_dev()
            cap = n_slots + 1
            lfix = self._buf("lfix%d" % par, (max(cap, 1),), torch.int64)
            like = self._buf("like%d" % par, (max(cap, 1),), torch.float32)
            self.eng.shard_frame_dev(d_desc.data_ptr(), q, sig_id, N, self.rank, self.world, allc.data_ptr(), self.total_rows,
                                     words.data_ptr(), lfix.data_ptr() if want_likelihood else None, lfix.shape[0],
                                     incremental=incremental, new_words_compared=new_words_compared, nndr=nndr,
                                     first_new_word_id=first_new_word_id)
            _, n_slots = self.eng.slots_dev()
            if want_likelihood and not defer:
                self._all_reduce_sum(lfix[:n_slots])
                self.eng.finalize_dev(lfix.data_ptr(), n_slots, like.data_ptr())
        if not want_likelihood:
            return words[:q], (prev if defer else None)
        if not defer:
            return words[:q], like[:n_slots]
        if self.comm is None:
            self.comm = torch.cuda.Stream(self.device)
        ready = torch.cuda.Event()
        ready.record(self.stream)
        self.comm.wait_event(ready)
        with torch.cuda.stream(self.comm):                # RCCL orders its own stream behind the current one: behind `ready`
            self._all_reduce_sum(lfix[:n_slots], stream=self.comm)
        done = torch.cuda.Event()
        done.record(self.comm)
        self._pending = (lfix, like, n_slots, done)
        return words[:q], prev

    def close(self):
        if getattr(self, "p2p", None) is not None:
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)                    # no rank unmaps an arena a peer's kernel may still write
            self.p2p.close()
            self.p2p = None
        self.eng.close()


# ---------------------------------------------------------------------------------------------------------------- native driver
_shard_lib = None


def load_shard():
    """liblcd_shard.so: the sharded frame as C++ host code over the C-ABI + RCCL (include/lcd_shard.h, rtabmap_amd/host/ShardedLcd.cpp)."""
    global _shard_lib
    if _shard_lib is None:
        import ctypes as C
        from . import build as _b
        from .capi import load
        load()                                                           # liblcd_hip.so first (the driver links against it)
        L = C.CDLL(_b.build_shard())
        vp = C.c_void_p
        frame_args = [vp, vp, C.c_int, C.c_int, C.c_float, C.c_int32, C.c_int32, C.c_float, C.c_int64, vp, vp, C.c_int64]
        L.lcd_shard_unique_id.argtypes = [vp]
        L.lcd_shard_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
        L.lcd_shard_comm_create_transport.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
        L.lcd_shard_comm_destroy.argtypes = [vp]
        L.lcd_shard_comm_destroy.restype = None
        L.lcd_shard_last_error.argtypes = [vp]
        L.lcd_shard_last_error.restype = C.c_char_p
        L.lcd_shard_frame.argtypes = frame_args
        L.lcd_shard_frame_deferred.argtypes = frame_args
        L.lcd_shard_flush.argtypes = [vp]
        L.lcd_shard_sig_remove.argtypes = [vp, C.c_int32]
        L.lcd_shard_set_growth.argtypes = [vp, C.c_int32, C.c_int32]
        L.lcd_shard_owner_of.argtypes = [vp, C.c_int32]
        L.lcd_shard_set_append.argtypes = [vp, C.c_int]
        _shard_lib = L
    return _shard_lib


class HostStagedTransport:
    """lcd_shard_transport over a torch.distributed process group of ANY backend, staged through the host (tests: two ranks sharing one
    GPU cannot talk RCCL to each other).  The callbacks complete before they return: they wait for `stream`, copy device -> host with the
    HIP runtime the process already has, exchange over the group, copy back."""

    def __init__(self, group=None):
        import ctypes as C
        self.C, self.group = C, group
        self.world = dist.get_world_size(group)
        self.hip = C.CDLL("libamdhip64.so")                            # the runtime torch loaded (same soname: the same library)
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
        REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

        class Transport(C.Structure):
            _fields_ = [("struct_size", C.c_int32), ("reserved", C.c_int32), ("user", C.c_void_p), ("all_gather", GATHER), ("all_reduce_sum_i64", REDUCE)]
        self._gather, self._reduce = GATHER(self._all_gather), REDUCE(self._all_reduce)      # (kept alive with the object)
        self.struct = Transport(C.sizeof(Transport), 0, None, self._gather, self._reduce)
        self.calls = {"all_gather": 0, "all_reduce": 0}

    def _all_gather(self, user, d_send, d_recv, nbytes, stream):
        try:
            self.calls["all_gather"] += 1
            if self.hip.hipStreamSynchronize(stream) != 0:
                return 1
            h = np.empty(nbytes, np.uint8)
            if self.hip.hipMemcpy(h.ctypes.data, d_send, nbytes, 2) != 0:
                return 2
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, torch.from_numpy(h), group=self.group)
            allb = torch.cat(parts).numpy()
            return 0 if self.hip.hipMemcpy(d_recv, allb.ctypes.data, nbytes * self.world, 1) == 0 else 3
        except Exception:                                                 # noqa: BLE001 -- no exception crosses the C boundary
            return 9

    def _all_reduce(self, user, d_buf, count, stream):
        try:
            self.calls["all_reduce"] += 1
            if self.hip.hipStreamSynchronize(stream) != 0:
                return 1
            h = np.empty(count, np.int64)
            if self.hip.hipMemcpy(h.ctypes.data, d_buf, count * 8, 2) != 0:
                return 2
            t = torch.from_numpy(h)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return 0 if self.hip.hipMemcpy(d_buf, h.ctypes.data, count * 8, 1) == 0 else 3
        except Exception:                                                 # noqa: BLE001
            return 9


_p2p_lib = None


def load_p2p():
    """liblcd_p2p.so: the one-shot peer-to-peer exchanges of include/lcd_p2p.h (rtabmap_amd/csrc/p2p_exchange.hip)."""
    global _p2p_lib
    if _p2p_lib is None:
        import ctypes as C
        from . import build as _b
        L = C.CDLL(_b.build_p2p())
        vp = C.c_void_p
        L.lcd_p2p_create.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(vp)]
        L.lcd_p2p_export.argtypes = [vp, vp]
        L.lcd_p2p_connect.argtypes = [vp, vp]
        L.lcd_p2p_destroy.argtypes = [vp]
        L.lcd_p2p_destroy.restype = None
        L.lcd_p2p_last_error.argtypes = [vp]
        L.lcd_p2p_last_error.restype = C.c_char_p
        L.lcd_p2p_set_wire.argtypes = [vp, C.c_int]
        L.lcd_p2p_set_timeout_ms.argtypes = [vp, C.c_int64]
        L.lcd_p2p_set_conservative_fences.argtypes = [vp, C.c_int]
        L.lcd_p2p_status.argtypes = [vp]
        L.lcd_p2p_status.restype = C.c_uint32
        L.lcd_p2p_clear_status.argtypes = [vp]
        L.lcd_p2p_clear_status.restype = None
        L.lcd_p2p_all_gather.argtypes = [vp, vp, vp, C.c_size_t, vp]
        L.lcd_p2p_all_reduce_sum_i64.argtypes = [vp, vp, C.c_size_t, vp]
        L.lcd_p2p_transport.argtypes = [vp, vp]
        _p2p_lib = L
    return _p2p_lib


class P2PTransport:
    """lcd_shard_transport over liblcd_p2p.so: kernels that write the peers' hipIpc-mapped arenas directly (the one-shot
    reduce-scatter + all-gather next to RCCL).  The process group (any backend) carries the 128-byte exports once, at construction;
    the per-frame exchanges never touch it.  Works between processes that share ONE GPU too (the test box)."""
    HANDLE_BYTES = 128

    def __init__(self, rank, world, gather_bytes_per_rank_max, reduce_count_max, group=None, wire="i64", timeout_ms=10000):
        import ctypes as C
        self.C, self.L, self.rank, self.world = C, load_p2p(), rank, world
        h = C.c_void_p()
        rc = self.L.lcd_p2p_create(rank, world, gather_bytes_per_rank_max, reduce_count_max, C.byref(h))
        if rc != 0:
            raise RuntimeError("lcd_p2p_create failed (%d)" % rc)
        self.h = h
        mine = (C.c_ubyte * self.HANDLE_BYTES)()
        self._ck(self.L.lcd_p2p_export(h, mine), "lcd_p2p_export")
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, bytes(mine), group=group)
            blob = b"".join(parts)
            self._ck(self.L.lcd_p2p_connect(h, (C.c_ubyte * len(blob)).from_buffer_copy(blob)), "lcd_p2p_connect")
        self.set_wire(wire)
        self._ck(self.L.lcd_p2p_set_timeout_ms(h, timeout_ms), "lcd_p2p_set_timeout_ms")

        class Transport(C.Structure):
            _fields_ = [("struct_size", C.c_int32), ("reserved", C.c_int32), ("user", C.c_void_p), ("all_gather", C.c_void_p), ("all_reduce_sum_i64", C.c_void_p)]
        self.struct = Transport()
        self._ck(self.L.lcd_p2p_transport(h, C.byref(self.struct)), "lcd_p2p_transport")

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s" % (what, self.L.lcd_p2p_last_error(self.h).decode()))

    def set_wire(self, wire):
        self._ck(self.L.lcd_p2p_set_wire(self.h, {"i64": 0, "f32": 1}[wire]), "lcd_p2p_set_wire")

    def set_conservative_fences(self, on=True):
        """lcd_p2p_set_conservative_fences: the compiler's release fences instead of acknowledged stores + relaxed flags (bring-up switch)"""
        self._ck(self.L.lcd_p2p_set_conservative_fences(self.h, 1 if on else 0), "lcd_p2p_set_conservative_fences")

    def all_gather(self, d_send_ptr, d_recv_ptr, bytes_per_rank, stream=None):
        self._ck(self.L.lcd_p2p_all_gather(self.h, d_send_ptr, d_recv_ptr, bytes_per_rank, stream), "lcd_p2p_all_gather")

    def all_reduce_sum_i64(self, d_buf_ptr, count, stream=None):
        self._ck(self.L.lcd_p2p_all_reduce_sum_i64(self.h, d_buf_ptr, count, stream), "lcd_p2p_all_reduce_sum_i64")

    def status(self):
        """LCD_P2P_TIMEOUT_* bits raised by completed kernels (0: every peer arrived in time)"""
        return int(self.L.lcd_p2p_status(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.lcd_p2p_destroy(self.h)
            self.h = None


class NativeShardComm:
    """One rank of the C++ driver around an Engine.  transport=None: RCCL -- the 128-byte RCCL id travels through torch.distributed (any
    backend) when world > 1, the only thing the process group is used for; the two per-frame exchanges are RCCL calls made by the C++ code.
    transport=HostStagedTransport(...) / P2PTransport(...): the caller's exchanges (lcd_shard_comm_create_transport)."""

    def __init__(self, eng, rank=0, world=1, group=None, transport=None):
        import ctypes as C
        self.L, self.eng, self.rank, self.world, self.transport = load_shard(), eng, rank, world, transport
        h = C.c_void_p()
        if transport is not None:
            rc = self.L.lcd_shard_comm_create_transport(eng.h, rank, world, C.byref(transport.struct), C.byref(h))
        else:
            idb = (C.c_ubyte * 128)()
            if world > 1:
                box = [None]
                if rank == 0:
                    assert self.L.lcd_shard_unique_id(idb) == 0, "ncclGetUniqueId failed"
                    box[0] = bytes(idb)
                dist.broadcast_object_list(box, src=0, group=group)
                idb = (C.c_ubyte * 128).from_buffer_copy(box[0])
            rc = self.L.lcd_shard_comm_create(eng.h, rank, world, idb, C.byref(h))
        if rc != 0:
            raise RuntimeError("lcd_shard_comm_create failed (%d)" % rc)
        self.h = h

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s" % (what, self.L.lcd_shard_last_error(self.h).decode()))

    def frame(self, d_desc_ptr, q, sig_id, N, total_live_rows, d_word_ids_ptr, d_like_ptr, like_capacity, incremental=True,
              new_words_compared=True, nndr=0.8, first_new_word_id=0, defer=False):
        """defer=True: d_like_ptr is written by the NEXT frame call (or flush()): the all-reduce runs under that frame's search"""
        flags = (1 if incremental else 0) | (2 if new_words_compared else 0)
        fn = self.L.lcd_shard_frame_deferred if defer else self.L.lcd_shard_frame
        self._ck(fn(self.h, d_desc_ptr, q, flags, nndr, sig_id, first_new_word_id, float(N), int(total_live_rows), d_word_ids_ptr, d_like_ptr,
                    like_capacity), "lcd_shard_frame")

    def flush(self):
        self._ck(self.L.lcd_shard_flush(self.h), "lcd_shard_flush")

    def sig_remove(self, sig_id):
        self._ck(self.L.lcd_shard_sig_remove(self.h, sig_id), "lcd_shard_sig_remove")

    def set_growth(self, first_incremental_id, block):
        self._ck(self.L.lcd_shard_set_growth(self.h, first_incremental_id, block), "lcd_shard_set_growth")

    def owner_of(self, word_id):
        return int(self.L.lcd_shard_owner_of(self.h, word_id))

    def set_append(self, on=True):
        """lcd_shard_set_append: the words a frame creates become rows of the owning rank's shard on the device (update()'s append)"""
        self._ck(self.L.lcd_shard_set_append(self.h, 1 if on else 0), "lcd_shard_set_append")

    def close(self):
        if getattr(self, "h", None):
            self.L.lcd_shard_comm_destroy(self.h)
            self.h = None
