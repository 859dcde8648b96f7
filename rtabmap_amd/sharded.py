"""Word-ID-range sharding of the loop-closure engine over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank owns a consecutive range of
the vocabulary rows and the references of those words; per frame there are exactly two exchanges:

    all-gather   q x 2 candidate records of 16 B per rank   (local exact 2-NN -> global 2-NN)
    all-reduce   n_slots int64 partial likelihood sums      (integer sum: order-free, equals the 1-GPU result bit for bit)

Everything else (same-frame resolution, registration of the frame, scoring of the owned words) runs replicated / locally
in the engine (lcd_shard_knn2_dev / lcd_shard_frame_dev / lcd_finalize_dev).  This module is plumbing: torch tensors hold
the exchange buffers, torch.distributed moves them.
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
    def __init__(self, dtype, dim, rank=None, world=None, device=0, group=None, stream=None, vocab_capacity=0, sig_capacity=0):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.device = torch.device("cuda", device)
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.eng = Engine(dtype, dim, device=device, vocab_capacity=vocab_capacity, sig_capacity=sig_capacity,
                          stream=self.stream.cuda_stream)
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.total_rows = 0
        self.lo = self.hi = 0            # owned word-id range (ids lo+1 .. hi when ids are 1..n in row order)
        self._bufs = {}

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

    def retire(self, sig_id):
        self.eng.sig_remove(sig_id)

    # ---- collectives (RCCL directly on device tensors; other backends are staged through the host: tests only)
    def _all_gather(self, out, inp):
        if self.world == 1:
            out.copy_(inp.reshape(out.shape))
        elif self.backend == "nccl":
            dist.all_gather_into_tensor(out, inp, group=self.group)
        else:
            self.stream.synchronize()
            parts = [torch.empty_like(inp, device="cpu") for _ in range(self.world)]
            dist.all_gather(parts, inp.cpu(), group=self.group)
            out.copy_(torch.cat(parts).reshape(out.shape).to(out.device))

    def _all_reduce_sum(self, t):
        if self.world == 1:
            return
        if self.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self.stream.synchronize()
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
    def frame(self, d_desc, sig_id, N, incremental=True, new_words_compared=True, nndr=0.8, want_likelihood=True, first_new_word_id=0):
        """d_desc: [q, dim] device tensor.  Returns (word ids int32 [q], likelihood float32 [n_slots]) device tensors."""
        q = d_desc.shape[0]
        if self.world == 1:
            # one rank owns everything: the sharded frame IS the single-GPU frame (fused launches, no exchange)
            with torch.cuda.stream(self.stream):
                words = self._buf("words", (q,), torch.int32)
                _, n_slots = self.eng.slots_dev()
                like = self._buf("like", (n_slots + 2,), torch.float32)
                self.eng.frame_dev(d_desc.data_ptr(), q, sig_id, N, words.data_ptr(), like.data_ptr() if want_likelihood else None,
                                   like.shape[0], incremental=incremental, new_words_compared=new_words_compared, nndr=nndr,
                                   first_new_word_id=first_new_word_id)
                _, n_slots = self.eng.slots_dev()
            return words[:q], like[:n_slots]
        with torch.cuda.stream(self.stream):
            cand = self._buf("cand", (q * 2 * 2,), torch.int64)                  # 16-byte records as 2 x int64
            allc = self._buf("allc", (self.world * q * 2 * 2,), torch.int64)
            words = self._buf("words", (q,), torch.int32)
            cand, allc = cand[: q * 4], allc[: self.world * q * 4]
            self.eng.shard_knn2_dev(d_desc.data_ptr(), q, cand.data_ptr())
            self._all_gather(allc, cand)
            _, n_slots = self.eng.slots_dev()
            cap = n_slots + 1
            lfix = self._buf("lfix", (max(cap, 1),), torch.int64)
            like = self._buf("like", (max(cap, 1),), torch.float32)
            self.eng.shard_frame_dev(d_desc.data_ptr(), q, sig_id, N, self.rank, self.world, allc.data_ptr(), self.total_rows,
                                     words.data_ptr(), lfix.data_ptr() if want_likelihood else None, lfix.shape[0],
                                     incremental=incremental, new_words_compared=new_words_compared, nndr=nndr,
                                     first_new_word_id=first_new_word_id)
            _, n_slots = self.eng.slots_dev()
            if want_likelihood:
                self._all_reduce_sum(lfix[:n_slots])
                self.eng.finalize_dev(lfix.data_ptr(), n_slots, like.data_ptr())
        return words[:q], like[:n_slots]

    def close(self):
        self.eng.close()
