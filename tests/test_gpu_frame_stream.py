"""GPU parity of the TIMED path: lcd_frame_dev(sig_id != 0, d_likelihood) frame after frame against the restated
Memory::update -> VWDictionary::addNewWords -> Memory::computeLikelihood (oracle) on the same descriptor stream.

What the stream exercises: device-side registration of the frame inside the frame-tail kernel (postings of existing AND of
new words, VWDictionary.cpp:880,1185), VWDictionary::update() between frames (lcd_vocab_append of the words the previous frame
created, lcd_vocab_remove + lcd_vocab_rebuild of the words Memory::cleanUnusedWords drops, Memory.cpp:6899), retirement of
the oldest signature (Memory::disableWordsRef), bulk-loaded and frame-registered signatures side by side, several bucket
seals, a bucket whose signatures all retire, recycled postings keys.  Per frame: word ids identical, likelihood within 1e-4
relative (1e-7 absolute) over ALL live signatures, retired slots exactly 0, the same best candidate; sampled nw identical."""
import numpy as np
import pytest
import torch

from rtabmap_amd import synth

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-7


def _revisit(rng, kind, history, vocab_rows, q, fresh_frac):
    """A frame that revisits an earlier frame: noisy copies of that frame's descriptors (so that words the earlier frame CREATED
    are matched again) plus some fresh descriptors."""
    if kind == "orb":
        fresh = rng.integers(0, 256, (q, vocab_rows.shape[1]), dtype=np.uint8)
    else:
        fresh = synth.vocab_surf(q, seed=int(rng.integers(1 << 30)))
    if not history:
        return fresh
    src = history[int(rng.integers(len(history)))]
    pick = rng.integers(0, src.shape[0], q)
    out = src[pick].copy()
    if kind == "orb":
        out ^= np.packbits(rng.random((q, out.shape[1] * 8)) < 0.02, axis=1)
    else:
        out += rng.standard_normal(out.shape).astype(np.float32) * np.float32(0.02)
        out /= np.linalg.norm(out, axis=1, keepdims=True)
    m = rng.random(q) < fresh_frac
    out[m] = fresh[m]
    return np.ascontiguousarray(out)


class _Mirror:
    """Host-side glue a caller of the C-ABI keeps: which words are rows of the device vocabulary, the words waiting for
    VWDictionary::update(), signature -> slot."""

    def __init__(self, eng, oracle_mem):
        self.eng, self.m = eng, oracle_mem
        self.rows = set()
        self.pending = []          # (word id, descriptor) created by the previous frame
        self.slot_of = {}
        self.n_slots = 0

    def pre_update(self):
        """What Memory::preUpdate does before addNewWords: cleanUnusedWords + VWDictionary::update()."""
        unused = set(self.m.vwd.get_unused_word_ids())
        gone = sorted(w for w in unused if w in self.rows)
        never_indexed = sorted(w for (w, _) in self.pending if w in unused)     # created by the last frame, unreferenced already
        self.pending = [(w, r) for (w, r) in self.pending if w not in unused]
        if gone or never_indexed:
            self.eng.vocab_remove(np.array(gone + never_indexed, np.int32))    # removeWords: rows are tombstoned, keys come back
            self.rows.difference_update(gone)
        if self.pending:
            ids = np.array([w for w, _ in self.pending], np.int32)
            self.eng.vocab_append(np.stack([r for _, r in self.pending]), ids)
            self.rows.update(ids.tolist())
            self.pending = []
        if gone:
            self.eng.vocab_rebuild()


def _run_stream(oracle, kind, n_frames, q, wm, pipeline=False, n_bulk=300, seed=5):
    import rtabmap_amd
    rng = np.random.default_rng(seed)
    if kind == "orb":
        base = synth.vocab_orb(1500, seed=31)
        dim, dt = 32, "u8"
    else:
        base = synth.vocab_surf(3000, seed=32)
        dim, dt = 64, "f32"
    n_base = base.shape[0]
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    eng = rtabmap_amd.Engine(dt, dim, sig_capacity=256, pipeline=pipeline)
    mir = _Mirror(eng, m)
    base_ids = np.arange(1, n_base + 1, dtype=np.int32)
    for i, r in zip(base_ids, base):
        m.vwd.add_word(int(i), r)
    m.vwd.update()
    eng.vocab_append(base, base_ids)
    mir.rows.update(base_ids.tolist())
    # a bulk-loaded memory (Memory::loadDataFromDb): one full bucket + a partly filled one
    words = synth.zipf_words(n_bulk, 90, n_base, seed=33)
    sig_ids = []
    for s in range(n_bulk):
        sid = m.add_signature(words[s])
        sig_ids.append(sid)
        mir.slot_of[sid] = mir.n_slots
        mir.n_slots += 1
    eng.sig_add_bulk(np.array(sig_ids, np.int32), np.arange(0, (n_bulk + 1) * 90, 90, dtype=np.int64), words.reshape(-1))
    live = list(sig_ids)
    history = [base[rng.integers(0, n_base, q)] for _ in range(3)]
    d_words = torch.zeros(q, dtype=torch.int32, device="cuda")
    cap = n_bulk + n_frames + 8
    d_like = torch.full((cap,), -7.0, dtype=torch.float32, device="cuda")
    checked_nw = 0
    for t in range(n_frames):
        desc = _revisit(rng, kind, history, base, q, fresh_frac=0.25)
        history.append(desc)
        if len(history) > 40:
            history.pop(int(rng.integers(0, 20)))
        mir.pre_update()
        first_new = m.vwd.last_word_id + 1
        sid, exp = m.update(desc)
        d = torch.from_numpy(desc).cuda()
        eng.frame_dev(d.data_ptr(), q, sid, float(m.num_signatures()), d_words.data_ptr(), d_like.data_ptr(), cap,
                      incremental=True, new_words_compared=True, nndr=0.8, first_new_word_id=first_new)
        eng.synchronize()
        got = d_words.cpu().numpy()
        mapped = np.where(got < 0, first_new - got - 1, got)
        assert mapped.tolist() == exp, "frame %d: word ids differ from addNewWords" % t
        # the words this frame created wait for the next VWDictionary::update()
        seen = set()
        for i, w in enumerate(got.tolist()):
            if w < 0 and w not in seen:
                seen.add(w)
                mir.pending.append((first_new - w - 1, desc[i]))
        mir.slot_of[sid] = mir.n_slots
        mir.n_slots += 1
        live.append(sid)
        ids = np.array(live, np.int32)
        oi, Lo = m.compute_likelihood(np.array(exp, np.int32), ids)
        assert oi.tolist() == sorted(live)
        Lh_all = d_like[: mir.n_slots].cpu().numpy()
        Lh = Lh_all[[mir.slot_of[s] for s in oi.tolist()]]
        np.testing.assert_allclose(Lh, Lo, rtol=RTOL, atol=ATOL, err_msg="frame %d" % t)
        dead = np.ones(mir.n_slots, bool)
        dead[[mir.slot_of[s] for s in live]] = False
        assert not Lh_all[dead].any(), "frame %d: a retired slot scored" % t
        others = Lo[:-1]                                  # the frame itself is the last (highest) id
        if others.size and others.max() > 0:
            k = int(np.argmax(Lh[:-1]))
            assert Lo[k] >= others.max() * (1 - RTOL), "frame %d: another best candidate" % t
        if t % 37 == 5:
            for w in rng.choice(np.array(sorted(mir.rows)), 12, replace=False).tolist() + [x for x in exp[:4] if x > 0]:
                refs = m.vwd.word_refs(int(w))
                assert eng.word_nrefs(int(w)) == (len(refs) if refs is not None else 0), "frame %d word %d" % (t, w)
                checked_nw += 1
        if len(live) > wm:
            old = live.pop(0)
            m.forget(old)
            eng.sig_remove(old)
    assert checked_nw > 0
    st = eng.stats()
    assert st["signatures"] == len(live)
    eng.close()
    return st


@pytest.mark.parametrize("kind", ["surf", "orb"])
def test_frame_dev_register_and_score_stream(oracle, kind):
    # 300 bulk-loaded + 540 frame-registered signatures, working memory of 270: buckets 0..2 are sealed, bucket 0 (and most of
    # bucket 1) retires completely, words come and go
    st = _run_stream(oracle, kind, n_frames=540, q=96, wm=270)
    assert st["buckets_sealed"] >= 3
    # postings keys are recycled: far fewer in use than words ever created
    assert st["word_slots"] < st["vocab_live"] + 20000, st           # (up to 16384 keys wait for the next batched check)


def test_frame_dev_stream_pipelined_handle(oracle):
    """The same stream on a pipelined handle (lcd_config.pipeline): the test reads every frame back, so each owed index stage is
    completed on its own -- identical results."""
    _run_stream(oracle, "surf", n_frames=70, q=128, wm=330, pipeline=1, seed=9)


# (72 000 words, 700 descriptors: 282 strips x 2 blocks of 512 queries -- the launches with persistent filter workgroups,
# knn_bf16_filter_kernel_p on the plain handle and frame_a_kernel_p on the pipelined one)
@pytest.mark.parametrize("n_words,q", [(6000, 200), (72000, 700)])
def test_pipelined_frames_enqueued_back_to_back(oracle, n_words, q):
    """Frames enqueued without waiting for each other (the bench's pattern) on a pipelined handle -- the tail of frame t - 1 rides in
    the filter launch of frame t, its scoring in the re-rank launch, retirements and the hypothesis keep their place: every frame's
    word ids, likelihood and hypothesis equal the unpipelined handle's, bit for bit."""
    import rtabmap_amd
    n_sig, T = 900, 12
    vocab = synth.vocab_surf(n_words, seed=41)
    words = synth.zipf_words(n_sig, q, n_words, seed=42)
    if n_words > 6000:      # every word referenced at least once: Memory::update would drop the others (cleanUnusedWords), this test never does
        words.reshape(-1)[-n_words:] = np.arange(1, n_words + 1, dtype=np.int32)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    frames = [torch.from_numpy(synth.frame_from_signature(vocab, words[(37 * t) % n_sig], seed=50 + t)).cuda() for t in range(T)]
    out = {}
    for pipe in (0, 1):
        eng = rtabmap_amd.Engine("f32", 64, sig_capacity=n_sig + T, pipeline=pipe)
        eng.vocab_append(vocab, ids)
        eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
        cap = n_sig + T
        d_w = torch.zeros((T, q), dtype=torch.int32, device="cuda")
        d_l = torch.zeros((T, cap), dtype=torch.float32, device="cuda")
        d_h = torch.zeros((T, 8), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for t in range(T):
            eng.frame_dev(frames[t].data_ptr(), q, n_sig + 1 + t, float(n_sig + 1 + t), d_w[t].data_ptr(), d_l[t].data_ptr(), cap,
                          first_new_word_id=n_words + 1 + t * q,    # new words are never read back here: an upper bound per frame
                          d_hypothesis_ptr=d_h[t].data_ptr() if t % 2 else None, exclude_recent=3)
            if t % 3 == 2:
                eng.sig_remove(1 + t // 3)
        eng.synchronize()
        out[pipe] = (d_w.cpu().numpy(), d_l.cpu().numpy(), d_h.cpu().numpy())
        eng.close()
    for k in range(3):
        np.testing.assert_array_equal(out[1][k], out[0][k])
    assert out[0][2][1::2, 0].all() and not out[0][2][0::2].any()       # a best candidate where one was asked for
    # and the first frame agrees with the oracle (later frames meet words the oracle has indexed meanwhile and this test never appends)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8)
    for i, r in zip(ids, vocab):
        m.vwd.add_word(int(i), r)
    m.vwd.update()
    for s in range(n_sig):
        m.add_signature(words[s])
    sid, exp = m.update(frames[0].cpu().numpy())
    got = out[0][0][0]
    assert [w if w > 0 else 0 for w in got.tolist()] == [w if w <= n_words else 0 for w in exp]
    live = np.array(m.signature_ids(), np.int32)
    oi, Lo = m.compute_likelihood(np.array(exp, np.int32), live)
    np.testing.assert_allclose(out[0][1][0][: n_sig + 1], Lo, rtol=RTOL, atol=ATOL)


def test_pipelined_frames_of_changing_size():
    """Descriptor counts that change from frame to frame (1 .. 700, growing and shrinking, across the 512-query block boundary) on a
    pipelined handle: every frame's scratch lives in its own set of the ring, and a larger frame must not move the buffers of the
    frames still in flight.  Word ids and likelihood equal the unpipelined handle's, bit for bit."""
    import rtabmap_amd
    n_words, n_sig, qs = 9000, 400, [40, 300, 90, 700, 33, 512, 513, 1, 640, 64, 700, 5, 256, 257, 100, 700]
    vocab = synth.vocab_surf(n_words, seed=81)
    words = synth.zipf_words(n_sig, 120, n_words, seed=82)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    T, qmax = len(qs), max(qs)
    frames = []
    for t, q in enumerate(qs):
        base = np.concatenate([synth.frame_from_signature(vocab, words[(29 * t + k) % n_sig], seed=90 + 7 * t + k) for k in range(-(-q // 120))])
        frames.append(torch.from_numpy(np.ascontiguousarray(base[:q])).cuda())
    out = {}
    for pipe in (0, 1):
        eng = rtabmap_amd.Engine("f32", 64, sig_capacity=n_sig + T, pipeline=pipe)
        eng.vocab_append(vocab, ids)
        eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * 120, 120, dtype=np.int64), words.reshape(-1))
        cap = n_sig + T
        d_w = torch.zeros((T, qmax), dtype=torch.int32, device="cuda")
        d_l = torch.zeros((T, cap), dtype=torch.float32, device="cuda")
        first = n_words + 1
        for t, q in enumerate(qs):
            eng.frame_dev(frames[t].data_ptr(), q, n_sig + 1 + t, float(n_sig + 1 + t), d_w[t].data_ptr(), d_l[t].data_ptr(), cap,
                          first_new_word_id=first)
            first += q
        eng.synchronize()
        out[pipe] = (d_w.cpu().numpy(), d_l.cpu().numpy())
        eng.close()
    np.testing.assert_array_equal(out[1][0], out[0][0])
    np.testing.assert_array_equal(out[1][1], out[0][1])


def test_pipelined_frames_across_growing_word_tables():
    """Frames whose descriptors are all new words (128 postings keys reserved per frame) on a small engine: the word-indexed tables
    (nw, dense ids, idf stamps) double several times while frames are in flight.  The reservation that moves them belongs to the
    frame whose decision loop is about to run; the registration and scoring arguments of the OLDER frames in the same launches must
    be built after it (they used to hold the freed tables: a memory fault after ~3 000 frames of the bench's stream -- a freed table
    is not always unmapped, so this test guards the results; `bench.py --steps 8000` is the run that faulted).  Word ids and
    likelihood equal the unpipelined handle's, bit for bit."""
    import rtabmap_amd
    n_words, n_sig, q, T = 600, 300, 128, 90
    vocab = synth.vocab_surf(n_words, seed=61)
    words = synth.zipf_words(n_sig, q, n_words, seed=62)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    rng = np.random.default_rng(63)
    frames = []
    for t in range(T):
        f = rng.standard_normal((q, 64)).astype(np.float32)
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        if t % 3 == 0:                                                # every third frame revisits a signature: some existing words too
            f[: q // 2] = synth.frame_from_signature(vocab, words[(17 * t) % n_sig], seed=70 + t)[: q // 2]
        frames.append(torch.from_numpy(f).cuda())
    out = {}
    for pipe in (0, 1):
        eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n_words, sig_capacity=n_sig + T, pipeline=pipe)
        eng.vocab_append(vocab, ids)
        eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
        cap = n_sig + T
        d_w = torch.zeros((T, q), dtype=torch.int32, device="cuda")
        d_l = torch.zeros((T, cap), dtype=torch.float32, device="cuda")
        bytes0 = eng.stats()["bytes_device"]
        for t in range(T):
            eng.frame_dev(frames[t].data_ptr(), q, n_sig + 1 + t, float(n_sig + 1 + t), d_w[t].data_ptr(), d_l[t].data_ptr(), cap,
                          first_new_word_id=n_words + 1 + t * q)
            if t % 4 == 3:
                eng.sig_remove(1 + t // 4)
        eng.synchronize()
        assert eng.stats()["bytes_device"] > bytes0                   # (the tables did grow)
        out[pipe] = (d_w.cpu().numpy(), d_l.cpu().numpy())
        eng.close()
    np.testing.assert_array_equal(out[1][0], out[0][0])
    np.testing.assert_array_equal(out[1][1], out[0][1])
    assert (out[0][0] < 0).sum() > T * q // 2                         # most descriptors became new words


@pytest.mark.parametrize("n_sig,pipeline,bench_mode", [(20000, False, False), (100000, True, False), (100000, True, True)])
def test_frame_dev_at_headline_sizes(oracle, n_sig, pipeline, bench_mode):
    """BASELINE.json's configuration: 49k SURF words, 500 descriptors per frame, a Zipf memory of 100 000 signatures x 500 words (and a
    20 000-signature one on a plain handle).  Frames through lcd_frame_dev with retirement of the oldest signature: ids identical,
    likelihood within 1e-4 over every slot, the same best candidate, sampled nw identical.  bench_mode = the step bench.py times, so that
    the driver's own `pytest -m gpu` covers the benchmarked configuration: the fp16 one-product matrix-core filter (LCD_KNN_F16),
    update()'s append on the device (append_new_words), six frames enqueued back to back (four in flight), nothing completed in between."""
    import rtabmap_amd
    if bench_mode:
        return _headline_bench_mode(oracle, n_sig)
    n_words, q = 49000, 500
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(n_sig, q, n_words, seed=100000)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8)
    for i, r in zip(ids, vocab):
        m.vwd.add_word(int(i), r)
    m.vwd.update()
    assert m.add_signatures_bulk(words) == 1                        # (the bulk constructor is pinned to the one-by-one path: tests/test_oracle_bulk.py)
    eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n_words + 1024, sig_capacity=n_sig + 64, pipeline=pipeline)
    eng.vocab_append(vocab, ids)
    eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
    cap = n_sig + 16
    d_words = torch.zeros(q, dtype=torch.int32, device="cuda")
    d_like = torch.zeros(cap, dtype=torch.float32, device="cuda")
    rng = np.random.default_rng(3)
    for t in range(3):
        src = int(rng.integers(100, n_sig))
        desc = synth.frame_from_signature(vocab, words[src], seed=900 + t)
        first_new = m.vwd.last_word_id + 1
        sid, exp = m.update(desc)
        eng.frame_dev(torch.from_numpy(desc).cuda().data_ptr(), q, sid, float(m.num_signatures()), d_words.data_ptr(), d_like.data_ptr(), cap,
                      first_new_word_id=first_new)
        eng.synchronize()
        got = d_words.cpu().numpy()
        assert np.where(got < 0, first_new - got - 1, got).tolist() == exp
        live = np.array(m.signature_ids(), np.int32)
        oi, Lo = m.compute_likelihood(np.array(exp, np.int32), live)
        Lh = d_like[: n_sig + t + 1].cpu().numpy()
        slots = np.where(oi <= n_sig, oi - 1, oi - 1)          # signature id s sits in slot s - 1 in this test
        np.testing.assert_allclose(Lh[slots], Lo, rtol=RTOL, atol=ATOL)
        assert int(np.argmax(Lh[slots][:-1])) == int(np.argmax(Lo[:-1]))
        for w in rng.integers(1, n_words + 1, 20).tolist():
            refs = m.vwd.word_refs(int(w))
            assert eng.word_nrefs(int(w)) == (len(refs) if refs is not None else 0)
        m.forget(t + 1)
        eng.sig_remove(t + 1)
    st = eng.stats()
    assert st["dense_words"] > 50 and st["buckets_sealed"] == n_sig // 256
    eng.close()


def _headline_bench_mode(oracle, n_sig, n_frames=8):
    import rtabmap_amd
    from rtabmap_amd import capi
    n_words, q = 49000, 500
    vocab = synth.vocab_surf(n_words)
    words = synth.zipf_words(n_sig, q, n_words, seed=100000)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    for i, r in zip(ids, vocab):
        m.vwd.add_word(int(i), r)
    m.vwd.update()
    assert m.add_signatures_bulk(words) == 1
    cap = n_sig + 16
    rng = np.random.default_rng(5)
    src0 = int(rng.integers(100, n_sig))
    frames, first_new, expected, likes = [], [], [], []
    for t in range(n_frames):
        # every other frame looks at the place of the frame before it again: the words that frame created are matched as rows of the vocabulary
        src = src0 if t % 2 else int(rng.integers(100, n_sig))
        src0 = src
        desc = synth.frame_from_signature(vocab, words[src], seed=(950 + t - 1) if t % 2 else (950 + t))
        if t % 2:
            desc = (desc + np.float32(1e-3) * np.random.default_rng(t).standard_normal(desc.shape).astype(np.float32)).astype(np.float32)
        first_new.append(m.vwd.last_word_id + 1)
        sid, exp = m.update(desc)
        frames.append(desc)
        expected.append(exp)
        live = np.array(m.signature_ids(), np.int32)
        likes.append(m.compute_likelihood(np.array(exp, np.int32), live))
        m.forget(t + 1)
    d_desc = [torch.from_numpy(f).cuda() for f in frames]
    # the launch shapes the engine selects by the row-growth estimate (shadow scores, rows instead of postings keys out of the decision loop, its straight
    # first round trip), as the engine picks them / forced on from the first frame / forced off: the same integers and the same likelihood every way
    # ... and with the words numbered ON THE DEVICE (LCD_NEW_WORD_IDS_AUTO: the caller does not know how many words the frames in flight created): the
    # reference's integers -- ++_lastWordId, VWDictionary.cpp:1188 -- without a renumbering
    for opts in ({}, {"shadow_rows": 2, "slots_from_rows": 2, "decision_straight": 2}, {"shadow_rows": 0, "slots_from_rows": 0, "decision_straight": 0},
                 {"shadow_rows": 2, "slots_from_rows": 0, "decision_straight": 2}, {"shadow_rows": 0, "slots_from_rows": 2, "decision_straight": 0},
                 {"auto_ids": 1}, {"auto_ids": 1, "shadow_rows": 0}):
        opts = dict(opts)
        auto = bool(opts.pop("auto_ids", 0))
        eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n_words + 8192, sig_capacity=n_sig + 64, pipeline=True, knn_mode="f16")
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.vocab_append(vocab, ids)
        eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
        d_words = torch.zeros((n_frames, q), dtype=torch.int32, device="cuda")
        d_like = torch.zeros((n_frames, cap), dtype=torch.float32, device="cuda")
        d_first = torch.zeros(n_frames, dtype=torch.int32, device="cuda")
        if auto:
            eng.set_option("next_word_id", first_new[0])
        torch.cuda.synchronize()
        for t in range(n_frames):
            eng.frame_dev(d_desc[t].data_ptr(), q, n_sig + 1 + t, float(n_sig + 1), d_words[t].data_ptr(), d_like[t].data_ptr(), cap,
                          first_new_word_id=capi.LCD_NEW_WORD_IDS_AUTO if auto else first_new[t], append_new_words=True,
                          d_first_new_word_id_ptr=d_first[t:].data_ptr())
            eng.sig_remove(t + 1)
        eng.synchronize()
        got, like = d_words.cpu().numpy(), d_like.cpu().numpy()
        assert d_first.cpu().numpy().tolist() == first_new, "the id of every frame's first new word, options %r auto %r" % (opts, auto)
        if auto:                                                    # the host learns ids and postings keys of the device's words from the rows
            for w in [first_new[1], first_new[2] - 1, first_new[-1]]:
                refs = m.vwd.word_refs(int(w))
                if refs is not None:
                    assert eng.word_nrefs(int(w)) == len(refs), "word %d" % w
        matched_new = 0
        for t in range(n_frames):
            mapped = np.where(got[t] < 0, first_new[t] - got[t] - 1, got[t])
            assert mapped.tolist() == expected[t], "frame %d, options %r" % (t, opts)
            if t % 2:
                matched_new += int(((got[t] > n_words)).sum())
            oi, Lo = likes[t]
            Lh = like[t][oi - 1]                                    # signature id s sits in slot s - 1 in this test
            np.testing.assert_allclose(Lh, Lo, rtol=RTOL, atol=ATOL, err_msg="frame %d, options %r" % (t, opts))
            assert int(np.argmax(Lh[:-1])) == int(np.argmax(Lo[:-1]))
        assert matched_new > 50, "the revisits must match words the frames before them created (rows appended on the device)"
        eng.close()


def test_hypothesis_from_the_device(oracle):
    """lcd_frame_dev's hypothesis output == Rtabmap::adjustLikelihood (restated) on the likelihood of the considered signatures +
    the best of them; the adjusted vector agrees entry by entry."""
    import ctypes as C
    import rtabmap_amd
    from rtabmap_amd.capi import LcdHypothesis
    n_words, n_sig, q = 5000, 1200, 300
    vocab = synth.vocab_surf(n_words, seed=61)
    words = synth.zipf_words(n_sig, q, n_words, seed=62)
    eng = rtabmap_amd.Engine("f32", 64, sig_capacity=n_sig + 8)
    eng.vocab_append(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
    for s in (5, 17, 300):
        eng.sig_remove(s)
    cap = n_sig + 8
    d_words = torch.zeros(q, dtype=torch.int32, device="cuda")
    d_like = torch.zeros(cap, dtype=torch.float32, device="cuda")
    d_adj = torch.zeros(cap + 1, dtype=torch.float32, device="cuda")
    d_hyp = torch.zeros(8, dtype=torch.int32, device="cuda")
    for ratio in (0.0, 0.5):
        for t, src in enumerate((700, 41)):
            desc = synth.frame_from_signature(vocab, words[src], seed=70 + t)
            sid = n_sig + 1 + t + (10 if ratio else 0)
            exclude = 20
            eng.frame_dev(torch.from_numpy(desc).cuda().data_ptr(), q, sid, float(n_sig), d_words.data_ptr(), d_like.data_ptr(), cap,
                          d_hypothesis_ptr=d_hyp.data_ptr(), d_adjusted_ptr=d_adj.data_ptr(), exclude_recent=exclude,
                          virtual_place_ratio=ratio)
            eng.synchronize()
            _, n_slots = eng.slots_dev()
            L = d_like[:n_slots].cpu().numpy()
            considered = np.ones(n_slots, bool)
            considered[n_slots - exclude:] = False
            considered[[4, 16, 299]] = False
            vec = np.concatenate([[0.0], L[considered]]).astype(np.float32)
            exp = oracle.adjust_likelihood(vec, ratio)
            h = LcdHypothesis.from_buffer_copy(d_hyp.cpu().numpy().tobytes())
            adj = d_adj[: n_slots + 1].cpu().numpy()
            np.testing.assert_allclose(adj[0], exp[0], rtol=1e-5)
            np.testing.assert_allclose(adj[1:][considered], exp[1:], rtol=1e-5, atol=1e-7)
            assert not adj[1:][~considered].any()
            best = int(np.flatnonzero(considered)[np.argmax(L[considered])])
            assert h.slot == best and h.sig_id == best + 1
            assert h.likelihood == L[best]
            np.testing.assert_allclose(h.adjusted, adj[1 + best], rtol=1e-6)
            np.testing.assert_allclose(h.virtual_place, exp[0], rtol=1e-5)
            assert h.n_positive == int((L[considered] > 0).sum())
    eng.close()


def test_two_handles_with_different_knn_modes(oracle):
    """The 2-NN mode belongs to the handle (lcd_config.knn_mode), not to the process: two engines side by side."""
    import rtabmap_amd
    v = synth.vocab_surf(5000, seed=81)
    qs = synth.queries_surf(v, 300, seed=82)
    ids = np.arange(1, 5001, dtype=np.int32)
    a = rtabmap_amd.Engine("f32", 64, knn_mode="valu")
    b = rtabmap_amd.Engine("f32", 64, knn_mode="bf16")
    c = rtabmap_amd.Engine("f32", 64, knn_mode="mfma32")
    for e in (a, b, c):
        e.vocab_append(v, ids)
    ra, rb, rc = a.knn2(qs), b.knn2(qs), c.knn2(qs)
    for r in (rb, rc):
        np.testing.assert_array_equal(r[0], ra[0])
        np.testing.assert_array_equal(r[1], ra[1])
    idx, dist = oracle.knn2_linear(v, qs)
    np.testing.assert_array_equal(ra[0], idx + 1)
    np.testing.assert_array_equal(ra[1], dist)
    assert a.stats()["knn_max_err_ratio"] == 0.0            # no filter ran on this handle
    assert b.stats()["knn_max_err_ratio"] > 0.0 and c.stats()["knn_max_err_ratio"] > 0.0
    for e in (a, b, c):
        e.close()


def test_bulk_load_matches_incremental_registration(oracle):
    """lcd_sig_add_bulk (one registration launch, batched sealing) and lcd_sig_add one by one give the same likelihood, and both
    match the oracle; negative idf (N smaller than nw) follows the reference's arithmetic too."""
    import rtabmap_amd
    n_words, n_sig, q = 4000, 1300, 150
    words = synth.zipf_words(n_sig, q, n_words, seed=91)
    words[5, :40] = 0                     # features without a word: they only count in ni
    words[6, :] = words[6, 0]             # one word 150 times
    sig_ids = np.arange(1, n_sig + 1, dtype=np.int32)
    offs = np.arange(0, (n_sig + 1) * q, q, dtype=np.int64)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
    for w in range(1, n_words + 1):
        m.vwd.add_word(w, np.zeros(64, np.float32))
    for s in range(n_sig):
        m.add_signature(words[s])
    a = rtabmap_amd.Engine("f32", 64)
    b = rtabmap_amd.Engine("f32", 64)
    a.sig_add_bulk(sig_ids[:700], offs[:701], words[:700].reshape(-1))
    a.sig_add_bulk(sig_ids[700:], offs[700:] - offs[700], words[700:].reshape(-1))
    for s in range(n_sig):
        b.sig_add(int(sig_ids[s]), words[s])
    for t in range(3):
        qw = synth.query_from_signature(words[100 + 300 * t], n_words, seed=t)
        _, Lo = m.compute_likelihood(qw, sig_ids)
        La = a.likelihood(qw, sig_ids, float(n_sig))
        Lb = b.likelihood(qw, sig_ids, float(n_sig))
        np.testing.assert_array_equal(La, Lb)          # integer accumulation: the layout does not change the bits
        np.testing.assert_allclose(La, Lo, rtol=RTOL, atol=ATOL)
        assert int(np.argmax(La)) == int(np.argmax(Lo))
    # N < nw for the frequent words: negative terms, as log10(N / nw) gives them in the reference
    qw = words[6]
    La = a.likelihood(qw, sig_ids, 3.0)
    has = (words == words[6, 0])
    nw = np.float32(has.any(axis=1).sum())
    idf = np.float32(np.log10(np.float32(3.0) / nw))
    exp = (has.sum(axis=1).astype(np.float32) * idf) / np.float32(q)
    assert idf < 0 and (La < 0).any()
    np.testing.assert_allclose(La, exp, rtol=1e-5, atol=1e-9)
    a.close(); b.close()


@pytest.mark.parametrize("kind,pipeline", [("surf", False), ("orb", False), ("surf", True)])
def test_frame_host_is_frame_dev_with_the_copies(kind, pipeline):
    """lcd_frame_host (ABI v5: host descriptors in, word ids + dense likelihood out, one synchronisation -- what the reference-interface
    mirror calls per frame) against lcd_frame_dev on a twin engine fed through device pointers: the same bits, frame after frame, with
    update()'s append on the device, a retirement per frame, and lcd_slot_count telling the caller how large its likelihood buffer must be."""
    import rtabmap_amd
    n_words, q, n_bulk, n_frames = 3000, 96, 40, 12
    rng = np.random.default_rng(91)
    base = synth.vocab_surf(n_words, seed=92) if kind == "surf" else synth.vocab_orb(n_words, seed=92)
    words = synth.zipf_words(n_bulk, q, n_words, seed=93)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    engs = []
    for p in (False, pipeline):
        e = rtabmap_amd.Engine("f32" if kind == "surf" else "u8", base.shape[1], sig_capacity=n_bulk + n_frames + 8, pipeline=p)
        e.vocab_append(base, ids)
        e.sig_add_bulk(np.arange(1, n_bulk + 1, dtype=np.int32), np.arange(0, (n_bulk + 1) * q, q, dtype=np.int64), words.reshape(-1))
        engs.append(e)
    dev, host = engs
    cap = n_bulk + n_frames + 8
    d_w = torch.zeros(q, dtype=torch.int32, device="cuda")
    d_l = torch.zeros(cap, dtype=torch.float32, device="cuda")
    history = [base[rng.integers(0, n_words, q)] for _ in range(2)]
    first_new = n_words + 1
    created = 0
    for t in range(n_frames):
        desc = _revisit(rng, kind, history, base, q, fresh_frac=0.3)
        history.append(desc)
        sid = n_bulk + 1 + t
        d = torch.from_numpy(desc).cuda()
        dev.frame_dev(d.data_ptr(), q, sid, float(n_bulk + 1), d_w.data_ptr(), d_l.data_ptr(), cap, first_new_word_id=first_new, append_new_words=True)
        dev.synchronize()
        exp_w, exp_l = d_w.cpu().numpy(), d_l[:sid].cpu().numpy()
        got_w, got_l = host.frame_host(desc, sid, float(n_bulk + 1), first_new_word_id=first_new, append_new_words=True)
        np.testing.assert_array_equal(got_w, exp_w, err_msg="frame %d" % t)
        assert got_l.shape[0] == sid
        np.testing.assert_array_equal(got_l, exp_l, err_msg="frame %d" % t)
        n_new = int(-exp_w.min()) if exp_w.min() < 0 else 0
        created += n_new
        first_new += n_new
        dev.sig_remove(t + 1)
        host.sig_remove(t + 1)
    assert created > 100
    assert dev.vocab_count() == host.vocab_count() == (n_words + created, n_words + created)
    for e in engs:
        e.close()
