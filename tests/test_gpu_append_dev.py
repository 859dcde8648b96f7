"""GPU parity of VWDictionary::update()'s append branch run ON THE DEVICE (lcd_frame_args.append_new_words): the words a frame creates
become vocabulary rows behind the decision loop, without lcd_vocab_append and without a host round trip, and the NEXT frame -- whose
matrix-core filter took its snapshot of the vocabulary before those rows existed on a pipelined handle -- still finds them (its re-rank
scans the appended rows exactly).  Against the restated Memory::update (preUpdate: cleanUnusedWords + VWDictionary::update(), then
addNewWords) + computeLikelihood on the same descriptor stream, frames enqueued back to back."""
import numpy as np
import pytest
import torch

from rtabmap_amd import synth
from test_gpu_frame_stream import _revisit, RTOL, ATOL

pytestmark = pytest.mark.gpu


def _stream(oracle, pipeline, n_words, q, n_frames, seed, sync_every=0, kind="surf", knn_mode=None, options=None, clean_every_frame=False, auto_ids=False):
    import rtabmap_amd
    from rtabmap_amd import capi
    rng = np.random.default_rng(seed)
    base = synth.vocab_surf(n_words, seed=seed + 1) if kind == "surf" else synth.vocab_orb(n_words, seed=seed + 1)
    n_bulk = max(40, (n_words + q - 1) // q + 2)
    words = synth.zipf_words(n_bulk, q, n_words, seed=seed + 2)
    words.reshape(-1)[-n_words:] = np.arange(1, n_words + 1, dtype=np.int32)       # every word referenced: cleanUnusedWords drops none
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    for i, r in zip(ids, base):
        m.vwd.add_word(int(i), r)
    m.vwd.update()
    for s in range(n_bulk):
        m.add_signature(words[s])
    # the oracle runs ahead: it tells which id the first new word of every frame gets (++_lastWordId, VWDictionary.cpp:1185)
    history = [base[rng.integers(0, n_words, q)] for _ in range(2)]
    frames, first_new, expected, likes = [], [], [], []
    for t in range(n_frames):
        desc = _revisit(rng, kind, history, base, q, fresh_frac=0.3)
        history.append(desc)
        first_new.append(m.vwd.last_word_id + 1)
        sid, exp = m.update(desc)
        assert sid == n_bulk + 1 + t
        frames.append(desc)
        expected.append(exp)
        live = np.array(m.signature_ids(), np.int32)
        likes.append(m.compute_likelihood(np.array(exp, np.int32), live)[1])
    assert not m.vwd.get_unused_word_ids()
    eng = rtabmap_amd.Engine("f32" if kind == "surf" else "u8", base.shape[1], sig_capacity=n_bulk + n_frames + 8, pipeline=pipeline, knn_mode=knn_mode)
    for key, value in (options or {}).items():
        eng.set_option(key, value)
    eng.vocab_append(base, ids)
    eng.sig_add_bulk(np.arange(1, n_bulk + 1, dtype=np.int32), np.arange(0, (n_bulk + 1) * q, q, dtype=np.int64), words.reshape(-1))
    cap = n_bulk + n_frames + 8
    d_desc = [torch.from_numpy(f).cuda() for f in frames]
    d_w = torch.zeros((n_frames, q), dtype=torch.int32, device="cuda")
    d_l = torch.zeros((n_frames, cap), dtype=torch.float32, device="cuda")
    d_first = torch.zeros(n_frames, dtype=torch.int32, device="cuda")
    if auto_ids:                                  # the device numbers the words (LCD_NEW_WORD_IDS_AUTO): the caller only says where the dictionary stands
        eng.set_option("next_word_id", first_new[0])
    torch.cuda.synchronize()
    for t in range(n_frames):
        eng.frame_dev(d_desc[t].data_ptr(), q, n_bulk + 1 + t, float(n_bulk + 1 + t), d_w[t].data_ptr(), d_l[t].data_ptr(), cap,
                      first_new_word_id=capi.LCD_NEW_WORD_IDS_AUTO if auto_ids else first_new[t], append_new_words=True,
                      d_first_new_word_id_ptr=d_first[t:].data_ptr())
        if clean_every_frame:
            eng.vocab_remove_unused_async()       # Memory::preUpdate of the next frame; no word is ever unused here: it must remove nothing
        if sync_every and t % sync_every == sync_every - 1:
            eng.synchronize()                     # completes the owed stages stand-alone: the appends of the drained frames included
    eng.synchronize()
    got, like = d_w.cpu().numpy(), d_l.cpu().numpy()
    assert d_first.cpu().numpy().tolist() == first_new, "the id of every frame's first new word as ++_lastWordId gives it (VWDictionary.cpp:1188)"
    n_created = 0
    for t in range(n_frames):
        mapped = np.where(got[t] < 0, first_new[t] - got[t] - 1, got[t])
        assert mapped.tolist() == expected[t], "frame %d: word ids differ from addNewWords over the updated vocabulary" % t
        n_created += len(set(w for w in got[t].tolist() if w < 0))
        np.testing.assert_allclose(like[t][: n_bulk + 1 + t], likes[t], rtol=RTOL, atol=ATOL, err_msg="frame %d" % t)
    assert n_created > 0
    # the vocabulary is what update() would have built: the base rows, then every created word in id order
    rows, live = eng.vocab_count()
    assert rows == live == n_words + n_created
    vr, vi = eng.vocab_read(n_words, n_created)
    assert vi.tolist() == sorted(vi.tolist()) and len(set(vi.tolist())) == n_created
    by_id = {}
    for t in range(n_frames):
        for i, w in enumerate(got[t].tolist()):
            if w < 0:
                by_id.setdefault(first_new[t] - w - 1, frames[t][i])                 # the first descriptor with the code created the word
    assert sorted(by_id) == vi.tolist()
    np.testing.assert_array_equal(vr, np.stack([by_id[i] for i in vi.tolist()]))
    # and the rows are searchable like any others: the 2-NN of a created word's descriptor is that word at distance 0
    probe = vr[:: max(1, n_created // 16)]
    kid, kd = eng.knn2(probe)
    assert kid[:, 0].tolist() == vi[:: max(1, n_created // 16)].tolist() and not kd[:, 0].any()
    st = eng.stats()
    assert st["vocab_rows"] == n_words + n_created
    assert st["clean_divergent_refs"] == 0        # nothing is retired in these streams: no enqueued clean can have tombstoned a word a frame in flight matched
    for key in (options or {}):
        eng.set_option(key, -1)                   # (process-wide options go back to their built-in values)
    eng.close()
    return n_created


def test_appended_rows_written_by_a_launch_of_their_own(oracle):
    # memories of 1024 sealed buckets and more leave the row writers of a deferred append to a kernel behind launch B (the scoring branch of
    # the fused launch keeps its registers that way); the option runs that path at this test's size
    assert _stream(oracle, True, n_words=3000, q=96, n_frames=30, seed=13, options={"append_split_buckets": 0, "append_from_rerank": 0}) > 200


@pytest.mark.parametrize("pipeline", [False, True])
def test_words_of_a_frame_in_flight_survive_the_enqueued_clean(oracle, pipeline):
    """cleanUnusedWords enqueued behind EVERY frame of a stream in which no word ever loses its last reference (nothing is retired, every
    base word is referenced): the reference's clean removes nothing (a word addNewWords creates is referenced as it is created,
    VWDictionary.cpp:1185-1195), so the stream must be the oracle's word for word -- the words a frame creates are matched, as positive
    ids, by the revisits a few frames later, and every created row is alive at the end.  On a pipelined handle the clean runs between a
    frame's row writers and its registration: it must not take the not yet referenced rows for unused words (the advisor's round-4
    finding)."""
    assert _stream(oracle, pipeline, n_words=3000, q=96, n_frames=40, seed=17, clean_every_frame=True) > 200


@pytest.mark.parametrize("pipeline", [False, True])
def test_append_new_words_on_the_device(oracle, pipeline):
    assert _stream(oracle, pipeline, n_words=3000, q=96, n_frames=30, seed=11) > 200


@pytest.mark.parametrize("who", [0, 1])
def test_rows_of_the_deferred_append_written_by_either_kind_of_workgroup(oracle, who):
    """the rows a frame appended are written by the re-rank workgroups of launch B from their staging area (the default) or by round 4's
    eight row-writer workgroups (lcd_set_option "append_from_rerank" = 0, kept for A/B runs and for frames without re-rank workgroups):
    same results"""
    assert _stream(oracle, True, n_words=3000, q=96, n_frames=30, seed=19, options={"append_from_rerank": who}) > 200


@pytest.mark.parametrize("n_words,q,n_frames,knn_mode,n_wr", [(3000, 96, 30, None, 3), (3000, 200, 12, "f16", 16), (72000, 700, 8, "f16", 24), (3000, 96, 30, "f16", 1)])
def test_rows_written_by_extra_workgroups_of_the_rerank_role(oracle, n_words, q, n_frames, knn_mode, n_wr):
    """lcd_set_option "row_writer_wgs" = n: the rows a frame appended are written by n extra workgroups of launch B's re-rank role (no re-rank
    workgroup then has the row stores at the end of its chain, and the kernel gets no third branch) -- word ids, likelihood and vocabulary as
    when the re-rank workgroups write them; one writer alone takes every row (several chunks of its staging area at 700 descriptors)"""
    assert _stream(oracle, True, n_words=n_words, q=q, n_frames=n_frames, seed=31, knn_mode=knn_mode, options={"row_writer_wgs": n_wr}) > 100


@pytest.mark.parametrize("n_words,q,n_frames,knn_mode,extra", [(3000, 96, 40, None, {}), (3000, 200, 16, "f16", {"row_writer_wgs": 3}), (72000, 700, 8, "f16", {"mirror_from_b": 1}),
                                                               (3000, 96, 40, "f16", {"clean": 1}), (150003, 300, 6, "f16", {})])
def test_new_rows_ranked_by_the_filter_as_shadow_rows(oracle, n_words, q, n_frames, knn_mode, extra):
    """lcd_set_option "shadow_rows" = 1: the words the previous frame created are descriptors of that frame; its query pre-split leaves them as rows
    of an operand table, the matrix-core filter of this frame ranks them in extra strips, and the re-rank keeps the candidates whose descriptor
    the previous decision loop's mask names -- no workgroup stages or scans the new rows.  Word ids, likelihood and vocabulary are the oracle's:
    streams that revisit the place of the frame before them (words matched one frame after they were created), bf16 and fp16 filters, frames of
    700 descriptors (three shadow strips, several mask words), a clean behind every frame, and a vocabulary whose filter is persistent (where the
    option changes nothing: the rows are staged as before)."""
    opts = {"shadow_rows": 2}                                          # always (the built-in, 1, waits until the stream has shown that it creates words)
    opts.update({k: v for k, v in extra.items() if k != "clean"})
    assert _stream(oracle, True, n_words=n_words, q=q, n_frames=n_frames, seed=41, knn_mode=knn_mode, options=opts, clean_every_frame=bool(extra.get("clean"))) > 100


@pytest.mark.parametrize("options", [{"shadow_rows": 0}, {"shadow_rows": 0, "row_writer_wgs": 0}, {"shadow_rows": 0, "row_writer_wgs": 0, "mirror_from_b": 0}])
@pytest.mark.parametrize("knn_mode", [None, "f16"])
def test_new_rows_staged_and_scanned_by_the_rerank(oracle, options, knn_mode):
    """the paths the shadow scores replaced stay in the library (vocabularies whose filter is persistent take them, and the options select them
    for A/B runs): every re-rank workgroup stages the rows the previous frame appended and scans them exactly -- rows written by the writer
    workgroups of the re-rank role, or (row_writer_wgs = 0) by the re-rank workgroups from their staging area; mirror stored by either launch"""
    assert _stream(oracle, True, n_words=3000, q=96, n_frames=30, seed=43, knn_mode=knn_mode, options=options) > 200


@pytest.mark.parametrize("options", [{"mirror_from_b": 1}, {"mirror_from_b": 1, "row_writer_wgs": 8}, {"mirror_from_b": 1, "append_from_rerank": 0},
                                     {"mirror_from_b": 1, "append_split_buckets": 0, "append_from_rerank": 0}])
def test_row_count_mirror_stored_by_launch_b(oracle, options):
    """lcd_set_option "mirror_from_b" = 1: the pinned row-count mirror the host plans its launches from is stored by the launch that writes the
    frame's rows (whichever kind of workgroup writes them), not at the end of the decision loop's chain -- same results, and the host's
    throttle (which reads the mirror's tag) keeps up over a stream longer than its eight-frame window"""
    assert _stream(oracle, True, n_words=3000, q=96, n_frames=40, seed=37, options=options) > 200


@pytest.mark.parametrize("n_words,q,n_frames,knn_mode", [(3000, 96, 30, None), (3000, 200, 12, "f16"), (72000, 700, 8, "f16")])
def test_pending_rows_read_from_the_cross_frame_tiles(oracle, n_words, q, n_frames, knn_mode):
    """lcd_set_option "cross_frame_tiles" = 1: the words the previous frame created are descriptors of that frame, so extra distance tiles of
    launch A compute this frame x the frame before it in the reference's arithmetic and the re-rank gathers its pending rows' distances from
    that matrix (each workgroup stages only the rows it writes) -- word ids, likelihood and vocabulary as with the staged rows"""
    assert _stream(oracle, True, n_words=n_words, q=q, n_frames=n_frames, seed=29, knn_mode=knn_mode, options={"cross_frame_tiles": 1}) > 100


@pytest.mark.parametrize("pipeline,n_words,q", [(False, 3000, 96), (True, 3000, 96), (True, 72000, 700)])
def test_append_new_words_with_the_fp16_filter(oracle, pipeline, n_words, q):
    """LCD_KNN_F16: the one-product fp16 matrix-core filter (operand tables in IEEE half, rows appended on the device split the same
    way, the re-rank's wider error bound) gives the same word ids, likelihood and vocabulary as the exact scan -- plain handle,
    pipelined frames, and the persistent filter over a vocabulary that grows while frames are in flight."""
    assert _stream(oracle, pipeline, n_words=n_words, q=q, n_frames=8 if n_words > 10000 else 30, seed=13, knn_mode="f16") > 100


def test_append_new_words_on_the_device_orb(oracle):
    """256-bit binary descriptors (Hamming scan, plain handle): the rows are copied as they are"""
    assert _stream(oracle, False, n_words=1500, q=96, n_frames=20, seed=31, kind="orb") > 100


@pytest.mark.parametrize("kind,pipeline,sync_every,n_words,q,n_frames", [("surf", True, 0, 3000, 96, 30), ("surf", False, 0, 3000, 96, 30), ("orb", False, 0, 1500, 96, 20),
                                                                         ("surf", True, 5, 2600, 120, 24), ("surf", True, 0, 72000, 700, 8)])
def test_new_words_numbered_on_the_device(oracle, kind, pipeline, sync_every, n_words, q, n_frames):
    """LCD_NEW_WORD_IDS_AUTO: a caller that never learns how many words the frames in flight created gets the reference's integers all the same --
    the id of a new word is its vocabulary row + (next word id - rows) at the start of the run of appending frames (every new word is one row and
    one id); drains in between, the plain handle's lazy chain (ORB) and vocabulary buffers that grow under the frames included."""
    assert _stream(oracle, pipeline, n_words=n_words, q=q, n_frames=n_frames, seed=41, sync_every=sync_every, kind=kind, auto_ids=True,
                   knn_mode="f16" if (kind == "surf" and pipeline) else None) > 50


def test_append_new_words_pipelined_with_drains_in_between(oracle):
    _stream(oracle, True, n_words=2600, q=120, n_frames=24, seed=23, sync_every=5)


def test_append_new_words_persistent_filter_and_growth(oracle):
    """72 000 words x 700 descriptors: the persistent filter launch (rows clamped to the device count), vocabulary buffers that grow
    while frames are in flight"""
    _stream(oracle, True, n_words=72000, q=700, n_frames=8, seed=5)


@pytest.mark.parametrize("kind,pipeline", [("surf", False), ("surf", True), ("orb", False)])
def test_device_resident_stream_with_clean_unused_words(oracle, kind, pipeline):
    """The whole of Memory::preUpdate on the device, frame after frame, with a small working memory so that words die: cleanUnusedWords
    (lcd_vocab_remove_unused, from the device's own reference counts) -> update() (the previous frame's appends are already rows) ->
    addNewWords + references + likelihood (lcd_frame_dev, append_new_words) -> the oldest signature leaves.  The removed word ids are
    exactly the oracle's getUnusedWords(), the word assignment and the likelihood its Memory::update / computeLikelihood."""
    import rtabmap_amd
    rng = np.random.default_rng(77)
    n_words, q, n_frames, wm = 2000, 96, 70, 12
    base = synth.vocab_surf(n_words, seed=78) if kind == "surf" else synth.vocab_orb(n_words, seed=78)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    for i, r in zip(ids, base):
        m.vwd.add_word(int(i), r)
    m.vwd.update()
    eng = rtabmap_amd.Engine("f32" if kind == "surf" else "u8", base.shape[1], sig_capacity=n_frames + 8, pipeline=pipeline)
    eng.vocab_append(base, ids)
    cap = n_frames + 8
    d_w = torch.zeros(q, dtype=torch.int32, device="cuda")
    d_l = torch.zeros(cap, dtype=torch.float32, device="cuda")
    history = [base[rng.integers(0, n_words, q)] for _ in range(2)]
    live, removed_total = [], 0
    for t in range(n_frames):
        desc = _revisit(rng, kind, history, base, q, fresh_frac=0.3)
        history.append(desc)
        if len(history) > 12:
            history.pop(0)
        unused_o = sorted(m.vwd.get_unused_word_ids())
        n, gone = eng.vocab_remove_unused(capacity=8192)
        assert n == len(unused_o) and sorted(gone.tolist()) == unused_o, "frame %d: cleanUnusedWords removes other words" % t
        removed_total += n
        if n and t % 9 == 0:
            eng.vocab_rebuild()                                        # the reference rebuilds whenever something was removed: same row order
        first_new = m.vwd.last_word_id + 1
        sid, exp = m.update(desc)
        d = torch.from_numpy(desc).cuda()
        eng.frame_dev(d.data_ptr(), q, sid, float(m.num_signatures()), d_w.data_ptr(), d_l.data_ptr(), cap, first_new_word_id=first_new,
                      append_new_words=True)
        eng.synchronize()
        got = d_w.cpu().numpy()
        assert np.where(got < 0, first_new - got - 1, got).tolist() == exp, "frame %d" % t
        live.append(sid)
        oi, Lo = m.compute_likelihood(np.array(exp, np.int32), np.array(live, np.int32))
        np.testing.assert_allclose(d_l[:sid].cpu().numpy()[oi - 1], Lo, rtol=RTOL, atol=ATOL, err_msg="frame %d" % t)
        if len(live) > wm:
            old = live.pop(0)
            m.forget(old)
            eng.sig_remove(old)
    assert removed_total > 500
    rows, n_live = eng.vocab_count()
    assert n_live == len(m.vwd.word_ids())                             # (the words the last retirement left unused go with the next preUpdate)
    eng.close()


@pytest.mark.parametrize("kind,pipeline", [("surf", False), ("surf", True), ("orb", False)])
def test_clean_unused_words_enqueued_without_a_drain(oracle, kind, pipeline):
    """lcd_vocab_remove_unused_async: the same Memory::preUpdate stream as above, but cleanUnusedWords is ONE kernel enqueued behind the
    frame (nothing comes back, nothing is synchronised by the call).  The frames are completed one by one here (lcd_synchronize after
    each), so the clean sits exactly where the reference runs it and everything downstream -- word assignment over the cleaned
    vocabulary, likelihood, the number of live words the host mirror reports once it has caught up -- must be the oracle's."""
    import rtabmap_amd
    rng = np.random.default_rng(177)
    n_words, q, n_frames, wm = 2000, 96, 70, 12
    base = synth.vocab_surf(n_words, seed=178) if kind == "surf" else synth.vocab_orb(n_words, seed=178)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    for i, r in zip(ids, base):
        m.vwd.add_word(int(i), r)
    m.vwd.update()
    eng = rtabmap_amd.Engine("f32" if kind == "surf" else "u8", base.shape[1], sig_capacity=n_frames + 8, pipeline=pipeline)
    eng.vocab_append(base, ids)
    cap = n_frames + 8
    d_w = torch.zeros(q, dtype=torch.int32, device="cuda")
    d_l = torch.zeros(cap, dtype=torch.float32, device="cuda")
    history = [base[rng.integers(0, n_words, q)] for _ in range(2)]
    live, removed_total = [], 0
    for t in range(n_frames):
        desc = _revisit(rng, kind, history, base, q, fresh_frac=0.3)
        history.append(desc)
        if len(history) > 12:
            history.pop(0)
        removed_total += len(m.vwd.get_unused_word_ids())
        eng.vocab_remove_unused_async()
        if t % 9 == 8:
            rows, n_live = eng.vocab_count()                           # completes what is owed: the host mirror catches up with the log
            assert n_live == len(m.vwd.word_ids()) - len(m.vwd.get_unused_word_ids()), "frame %d" % t
            eng.vocab_rebuild()
        first_new = m.vwd.last_word_id + 1
        sid, exp = m.update(desc)
        d = torch.from_numpy(desc).cuda()
        eng.frame_dev(d.data_ptr(), q, sid, float(m.num_signatures()), d_w.data_ptr(), d_l.data_ptr(), cap, first_new_word_id=first_new,
                      append_new_words=True)
        eng.synchronize()
        got = d_w.cpu().numpy()
        assert np.where(got < 0, first_new - got - 1, got).tolist() == exp, "frame %d" % t
        live.append(sid)
        oi, Lo = m.compute_likelihood(np.array(exp, np.int32), np.array(live, np.int32))
        np.testing.assert_allclose(d_l[:sid].cpu().numpy()[oi - 1], Lo, rtol=RTOL, atol=ATOL, err_msg="frame %d" % t)
        if len(live) > wm:
            old = live.pop(0)
            m.forget(old)
            eng.sig_remove(old)
    assert removed_total > 500
    n, _ = eng.vocab_remove_unused(capacity=0)                         # what the last retirement left unused
    assert n == len(m.vwd.get_unused_word_ids())
    rows, n_live = eng.vocab_count()
    assert n_live == len(m.vwd.word_ids()) - n
    eng.close()


def test_clean_unused_words_behind_frames_in_flight(oracle):
    """The pipelined stream the bench times with `update()` in the step: frame (append on the device), retirement, cleanUnusedWords -- all
    enqueued, four frames in flight, nothing completed for 150 frames.  The clean then runs behind frames that took their snapshot of the
    vocabulary earlier (lcd.h: a word such a frame still matched keeps that signature's references), so the word assignment is the
    DEVICE's; what must hold whatever the interleaving: the likelihood is Memory::computeLikelihood of a memory that registered exactly the
    words the device reported (postings keys are recycled ~5 times over in this stream: a key handed out twice, or the key of a live row
    handed out again, shows up here), no live row is left without a reference except the last frames' leftovers, and every tombstoned
    row's word is gone from the host mirror."""
    import rtabmap_amd
    rng = np.random.default_rng(277)
    n_words, q, n_frames, wm = 3000, 200, 150, 6
    base = synth.vocab_surf(n_words, seed=278)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    eng = rtabmap_amd.Engine("f32", 64, sig_capacity=n_frames + 8, pipeline=True)
    eng.vocab_append(base, ids)
    cap = n_frames + 8
    frames, history = [], [base[rng.integers(0, n_words, q)] for _ in range(2)]
    for t in range(n_frames):
        desc = _revisit(rng, "surf", history, base, q, fresh_frac=0.3)
        history.append(desc)
        if len(history) > 8:
            history.pop(0)
        frames.append(desc)
    d_desc = [torch.from_numpy(f).cuda() for f in frames]
    d_w = torch.zeros((n_frames, q), dtype=torch.int32, device="cuda")
    d_l = torch.zeros((4, cap), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for t in range(n_frames):
        eng.frame_dev(d_desc[t].data_ptr(), q, t + 1, float(min(t + 1, wm + 1)), d_w[t].data_ptr(), d_l[t % 4].data_ptr(), cap,
                      first_new_word_id=n_words + 1 + t * q, append_new_words=True)
        if t >= wm:
            eng.sig_remove(t + 1 - wm)
        eng.vocab_remove_unused_async()
    eng.synchronize()
    got = d_w.cpu().numpy()
    like = d_l[(n_frames - 1) % 4].cpu().numpy()
    # the oracle's memory registers what the device decided (ids of new words: first_new + k)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    for i, r in zip(ids, base):
        m.vwd.add_word(int(i), r)
    known = set(ids.tolist())
    for t in range(n_frames):
        w = np.where(got[t] < 0, n_words + 1 + t * q - got[t] - 1, got[t]).astype(np.int32)
        assert (w > 0).all()
        for j in np.flatnonzero(got[t] < 0).tolist():
            if int(w[j]) not in known:
                known.add(int(w[j]))
                m.vwd.add_word(int(w[j]), frames[t][j])
        assert m.add_signature_with_id(t + 1, w) == t + 1
        if t >= wm and t < n_frames - 1:
            m.forget(t + 1 - wm)
    live = np.array(m.signature_ids(), np.int32)
    oi, Lo = m.compute_likelihood(w, live)
    np.testing.assert_allclose(like[oi - 1], Lo, rtol=RTOL, atol=ATOL)
    dead = np.ones(n_frames, bool)
    dead[oi - 1] = False
    assert not like[:n_frames][dead].any()
    rows, n_live = eng.vocab_count()
    st = eng.stats()
    assert rows > n_words + 1000 and n_live < rows - 1000, "the stream must create and remove thousands of words"
    assert st["word_slots"] < 3 * n_live + 4 * q * 40, "postings keys are recycled, not leaked: %d keys for %d live words" % (st["word_slots"], n_live)
    # the live rows are referenced words (up to what the last retirements left behind): one more clean removes little
    vr, vi = eng.vocab_read(0, rows)
    live_ids = vi[vi != 0]
    assert live_ids.shape[0] == n_live and len(set(live_ids.tolist())) == n_live
    n_left, _ = eng.vocab_remove_unused(capacity=0)
    assert n_left <= 4 * q
    refs_alive = set()
    for t in range(max(0, n_frames - wm - 1), n_frames):
        refs_alive.update(np.where(got[t] < 0, n_words + 1 + t * q - got[t] - 1, got[t]).tolist())
    rows2, n_live2 = eng.vocab_count()
    vr2, vi2 = eng.vocab_read(0, rows2)
    assert set(vi2[vi2 != 0].tolist()) <= refs_alive, "a live row whose word no live signature references"
    eng.close()


def test_rows_appended_on_the_device_keep_their_postings_keys(oracle):
    """A word that a frame appends on the device holds its postings key in the row (row_wslot) whether or not anything references it: the
    batched check of superseded reservations must not hand that key to another word.  Frames WITHOUT a signature (sig_id = 0: no
    references at all) append ~70 words each for 90 frames -- more than 16 384 reserved keys, so several batched checks run -- then
    signatures are registered by word id (the host path: its id -> key table must name the row's key, not a second one) and by later
    frames that match the appended rows; the likelihood over all of them is the oracle's."""
    import rtabmap_amd
    rng = np.random.default_rng(377)
    n_words, q, n_frames = 2500, 256, 90
    base = synth.vocab_surf(n_words, seed=378)
    ids = np.arange(1, n_words + 1, dtype=np.int32)
    eng = rtabmap_amd.Engine("f32", 64, sig_capacity=64, pipeline=True)
    eng.vocab_append(base, ids)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    for i, r in zip(ids, base):
        m.vwd.add_word(int(i), r)
    m.vwd.update()
    frames = [np.ascontiguousarray(np.where((rng.random((q, 1)) < 0.3), synth.vocab_surf(q, seed=900 + t), base[rng.integers(0, n_words, q)]))
              for t in range(n_frames)]
    d_desc = [torch.from_numpy(f).cuda() for f in frames]
    d_w = torch.zeros((n_frames, q), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    firsts = []
    for t in range(n_frames):
        firsts.append(n_words + 1 + t * q)
        eng.frame_dev(d_desc[t].data_ptr(), q, 0, 1.0, d_w[t].data_ptr(), 0, 0, first_new_word_id=firsts[t], append_new_words=True)
    eng.synchronize()
    got = d_w.cpu().numpy()
    created = {}                                                     # word id -> descriptor
    per_frame_words = []
    for t in range(n_frames):
        w = np.where(got[t] < 0, firsts[t] - got[t] - 1, got[t]).astype(np.int32)
        per_frame_words.append(w)
        for j in np.flatnonzero(got[t] < 0).tolist():
            created.setdefault(int(w[j]), frames[t][j])
    assert len(created) > 4000
    rows, n_live = eng.vocab_count()
    assert rows == n_live == n_words + len(created)
    for wid in sorted(created):
        m.vwd.add_word(wid, created[wid])
    m.vwd.update()
    # signatures registered through the host path, naming appended words by id
    sigs = []
    for s in range(12):
        w = per_frame_words[(7 * s) % n_frames]
        sid = m.add_signature(w)
        eng.sig_add(sid, w)
        sigs.append(sid)
    # ... and through the device path: a revisit of an early frame matches the rows that frame appended
    cap = 64
    d_l = torch.zeros(cap, dtype=torch.float32, device="cuda")
    d_w1 = torch.zeros(q, dtype=torch.int32, device="cuda")
    # Memory::update starts with cleanUnusedWords: the words nobody references go, on both sides (rows appended on the device included)
    unused_o = sorted(m.vwd.get_unused_word_ids())
    n_gone, gone = eng.vocab_remove_unused(capacity=1 << 20)
    assert n_gone == len(unused_o) and sorted(gone.tolist()) == unused_o
    first_new = m.vwd.last_word_id + 1
    sid, exp = m.update(frames[3])
    eng.frame_dev(d_desc[3].data_ptr(), q, sid, float(m.num_signatures()), d_w1.data_ptr(), d_l.data_ptr(), cap, first_new_word_id=first_new,
                  append_new_words=True)
    eng.synchronize()
    g = d_w1.cpu().numpy()
    assert np.where(g < 0, first_new - g - 1, g).tolist() == exp
    sigs.append(sid)
    oi, Lo = m.compute_likelihood(np.array(exp, np.int32), np.array(sigs, np.int32))
    np.testing.assert_allclose(d_l[: len(sigs)].cpu().numpy(), Lo, rtol=RTOL, atol=ATOL)
    for wid in [w for w in list(created)[:: max(1, len(created) // 60)] if w not in set(unused_o)]:
        assert eng.word_nrefs(wid) == len(m.vwd.word_refs(wid))
    eng.close()
