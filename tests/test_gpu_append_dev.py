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


def _stream(oracle, pipeline, n_words, q, n_frames, seed, sync_every=0, kind="surf"):
    import rtabmap_amd
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
    eng = rtabmap_amd.Engine("f32" if kind == "surf" else "u8", base.shape[1], sig_capacity=n_bulk + n_frames + 8, pipeline=pipeline)
    eng.vocab_append(base, ids)
    eng.sig_add_bulk(np.arange(1, n_bulk + 1, dtype=np.int32), np.arange(0, (n_bulk + 1) * q, q, dtype=np.int64), words.reshape(-1))
    cap = n_bulk + n_frames + 8
    d_desc = [torch.from_numpy(f).cuda() for f in frames]
    d_w = torch.zeros((n_frames, q), dtype=torch.int32, device="cuda")
    d_l = torch.zeros((n_frames, cap), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for t in range(n_frames):
        eng.frame_dev(d_desc[t].data_ptr(), q, n_bulk + 1 + t, float(n_bulk + 1 + t), d_w[t].data_ptr(), d_l[t].data_ptr(), cap,
                      first_new_word_id=first_new[t], append_new_words=True)
        if sync_every and t % sync_every == sync_every - 1:
            eng.synchronize()                     # completes the owed stages stand-alone: the appends of the drained frames included
    eng.synchronize()
    got, like = d_w.cpu().numpy(), d_l.cpu().numpy()
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
    eng.close()
    return n_created


@pytest.mark.parametrize("pipeline", [False, True])
def test_append_new_words_on_the_device(oracle, pipeline):
    assert _stream(oracle, pipeline, n_words=3000, q=96, n_frames=30, seed=11) > 200


def test_append_new_words_on_the_device_orb(oracle):
    """256-bit binary descriptors (Hamming scan, plain handle): the rows are copied as they are"""
    assert _stream(oracle, False, n_words=1500, q=96, n_frames=20, seed=31, kind="orb") > 100


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
