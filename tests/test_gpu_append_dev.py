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


def _stream(oracle, pipeline, n_words, q, n_frames, seed, sync_every=0):
    import rtabmap_amd
    rng = np.random.default_rng(seed)
    base = synth.vocab_surf(n_words, seed=seed + 1)
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
        desc = _revisit(rng, "surf", history, base, q, fresh_frac=0.3)
        history.append(desc)
        first_new.append(m.vwd.last_word_id + 1)
        sid, exp = m.update(desc)
        assert sid == n_bulk + 1 + t
        frames.append(desc)
        expected.append(exp)
        live = np.array(m.signature_ids(), np.int32)
        likes.append(m.compute_likelihood(np.array(exp, np.int32), live)[1])
    assert not m.vwd.get_unused_word_ids()
    eng = rtabmap_amd.Engine("f32", 64, sig_capacity=n_bulk + n_frames + 8, pipeline=pipeline)
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


def test_append_new_words_pipelined_with_drains_in_between(oracle):
    _stream(oracle, True, n_words=2600, q=120, n_frames=24, seed=23, sync_every=5)


def test_append_new_words_persistent_filter_and_growth(oracle):
    """72 000 words x 700 descriptors: the persistent filter launch (rows clamped to the device count), vocabulary buffers that grow
    while frames are in flight"""
    _stream(oracle, True, n_words=72000, q=700, n_frames=8, seed=5)
