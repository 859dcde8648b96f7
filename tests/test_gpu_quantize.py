"""GPU parity: lcd_quantize / lcd_find_nn (the addNewWords / findNN decision loops on the device) vs the oracle's
restated VWDictionary on identical descriptor streams.  Word assignments must be identical (ORB and SURF)."""
import numpy as np
import pytest
import torch  # before liblcd_hip.so is loaded: one HIP runtime per process (rtabmap_amd/capi.py)

from rtabmap_amd import synth

pytestmark = pytest.mark.gpu


def _frames(kind, n_frames, q, base_n=400):
    if kind == "orb":
        base = synth.vocab_orb(base_n, seed=77)
        return [synth.queries_orb(base, q, seed=100 + t, frac_known=0.8, flip=0.05) for t in range(n_frames)]
    base = synth.vocab_surf(base_n, seed=78)
    return [synth.queries_surf(base, q, seed=200 + t, frac_known=0.8, sigma=0.03) for t in range(n_frames)]


@pytest.mark.parametrize("kind", ["orb", "surf"])
@pytest.mark.parametrize("together", [True, False])
def test_quantize_stream_matches_oracle(oracle, kind, together):
    """Frame by frame: the engine's vocabulary is loaded with exactly the oracle's indexed rows (same order), then
    lcd_quantize must reproduce addNewWords() -- including the same-frame new-word dependency chain."""
    import rtabmap_amd
    frames = _frames(kind, 10, 150)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=together)
    eng = rtabmap_amd.Engine("u8" if kind == "orb" else "f32", frames[0].shape[1])
    words = {}
    total_new = 0
    for t, desc in enumerate(frames):
        # what Memory::update does before addNewWords: cleanUnusedWords + update()
        m.preupdate = None
        unused = m.vwd.get_unused_word_ids()
        if unused:
            m.vwd.remove_words(unused)
            for w in unused:
                words.pop(w)
        m.vwd.update()
        index_ids = np.array(m.vwd.index_ids(), np.int32)
        eng.vocab_clear()
        if len(index_ids):
            eng.vocab_append(np.stack([words[i] for i in index_ids]), index_ids)
        last_id = m.vwd.last_word_id
        got, n_new = eng.quantize(desc, incremental=True, new_words_compared=together, nndr=0.8)
        exp = m.vwd.add_new_words(desc, t + 1)
        mapped = np.where(got < 0, last_id - got, got)          # -(k+1) -> last_id + k + 1
        assert mapped.tolist() == exp, "frame %d" % t
        assert n_new == len({e for e in exp if e > last_id})
        total_new += n_new
        for i, w in enumerate(exp):
            if w > last_id and w not in words:
                words[w] = desc[i].copy()
        if t >= 3:                                              # forget an old frame: words become unused
            for w in m.vwd.word_ids():
                m.vwd.remove_all_word_ref(w, t - 2)
    assert total_new > 0
    eng.close()


def test_quantize_first_frames_empty_dictionary(oracle):
    """Empty / 1-word dictionaries: no indexed search (VWDictionary.cpp:1015), everything hinges on same-frame words."""
    import rtabmap_amd
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (5, 32), dtype=np.uint8)
    desc = np.repeat(base, 6, axis=0)                           # exact duplicates inside one frame
    desc[7] ^= 1
    m = oracle.OracleVWDictionary(strategy=oracle.kNNBruteForce, nndr=0.8)
    eng = rtabmap_amd.Engine("u8", 32)
    got, n_new = eng.quantize(desc)
    exp = m.add_new_words(desc, 1)
    assert np.where(got < 0, -got, got).tolist() == exp
    assert n_new == max(exp)
    eng.close()


@pytest.mark.parametrize("kind", ["orb", "surf"])
def test_quantize_fixed_dictionary(oracle, kind):
    import rtabmap_amd
    if kind == "orb":
        v = synth.vocab_orb(3000); q = synth.queries_orb(v, 200); eng = rtabmap_amd.Engine("u8", 32)
    else:
        v = synth.vocab_surf(3000); q = synth.queries_surf(v, 200); eng = rtabmap_amd.Engine("f32", 64)
    ids = np.arange(1, 3001, dtype=np.int32)
    eng.vocab_append(v, ids)
    m = oracle.OracleVWDictionary(strategy=oracle.kNNBruteForce, incremental=False)
    for i, r in zip(ids, v):
        m.add_word(int(i), r)
    m.update()
    got, n_new = eng.quantize(q, incremental=False)
    assert n_new == 0 and got.tolist() == m.add_new_words(q, 1)
    eng.close()


@pytest.mark.parametrize("kind", ["orb", "surf"])
@pytest.mark.parametrize("n_extra", [0, 1, 40])
def test_find_nn_matches_oracle(oracle, kind, n_extra):
    import rtabmap_amd
    if kind == "orb":
        v = synth.vocab_orb(2000); q = synth.queries_orb(v, 120, flip=0.03)
        extra = synth.queries_orb(v, max(n_extra, 1), seed=9, frac_known=0.0)[:n_extra]
        eng = rtabmap_amd.Engine("u8", 32)
    else:
        v = synth.vocab_surf(2000); q = synth.queries_surf(v, 120, sigma=0.02)
        extra = synth.vocab_surf(max(n_extra, 1), seed=9)[:n_extra]
        eng = rtabmap_amd.Engine("f32", 64)
    if n_extra:
        q[:n_extra] = extra                                     # some queries ARE not-yet-indexed words
    ids = np.arange(1, 2001, dtype=np.int32)
    extra_ids = np.arange(5001, 5001 + n_extra, dtype=np.int32)
    eng.vocab_append(v, ids)
    m = oracle.OracleVWDictionary(strategy=oracle.kNNBruteForce, nndr=0.8)
    for i, r in zip(ids, v):
        m.add_word(int(i), r)
    m.update()
    for i, r in zip(extra_ids, extra):
        m.add_word(int(i), r)                                   # stays in _notIndexedWords
    got = eng.find_nn(q, extra, extra_ids, incremental=True, nndr=0.8)
    assert got.tolist() == m.find_nn(q)
    eng.close()


def test_quantize_large_frame_more_than_1024_descriptors(oracle):
    """Kp/MaxFeatures above the decision kernel's workgroup size: every thread resolves several descriptors."""
    import rtabmap_amd
    v = synth.vocab_surf(3000, seed=11)
    q = synth.queries_surf(v, 1500, seed=12, frac_known=0.6, sigma=0.03)
    q[700:760] = q[100:160]                                   # same-frame duplicates far apart in the frame
    ids = np.arange(1, 3001, dtype=np.int32)
    eng = rtabmap_amd.Engine("f32", 64)
    eng.vocab_append(v, ids)
    m = oracle.OracleVWDictionary(strategy=oracle.kNNBruteForce, nndr=0.8)
    for i, r in zip(ids, v):
        m.add_word(int(i), r)
    m.update()
    got, n_new = eng.quantize(q, incremental=True, new_words_compared=True, nndr=0.8)
    exp = m.add_new_words(q, 1)
    assert np.where(got < 0, 3000 - got, got).tolist() == exp
    assert n_new == len({e for e in exp if e > 3000}) > 100
    eng.close()


@pytest.mark.parametrize("mode", ["bf16", "f16", "mfma32"])
def test_frame_dev_redoes_uncertifiable_queries_inside_the_tail_launch(oracle, monkeypatch, mode):
    """lcd_frame_dev has no launch of its own for the exact redo of queries the filter certificate rejects: extra workgroups
    of the frame-tail launch do it and the decision workgroup waits for them.  A vocabulary with a run of identical rows makes
    some queries of every frame uncertifiable; the word assignment must still be the reference's, frame after frame (the
    counters the tail resets must be clean for the next frame), and identical to lcd_quantize's (stand-alone redo kernel)."""
    import rtabmap_amd
    n = 6000
    v = synth.vocab_surf(n, seed=21)
    v[3000:3040] = v[77]                                      # 41 identical rows: more equal candidates than a row block keeps
    ids = np.arange(1, n + 1, dtype=np.int32)
    eng = rtabmap_amd.Engine("f32", 64, sig_capacity=64, knn_mode=mode)
    eng.vocab_append(v, ids)
    d_words = torch.zeros(400, dtype=torch.int32, device="cuda")
    for t in range(3):
        m = oracle.OracleVWDictionary(strategy=oracle.kNNBruteForce, nndr=0.8)     # a fresh reference dictionary per frame
        for i, r in zip(ids, v):
            m.add_word(int(i), r)
        m.update()
        q = synth.queries_surf(v, 400, seed=300 + t, frac_known=0.7, sigma=0.03)
        q[5] = v[77]
        q[6] = v[77] + np.float32(1e-4)
        q[200:230] = v[77] + (np.arange(30, dtype=np.float32)[:, None] * np.float32(2e-5))
        q[390] = q[5]                                         # same-frame duplicate of an uncertifiable query
        assert eng.knn2(q)[0].shape == (400, 2) and eng.stats()["knn_last_fallback_queries"] >= 1   # the premise of the test
        d = torch.from_numpy(q).cuda()
        eng.frame_dev(d.data_ptr(), 400, 0, 10.0, d_words.data_ptr(), 0, 0, incremental=True, new_words_compared=True,
                      nndr=0.8)                               # sig_id 0, no likelihood: the dictionary and the index stay as they are
        torch.cuda.synchronize()
        got = d_words.cpu().numpy()
        exp = m.add_new_words(q, 1000 + t)
        assert np.where(got < 0, n - got, got).tolist() == exp, "frame %d" % t
        got_q, _ = eng.quantize(q, incremental=True, new_words_compared=True, nndr=0.8)
        assert got_q.tolist() == got.tolist()
        assert eng.stats()["knn_last_fallback_queries"] >= 1
    eng.close()
