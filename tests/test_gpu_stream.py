"""GPU parity, end to end through the HOST MIRROR: the same descriptor stream is pushed through the oracle's restated
Memory/VWDictionary and through MemoryHip/VWDictionaryHip (C++ over the C-ABI).  Per frame: identical word ids, identical
dictionary bookkeeping, identical index (tie-break) order; likelihood within 1e-4 relative with the same arg-max."""
import os

import numpy as np
import pytest

from rtabmap_amd import synth

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-7


def _frames(kind, n_frames, q, base_n=500):
    if kind == "orb":
        base = synth.vocab_orb(base_n, seed=77)
        return [synth.queries_orb(base, q, seed=100 + t, frac_known=0.8, flip=0.05) for t in range(n_frames)]
    base = synth.vocab_surf(base_n, seed=78)
    return [synth.queries_surf(base, q, seed=200 + t, frac_known=0.8, sigma=0.03) for t in range(n_frames)]


def _same_state(o, h):
    assert h.vwd.visual_words == o.vwd.visual_words
    assert h.vwd.not_indexed_words == o.vwd.not_indexed_words
    assert h.vwd.indexed_words == o.vwd.indexed_words
    assert h.vwd.total_active_references == o.vwd.total_active_references
    assert h.vwd.unused_words == o.vwd.unused_words
    assert h.vwd.index_ids() == o.vwd.index_ids()


@pytest.mark.parametrize("kind", ["orb", "surf"])
@pytest.mark.parametrize("together", [True, False])
def test_incremental_stream(oracle, kind, together):
    from rtabmap_amd.vwdictionary import MemoryHip
    frames = _frames(kind, 14, 160)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=together)
    h = MemoryHip(nndr=0.8, new_words_compared_together=together)
    W = 6                                            # working-memory size: older frames are forgotten (words get removed)
    for t, desc in enumerate(frames):
        nq = None if t % 5 else desc.shape[0] - 7   # some frames keep features that are not quantised (ids -1,-2,..)
        so, ido = o.update(desc, nq)
        sh, idh = h.update(desc, nq)
        assert so == sh and idh == ido, "frame %d" % t
        _same_state(o, h)
        ids = np.array(o.signature_ids(), np.int32)
        oi, Lo = o.compute_likelihood(np.array(ido, np.int32), ids)
        hi, Lh = h.compute_likelihood(np.array(idh, np.int32), ids)
        assert oi.tolist() == hi.tolist()
        np.testing.assert_allclose(Lh, Lo, rtol=RTOL, atol=ATOL)
        if Lo[:-1].size and Lo[:-1].max() > 0:
            assert int(np.argmax(Lh[:-1])) == int(np.argmax(Lo[:-1]))
        if so > W:
            o.forget(so - W)
            h.forget(so - W)
            assert h.get_ni(so - W) == o.get_ni(so - W) == desc.shape[0]
    for w in o.vwd.word_ids()[::9]:
        assert h.vwd.word_refs(w) == o.vwd.word_refs(w)
    h.close()


@pytest.mark.parametrize("kind", ["orb", "surf"])
def test_find_nn_after_stream(oracle, kind):
    """enableWordsRef path: re-activated words are matched with findNN against indexed + not yet indexed words."""
    from rtabmap_amd.vwdictionary import MemoryHip
    frames = _frames(kind, 5, 120)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
    h = MemoryHip()
    for desc in frames[:4]:
        o.update(desc)
        h.update(desc)
    # after the last addNewWords the new words are not indexed yet: findNN must see them too
    assert h.vwd.not_indexed_words == o.vwd.not_indexed_words > 0
    q = frames[4].copy()
    q[:30] = frames[3][:30]                           # exact copies of descriptors that just became words
    assert h.vwd.find_nn(q) == o.vwd.find_nn(q)
    h.close()


def test_fixed_dictionary_from_text_file(oracle, tmp_path):
    """Kp/IncrementalDictionary=false + Kp/DictionaryPath (stand-in for data/Dictionary49k.txt, same text format)."""
    from rtabmap_amd.vwdictionary import MemoryHip
    vocab = synth.vocab_surf(3000)
    path = os.path.join(str(tmp_path), "Dictionary3k.txt")
    synth.write_dictionary_text(path, vocab)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce, incremental=False)
    assert o.vwd.load_fixed_text(path) == 3000
    h = MemoryHip(incremental=False, dictionary_path=path)
    assert h.vwd.visual_words == 3000 and h.vwd.indexed_words == 3000
    for t in range(3):
        q = synth.queries_surf(vocab, 200, seed=300 + t)
        so, ido = o.update(q)
        sh, idh = h.update(q)
        assert idh == ido and len(idh) == 200
    ids = np.array(o.signature_ids(), np.int32)
    oi, Lo = o.compute_likelihood(np.array(ido, np.int32), ids)
    hi, Lh = h.compute_likelihood(np.array(idh, np.int32), ids)
    np.testing.assert_allclose(Lh, Lo, rtol=RTOL, atol=ATOL)
    # round trip of the exported dictionary (exportDictionary writes %f: 6 decimals)
    out = os.path.join(str(tmp_path), "exported.txt")
    refs = os.path.join(str(tmp_path), "refs.txt")
    h.vwd.export_text(refs, out)
    o2 = os.path.join(str(tmp_path), "exported_oracle.txt")
    o.vwd.export_text(os.path.join(str(tmp_path), "refs_oracle.txt"), o2)
    assert open(out).read() == open(o2).read()
    assert open(refs).read() == open(os.path.join(str(tmp_path), "refs_oracle.txt")).read()
    h.close()


def test_error_paths_return_empty_like_the_reference(oracle):
    from rtabmap_amd.vwdictionary import VWDictionaryHip
    d = VWDictionaryHip()
    assert d.add_new_words(np.zeros((0, 64), np.float32), 1) == []            # "Descriptors size is null!"
    assert len(d.add_new_words(synth.vocab_surf(5), 1)) == 5
    assert d.add_new_words(np.zeros((3, 32), np.float32), 2) == []            # size mismatch with the dictionary
    assert d.add_new_words(np.zeros((3, 64), np.uint8), 2) == []              # type mismatch
    assert d.find_nn(np.zeros((2, 32), np.float32)) == [0, 0]
    f = VWDictionaryHip(incremental=False)
    assert f.add_new_words(synth.vocab_surf(4), 1) == []                      # fixed dictionary without words
    d.close(); f.close()
