"""GPU parity: the device inverted index + TF-IDF likelihood (lcd_sig_add / lcd_sig_remove / lcd_likelihood) vs the
oracle's restated Memory::computeLikelihood, and vs the reference's MATLAB golden vector.

Tolerance (BASELINE.json north_star): <= 1e-4 relative on likelihood scores (abs floor 1e-7); the arg-max candidate
must be identical.  In practice the fixed-point accumulation agrees to ~1e-6."""
import numpy as np
import pytest
import torch  # before liblcd_hip.so is loaded: one HIP runtime per process (rtabmap_amd/capi.py)

from helpers import update_common_signature
from rtabmap_amd import synth

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-7


def _engine():
    import rtabmap_amd
    return rtabmap_amd.Engine("f32", 64)


def test_golden_vector_through_the_engine(golden2010):
    """archive/2010-LoopClosure/Tests/TestComputeLikelihood.m:23-27 reproduced by the HIP engine."""
    mem = golden2010["signatures"].astype(np.int64)
    dic = golden2010["dictionary"].astype(np.int64)
    row, _ = update_common_signature(mem, dic)
    mem[0] = row
    eng = _engine()
    for r in mem:
        eng.sig_add(int(r[0]), r[1:][r[1:] != 0].astype(np.int32))
    sign = mem[-1]
    L = eng.likelihood(sign[1:][sign[1:] != 0].astype(np.int32), mem[:, 0].astype(np.int32), N=float(mem.shape[0]))
    assert np.floor(L.astype(np.float64) * 1000).astype(int).tolist() == golden2010["golden_likelihood_floor1000"].tolist()
    eng.close()


def _build(oracle, n_sig, n_words, per_sig, uniform, seed=1):
    words = synth.zipf_words(n_sig, per_sig, n_words, seed=seed, uniform=uniform)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
    for w in range(1, n_words + 1):
        m.vwd.add_word(w, np.zeros(1, np.float32))
    eng = _engine()
    ids = []
    for s in range(n_sig):
        sid = m.add_signature(words[s])
        ids.append(sid)
    offsets = np.arange(0, (n_sig + 1) * per_sig, per_sig, dtype=np.int64)
    eng.sig_add_bulk(np.array(ids, np.int32), offsets, words.reshape(-1))
    return m, eng, words, np.array(ids, np.int32)


def _compare(m, eng, qwords, ids, N):
    oid, exp = m.compute_likelihood(qwords, ids)
    got = eng.likelihood(qwords, oid, N)
    np.testing.assert_allclose(got, exp, rtol=RTOL, atol=ATOL)
    if exp.max() > 0:
        assert int(np.argmax(got)) == int(np.argmax(exp))
    return got, exp


@pytest.mark.parametrize("uniform", [False, True])
@pytest.mark.parametrize("n_sig", [300, 2500])          # below one bucket / sealed buckets + an open one
def test_likelihood_matches_oracle(oracle, n_sig, uniform):
    n_words = 3000
    m, eng, words, ids = _build(oracle, n_sig, n_words, 120, uniform)
    for t in range(4):
        qw = synth.query_from_signature(words[(17 * t + 5) % n_sig], n_words, seed=t)
        got, exp = _compare(m, eng, qw, ids, float(n_sig))
        assert exp.max() > 0
    # only a subset of ids, unknown ids and the virtual place score 0 (Memory.cpp:2271-2272)
    sub = np.concatenate([[-1], ids[::7], [10 ** 6]]).astype(np.int32)
    oid, exp = m.compute_likelihood(qw, sub)
    got = eng.likelihood(qw, oid, float(n_sig))
    np.testing.assert_allclose(got, exp, rtol=RTOL, atol=ATOL)
    assert got[0] == 0 and got[-1] == 0
    # nw of a few words
    for w in (1, 2, 50, n_words):
        refs = m.vwd.word_refs(w)
        assert eng.word_nrefs(w) == len(refs)
    eng.close()


def test_likelihood_after_retiring_signatures(oracle):
    """Memory::forget -> disableWordsRef: retired signatures stop counting in nw and score 0."""
    n_sig, n_words = 2300, 2000
    m, eng, words, ids = _build(oracle, n_sig, n_words, 100, False, seed=4)
    gone = ids[5:1400:3]
    for s in gone:
        m.forget(int(s))
        eng.sig_remove(int(s))
    assert eng.sig_count()[0] == n_sig - len(gone)
    live = np.array(sorted(set(ids.tolist()) - set(gone.tolist())), np.int32)
    qw = synth.query_from_signature(words[n_sig - 10], n_words, seed=9)
    N = float(m.num_signatures())
    _compare(m, eng, qw, live, N)
    got = eng.likelihood(qw, gone.astype(np.int32), N)
    assert (got == 0).all()
    for w in (1, 3, 77):
        assert eng.word_nrefs(w) == len(m.vwd.word_refs(w))
    # retire a whole sealed bucket (slots 0..1023): its postings are dropped, results unchanged
    for s in ids[:1024]:
        if s not in set(gone.tolist()):
            m.forget(int(s)); eng.sig_remove(int(s))
    live2 = np.array([s for s in live if s > 1024], np.int32)
    _compare(m, eng, qw, live2, float(m.num_signatures()))
    eng.close()


def test_likelihood_duplicates_and_invalid_words(oracle):
    """counts (nwi), ni incl. features without a word (ids <= 0), duplicate query words, words unknown to the index."""
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
    for w in range(1, 50):
        m.vwd.add_word(w, np.zeros(1, np.float32))
    eng = _engine()
    rng = np.random.default_rng(0)
    sigs = []
    for s in range(60):
        w = rng.integers(-3, 50, 40).astype(np.int32)      # includes 0 and negative "no word" entries
        w[w == 0] = -1
        sid = m.add_signature(w)
        eng.sig_add(sid, w)                                 # ni defaults to len(w), like Signature::getWords().size()
        sigs.append(sid)
    q = np.array([5, 5, 5, 7, -1, 49, 48, 1], np.int32)
    _compare(m, eng, q, np.array(sigs, np.int32), 60.0)
    eng.close()


def test_adjust_likelihood_matches_oracle(oracle):
    rng = np.random.default_rng(1)
    eng = _engine()
    for n in (2, 10, 5000):
        L = (rng.random(n).astype(np.float32) ** 4)
        L[rng.random(n) < 0.3] = 0
        for ratio in (0.0, 1.0):
            exp = oracle.adjust_likelihood(L, ratio)
            got = eng.adjust_likelihood(L, ratio)
            np.testing.assert_allclose(got, exp, rtol=1e-4, atol=1e-6)
            d = torch.from_numpy(L.copy()).cuda()                       # device-resident variant: same kernel, no host round trip
            eng.adjust_likelihood_dev(d.data_ptr(), L.shape[0], ratio)
            eng.synchronize()
            np.testing.assert_array_equal(d.cpu().numpy(), got)
    z = eng.adjust_likelihood(np.zeros(5, np.float32))
    assert z.tolist() == [2.0, 1.0, 1.0, 1.0, 1.0]
    eng.close()
