"""Pins the CPU oracle's TF-IDF restatement (Memory::computeLikelihood, Memory.cpp:2215-2291) against the reference's
own known-answer test: archive/2010-LoopClosure/Tests/TestComputeLikelihood.m / TestUpdateCommonSignature.m.
CPU only."""
import numpy as np

from helpers import matlab_compute_likelihood, update_common_signature


def _updated(golden2010):
    mem = golden2010["signatures"].astype(np.int64)
    dic = golden2010["dictionary"].astype(np.int64)
    row, dic2 = update_common_signature(mem, dic)
    mem2 = mem.copy()
    mem2[0] = row
    return mem2, dic2


def test_update_common_signature_matches_golden(golden2010):
    mem2, _ = _updated(golden2010)
    common = np.sort(mem2[0, 1:][mem2[0, 1:] != 0])
    assert common.tolist() == golden2010["golden_common_words"].tolist()


def test_query_row_matches_golden(golden2010):
    mem2, _ = _updated(golden2010)
    q = golden2010["query_row"]          # MATLAB's dlmread yields one more (zero) column than our parser
    w = mem2.shape[1]
    assert mem2[-1].tolist() == q[:w].tolist() and not q[w:].any()


def test_matlab_model_reproduces_golden(golden2010):
    """The numpy restatement of the MATLAB model itself reproduces the stored vector (fixture sanity)."""
    mem2, dic2 = _updated(golden2010)
    L = matlab_compute_likelihood(mem2[-1], mem2, dic2)
    assert np.floor(L * 1000).astype(int).tolist() == golden2010["golden_likelihood_floor1000"].tolist()


def _build_memory(oracle, mem2):
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
    words = np.unique(mem2[:, 1:])
    for w in words[words > 0]:
        m.vwd.add_word(int(w), np.zeros(1, np.float32))
    for row in mem2:
        m.add_signature_with_id(int(row[0]), row[1:][row[1:] != 0].astype(np.int32))
    return m


def test_oracle_compute_likelihood_matches_golden(oracle, golden2010):
    mem2, dic2 = _updated(golden2010)
    m = _build_memory(oracle, mem2)
    assert m.num_signatures() == 83
    # the references rebuilt through addWordRef equal the (updated) dictionary fixture
    for r in dic2[::97]:
        refs = m.vwd.word_refs(int(r[0]))
        vals, cnts = np.unique(r[1:][r[1:] != 0], return_counts=True)
        assert refs == dict(zip(vals.tolist(), cnts.tolist()))
    sign = mem2[-1]
    ids = mem2[:, 0].astype(np.int32)
    out_ids, L = m.compute_likelihood(sign[1:][sign[1:] != 0].astype(np.int32), ids)
    assert out_ids.tolist() == sorted(ids.tolist())        # std::map order: -1 first
    golden = golden2010["golden_likelihood_floor1000"]
    # the fixture rows are already in ascending id order (-1, 1..82)
    assert ids.tolist() == sorted(ids.tolist())
    got = np.floor(L.astype(np.float64) * 1000).astype(int)
    assert got.tolist() == golden.tolist()
    # and agrees with the float64 MATLAB model to float32 accuracy
    Lm = matlab_compute_likelihood(mem2[-1], mem2, dic2)
    np.testing.assert_allclose(L, Lm, rtol=2e-6, atol=1e-7)


def test_oracle_likelihood_only_scores_requested_ids(oracle, golden2010):
    """Memory.cpp:2271-2272: only signatures present in `ids` receive a score; N still counts every signature."""
    mem2, _ = _updated(golden2010)
    m = _build_memory(oracle, mem2)
    sign = mem2[-1]
    words = sign[1:][sign[1:] != 0].astype(np.int32)
    all_ids = mem2[:, 0].astype(np.int32)
    _, full = m.compute_likelihood(words, all_ids)
    sub = all_ids[::3]
    sub_ids, part = m.compute_likelihood(words, sub)
    assert sub_ids.tolist() == sub.tolist()
    np.testing.assert_array_equal(part, full[::3])


def test_adjust_likelihood_formula(oracle):
    """Rtabmap.cpp:5691-5760 vs the closed form (mean/sample-stddev over positive entries after the virtual place)."""
    L = np.array([0.0, 0.3, 0.4, 0.2, 0.9, 0.0], np.float32)
    out = oracle.adjust_likelihood(L)
    vals = L[1:][L[1:] > 0]
    mean = np.float32(vals.sum(dtype=np.float32) / np.float32(len(vals)))
    std = np.float32(np.sqrt(np.float32(((vals - mean) ** 2).sum(dtype=np.float32) / np.float32(len(vals) - 1))))
    exp = np.ones_like(L)
    for i in range(1, len(L)):
        if L[i] > mean + std:
            exp[i] = (L[i] - (std - np.float32(1e-4))) / mean
    exp[0] = mean / std + 1.0
    np.testing.assert_allclose(out, exp, rtol=1e-6)
    # all zeros -> every entry 1, virtual place 2 (TestAdjustLikelihood.m first case)
    z = oracle.adjust_likelihood(np.zeros(5, np.float32))
    assert z.tolist() == [2.0, 1.0, 1.0, 1.0, 1.0]


def test_log10_overload_does_not_matter_at_the_parity_bound(oracle, golden2010):
    """Memory.cpp:2266 calls the unqualified log10 on a float ratio: log10f with a current standard library (the float overload
    is in the global namespace), log10(double) rounded to float with an old one.  Both readings of the restated computeLikelihood
    reproduce the reference's golden vector, and on a 3 000-signature Zipf memory they differ by far less than the 1e-4 relative
    bound the device is held to -- so the choice (log10f, also what the device computes) cannot decide a parity test."""
    from rtabmap_amd import synth
    mem2, _ = _updated(golden2010)
    m = _build_memory(oracle, mem2)
    sign = mem2[-1]
    words = sign[1:][sign[1:] != 0].astype(np.int32)
    ids = mem2[:, 0].astype(np.int32)
    try:
        _, Lf = m.compute_likelihood(words, ids)
        oracle.set_log10_double(True)
        _, Ld = m.compute_likelihood(words, ids)
        assert np.floor(Ld * 1000).astype(int).tolist() == np.floor(Lf * 1000).astype(int).tolist()
        np.testing.assert_allclose(Ld, Lf, rtol=2e-6, atol=1e-9)
        oracle.set_log10_double(False)
        big = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
        vocab = synth.vocab_surf(2000, seed=3)
        for w in range(1, 2001):
            big.vwd.add_word(w, vocab[w - 1])
        big.vwd.update()
        sw = synth.zipf_words(3000, 200, 2000, seed=4)
        big.add_signatures_bulk(sw)
        all_ids = np.arange(1, 3001, dtype=np.int32)
        _, a = big.compute_likelihood(sw[5], all_ids)
        oracle.set_log10_double(True)
        _, b = big.compute_likelihood(sw[5], all_ids)
        rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-7)
        assert rel.max() < 5e-6 and int(np.argmax(a)) == int(np.argmax(b))
    finally:
        oracle.set_log10_double(False)
