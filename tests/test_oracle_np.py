"""CPU: the numpy restatement of the TF-IDF likelihood for dense signature matrices (oracle/tfidf_np.py, used by bench.py's parity legs
at memory sizes the std::map oracle cannot hold) against the C++ oracle's Memory::computeLikelihood, which the reference's golden vector
pins (tests/test_oracle_golden.py)."""
import numpy as np

from rtabmap_amd import synth


def test_dense_numpy_tfidf_equals_the_cpp_oracle(oracle):
    from oracle import tfidf_np
    n_words, n_sig, q = 900, 700, 60
    words = synth.zipf_words(n_sig, q, n_words, seed=4)
    words[5, :7] = 0                                                   # features without a word count in ni only
    words[9, :] = words[9, 0]                                          # one word 60 times
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8)
    base = synth.vocab_surf(n_words, seed=3)
    for w in range(1, n_words + 1):
        m.vwd.add_word(w, base[w - 1])
    for s in range(n_sig):
        assert m.add_signature_with_id(s + 1, words[s]) == s + 1
    ids = np.arange(1, n_sig + 1, dtype=np.int32)
    for t in (0, 9, 123, n_sig - 1):
        query = np.concatenate([words[t], [0, -3, words[(t + 1) % n_sig][0]]]).astype(np.int32)
        oi, Lo = m.compute_likelihood(query, ids)
        Ln = tfidf_np.compute_likelihood_dense(words, query)
        assert oi.tolist() == ids.tolist()
        # same float32 operations in the same order; log10 may differ from glibc's log10f in the last bit
        np.testing.assert_allclose(Ln, Lo, rtol=2e-6, atol=1e-9)
        assert int(np.argmax(Ln)) == int(np.argmax(Lo))
    # a prefix of the memory with an explicit N (the replay scores frame t against the t + 1 signatures that exist then)
    m2 = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8)
    for w in range(1, n_words + 1):
        m2.vwd.add_word(w, base[w - 1])
    for s in range(200):
        m2.add_signature_with_id(s + 1, words[s])
    oi, Lo = m2.compute_likelihood(words[199], ids[:200])
    np.testing.assert_allclose(tfidf_np.compute_likelihood_dense(words[:200], words[199]), Lo, rtol=2e-6, atol=1e-9)
