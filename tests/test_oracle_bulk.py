"""CPU: the oracle's fast bulk constructor (test glue for 100k-signature memories) builds exactly the state that the
call-by-call construction through addWordRef builds -- references, counters, ni and likelihood."""
import numpy as np

from rtabmap_amd import synth


def test_bulk_constructor_equals_call_by_call(oracle):
    n_words, n_sig, q = 700, 400, 60
    words = synth.zipf_words(n_sig, q, n_words, seed=11)
    words[3, :10] = 0                      # features without a word
    words[4, :] = words[4, 0]              # one word many times
    words[5, :5] = n_words + 50            # unknown word ids are ignored by addWordRef
    mems = []
    for bulk in (False, True):
        m = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
        for w in range(1, n_words + 1):
            m.vwd.add_word(w, np.zeros(8, np.float32))
        m.add_signature(words[0])          # an existing memory in front of the bulk part
        if bulk:
            assert m.add_signatures_bulk(words[1:]) == 2
        else:
            for s in range(1, n_sig):
                m.add_signature(words[s])
        mems.append(m)
    a, b = mems
    assert a.signature_ids() == b.signature_ids()
    assert a.vwd.total_active_references == b.vwd.total_active_references
    assert a.vwd.unused_words == b.vwd.unused_words
    for w in range(1, n_words + 1):
        assert a.vwd.word_refs(w) == b.vwd.word_refs(w)
    ids = np.array(a.signature_ids(), np.int32)
    for s in (0, 4, 77):
        assert a.get_ni(s + 1) == b.get_ni(s + 1) == q
        ia, La = a.compute_likelihood(words[s], ids)
        ib, Lb = b.compute_likelihood(words[s], ids)
        assert ia.tolist() == ib.tolist()
        np.testing.assert_array_equal(La, Lb)
    # forgetting works on the bulk-built state too
    for m in mems:
        m.forget(7)
    assert a.vwd.word_refs(int(words[6, 0])) == b.vwd.word_refs(int(words[6, 0]))
