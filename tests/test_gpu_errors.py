"""GPU: capacity growth and the error statuses of the C-ABI (the reference logs an error and returns an empty result;
the engine returns a status and never aborts)."""
import numpy as np
import pytest

from rtabmap_amd import synth

pytestmark = pytest.mark.gpu


def test_capacity_growth_of_vocabulary_and_signatures(oracle):
    import rtabmap_amd
    eng = rtabmap_amd.Engine("u8", 32, vocab_capacity=16, sig_capacity=8)
    v = synth.vocab_orb(5000, seed=2)
    ids = np.arange(1, 5001, dtype=np.int32)
    for a in range(0, 5000, 700):
        eng.vocab_append(v[a:a + 700], ids[a:a + 700])
    q = synth.queries_orb(v, 90, seed=3)
    got_ids, got_d = eng.knn2(q)
    idx, d = oracle.knn2_linear(v, q)
    np.testing.assert_array_equal(got_ids, ids[idx])
    np.testing.assert_array_equal(got_d, d)
    words = synth.zipf_words(900, 50, 5000, seed=4)
    for s in range(900):
        eng.sig_add(s + 1, words[s])
    assert eng.sig_count()[0] == 900
    L = eng.likelihood(words[123], np.arange(1, 901, dtype=np.int32), 900.0)
    assert int(np.argmax(L)) == 123
    eng.close()


def test_error_statuses():
    import rtabmap_amd
    from rtabmap_amd.capi import LcdError
    eng = rtabmap_amd.Engine("f32", 64)
    v = synth.vocab_surf(10)
    eng.vocab_append(v, np.arange(1, 11, dtype=np.int32))
    with pytest.raises(LcdError) as e:
        eng.vocab_append(v[:1], np.array([3], np.int32))          # word already present
    assert e.value.status == 4
    with pytest.raises(LcdError):
        eng.vocab_append(v[:1], np.array([0], np.int32))          # ids must be > 0
    with pytest.raises(LcdError):
        eng.vocab_remove(np.array([77], np.int32))                # unknown word
    eng.sig_add(5, np.array([1, 2, 3], np.int32))
    with pytest.raises(LcdError):
        eng.sig_add(5, np.array([1], np.int32))                   # signature registered twice
    with pytest.raises(LcdError):
        eng.sig_remove(6)                                         # unknown signature
    with pytest.raises(LcdError):
        eng.sig_add(0, np.array([1], np.int32))                   # id 0 is invalid
    import torch
    d_q = torch.zeros(4 * 64 + 4, dtype=torch.float32, device="cuda")
    d_w = torch.zeros(8, dtype=torch.int32, device="cuda")
    d_d = torch.zeros(8, dtype=torch.float32, device="cuda")
    with pytest.raises(LcdError) as e:
        eng.knn2_dev(d_q.data_ptr() + 4, 4, d_w.data_ptr(), d_d.data_ptr())   # device descriptors are read as 16-byte vectors
    assert "aligned" in str(e.value)
    with pytest.raises(LcdError) as e:
        eng.frame_dev(d_q.data_ptr() + 8, 4, 9, 1.0, d_w.data_ptr(), None, 0)
    assert "aligned" in str(e.value)
    # the handle is still usable after errors
    ids, d = eng.knn2(v[:2])
    assert ids[:, 0].tolist() == [1, 2] and (d[:, 0] == 0).all()
    # empty inputs are not errors
    assert eng.likelihood(np.zeros(0, np.int32), np.array([5], np.int32), 1.0).tolist() == [0.0]
    eng.close()
