"""GPU parity, randomised: many small random configurations of the 2-NN (all three f32 modes + Hamming) and of the quantiser
against the oracle -- sizes around the tile / workgroup / query-chunk boundaries of the kernels, tombstones, duplicates.
LCD_FUZZ_ITERS raises the number of cases (default: a quick pass)."""
import os

import numpy as np
import pytest
import torch  # noqa: F401  before liblcd_hip.so is loaded: one HIP runtime per process (rtabmap_amd/capi.py)

from rtabmap_amd import synth

pytestmark = pytest.mark.gpu
ITERS = int(os.environ.get("LCD_FUZZ_ITERS", "12"))
EDGE_N = [1, 2, 3, 31, 32, 33, 255, 256, 257, 511, 512, 513, 1023, 1025, 2047, 2049, 6143, 6145, 8191, 8193]
EDGE_Q = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 511, 512, 513, 640]


def _pick(rng, edges, lo, hi):
    return int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(lo, hi))


@pytest.mark.parametrize("mode", ["bf16", "f16", "mfma32", "valu"])
def test_fuzz_knn2_f32(oracle, monkeypatch, mode):
    import rtabmap_amd
    rng = np.random.default_rng({"bf16": 1, "mfma32": 2, "valu": 3, "f16": 5}[mode])
    for it in range(ITERS):
        n, q = _pick(rng, EDGE_N, 1, 9000), _pick(rng, EDGE_Q, 1, 700)
        v = synth.vocab_surf(n, seed=1000 + it)
        qs = synth.queries_surf(v, q, seed=2000 + it, frac_known=float(rng.random()))
        if n > 8 and rng.random() < 0.5:                       # duplicates: ties must go to the lower row
            v[rng.integers(0, n, 4)] = v[rng.integers(0, n)]
            qs[rng.integers(0, q)] = v[rng.integers(0, n)]
        ids = rng.permutation(np.arange(1, n + 1)).astype(np.int32) if rng.random() < 0.3 else np.arange(1, n + 1, dtype=np.int32)
        eng = rtabmap_amd.Engine("f32", 64, knn_mode=mode)
        eng.vocab_append(v, ids)
        removed = None
        if n > 4 and rng.random() < 0.5:                       # tombstones
            dead = rng.choice(n, size=int(rng.integers(1, max(2, n // 3))), replace=False)
            eng.vocab_remove(ids[dead])
            removed = np.zeros(n, np.uint8); removed[dead] = 1
        got_ids, got_d = eng.knn2(qs)
        idx, d = oracle.knn2_linear(v, qs, removed=removed)
        exp_ids = np.where(idx >= 0, ids[np.maximum(idx, 0)], 0).astype(np.int32)
        np.testing.assert_array_equal(got_ids, exp_ids, err_msg="case %d: n=%d q=%d" % (it, n, q))
        np.testing.assert_array_equal(got_d, d, err_msg="case %d: n=%d q=%d" % (it, n, q))
        if mode != "valu":
            assert eng.stats()["knn_max_err_ratio"] < 0.5
        eng.close()


def test_fuzz_knn2_hamming(oracle):
    import rtabmap_amd
    rng = np.random.default_rng(4)
    for it in range(ITERS):
        n, q = _pick(rng, EDGE_N, 1, 9000), _pick(rng, EDGE_Q, 1, 700)
        nbytes = int(rng.choice([16, 32, 32, 32, 61, 64]))
        v = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
        qs = v[rng.integers(0, n, q)].copy()
        flip = rng.random((q, nbytes)) < 0.1
        qs[flip] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
        ids = np.arange(1, n + 1, dtype=np.int32)
        eng = rtabmap_amd.Engine("u8", nbytes)
        eng.vocab_append(v, ids)
        got_ids, got_d = eng.knn2(qs)
        # the engine replaces the brute-force strategies: cv::NORM_HAMMING over every byte (rtflann's functor drops size % 8 bytes)
        idx, d = oracle.knn2_linear(v, qs, metric=oracle.METRIC_HAMMING_CV)
        msg = "case %d: n=%d q=%d nbytes=%d" % (it, n, q, nbytes)
        np.testing.assert_array_equal(got_d, d, err_msg=msg)
        np.testing.assert_array_equal(got_ids, np.where(idx >= 0, ids[np.maximum(idx, 0)], 0).astype(np.int32), err_msg=msg)
        eng.close()


@pytest.mark.parametrize("mode", ["bf16", "f16", "valu"])
def test_fuzz_quantize_and_frame(oracle, monkeypatch, mode):
    """lcd_quantize and the fused lcd_frame_dev tail against the restated addNewWords on random frames."""
    import rtabmap_amd
    rng = np.random.default_rng({"bf16": 7, "valu": 8, "f16": 9}[mode])
    for it in range(max(4, ITERS // 2)):
        n, q = _pick(rng, [2, 3, 255, 256, 257, 600, 2500], 2, 3000), _pick(rng, [1, 2, 63, 64, 65, 300, 512, 513], 1, 600)
        v = synth.vocab_surf(n, seed=3000 + it)
        qs = synth.queries_surf(v, q, seed=4000 + it, frac_known=float(rng.uniform(0.2, 1.0)), sigma=float(rng.uniform(0.005, 0.06)))
        if q > 10:
            qs[q // 2:q // 2 + 3] = qs[0:3]                       # same-frame duplicates
        together = bool(rng.random() < 0.7)
        nndr = float(rng.choice([0.6, 0.8, 0.95]))
        ids = np.arange(1, n + 1, dtype=np.int32)
        eng = rtabmap_amd.Engine("f32", 64, sig_capacity=64, knn_mode=mode)
        eng.vocab_append(v, ids)
        m = oracle.OracleVWDictionary(strategy=oracle.kNNBruteForce, nndr=nndr, new_words_compared_together=together)
        for i, r in zip(ids, v):
            m.add_word(int(i), r)
        m.update()
        got, n_new = eng.quantize(qs, incremental=True, new_words_compared=together, nndr=nndr)
        d = torch.from_numpy(qs).cuda()
        d_words = torch.zeros(q, dtype=torch.int32, device="cuda")
        eng.frame_dev(d.data_ptr(), q, 0, 10.0, d_words.data_ptr(), 0, 0, incremental=True, new_words_compared=together, nndr=nndr)
        torch.cuda.synchronize()
        exp = m.add_new_words(qs, 1)
        msg = "case %d: n=%d q=%d together=%s nndr=%g" % (it, n, q, together, nndr)
        assert np.where(got < 0, n - got, got).tolist() == exp, msg
        assert d_words.cpu().numpy().tolist() == got.tolist(), msg
        assert n_new == len({e for e in exp if e > n}), msg
        eng.close()
