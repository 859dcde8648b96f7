"""CPU restatements of three small formulas the HIP kernels rely on, checked exhaustively / on random inputs (the kernels
themselves are checked on the GPU against the oracle; these pin the arithmetic identities their comments claim)."""
import itertools

import numpy as np


def test_to_fixed_two_exact_halves():
    """tfidf.hip::to_fixed: floor(t * 2^48) == (uint32(floor(t * 2^16)) << 32) | uint32(frac(t * 2^16) * 2^32) in float32."""
    rng = np.random.default_rng(0)
    t = np.concatenate([rng.uniform(0, 1, 200000), rng.uniform(0, 32767, 50000), 2.0 ** rng.uniform(-60, 15, 200000),
                        [0.0, 1e-45, 32767.998]]).astype(np.float32)
    bits = t.view(np.uint32)
    mant = ((bits & 0x7FFFFF) | 0x800000).astype(np.uint64)
    e = (bits >> 23).astype(np.int64)
    shift = e - (127 + 23 - 48)
    ref = np.where(e == 0, 0, np.where(shift >= 0, mant << np.clip(shift, 0, 39).astype(np.uint64),
                                       np.where(shift > -24, mant >> np.clip(-shift, 0, 63).astype(np.uint64), 0))).astype(np.uint64)
    s = (t * np.float32(65536.0)).astype(np.float32)
    fl = np.floor(s).astype(np.float32)
    rem = (s - fl).astype(np.float32)
    lo = np.floor((rem * np.float32(4294967296.0)).astype(np.float32).astype(np.float64)).astype(np.uint64)
    new = (fl.astype(np.uint64) << np.uint64(32)) | lo
    assert (new == ref).all()


def test_top3_insertion_with_min_and_two_medians():
    """knn_mfma_kernels.hip::top3_push32: from the OLD sorted triple, k0' = min(k0, k), k1' = med3(k0, k1, k),
    k2' = med3(k1, k2, k) is the sorted insertion keeping the three smallest."""
    med3 = lambda a, b, c: sorted((a, b, c))[1]
    vals = range(-3, 4)
    for k0, k1, k2 in itertools.combinations_with_replacement(vals, 3):
        for k in vals:
            got = (min(k0, k), med3(k0, k1, k), med3(k1, k2, k))
            assert list(got) == sorted((k0, k1, k2, k))[:3]


def test_third_of_two_sorted_triples():
    """knn_mfma_kernels.hip::third_of_two_triples: min(a2, b2, max(a1, b0), max(a0, b1)) is the third smallest of the six."""
    rng = np.random.default_rng(1)
    for _ in range(20000):
        a = sorted(rng.integers(0, 12, 3).tolist())
        b = sorted(rng.integers(0, 12, 3).tolist())
        got = min(a[2], b[2], max(a[1], b[0]), max(a[0], b[1]))
        assert got == sorted(a + b)[2]
