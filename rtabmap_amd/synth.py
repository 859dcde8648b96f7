"""Deterministic synthetic inputs for the loop-closure hot path.

numpy only; used by tests/ and bench.py.  The reference's data/Dictionary49k.txt is a missing large blob
(/root/reference/.MISSING_LARGE_BLOBS), so Vocab-SURF(N) written in the reference's dictionary text format
(VWDictionary.cpp:1655,1680-1688) is its stand-in.
"""
import numpy as np


def vocab_surf(n, seed=49000, dim=64):
    """Unit-norm SURF-like rows: per 4-tuple (sum dx, sum dy, sum|dx|, sum|dy|) the last two are >= |first two|."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, dim), dtype=np.float32)
    v4 = v.reshape(n, dim // 4, 4)
    v4[:, :, 2] = np.abs(v4[:, :, 2]) + np.abs(v4[:, :, 0])
    v4[:, :, 3] = np.abs(v4[:, :, 3]) + np.abs(v4[:, :, 1])
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return np.ascontiguousarray(v, dtype=np.float32)


def queries_surf(vocab, q, seed=500, frac_known=0.7, sigma=0.05):
    """70 % noisy copies of vocabulary rows (NNDR accepts), 30 % fresh draws (NNDR rejects)."""
    rng = np.random.default_rng(seed)
    n, dim = vocab.shape
    out = np.empty((q, dim), np.float32)
    known = rng.random(q) < frac_known
    src = rng.integers(0, n, q)
    noise = rng.standard_normal((q, dim), dtype=np.float32) * np.float32(sigma)
    fresh = vocab_surf(q, seed=seed + 7919, dim=dim)
    out[known] = vocab[src[known]] + noise[known]
    out[~known] = fresh[~known]
    out /= np.linalg.norm(out, axis=1, keepdims=True)
    return np.ascontiguousarray(out, dtype=np.float32)


def vocab_orb(n, seed=200000, nbytes=32):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, nbytes), dtype=np.uint8)


def queries_orb(vocab, q, seed=501, frac_known=0.7, flip=0.1):
    rng = np.random.default_rng(seed)
    n, nb = vocab.shape
    known = rng.random(q) < frac_known
    src = rng.integers(0, n, q)
    out = rng.integers(0, 256, (q, nb), dtype=np.uint8)
    flips = np.packbits(rng.random((q, nb * 8)) < flip, axis=1)
    out[known] = vocab[src[known]] ^ flips[known]
    return np.ascontiguousarray(out)


def zipf_words(n_sig, words_per_sig, n_words, seed=100000, s=1.0, uniform=False):
    """Word ids (1..n_words) of n_sig signatures: Zipf(s) over the vocabulary (heavy-tailed like real BoW) or uniform."""
    rng = np.random.default_rng(seed)
    if uniform:
        return rng.integers(1, n_words + 1, (n_sig, words_per_sig), dtype=np.int32)
    ranks = np.arange(1, n_words + 1, dtype=np.float64)
    p = ranks ** (-s)
    cdf = np.cumsum(p / p.sum())
    u = rng.random((n_sig, words_per_sig))
    ids = np.searchsorted(cdf, u).astype(np.int32) + 1
    # decouple frequency rank from word id (ids are creation order in the reference, not popularity order)
    perm = rng.permutation(n_words).astype(np.int32) + 1
    return perm[np.minimum(ids, n_words) - 1]


def query_from_signature(sig_words, n_words, seed, resample=0.3):
    """A query frame = the words of an earlier signature with 30 % resampled (the expected top candidate is known)."""
    rng = np.random.default_rng(seed)
    out = sig_words.copy()
    m = rng.random(out.shape[0]) < resample
    out[m] = rng.integers(1, n_words + 1, int(m.sum()), dtype=np.int32)
    return out


def frame_from_signature(vocab, sig_words, seed, resample=0.3, sigma=0.05):
    """Descriptors of a revisit of an earlier place: descriptor i is a noisy copy of the vocabulary row of the i-th word
    of that signature (quantisation maps it back to the word), except `resample` of them which are fresh descriptors
    (NNDR rejects them: would-be new words).  vocab row r holds word id r + 1.  Works for float and binary vocabularies."""
    rng = np.random.default_rng(seed)
    q = sig_words.shape[0]
    fresh = rng.random(q) < resample
    rows = vocab[np.clip(sig_words, 1, vocab.shape[0]) - 1]
    if vocab.dtype == np.uint8:
        out = rows ^ np.packbits(rng.random((q, vocab.shape[1] * 8)) < sigma, axis=1)
        out[fresh] = rng.integers(0, 256, (int(fresh.sum()), vocab.shape[1]), dtype=np.uint8)
        return np.ascontiguousarray(out)
    out = rows + rng.standard_normal(rows.shape, dtype=np.float32) * np.float32(sigma)
    out[fresh] = vocab_surf(int(fresh.sum()), seed=seed + 104729, dim=vocab.shape[1]) if fresh.any() else out[fresh]
    out /= np.linalg.norm(out, axis=1, keepdims=True)
    return np.ascontiguousarray(out, dtype=np.float32)


def write_dictionary_text(path, vocab, first_id=1):
    """The reference's dictionary text format (VWDictionary.cpp:1655,1680-1688): '%d ' then '%f ' per value."""
    with open(path, "w") as f:
        f.write("WordID Descriptors...%d\n" % vocab.shape[1])
        for i, row in enumerate(vocab):
            f.write("%d " % (first_id + i))
            f.write("".join("%f " % x for x in row))
            f.write("\n")
