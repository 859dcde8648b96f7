"""oracle/tfidf_np.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of Memory::computeLikelihood's TF-IDF branch (reference corelib/src/Memory.cpp:2215-2291) for memories too large for
the std::map containers of oracle/lcd_oracle.cpp (a million signatures of 500 words are 500 M map nodes): the signatures are a dense
[n_sig x q] int32 matrix of word ids (row s = signature id s + 1, ids <= 0 = features without a word: they only count in ni,
Memory.cpp:4955-4968), every signature is live.  Same arithmetic in the same order:

    for every unique word id w > 0 of the query, ASCENDING (uUniqueKeys, :2249):   nw = number of signatures that hold w (:2262)
        logNnw = log10f(N / nw) as float32 (:2266), skipped when 0 (:2267)
        for every signature s that holds w:   L[s] += (nwi * logNnw) / ni   -- three float32 operations, accumulated in float32 (:2275-2279)

Pinned against the C++ oracle (which is pinned against the reference's golden vector) by tests/test_oracle_np.py.  Only bench.py's
parity legs and tests/ use it."""
import numpy as np


def compute_likelihood_dense(sig_words, query_words, N=None, chunk=1 << 22):
    """sig_words: [n_sig, q] int32; query_words: the query signature's word ids.  Returns float32 L[n_sig] (signature s + 1 at index s)."""
    sig_words = np.asarray(sig_words)
    n_sig, q = sig_words.shape
    qw = np.unique(np.asarray(query_words, dtype=np.int64))
    qw = qw[qw > 0]                                                   # "if(*i>0)" (:2252)
    L = np.zeros(n_sig, np.float32)
    if n_sig == 0 or qw.size == 0:
        return L
    Nf = np.float32(n_sig if N is None else N)
    ni = np.float32(q)                                                # getNi: every feature of the signature counts
    top = int(max(int(qw.max()), int(sig_words.max()))) + 1
    lut = np.zeros(top + 1, np.int32)
    lut[qw] = np.arange(1, qw.size + 1, dtype=np.int32)               # ascending word id <-> ascending index
    flat = sig_words.reshape(-1)
    sig_parts, w_parts = [], []
    for a in range(0, flat.shape[0], chunk):                          # (bounded temporaries: the matrix may hold 5e8 entries)
        blk = flat[a:a + chunk]
        hit = lut[np.clip(blk, 0, top)]
        idx = np.flatnonzero(hit)
        if idx.size:
            sig_parts.append(((idx + a) // q).astype(np.int32))
            w_parts.append(hit[idx] - 1)
    if not sig_parts:
        return L
    sig = np.concatenate(sig_parts)
    wi = np.concatenate(w_parts)
    order = np.argsort(wi, kind="stable")                             # word-major, signatures ascending inside a word (std::map order)
    sig, wi = sig[order], wi[order]
    bounds = np.searchsorted(wi, np.arange(qw.size + 1))
    for k in range(qw.size):                                          # ascending word id: the accumulation order of the reference
        s = sig[bounds[k]:bounds[k + 1]]
        if s.size == 0:
            continue
        us, cnt = np.unique(s, return_counts=True)                    # refs(w): signature -> nwi
        nw = np.float32(us.size)
        logNnw = np.log10(Nf / nw, dtype=np.float32)
        if logNnw == 0:
            continue
        L[us] += (cnt.astype(np.float32) * logNnw) / ni
    return L
