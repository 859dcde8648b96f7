"""GPU parity across the growth paths of the device tables: a re-layout must carry what sealed buckets / older signatures left.

* word-major directory of the sparse postings (tfidf.h, dir2): more buckets than its row width AND more word slots than its rows,
  while sealed buckets already have records in it;
* Bayes filter state (bayes.h): more signature slots than allocated, with neighbour lists and a posterior in place.
Both against the oracle, like tests/test_gpu_likelihood.py and tests/test_gpu_bayes.py."""
import numpy as np
import pytest
import torch  # noqa: F401  (before liblcd_hip.so is loaded)

from bayes_model import DEFAULT_LC, Graph, csr_lists, random_adjusted
from rtabmap_amd import synth

pytestmark = pytest.mark.gpu


def test_directory_growth_keeps_sealed_buckets(oracle):
    import rtabmap_amd
    n1, n2, per, w1, w2 = 20000, 30000, 24, 60000, 140000
    eng = rtabmap_amd.Engine("f32", 64, sig_capacity=256)          # small hints: 128 buckets per directory row, 131 072 word slots
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
    for w in range(1, w2 + 1):
        m.vwd.add_word(w, np.zeros(1, np.float32))
    words1 = synth.zipf_words(n1, per, w1, seed=11, uniform=True)
    words2 = synth.zipf_words(n2, per, w2, seed=12, uniform=True)

    def add(words, first):
        ids = np.arange(first, first + words.shape[0], dtype=np.int32)
        assert m.add_signatures_bulk(words) == first
        eng.sig_add_bulk(ids, np.arange(0, (words.shape[0] + 1) * per, per, dtype=np.int64), words.reshape(-1))
        return ids

    def check(ids, sources, N):
        for k, src in enumerate(sources):
            qw = synth.query_from_signature(src, w2, seed=100 + k)
            oid, exp = m.compute_likelihood(qw, ids)
            got = eng.likelihood(qw, oid, N)
            np.testing.assert_allclose(got, exp, rtol=1e-4, atol=1e-7)
            assert exp.max() > 0 and int(np.argmax(got)) == int(np.argmax(exp))

    ids1 = add(words1, 1)
    st1 = eng.stats()
    check(ids1, [words1[5], words1[7777], words1[19999]], float(n1))
    ids2 = add(words2, n1 + 1)                                     # 196 buckets > 128, word slots > 131 072: both directions grow
    st2 = eng.stats()
    assert st2["buckets_sealed"] == (n1 + n2) // 256 > 128 and st2["word_slots"] > 131072 > st1["word_slots"]
    allids = np.concatenate([ids1, ids2])
    # places of the first load (their buckets were sealed before the re-layout) and of the second
    check(allids, [words1[5], words1[7777], words1[19999], words2[3], words2[29999]], float(n1 + n2))
    eng.close()


def test_bayes_state_growth(oracle):
    import rtabmap_amd
    n1, n2, q = 3000, 2500, 6
    n = n1 + n2
    words = synth.zipf_words(n, q, 500, seed=4)
    eng = rtabmap_amd.Engine("f32", 64, sig_capacity=64)
    g = Graph(n, loops=[(2900, 40), (4000, 1500), (5200, 2999)])
    depth = DEFAULT_LC.shape[0] - 1
    eng.bayes_configure(DEFAULT_LC, 0.9)
    ob = oracle.OracleBayesFilter(DEFAULT_LC, 0.9)
    for s in range(1, n + 1):
        d = g.neighbors(s, depth)
        ob.set_neighbors(s, sorted(d), [d[k] for k in sorted(d)])
    d_adj = torch.zeros(n + 1, dtype=torch.float32, device="cuda")
    d_post = torch.zeros(n + 1, dtype=torch.float32, device="cuda")
    rng = np.random.default_rng(3)

    def load(first, last):
        ids = np.arange(first, last + 1, dtype=np.int32)
        eng.sig_add_bulk(ids, np.arange(0, (ids.shape[0] + 1) * q, q, dtype=np.int64), words[first - 1:last].reshape(-1))
        off, nbr, mg = csr_lists(g, ids, depth, keep=lambda k: k <= last)
        eng.bayes_set_neighbors(ids, off, nbr, mg)

    def update(upto):
        ids = [-1] + list(range(1, upto + 1))
        like = random_adjusted(len(ids), rng)
        adj = np.zeros(n + 1, np.float32)
        adj[: upto + 1] = like
        d_adj.copy_(torch.from_numpy(adj))
        _, n_slots = eng.slots_dev()
        eng.bayes_update_dev(d_adj.data_ptr(), n_slots - upto, d_post.data_ptr(), None)
        eng.synchronize()
        ob.set_stm(list(range(upto + 1, n + 1)))
        exp = ob.compute_posterior(ids, like, dense=False)
        got = d_post[: upto + 1].cpu().numpy()
        r = float(np.median(got[exp > 0].astype(np.float64) / exp[exp > 0]))
        assert abs(r - 1.0) <= max(len(ids) * 2.0 ** -24, 2e-6)
        np.testing.assert_allclose(got, exp.astype(np.float64) * r, rtol=2e-5, atol=1e-12)

    load(1, n1)
    update(2000)
    update(n1)
    load(n1 + 1, n)                                                # 5 500 slots > the 4 096 allocated: lists and posterior move
    update(n1 + 10)
    update(n)
    eng.close()
