"""GPU parity of the Bayes filter (lcd_bayes_*, the d_posterior / d_bayes outputs of lcd_frame_dev) against the oracle's restatement of
BayesFilter::computePosterior (reference BayesFilter.cpp:145-235, generatePrediction :273-420, normalize :434-500) and of the
hypothesis selection (Rtabmap.cpp:2147-2158), on the same pose graph, the same adjusted likelihoods and the same changes of the
working memory (signatures leaving the short-term memory, retirements).

Tolerance: every posterior entry within 2e-5 relative (+1e-12 absolute) of the oracle's after the normalisation constants are
divided out, and the constants within m * 2^-24 of each other -- the reference adds the m unnormalised entries into a float one by
one before dividing (BayesFilter.cpp:205-230) and multiplies through cv::gemm; the device sums in double in a fixed order.  The
selected hypothesis must be the oracle's unless the oracle's best two posteriors are closer than that tolerance."""
import ctypes as C

import numpy as np
import pytest
import torch

from bayes_model import DEFAULT_LC, Graph, csr_lists, prediction_lc_as_parsed, random_adjusted, random_graph
from rtabmap_amd import synth

pytestmark = pytest.mark.gpu
RTOL, ETOL, ATOL = 1e-4, 2e-5, 1e-12


def _engine_with_signatures(n_sig, q=8, pipeline=False):
    import rtabmap_amd
    n_words = 600
    eng = rtabmap_amd.Engine("f32", 64, sig_capacity=n_sig + 64, pipeline=pipeline)
    eng.vocab_append(synth.vocab_surf(n_words, seed=3), np.arange(1, n_words + 1, dtype=np.int32))
    words = synth.zipf_words(n_sig, q, n_words, seed=4)
    eng.sig_add_bulk(np.arange(1, n_sig + 1, dtype=np.int32), np.arange(0, (n_sig + 1) * q, q, dtype=np.int64), words.reshape(-1))
    return eng


def _pick(d_post, ids):
    """posterior entries of `ids` (-1 first) from a device vector laid out [virtual place, slot 0, slot 1, ...]; slot = id - 1"""
    got = d_post.cpu().numpy()
    return np.concatenate([[got[0]], got[np.asarray(ids[1:], np.int64)]]) if len(ids) > 1 else got[:1]


def _result(d_res):
    from rtabmap_amd.capi import LcdBayesResult
    return LcdBayesResult.from_buffer_copy(d_res.cpu().numpy().tobytes())


def _check(ids, post_o, post_d, res, ctx):
    """Entry by entry within ETOL once the two normalisation constants are divided out; the constants themselves within the error
    the reference's own accumulation carries: it adds the m unnormalised entries into a float one by one (BayesFilter.cpp:205-218),
    up to m * 2^-24 relative, while the device sums in double (measured: 6e-4 at m = 50 000, all of it in the reference's sum)."""
    m = len(ids)
    pos = post_o > 0
    r = float(np.median(post_d[pos].astype(np.float64) / post_o[pos].astype(np.float64))) if pos.any() else 1.0
    assert abs(r - 1.0) <= max(m * 2.0 ** -24, 2e-6), (ctx, r)
    np.testing.assert_allclose(post_d, post_o.astype(np.float64) * r, rtol=ETOL, atol=ATOL, err_msg=str(ctx))
    gtol = max(m * 2.0 ** -24, RTOL)
    hid, hval = __import__("oracle").OracleBayesFilter.hypothesis(ids, post_o)
    assert res.n_considered == m - 1
    np.testing.assert_allclose(res.value, hval, rtol=gtol, atol=gtol)
    np.testing.assert_allclose(res.virtual_place, post_o[0], rtol=gtol, atol=ATOL)
    if hid == 0:
        assert res.sig_id == 0 and res.slot == -1
        return
    po = np.asarray(post_o[1:], np.float64)
    top = np.sort(po)[::-1]
    if len(top) > 1 and top[0] - top[1] <= 4 * ETOL * top[0]:
        assert res.sig_id in [ids[1 + k] for k in np.flatnonzero(po >= top[0] * (1 - 8 * ETOL))]
    else:
        assert res.sig_id == hid, ctx
        assert res.slot == hid - 1


@pytest.mark.parametrize("n_sig,seed", [(3000, 0), (100000, 1)])
def test_posterior_on_a_static_graph(oracle, n_sig, seed):
    """Every signature's list handed over at once (listed from both ends: the engine keeps one entry per pair); the working memory
    grows, the short-term memory slides, signatures retire."""
    rng = np.random.default_rng(seed)
    g = random_graph(n_sig, n_sig // 150, rng)
    depth = DEFAULT_LC.shape[0] - 1
    eng = _engine_with_signatures(n_sig)
    eng.bayes_configure(DEFAULT_LC, 0.9)
    ob = oracle.OracleBayesFilter(DEFAULT_LC, 0.9)
    all_ids = np.arange(1, n_sig + 1, dtype=np.int32)
    off, nbr, mg = csr_lists(g, all_ids, depth)
    eng.bayes_set_neighbors(all_ids, off, nbr, mg)
    for i, s in enumerate(all_ids.tolist()):
        ob.set_neighbors(s, nbr[off[i]:off[i + 1]], mg[off[i]:off[i + 1]])
    d_adj = torch.zeros(n_sig + 1, dtype=torch.float32, device="cuda")
    d_post = torch.zeros(n_sig + 1, dtype=torch.float32, device="cuda")
    d_res = torch.zeros(8, dtype=torch.int32, device="cuda")
    retired = set()
    n_updates = 8 if n_sig <= 3000 else 4
    for t in range(n_updates):
        exclude = [n_sig // 2, n_sig // 3, 40, 25, 25, 10, 0, 0][t] if n_sig <= 3000 else [n_sig // 2, 2000, 30, 30][t]
        if t in (2, 4):
            gone = rng.choice(np.arange(1, n_sig - n_sig // 2), size=n_sig // 40, replace=False).tolist()
            for s in gone:
                if s not in retired:
                    eng.sig_remove(int(s))
                    retired.add(int(s))
        considered = [s for s in range(1, n_sig - exclude + 1) if s not in retired]
        ids = [-1] + considered
        like = random_adjusted(len(ids), rng)
        adj = np.zeros(n_sig + 1, np.float32)
        adj[0] = like[0]
        adj[np.asarray(considered)] = like[1:]                     # entry 1 + slot, slot = id - 1
        d_adj.copy_(torch.from_numpy(adj))
        eng.bayes_update_dev(d_adj.data_ptr(), exclude, d_post.data_ptr(), d_res.data_ptr())
        eng.synchronize()
        ob.set_stm(list(range(n_sig - exclude + 1, n_sig + 1)))
        post_o = ob.compute_posterior(ids, like, dense=False)
        got = d_post.cpu().numpy()
        post_d = np.concatenate([[got[0]], got[np.asarray(considered)]])
        _check(ids, post_o, post_d, _result(d_res), ("update", t))
        mask = np.ones(n_sig + 1, bool)
        mask[0] = False
        mask[np.asarray(considered)] = False
        assert not got[mask].any(), "a signature outside the likelihood has a posterior"
        # the same values through the host accessor
        some = [-1] + [considered[k] for k in rng.integers(0, len(considered), 20)] + [n_sig]
        np.testing.assert_array_equal(eng.bayes_posterior(some), [got[0]] + [got[s] for s in some[1:-1]] + [got[n_sig] if exclude == 0 else 0.0])
    eng.close()


def test_lists_entered_one_signature_at_a_time(oracle):
    """A signature's list arrives when it enters the working memory and names only signatures that exist by then; the engine enters
    it into the older signatures' lists too (BayesFilter::updatePrediction :581-592).  Result == the filter on the complete lists."""
    rng = np.random.default_rng(7)
    n_sig = 900
    g = random_graph(n_sig, 12, rng)
    depth = DEFAULT_LC.shape[0] - 1
    eng = _engine_with_signatures(n_sig)
    eng.bayes_configure(DEFAULT_LC, 0.9)
    ob = oracle.OracleBayesFilter(DEFAULT_LC, 0.9)
    for s in range(1, n_sig + 1):
        d = g.neighbors(s, depth)
        ob.set_neighbors(s, sorted(d), [d[k] for k in sorted(d)])
    d_adj = torch.zeros(n_sig + 1, dtype=torch.float32, device="cuda")
    d_post = torch.zeros(n_sig + 1, dtype=torch.float32, device="cuda")
    d_res = torch.zeros(8, dtype=torch.int32, device="cuda")
    entered = 0
    for t, upto in enumerate([1, 2, 5, 60, 61, 300, 301, 640, 900]):
        new = np.arange(entered + 1, upto + 1, dtype=np.int32)
        off, nbr, mg = csr_lists(g, new, depth, keep=lambda k: k <= upto)
        # every pair once: a list names the signatures that entered before it (and itself)
        eng.bayes_set_neighbors(new, off, nbr, mg)
        entered = upto
        ids = [-1] + list(range(1, upto + 1))
        like = random_adjusted(len(ids), rng)
        adj = np.zeros(n_sig + 1, np.float32)
        adj[: upto + 1] = like
        d_adj.copy_(torch.from_numpy(adj))
        eng.bayes_update_dev(d_adj.data_ptr(), n_sig - upto, d_post.data_ptr(), d_res.data_ptr())
        eng.synchronize()
        ob.set_stm(list(range(upto + 1, n_sig + 1)))
        post_o = ob.compute_posterior(ids, like, dense=False)
        _check(ids, post_o, d_post[: upto + 1].cpu().numpy(), _result(d_res), ("entered", upto))
    eng.close()


@pytest.mark.parametrize("lc,vp", [([0.1, 0.3, 0.2, 0.1], 0.9), ([0.2, 0.5, 0.2, 0.05, 0.05], 0.0), ([0.1, 0.24, 0.18, 0.1, 0.04, 0.01, 0.33], 0.9)])
def test_other_prediction_patterns(oracle, lc, vp):
    """Patterns whose values sum to less than 1 (every other place gets a share, normalize :448-465), a virtual-place prior of 0
    (:394-410), and the MATLAB model's pattern: against the oracle's dense, statement-by-statement evaluation."""
    rng = np.random.default_rng(11)
    n_sig = 260
    lcp = prediction_lc_as_parsed(lc)
    g = random_graph(n_sig, 4, rng)
    depth = min(lcp.shape[0] - 1, 5)
    eng = _engine_with_signatures(n_sig)
    eng.bayes_configure(lcp, vp)
    ob = oracle.OracleBayesFilter(lcp, vp)
    all_ids = np.arange(1, n_sig + 1, dtype=np.int32)
    off, nbr, mg = csr_lists(g, all_ids, depth)
    eng.bayes_set_neighbors(all_ids, off, nbr, mg)
    for i, s in enumerate(all_ids.tolist()):
        ob.set_neighbors(s, nbr[off[i]:off[i + 1]], mg[off[i]:off[i + 1]])
    d_adj = torch.zeros(n_sig + 1, dtype=torch.float32, device="cuda")
    d_post = torch.zeros(n_sig + 1, dtype=torch.float32, device="cuda")
    d_res = torch.zeros(8, dtype=torch.int32, device="cuda")
    for t, exclude in enumerate([200, 100, 30, 30, 0]):
        upto = n_sig - exclude
        ids = [-1] + list(range(1, upto + 1))
        like = random_adjusted(len(ids), rng)
        adj = np.zeros(n_sig + 1, np.float32)
        adj[: upto + 1] = like
        d_adj.copy_(torch.from_numpy(adj))
        eng.bayes_update_dev(d_adj.data_ptr(), exclude, d_post.data_ptr(), d_res.data_ptr())
        eng.synchronize()
        ob.set_stm(list(range(upto + 1, n_sig + 1)))
        post_o = ob.compute_posterior(ids, like, dense=True)
        _check(ids, post_o, d_post[: upto + 1].cpu().numpy(), _result(d_res), (lc, t))
    eng.close()


def test_reset_and_argument_errors(oracle):
    from rtabmap_amd.capi import LcdError
    eng = _engine_with_signatures(300)
    d_adj = torch.ones(301, dtype=torch.float32, device="cuda")
    with pytest.raises(LcdError):
        eng.bayes_update_dev(d_adj.data_ptr(), 0)                       # not configured
    with pytest.raises(LcdError):
        eng.bayes_configure([0.5], 0.9)                                 # fewer than two values (BayesFilter.cpp:81-84)
    with pytest.raises(LcdError):
        eng.bayes_configure([0.1, 1.5], 0.9)                            # a value outside [0, 1] (:97-104)
    eng.bayes_configure(DEFAULT_LC, 0.9)
    with pytest.raises(LcdError):
        eng.bayes_set_neighbors([1], [0, 1], [2], [17])                 # margin beyond the pattern's levels (UASSERT :263)
    eng.bayes_set_neighbors([5000], [0, 1], [1], [0])                   # a signature the engine does not hold (one without words has no
                                                                        # slot): skipped -- it has no likelihood and no posterior
    g = Graph(300)
    ids = np.arange(1, 301, dtype=np.int32)
    off, nbr, mg = csr_lists(g, ids, DEFAULT_LC.shape[0] - 1)
    eng.bayes_set_neighbors(ids, off, nbr, mg)
    ob = oracle.OracleBayesFilter(DEFAULT_LC, 0.9)
    for i, s in enumerate(ids.tolist()):
        ob.set_neighbors(s, nbr[off[i]:off[i + 1]], mg[off[i]:off[i + 1]])
    d_post = torch.zeros(301, dtype=torch.float32, device="cuda")
    rng = np.random.default_rng(2)
    lids = [-1] + ids.tolist()
    for rnd in range(2):
        for t in range(3):
            like = random_adjusted(301, rng)
            d_adj.copy_(torch.from_numpy(like))
            eng.bayes_update_dev(d_adj.data_ptr(), 0, d_post.data_ptr(), None)
            eng.synchronize()
            np.testing.assert_allclose(d_post.cpu().numpy(), ob.compute_posterior(lids, like), rtol=RTOL, atol=ATOL)   # m = 301: 2e-5 of slack
        eng.bayes_reset()                                               # BayesFilter::reset: the next update starts from ones again
        ob.reset()
        eng.bayes_set_neighbors(ids, off, nbr, mg)
    eng.close()


@pytest.mark.parametrize("pipeline", [False, True])
def test_frame_dev_carries_the_filter(oracle, pipeline):
    """lcd_frame_dev with d_posterior / d_bayes: quantisation -> registration -> likelihood -> adjustLikelihood -> Bayes update ->
    hypothesis, frame after frame, nothing but 32 bytes leaving the device; against the oracle chain Memory::update +
    computeLikelihood + adjustLikelihood + BayesFilter::computePosterior."""
    import rtabmap_amd
    rng = np.random.default_rng(9)
    n_words, n_bulk, q, n_frames, stm = 3000, 400, 120, 40, 10
    vocab = synth.vocab_surf(n_words, seed=21)
    words = synth.zipf_words(n_bulk, q, n_words, seed=22)
    m = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True, incremental=False)
    for w in range(1, n_words + 1):
        m.vwd.add_word(w, vocab[w - 1])
    m.vwd.update()
    for s in range(n_bulk):
        assert m.add_signature(words[s]) == s + 1
    eng = rtabmap_amd.Engine("f32", 64, sig_capacity=n_bulk + n_frames + 8, pipeline=pipeline)
    eng.vocab_append(vocab, np.arange(1, n_words + 1, dtype=np.int32))
    eng.sig_add_bulk(np.arange(1, n_bulk + 1, dtype=np.int32), np.arange(0, (n_bulk + 1) * q, q, dtype=np.int64), words.reshape(-1))
    total = n_bulk + n_frames
    g = random_graph(total, 6, rng)
    depth = DEFAULT_LC.shape[0] - 1
    eng.bayes_configure(DEFAULT_LC, 0.9)
    ob = oracle.OracleBayesFilter(DEFAULT_LC, 0.9)
    ids0 = np.arange(1, n_bulk + 1, dtype=np.int32)
    off, nbr, mg = csr_lists(g, ids0, depth, keep=lambda k: k <= n_bulk)
    eng.bayes_set_neighbors(ids0, off, nbr, mg)
    for s in range(1, total + 1):
        d = g.neighbors(s, depth)
        ob.set_neighbors(s, sorted(d), [d[k] for k in sorted(d)])
    cap = total + 8
    bufs = []
    pdepth = eng.pipeline_depth()                                        # a pipelined handle writes a frame's outputs during the next `pdepth` calls
    for _ in range(pdepth + 1):
        bufs.append(dict(words=torch.zeros(q, dtype=torch.int32, device="cuda"), like=torch.zeros(cap, dtype=torch.float32, device="cuda"),
                         post=torch.zeros(cap + 1, dtype=torch.float32, device="cuda"), res=torch.zeros(8, dtype=torch.int32, device="cuda"),
                         desc=torch.zeros((q, 64), dtype=torch.float32, device="cuda")))
    expected = []
    gone, gone_before = set(), set()
    for t in range(n_frames):
        sid = n_bulk + 1 + t
        src = int(rng.integers(60, n_bulk - 50))
        desc = synth.frame_from_signature(vocab, words[src], seed=100 + t)
        b = bufs[t % len(bufs)]
        b["desc"].copy_(torch.from_numpy(desc))
        osid, exp_words = m.update(desc)
        assert osid == sid
        eng.frame_dev(b["desc"].data_ptr(), q, sid, float(m.num_signatures()), b["words"].data_ptr(), b["like"].data_ptr(), cap,
                      incremental=False, exclude_recent=stm, d_posterior_ptr=b["post"].data_ptr(), d_bayes_ptr=b["res"].data_ptr())
        # the new signature's list names what exists by now; the engine completes the older signatures' lists.  On a pipelined
        # handle the call is queued behind the frame's owed index stage (no drain)
        off, nbr, mg = csr_lists(g, [sid], depth, keep=lambda k: k <= sid)
        eng.bayes_set_neighbors([sid], off, nbr, mg)
        # oracle: likelihood over the working memory (everything but the newest `stm` signatures), adjusted, filtered
        wm = [s for s in range(1, sid - stm + 1) if s not in gone_before]
        oi, Lo = m.compute_likelihood(np.array(exp_words, np.int32), np.array(wm, np.int32))
        vec = oracle.adjust_likelihood(np.concatenate([[0.0], Lo]).astype(np.float32), 0.0)
        ob.set_stm(list(range(sid - stm + 1, sid + 1)))
        expected.append(([-1] + wm, ob.compute_posterior([-1] + wm, vec, dense=False)))
        # Memory::forget of an old signature every other frame (queued behind the owed stage on a pipelined handle): it leaves the
        # likelihood and the filter from the next frame on
        if t % 2 == 1:
            old = 3 + t
            m.forget(old)
            eng.sig_remove(old)
            gone.add(old)
        gone_before = set(gone)
        torch.cuda.synchronize()
        done = t - pdepth                                                # a pipelined frame's index stage runs inside the later calls
        if done >= 0:
            pids, ppost = expected[done]
            pb = bufs[done % len(bufs)]
            _check(pids, ppost, _pick(pb["post"], pids), _result(pb["res"]), ("frame", done))
    eng.synchronize()
    for done in range(max(n_frames - pdepth, 0), n_frames - 1):           # the frames the last calls left owed
        pids, ppost = expected[done]
        _check(pids, ppost, _pick(bufs[done % len(bufs)]["post"], pids), _result(bufs[done % len(bufs)]["res"]), ("frame", done))
    pids, ppost = expected[-1]
    pb = bufs[(n_frames - 1) % len(bufs)]
    _check(pids, ppost, _pick(pb["post"], pids), _result(pb["res"]), ("last frame",))
    st = eng.stats()
    assert st["frame_calls"] == n_frames
    eng.close()


def test_relisting_the_same_neighbours_does_not_grow_the_table():
    """A signature listed again and again (its own list starts over, its entry in each neighbour's list is REPLACED): the host's
    upper bound of the list lengths counts every replacement, the true lengths stay what they are.  9 000 re-listings used to
    double the table's width until lcd_bayes_set_neighbors failed for good ('longer than 8192 entries')."""
    eng = _engine_with_signatures(64)
    eng.bayes_configure(DEFAULT_LC, 0.9)
    sig = np.array([10], np.int32)
    off = np.array([0, 5], np.int64)
    nbr = np.array([8, 9, 10, 11, 12], np.int32)
    mg = np.array([2, 1, 0, 1, 2], np.int32)
    prep = eng.bayes_neighbors_prepared(sig, off, nbr, mg)
    eng.bayes_set_neighbors_prepared(prep)                            # (allocates the table)
    before = eng.stats()["bytes_device"]
    for _ in range(9000):
        eng.bayes_set_neighbors_prepared(prep)
    eng.synchronize()
    assert eng.stats()["bytes_device"] == before                      # the neighbour table kept its width
    eng.close()
