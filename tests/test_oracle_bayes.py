"""The oracle's BayesFilter restatement (oracle/lcd_oracle.cpp) against the reference's own known-answer table
(archive/2010-LoopClosure/Tests/TestBayesFilter.m:32) and against itself (dense literal evaluation vs sparse evaluation)."""
import numpy as np
import pytest

import oracle as O
from bayes_model import DEFAULT_LC, DeviceModel, Graph, csr_lists, prediction_lc_as_parsed, random_adjusted, random_graph

# TestBayesFilter.m:32: floor(posterior * 1000) after each of 10 updates with likelihood = 1, predictionNP = 0.9,
# predictionLC = [0.1 0.24 0.18 0.18 0.1 0.1 0.04 0.04 0.01 0.01] (MATLAB format: one value per side and level)
GOLDEN = np.array([
    [1000, 0, 0, 0, 0, 0, 0, 0, 0, 0], [900, 99, 0, 0, 0, 0, 0, 0, 0, 0], [820, 117, 62, 0, 0, 0, 0, 0, 0, 0],
    [756, 111, 82, 50, 0, 0, 0, 0, 0, 0], [704, 103, 84, 67, 40, 0, 0, 0, 0, 0], [663, 96, 82, 69, 54, 32, 0, 0, 0, 0],
    [631, 90, 79, 69, 58, 44, 26, 0, 0, 0], [604, 84, 76, 68, 58, 48, 36, 21, 0, 0], [583, 79, 73, 66, 58, 49, 40, 30, 17, 0],
    [567, 74, 69, 64, 58, 50, 41, 33, 25, 14]])


@pytest.mark.parametrize("dense", [True, False])
def test_matlab_known_answers(dense):
    # the C++ prediction format has one value per level; a sixth value that no list ever reaches (the harness answers
    # getNeighborsId to depth 4, as the MATLAB model does) brings the total to 1 like the MATLAB pattern's sum
    lc = prediction_lc_as_parsed([0.1, 0.24, 0.18, 0.1, 0.04, 0.01, 0.33])
    b = O.OracleBayesFilter(lc, 0.9)
    for it in range(1, 11):
        places = list(range(1, it))
        g = Graph(max(it - 1, 1))
        for s in places:
            d = {k: v for k, v in g.neighbors(s, 5).items() if k < it}
            b.set_neighbors(s, sorted(d), [d[k] for k in sorted(d)])
        ids = [-1] + places
        post = b.compute_posterior(ids, np.ones(len(ids), np.float32), dense=dense)
        got = np.zeros(10)
        got[:len(ids)] = np.floor(post.astype(np.float64) * 1000.0)
        # the table was produced in double precision; the reference's C++ (and this restatement) computes in float:
        # a value that sits on an integer boundary (0.1 * 1000) may fall on either side
        assert np.all(np.abs(got - GOLDEN[it - 1]) <= 1), (it, got, GOLDEN[it - 1])
        assert abs(float(post.sum()) - 1.0) < 1e-5


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_dense_and_sparse_evaluations_agree(seed):
    rng = np.random.default_rng(seed)
    n = 120
    g = random_graph(n, 8, rng)
    depth = DEFAULT_LC.shape[0] - 1
    bd, bs = O.OracleBayesFilter(DEFAULT_LC), O.OracleBayesFilter(DEFAULT_LC)
    for s in range(1, n + 1):
        d = g.neighbors(s, depth)
        for b in (bd, bs):
            b.set_neighbors(s, sorted(d), [d[k] for k in sorted(d)])
    stm = 5
    for t in range(10, n + 1, 7):
        wm = [s for s in range(1, t - stm + 1) if (s * 7 + seed) % 11 != 0]      # some signatures were transferred out
        ids = [-1] + wm
        for b in (bd, bs):
            b.set_stm(list(range(t - stm + 1, t + 1)))
        like = random_adjusted(len(ids), rng)
        pd = bd.compute_posterior(ids, like, dense=True)
        ps = bs.compute_posterior(ids, like, dense=False)
        assert np.array_equal(pd, ps)
        assert abs(float(pd.sum()) - 1.0) < 1e-4
        hid, hval = O.OracleBayesFilter.hypothesis(ids, pd)
        assert hid == ids[1 + int(np.argmax(pd[1:] + np.arange(len(wm)) * 1e-12))] or pd[1:].max() == 0
        assert abs(hval - (1.0 - pd[0])) < 1e-6


def test_fill_of_all_other_places_dense():
    # a prediction whose values sum to less than 1 spreads the rest over every other place (normalize :448-465): dense evaluation only
    lc = prediction_lc_as_parsed([0.1, 0.3, 0.2, 0.1])
    b = O.OracleBayesFilter(lc, 0.9)
    g = Graph(30)
    for s in range(1, 31):
        d = g.neighbors(s, 3)
        b.set_neighbors(s, sorted(d), [d[k] for k in sorted(d)])
    ids = [-1] + list(range(1, 31))
    rng = np.random.default_rng(5)
    for _ in range(3):
        post = b.compute_posterior(ids, random_adjusted(len(ids), rng), dense=True)
        assert abs(float(post.sum()) - 1.0) < 1e-5 and np.all(post > 0)
    with pytest.raises(RuntimeError):
        O.OracleBayesFilter(lc, 0.9).compute_posterior(ids, np.ones(len(ids), np.float32), dense=False)


@pytest.mark.parametrize("lc,vp,dense,depth_cap", [(None, 0.9, False, 99), ([0.1, 0.3, 0.2, 0.1], 0.9, True, 99), ([0.2, 0.5, 0.2, 0.05, 0.05], 0.0, True, 99),
                                                  ([0.1, 0.24, 0.18, 0.1, 0.04, 0.01, 0.33], 0.9, True, 5)])
def test_device_algorithm_model_against_the_oracle(lc, vp, dense, depth_cap):
    """bayes.hip's evaluation order (columns -> rows gathered from symmetric lists -> normalise), modelled in numpy, gives the
    reference's posterior within float rounding for every prediction pattern -- including the ones that fill all other places."""
    lcp = DEFAULT_LC if lc is None else prediction_lc_as_parsed(lc)
    rng = np.random.default_rng(0)
    n_sig = 300 if lc is None else 160
    g = random_graph(n_sig, 6, rng)
    depth = min(lcp.shape[0] - 1, depth_cap)
    ob, dv = O.OracleBayesFilter(lcp, vp), DeviceModel(n_sig, lcp, vp)
    for s in range(1, n_sig + 1):
        d = g.neighbors(s, depth)
        ob.set_neighbors(s, sorted(d), [d[k] for k in sorted(d)])
        for k, m in d.items():
            dv.link(s - 1, k - 1, m)
    for exclude in [n_sig // 2, n_sig // 3, 20, 20, 0, 0]:
        upto = n_sig - exclude
        ids = [-1] + list(range(1, upto + 1))
        like = random_adjusted(len(ids), rng)
        adj = np.zeros(n_sig + 1, np.float32)
        adj[: upto + 1] = like
        inset = np.zeros(n_sig, bool)
        inset[:upto] = True
        ob.set_stm(list(range(upto + 1, n_sig + 1)))
        po = ob.compute_posterior(ids, like, dense=dense)
        pd = dv.update(adj, inset)[: upto + 1]
        np.testing.assert_allclose(pd, po, rtol=2e-5, atol=1e-12)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_incremental_prediction_update_equals_full_regeneration(seed):
    """Bayes/FullPredictionUpdate = false (the reference's default: updatePrediction patches last call's matrix from the cached
    neighbour maps, BayesFilter.cpp:502-706) against a full regeneration on every call -- what the device does -- while signatures
    enter the working memory, leave the short-term memory and are transferred out.  With the default pattern the posteriors are
    equal bit for bit: evaluating the prediction from the neighbour lists every frame IS the reference's default behaviour."""
    rng = np.random.default_rng(seed)
    n = 150
    g = random_graph(n, 7, rng)
    depth = DEFAULT_LC.shape[0] - 1
    full, inc = O.OracleBayesFilter(DEFAULT_LC), O.OracleBayesFilter(DEFAULT_LC)
    for s in range(1, n + 1):
        d = g.neighbors(s, depth)
        for b in (full, inc):
            b.set_neighbors(s, sorted(d), [d[k] for k in sorted(d)])
    stm, gone = 6, set()
    for t in range(12, n + 1, 3):
        if t % 4 == 0:
            gone.add(int(rng.integers(1, max(t - stm - 5, 2))))               # Memory::forget / transfer of an old signature
        wm = [s for s in range(1, t - stm + 1) if s not in gone]
        ids = [-1] + wm
        for b in (full, inc):
            b.set_stm(list(range(t - stm + 1, t + 1)))
        like = random_adjusted(len(ids), rng)
        pf = full.compute_posterior(ids, like, dense=True)
        pi = inc.compute_posterior(ids, like, incremental=True)
        assert np.array_equal(pf, pi), (t, float(np.max(np.abs(pf - pi))))
    # the same id set twice: the cached matrix is reused (:276-281)
    pf = full.compute_posterior(ids, like, dense=True)
    pi = inc.compute_posterior(ids, like, incremental=True)
    assert np.array_equal(pf, pi)
