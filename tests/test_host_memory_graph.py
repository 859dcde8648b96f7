"""CPU: the part of Memory that BayesFilter talks to, in the C++ host mirror (rtabmap_amd/host/MemoryHip) -- no device call is made.

MemoryHip::getNeighborsId restates Memory::getNeighborsId(id, depth, 0, false, false, true, true) (reference Memory.cpp:1703-1893)
level by level; the harness of the Bayes tests (tests/bayes_model.Graph) states the same thing as a 0-1 shortest-path search.  The
two are compared on random graphs with loop closures and forgotten nodes; addLink's return values follow Memory.cpp:3877-3935."""
import collections

import numpy as np
import pytest


def _search(odom, loop, alive, sid, max_depth):
    if sid not in alive:
        return {}
    dist = {sid: 0}
    dq = collections.deque([sid])
    while dq:
        u = dq.popleft()
        d = dist[u]
        for v in loop[u]:
            if v in alive and (v not in dist or dist[v] > d):
                dist[v] = d
                dq.appendleft(v)
        if d + 1 < max_depth:
            for v in odom[u]:
                if v in alive and (v not in dist or dist[v] > d + 1):
                    dist[v] = d + 1
                    dq.append(v)
    return dist


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_get_neighbors_id_matches_the_shortest_path_statement(seed):
    from rtabmap_amd.vwdictionary import MemoryHip
    rng = np.random.default_rng(seed)
    n = 120
    m = MemoryHip()
    odom, loop = collections.defaultdict(set), collections.defaultdict(set)
    for s in range(1, n + 1):
        assert m.add_signature([], s) == s
    for s in range(1, n):
        if s % 37:                                        # a few breaks in the odometry chain (new map)
            assert m.add_link(s, s + 1, neighbor=True)
            odom[s].add(s + 1); odom[s + 1].add(s)
    for _ in range(40):
        a, b = int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))
        if a == b:
            assert not m.add_link(a, b)
            continue
        linked = b in odom[a] or b in loop[a]
        assert m.add_link(a, b)                           # "already linked" is not an error (true, nothing added)
        if not linked:
            loop[a].add(b); loop[b].add(a)
    alive = set(range(1, n + 1))
    for depth in (1, 2, 4, 17):
        for s in rng.integers(1, n + 1, 25).tolist():
            assert m.get_neighbors_id(s, depth) == _search(odom, loop, alive, s, depth)
    # nodes that left the memory are neither reported nor walked through (maxCheckedInDatabase = 0)
    gone = set(rng.choice(np.arange(1, n + 1), 25, replace=False).tolist())
    for s in gone:
        m.forget(s)
    alive -= gone
    assert not m.add_link(next(iter(gone)), next(iter(alive)))          # a missing signature: false
    for depth in (2, 6, 17):
        for s in range(1, n + 1, 3):
            assert m.get_neighbors_id(s, depth) == _search(odom, loop, alive, s, depth)
    assert m.get_neighbors_id(-1, 3) == {} and m.get_neighbors_id(0, 3) == {}
    assert m.working_mem()[0] == -1 and set(m.working_mem()[1:]) == alive and m.st_mem() == []
    m.close()


def test_bayes_filter_hip_parameters_and_refusals():
    """BayesFilter::setPredictionLC / parseParameters / computePosterior's refusals (BayesFilter.cpp:56-122, :149-165), no device."""
    from rtabmap_amd.vwdictionary import BayesFilterHip, MemoryHip
    from bayes_model import DEFAULT_LC
    b = BayesFilterHip()
    np.testing.assert_array_equal(b.get_prediction_lc(), DEFAULT_LC)     # parsed through float, stored as double
    b.set_prediction_lc("0.1 0.5  0.25 0.1")                            # empty tokens are dropped
    np.testing.assert_array_equal(b.get_prediction_lc(), np.array([0.1, 0.5, 0.25, 0.1], np.float32).astype(np.float64))
    b.set_prediction_lc("0.5")                                          # fewer than two values: refused, the old ones stay
    assert b.get_prediction_lc().shape[0] == 4
    b.set_prediction_lc("0.1 1.5 0.2")                                  # out of range: refused
    assert b.get_prediction_lc().shape[0] == 4
    m = MemoryHip()
    ids, post = b.compute_posterior(m, [], [])                          # "likelihood is empty!": the (empty) posterior, unchanged
    assert ids.shape[0] == 0
    ids, post = b.compute_posterior(m, [-1], [1.0])                     # no engine yet (no descriptor seen): loud, unchanged
    assert ids.shape[0] == 0 and "engine" in b.last_error()
    b.close(); m.close()
