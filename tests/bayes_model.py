"""Test harness for the Bayes filter: the part of Memory the filter talks to (a pose graph and Memory::getNeighborsId), in Python.

graph: signatures 1..n in odometry order (neighbour links i <-> i+1) plus loop-closure links.  neighbors() is the traversal of
Memory::getNeighborsId(id, maxGraphDepth, 0, incrementMarginOnLoop=false, ignoreLoopIds=false, ...) (reference Memory.cpp:1703-1890):
breadth first, a loop-closure link does not increase the margin, margins 0 .. maxGraphDepth - 1, the signature itself at margin 0.
"""
import collections

import numpy as np

import oracle as O


def prediction_lc_as_parsed(values):
    """BayesFilter::setPredictionLC parses the string with uStr2Float (float) and stores doubles (BayesFilter.cpp:88-92)."""
    return np.asarray(values, dtype=np.float32).astype(np.float64)


DEFAULT_LC = prediction_lc_as_parsed(O.DEFAULT_PREDICTION_LC)


class Graph:
    def __init__(self, n, loops=()):
        self.n = n
        self.odom = collections.defaultdict(set)
        self.loop = collections.defaultdict(set)
        for i in range(1, n):
            self.odom[i].add(i + 1)
            self.odom[i + 1].add(i)
        for a, b in loops:
            if a != b:
                self.loop[a].add(b)
                self.loop[b].add(a)

    def neighbors(self, sid, max_depth):
        """{id: margin}: 0-1 breadth-first search (loop links cost 0, odometry links 1), margins < max_depth."""
        dist = {sid: 0}
        dq = collections.deque([sid])
        while dq:
            u = dq.popleft()
            d = dist[u]
            for v in self.loop[u]:
                if v not in dist or dist[v] > d:
                    dist[v] = d
                    dq.appendleft(v)
            if d + 1 < max_depth:
                for v in self.odom[u]:
                    if v not in dist or dist[v] > d + 1:
                        dist[v] = d + 1
                        dq.append(v)
        return dist


def random_graph(n, n_loops, rng):
    loops = []
    for _ in range(n_loops):
        a = int(rng.integers(20, n + 1)) if n > 20 else int(rng.integers(1, n + 1))
        b = int(rng.integers(1, max(a - 10, 2)))
        loops.append((a, b))
    return Graph(n, loops)


def csr_lists(graph, ids, max_depth, keep=None):
    """(offsets, nbr ids, margins) of the neighbour lists of `ids`, ascending neighbour id (std::map order); keep: filter on ids."""
    off = [0]
    nbr, mg = [], []
    for s in ids:
        d = graph.neighbors(int(s), max_depth)
        for k in sorted(d):
            if keep is None or keep(k):
                nbr.append(k)
                mg.append(d[k])
        off.append(len(nbr))
    return np.asarray(off, np.int64), np.asarray(nbr, np.int32), np.asarray(mg, np.int32)


def random_adjusted(m, rng):
    """An adjusted likelihood as Rtabmap::adjustLikelihood leaves it: 1 for most signatures, a few above, the virtual place first."""
    like = np.ones(m, np.float32)
    k = max(1, m // 50)
    hot = rng.choice(np.arange(1, m), size=min(k, m - 1), replace=False) if m > 1 else np.zeros(0, np.int64)
    like[hot] = (1.0 + rng.gamma(2.0, 2.0, size=hot.shape[0])).astype(np.float32)
    like[0] = np.float32(1.0 + rng.random() * 2.0)
    return like
