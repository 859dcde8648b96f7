"""Pins the restated distance functors / exact 2-NN / VWDictionary(kNNFlannNaive) of the oracle against the reference's
OWN rtflann compiled in place (oracle/_ref/librtflann_ref.so, built by oracle/Makefile from /root/reference).  CPU only."""
import numpy as np
import pytest

from rtabmap_amd import synth


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        try:
            oracle.build(ref=True)
        except Exception:
            pass
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/librtflann_ref.so not built and /root/reference absent")
    return oracle.ref()


def test_distance_functors_bitwise(oracle, ref):
    import ctypes as C
    rng = np.random.default_rng(1)
    for dim in (64, 128, 61, 3):
        a = rng.standard_normal((200, dim)).astype(np.float32)
        b = rng.standard_normal((200, dim)).astype(np.float32)
        for i in range(200):
            pa, pb = a[i].ctypes.data_as(C.c_void_p), b[i].ctypes.data_as(C.c_void_p)
            assert oracle.lib().orc_dist_l2(pa, pb, dim) == ref.ref_dist_l2(pa, pb, dim)
            assert oracle.lib().orc_dist_l1(pa, pb, dim) == ref.ref_dist_l1(pa, pb, dim)
    for nb in (32, 64, 8):
        a = rng.integers(0, 256, (200, nb), dtype=np.uint8)
        b = rng.integers(0, 256, (200, nb), dtype=np.uint8)
        for i in range(200):
            pa, pb = a[i].ctypes.data_as(C.c_void_p), b[i].ctypes.data_as(C.c_void_p)
            d = oracle.lib().orc_dist_hamming(pa, pb, nb)
            assert d == ref.ref_dist_hamming(pa, pb, nb)
            assert d == int(np.unpackbits(a[i] ^ b[i]).sum())
    # sizes that are not a multiple of 8 bytes (AKAZE: 61): the real rtflann functor ignores the trailing size % 8 bytes
    # (dist.h:555-579) and so does its restatement; cv::NORM_HAMMING (brute-force strategies, same-frame comparison) counts every
    # byte -- the oracle's METRIC_HAMMING_CV, which is what the engine is checked against
    for nb in (61, 33, 7, 12):
        a = rng.integers(0, 256, (100, nb), dtype=np.uint8)
        b = rng.integers(0, 256, (100, nb), dtype=np.uint8)
        for i in range(100):
            pa, pb = a[i].ctypes.data_as(C.c_void_p), b[i].ctypes.data_as(C.c_void_p)
            d = oracle.lib().orc_dist_hamming(pa, pb, nb)
            assert d == ref.ref_dist_hamming(pa, pb, nb)
            assert d == int(np.unpackbits(a[i, :nb // 8 * 8] ^ b[i, :nb // 8 * 8]).sum())
        full = oracle.dist_matrix(a, b, metric=oracle.METRIC_HAMMING_CV)
        trunc = oracle.dist_matrix(a, b, metric=oracle.METRIC_HAMMING)
        bits = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2)
        np.testing.assert_array_equal(full, bits.astype(np.float32))
        assert (trunc <= full).all() and (trunc < full).any()


@pytest.mark.parametrize("kind", ["surf", "orb", "orb_ties"])
def test_exact_2nn_matches_rtflann_linear(oracle, ref, kind):
    if kind == "surf":
        v = synth.vocab_surf(5000)
        q = synth.queries_surf(v, 300)
    elif kind == "orb":
        v = synth.vocab_orb(5000)
        q = synth.queries_orb(v, 300)
    else:   # many exact distance ties: short codes drawn from a tiny alphabet, duplicated rows
        rng = np.random.default_rng(5)
        v = rng.integers(0, 4, (3000, 8), dtype=np.uint8)
        v[1500:] = v[:1500]
        q = rng.integers(0, 4, (200, 8), dtype=np.uint8)
    idx, dist = oracle.knn2_linear(v, q)
    r = oracle.RefIndex(v)
    ridx, rdist = r.knn(q)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)
    if kind == "orb_ties":
        assert (dist[:, 0] == dist[:, 1]).any()           # the tie-break rule is actually exercised
        assert (idx[:, 0] < 1500).all()                   # the earlier duplicate always wins


def test_exact_2nn_with_removed_points(oracle, ref):
    v = synth.vocab_orb(2000, seed=3)
    q = synth.queries_orb(v, 100, seed=4)
    removed = np.zeros(2000, np.uint8)
    r = oracle.RefIndex(v)
    rng = np.random.default_rng(9)
    for i in rng.choice(2000, 300, replace=False):
        removed[i] = 1
        r.remove(int(i))
    idx, dist = oracle.knn2_linear(v, q, removed=removed)
    ridx, rdist = r.knn(q)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)
    # appended points get the next indices and removed ones stay skipped, also across FLANN's x2 rebuild
    extra = synth.vocab_orb(2500, seed=11)
    r.add(extra, rebuild=2.0)
    allv = np.vstack([v, extra])
    rem2 = np.concatenate([removed, np.zeros(2500, np.uint8)])
    idx, dist = oracle.knn2_linear(allv, q, removed=rem2)
    ridx, rdist = r.knn(q)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dist, rdist)


def test_kdtree_is_only_a_speed_baseline(oracle, ref):
    """The reference default (4 randomized kd-trees, 32 checks) is approximate: it must not be used as a parity oracle."""
    v = synth.vocab_surf(20000)
    q = synth.queries_surf(v, 200)
    exact, _ = oracle.knn2_linear(v, q)
    kd = oracle.RefIndex(v, algo=oracle.ALGO_KDTREE, trees=4)
    approx, _ = kd.knn(q, checks=32)
    agree = (exact[:, 0] == approx[:, 0]).mean()
    assert 0.2 < agree <= 1.0           # randomly seeded trees (kdtree_index.h:681-683): the agreement varies run to run


def _sim_flann_naive(oracle, frames, nndr):
    """VWDictionary with Kp/NNStrategy=0 driven 'by hand' on the REAL rtflann LINEAR index: the same steps as
    VWDictionary::update()/addNewWords() (VWDictionary.cpp:499-570, 1015-1209) but the indexed 2-NN comes from
    rtflann::Index<>::knnSearch itself.  Used to pin the oracle's strategy-0 restatement."""
    index = None
    index_to_id, id_to_index = {}, {}
    words = {}          # id -> descriptor
    refs = {}           # id -> {sig: count}
    not_indexed, removed_indexed = [], []
    last_id = 0
    next_index = 0
    out = []
    for sig, desc in enumerate(frames, start=1):
        # Memory::preUpdate: cleanUnusedWords + update()
        unused = [w for w in sorted(words) if not refs[w]]
        for w in unused:
            del words[w], refs[w]
            if w in not_indexed:
                not_indexed.remove(w)
            else:
                removed_indexed.append(w)
        first = not removed_indexed and len(words) == len(not_indexed)
        if not_indexed or not words or removed_indexed:
            if not first and words:
                for w in sorted(removed_indexed):
                    index.remove(id_to_index[w])
                    del index_to_id[id_to_index[w]], id_to_index[w]
                for w in sorted(not_indexed):
                    if index is None:
                        index = oracle.RefIndex(words[w][None, :]); idx = 0; next_index = 1
                    else:
                        index.add(words[w][None, :]); idx = next_index; next_index += 1
                    index_to_id[idx] = w; id_to_index[w] = idx
            else:
                index, index_to_id, id_to_index = None, {}, {}
                if words:
                    ids = sorted(words)
                    index = oracle.RefIndex(np.stack([words[w] for w in ids]))
                    next_index = len(ids)
                    for i, w in enumerate(ids):
                        index_to_id[i] = w; id_to_index[w] = i
        not_indexed, removed_indexed = [], []
        # addNewWords
        ids_out = []
        new_rows, new_ids = [], []
        if index is not None:
            kidx, kd = index.knn(desc)
        for i in range(desc.shape[0]):
            full = []
            if index is not None:
                for j in range(2):
                    wid = index_to_id.get(int(kidx[i, j]), 0)
                    if kd[i, j] >= 0 and wid:
                        full.append((float(kd[i, j]), wid))
                    else:
                        break
            if new_rows:
                nidx, nd = oracle.knn2_linear(np.stack(new_rows), desc[i:i + 1])
                for j in range(2 if len(new_rows) > 1 else 1):
                    if nidx[0, j] >= 0:
                        full.append((float(nd[0, j]), new_ids[int(nidx[0, j])]))
            full.sort(key=lambda t: t[0])        # stable: equal keys keep insertion order (std::multimap)
            bad = len(full) < 2 or np.float32(full[0][0]) > np.float32(nndr) * np.float32(full[1][0])
            if bad:
                last_id += 1
                words[last_id] = desc[i].copy(); refs[last_id] = {sig: 1}
                not_indexed.append(last_id); new_rows.append(desc[i]); new_ids.append(last_id)
                ids_out.append(last_id)
            else:
                w = full[0][1]
                refs[w][sig] = refs[w].get(sig, 0) + 1
                ids_out.append(w)
        out.append(ids_out)
        # forget an old signature now and then so that words become unused and get removed
        if sig > 3:
            old = sig - 3
            for w in refs:
                refs[w].pop(old, None)
    return out


@pytest.mark.parametrize("kind", ["orb", "surf"])
def test_vwdictionary_flann_naive_matches_real_rtflann(oracle, ref, kind):
    rng = np.random.default_rng(21)
    frames = []
    if kind == "orb":
        base = synth.vocab_orb(300, seed=77, nbytes=32)
        for t in range(8):
            frames.append(synth.queries_orb(base, 120, seed=100 + t, frac_known=0.8, flip=0.05))
    else:
        base = synth.vocab_surf(300, seed=78)
        for t in range(8):
            frames.append(synth.queries_surf(base, 120, seed=200 + t, frac_known=0.8, sigma=0.03))
    expect = _sim_flann_naive(oracle, frames, 0.8)
    m = oracle.OracleMemory(strategy=oracle.kNNFlannNaive, nndr=0.8)
    for t, desc in enumerate(frames):
        sid, ids = m.update(desc)
        assert ids == expect[t], "frame %d" % t
        if sid > 3:
            m.forget(sid - 3)
    del rng
