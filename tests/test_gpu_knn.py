"""GPU parity: exact 2-NN / self distances of the HIP engine (through the C-ABI) vs the CPU oracle.
Bit-exact for Hamming (ids and distances) AND for squared L2 (the kernel reproduces rtflann's summation order)."""
import numpy as np
import pytest

from rtabmap_amd import synth

pytestmark = pytest.mark.gpu


def _engine(dtype, dim, **kw):
    import rtabmap_amd
    return rtabmap_amd.Engine(dtype, dim, **kw)


def _check(eng, oracle, vocab, ids, queries, removed=None):
    got_ids, got_d = eng.knn2(queries)
    # u8: the engine stands in for the brute-force strategies, i.e. cv::NORM_HAMMING over every byte (rtflann's functor ignores the
    # size % 8 trailing bytes; the two agree for every size that is a multiple of 8)
    metric = oracle.METRIC_HAMMING_CV if vocab.dtype == np.uint8 else None
    idx, d = oracle.knn2_linear(vocab, queries, removed=removed, metric=metric)
    exp_ids = np.where(idx >= 0, ids[np.maximum(idx, 0)], 0).astype(np.int32)
    np.testing.assert_array_equal(got_ids, exp_ids)
    np.testing.assert_array_equal(got_d, d)          # bit-exact, also for float32 L2


# the last two: more strips of 256 words than compute units -> the persistent filter workgroups (knn_bf16_filter_body_p), 2 and 1
# blocks of 512 queries, a ragged last strip
@pytest.mark.parametrize("mode", ["bf16", "f16"])       # the library's default filter and the one bench.py runs
@pytest.mark.parametrize("n,q", [(49000, 500), (49000, 1000), (5000, 77), (257, 64), (3, 5), (70001, 1000), (150003, 300)])
def test_knn2_surf_bit_exact(oracle, n, q, mode):
    v = synth.vocab_surf(n)
    qs = synth.queries_surf(v, q)
    ids = np.arange(1, n + 1, dtype=np.int32)
    eng = _engine("f32", 64, knn_mode=mode)
    eng.vocab_append(v, ids)
    _check(eng, oracle, v, ids, qs)
    eng.close()


@pytest.mark.parametrize("n,q", [(200000, 500), (4097, 130)])
def test_knn2_orb_bit_exact(oracle, n, q):
    v = synth.vocab_orb(n)
    qs = synth.queries_orb(v, q)
    ids = np.arange(1, n + 1, dtype=np.int32)
    eng = _engine("u8", 32)
    eng.vocab_append(v, ids)
    _check(eng, oracle, v, ids, qs)
    eng.close()


def test_knn2_ties_lowest_row_wins(oracle):
    rng = np.random.default_rng(5)
    v = rng.integers(0, 4, (6000, 32), dtype=np.uint8)
    v[3000:] = v[:3000]                                  # every row has an exact duplicate later on
    q = rng.integers(0, 4, (200, 32), dtype=np.uint8)
    ids = np.arange(1, 6001, dtype=np.int32)
    eng = _engine("u8", 32)
    eng.vocab_append(v, ids)
    got_ids, got_d = eng.knn2(q)
    _check(eng, oracle, v, ids, q)
    assert (got_d[:, 0] == got_d[:, 1]).any() and (got_ids[:, 0] <= 3000).all()
    # float duplicates as well
    vf = synth.vocab_surf(2000)
    vf[1000:] = vf[:1000]
    qf = synth.queries_surf(vf, 100)
    e2 = _engine("f32", 64)
    e2.vocab_append(vf, np.arange(1, 2001, dtype=np.int32))
    _check(e2, oracle, vf, np.arange(1, 2001, dtype=np.int32), qf)
    eng.close(); e2.close()


def test_knn2_tombstones_append_rebuild(oracle):
    v = synth.vocab_orb(3000, seed=3)
    q = synth.queries_orb(v, 150, seed=4)
    ids = np.arange(10, 3010, dtype=np.int32)
    eng = _engine("u8", 32)
    eng.vocab_append(v[:2000], ids[:2000])
    eng.vocab_append(v[2000:], ids[2000:])               # second append keeps the order
    rng = np.random.default_rng(9)
    dead = np.sort(rng.choice(3000, 700, replace=False))
    eng.vocab_remove(ids[dead])
    removed = np.zeros(3000, np.uint8); removed[dead] = 1
    _check(eng, oracle, v, ids, q, removed=removed)      # tombstoned rows are never returned
    assert eng.vocab_count() == (3000, 2300)
    # rows re-added with LOWER ids than existing ones go to the end (append branch) ...
    extra = synth.vocab_orb(50, seed=12)
    extra_ids = np.arange(1, 51, dtype=np.int32) * 0 + np.arange(3100, 3150, dtype=np.int32)
    low_ids = np.array([3, 5, 7], np.int32)
    eng.vocab_append(extra, extra_ids)
    eng.vocab_append(v[dead[:3]], low_ids)
    allv = np.vstack([v, extra, v[dead[:3]]])
    allids = np.concatenate([ids, extra_ids, low_ids])
    rem = np.concatenate([removed, np.zeros(53, np.uint8)])
    _check(eng, oracle, allv, allids, q, removed=rem)
    # ... and a rebuild compacts and orders by ascending word id (VWDictionary.cpp:636-660)
    eng.vocab_rebuild()
    keep = rem == 0
    order = np.argsort(allids[keep], kind="stable")
    rv, rids = allv[keep][order], allids[keep][order]
    rows, got_ids = eng.vocab_read(0, rv.shape[0])
    np.testing.assert_array_equal(got_ids, rids)
    np.testing.assert_array_equal(rows, rv)
    _check(eng, oracle, rv, rids, q)
    eng.close()


def test_knn2_tiny_and_empty_vocabulary(oracle):
    eng = _engine("f32", 64)
    q = synth.queries_surf(synth.vocab_surf(10), 7)
    ids, d = eng.knn2(q)
    assert (ids == 0).all() and (d == -1).all()
    v = synth.vocab_surf(1, seed=1)
    eng.vocab_append(v, np.array([42], np.int32))
    ids, d = eng.knn2(q)
    assert (ids[:, 0] == 42).all() and (ids[:, 1] == 0).all() and (d[:, 1] == -1).all()
    np.testing.assert_array_equal(d[:, 0], oracle.dist_matrix(q, v)[:, 0])
    eng.close()


@pytest.mark.parametrize("dtype,dim", [("f32", 128), ("f32", 61), ("f32", 3), ("u8", 64), ("u8", 16), ("u8", 8), ("u8", 24), ("u8", 61), ("u8", 33)])
def test_knn2_other_descriptor_sizes(oracle, dtype, dim):
    rng = np.random.default_rng(dim)
    if dtype == "f32":
        v = rng.standard_normal((1500, dim)).astype(np.float32)
        q = rng.standard_normal((70, dim)).astype(np.float32)
    else:
        v = rng.integers(0, 256, (1500, dim), dtype=np.uint8)
        q = rng.integers(0, 256, (70, dim), dtype=np.uint8)
    ids = np.arange(1, 1501, dtype=np.int32)
    eng = _engine(dtype, dim)
    eng.vocab_append(v, ids)
    _check(eng, oracle, v, ids, q)
    eng.close()


@pytest.mark.parametrize("kind", ["surf", "orb"])
def test_selfdist_bit_exact_and_symmetric(oracle, kind):
    if kind == "surf":
        q = synth.queries_surf(synth.vocab_surf(1000), 300)
        eng = _engine("f32", 64)
    else:
        q = synth.queries_orb(synth.vocab_orb(1000), 300)
        eng = _engine("u8", 32)
    D = eng.selfdist(q)
    np.testing.assert_array_equal(D, oracle.dist_matrix(q, q))
    np.testing.assert_array_equal(D, D.T)
    eng.close()


@pytest.mark.parametrize("many_clusters", [False, True])
def test_knn2_mfma_filter_matches_exact_scan_and_falls_back_on_clusters(oracle, monkeypatch, many_clusters):
    """f32 dim-64 vocabularies >= 256 rows go through an MFMA filter (bf16x3 by default, f32 MFMA on request) + exact re-rank.
    Results must be the exact scan's, bit for bit; a run of identical rows next to each other (more equal candidates than a
    row block keeps) cannot be certified and must take the exact-scan fallback, scattered copies are certified."""
    import rtabmap_amd
    v = synth.vocab_surf(20000, seed=5)
    q = synth.queries_surf(v, 300, seed=6)
    v[::500] = v[123]                    # 40 copies of one row scattered over the vocabulary ...
    v[7000:7040] = v[123]                # ... and 40 more in one run
    q[7] = v[123]
    q[8] = v[123] + np.float32(1e-4)
    if many_clusters:                    # many uncertifiable queries at once
        q[100:160] = v[123] + (np.arange(60, dtype=np.float32)[:, None] * np.float32(1e-5))
    ids = np.arange(1, 20001, dtype=np.int32)
    res = {}
    for mode in ("bf16", "f16", "mfma32", "valu"):
        eng = rtabmap_amd.Engine("f32", 64, knn_mode=mode)
        eng.vocab_append(v, ids)
        res[mode] = eng.knn2(q)
        st = eng.stats()
        fb = st["knn_last_fallback_queries"]
        if mode == "f16":                # (the one-product fp16 filter's wider error bound sends a few more queries to the exact scan)
            assert (33 <= fb < 200) if many_clusters else (1 <= fb <= 100), (mode, fb)
        elif mode != "valu":
            assert (33 <= fb < 120) if many_clusters else (1 <= fb <= 32), (mode, fb)   # the cluster queries, not everything
        else:
            assert fb == 0
        _check(eng, oracle, v, ids, q)
        # tombstones in MFMA mode: remove the current nearest rows of every query, search again
        dead = np.unique(res[mode][0][:, 0])
        eng.vocab_remove(dead)
        removed = np.zeros(20000, np.uint8); removed[dead - 1] = 1
        _check(eng, oracle, v, ids, q, removed=removed)
        eng.vocab_rebuild()
        keep = removed == 0
        _check(eng, oracle, v[keep], ids[keep], q)
        if mode != "valu":               # the filter's error bound must hold with room to spare
            assert 0.0 < eng.stats()["knn_max_err_ratio"] < 0.5, eng.stats()["knn_max_err_ratio"]
        eng.close()
    np.testing.assert_array_equal(res["bf16"][0], res["valu"][0])
    np.testing.assert_array_equal(res["bf16"][1], res["valu"][1])
    np.testing.assert_array_equal(res["mfma32"][0], res["valu"][0])
    np.testing.assert_array_equal(res["mfma32"][1], res["valu"][1])
    np.testing.assert_array_equal(res["f16"][0], res["valu"][0])
    np.testing.assert_array_equal(res["f16"][1], res["valu"][1])


@pytest.mark.parametrize("mode", ["bf16", "mfma32", "f16"])
def test_knn2_filter_error_bound_on_wide_range_descriptors(oracle, monkeypatch, mode):
    """The filter certificate rests on |filter score - exact distance| <= eps.  Descriptors with a wide dynamic range (large
    and tiny components, mixed signs, non-unit norms, queries far from and equal to rows) must stay inside it, and the answers
    must still be the exact scan's."""
    import rtabmap_amd
    rng = np.random.default_rng(11)
    n, qn = 4096, 257
    v = (rng.standard_normal((n, 64)) * np.exp(rng.uniform(-6, 6, (n, 64)))).astype(np.float32)
    v[::7] *= np.float32(1e3)
    v[1::7] *= np.float32(1e-3)
    q = v[rng.integers(0, n, qn)] * (1 + rng.standard_normal((qn, 64)).astype(np.float32) * np.float32(1e-3))
    q = q.astype(np.float32)
    q[:16] = v[:16]
    q[16:32] = (rng.standard_normal((16, 64)) * 50).astype(np.float32)
    ids = np.arange(1, n + 1, dtype=np.int32)
    eng = rtabmap_amd.Engine("f32", 64, knn_mode=mode)
    eng.vocab_append(v, ids)
    _check(eng, oracle, v, ids, q)
    r = eng.stats()["knn_max_err_ratio"]
    if mode == "f16":                    # components beyond half's range: every query must have gone to the exact scan (and come back exact)
        assert eng.stats()["knn_last_fallback_queries"] == qn
        v2 = (v / np.float32(2.0 ** 22)).astype(np.float32)          # the same descriptors scaled into range: filter + certificate again
        eng2 = rtabmap_amd.Engine("f32", 64, knn_mode=mode)
        eng2.vocab_append(v2, ids)
        q2 = (q / np.float32(2.0 ** 22)).astype(np.float32)
        _check(eng2, oracle, v2, ids, q2)
        eng2.close()
    else:
        assert 0.0 < r < 0.5, r
    eng.close()


@pytest.mark.parametrize("units", [24, 7, 0])
def test_knn2_persistent_filter_many_strips_per_workgroup(oracle, units):
    """lcd_set_option("filter_units") plans the persistent filter for fewer compute units than the chip has: every workgroup walks
    23 (24 units: 12 workgroups per block of 512 queries) or 92 (7 units: 3 workgroups) strips through the two LDS strip buffers,
    the last of them ragged, unequal strip counts between workgroups; 0 switches the persistent launch off.  Bit-exact against the
    oracle each time, tombstoned rows included."""
    n, q = 70001, 1000
    v = synth.vocab_surf(n, seed=21)
    qs = synth.queries_surf(v, q, seed=22)
    ids = np.arange(1, n + 1, dtype=np.int32)
    eng = _engine("f32", 64)
    eng.set_option("filter_units", units)
    eng.vocab_append(v, ids)
    _check(eng, oracle, v, ids, qs)
    rng = np.random.default_rng(23)
    gone = np.unique(rng.integers(0, n, 3000))
    eng.vocab_remove(ids[gone])
    removed = np.zeros(n, np.uint8)
    removed[gone] = 1
    _check(eng, oracle, v, ids, qs, removed=removed)
    eng.close()


def test_knn2_one_million_words_properties(oracle):
    """BASELINE.json config 4 scale (1M SURF words; one GPU's HBM holds them easily).  The CPU oracle needs minutes for a whole
    frame at this size, so every query is checked through size-independent properties -- a query that IS a vocabulary row comes
    back as that row at distance exactly 0 (lowest row among duplicates); a slightly perturbed row comes back as that row; the
    MFMA-filter path (thousands of row blocks: the re-rank's general loops) agrees bit for bit with the exact VALU scan -- and a
    sample of 48 queries (exact rows, perturbed rows, the one with the duplicate) against the oracle's linear scan, bit for bit."""
    import rtabmap_amd
    n, q = 1_000_000, 500
    rng = np.random.default_rng(4)
    v = rng.standard_normal((n, 64)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    pick = rng.integers(0, n, q)
    qs = v[pick].copy()
    qs[250:] += rng.standard_normal((q - 250, 64)).astype(np.float32) * np.float32(0.01)
    v[999_999] = v[pick[0]]                                   # a duplicate far away: the lower row must win the tie
    ids = np.arange(1, n + 1, dtype=np.int32)
    res = {}
    for mode in ("bf16", "f16", "valu"):
        eng = rtabmap_amd.Engine("f32", 64, vocab_capacity=n, knn_mode=mode)
        for a in range(0, n, 250_000):
            eng.vocab_append(v[a:a + 250_000], ids[a:a + 250_000])
        res[mode] = eng.knn2(qs)
        eng.close()
    w, d = res["bf16"]
    np.testing.assert_array_equal(w, res["valu"][0])
    np.testing.assert_array_equal(d, res["valu"][1])
    np.testing.assert_array_equal(res["f16"][0], res["valu"][0])
    np.testing.assert_array_equal(res["f16"][1], res["valu"][1])
    first_row = {}
    for r in pick[:250]:
        first_row.setdefault(int(r), int(r))
    exp = np.array([min(int(r), 999_999) if i == 0 else int(r) for i, r in enumerate(pick)], dtype=np.int64)
    np.testing.assert_array_equal(w[:, 0], exp + 1)
    assert (d[:250, 0] == 0.0).all() and (d[:, 1] > d[:, 0]).sum() >= q - 1
    assert w[0, 1] == 1_000_000 and d[0, 1] == 0.0            # the duplicate is the second neighbour, also at distance 0
    sample = np.concatenate([np.arange(0, 16), np.arange(242, 258), np.arange(q - 16, q)])
    idx, d_ref = oracle.knn2_linear(v, qs[sample], threads=8)
    np.testing.assert_array_equal(w[sample], ids[idx])
    np.testing.assert_array_equal(d[sample], d_ref)


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_knn2_keys_that_stand_for_four_rows(oracle, mode):
    """The bf16 / fp16 filters keep one key per four consecutive rows and the re-rank evaluates the four rows of a kept key exactly
    (push_group4).  What that must get right, on a SURF vocabulary whose size is no multiple of four: both neighbours of a query inside ONE
    group (an exact duplicate and a near copy next to it), a tombstone between them, the nearest row the LAST row of the table (the group's
    other rows do not exist), a group whose best row is a tombstone (its other rows must not be dropped with it), and an exact tie between
    rows of one group (the lower row wins)."""
    n = 20003
    rng = np.random.default_rng(41)
    v = synth.vocab_surf(n, seed=42)
    q = synth.queries_surf(v, 240, seed=43, frac_known=0.5)

    def near(row, sigma):
        x = v[row] + sigma * rng.standard_normal(64).astype(np.float32)
        return (x / np.linalg.norm(x)).astype(np.float32)
    removed = np.zeros(n, np.uint8)
    for k in range(40):                                   # group base 4 * g: rows g4 .. g4 + 3
        g4 = 4 * int(rng.integers(10, n // 4 - 10))
        kind = k % 5
        if kind == 0:                                     # duplicate + near copy inside one group
            q[k] = near(g4 + 1, 0.0); v[g4 + 2] = near(g4 + 1, 0.01)
        elif kind == 1:                                   # ... with a tombstone between them
            q[k] = near(g4, 0.0); v[g4 + 2] = near(g4, 0.01); removed[g4 + 1] = 1
        elif kind == 2:                                   # the group's best row is a tombstone: the second best of the group must survive
            q[k] = near(g4 + 3, 0.0); removed[g4 + 3] = 1; v[g4] = near(g4 + 3, 0.02)
        elif kind == 3:                                   # an exact tie inside the group
            v[g4 + 2] = v[g4 + 1]; q[k] = near(g4 + 1, 0.005)
        else:                                             # the table's last row, alone in its group (n % 4 == 3: rows n-3 .. n-1 exist)
            q[k] = near(n - 1, 0.003)
    ids = np.arange(1, n + 1, dtype=np.int32)
    eng = _engine("f32", 64, knn_mode=mode)
    eng.vocab_append(v, ids)
    eng.vocab_remove(ids[removed == 1])
    _check(eng, oracle, v, ids, q, removed=removed)
    assert eng.stats()["knn_last_fallback_queries"] < 40, "the planted cases must go through the filter + re-rank, not through the exact redo"
    eng.close()
