"""GPU parity, end to end through the HOST MIRROR: the same descriptor stream is pushed through the oracle's restated
Memory/VWDictionary and through MemoryHip/VWDictionaryHip (C++ over the C-ABI).  Per frame: identical word ids, identical
dictionary bookkeeping, identical index (tie-break) order; likelihood within 1e-4 relative with the same arg-max."""
import os

import numpy as np
import pytest

from rtabmap_amd import synth

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-7


def _frames(kind, n_frames, q, base_n=500):
    if kind == "orb":
        base = synth.vocab_orb(base_n, seed=77)
        return [synth.queries_orb(base, q, seed=100 + t, frac_known=0.8, flip=0.05) for t in range(n_frames)]
    base = synth.vocab_surf(base_n, seed=78)
    return [synth.queries_surf(base, q, seed=200 + t, frac_known=0.8, sigma=0.03) for t in range(n_frames)]


def _same_state(o, h):
    assert h.vwd.visual_words == o.vwd.visual_words
    assert h.vwd.not_indexed_words == o.vwd.not_indexed_words
    assert h.vwd.indexed_words == o.vwd.indexed_words
    assert h.vwd.total_active_references == o.vwd.total_active_references
    assert h.vwd.unused_words == o.vwd.unused_words
    assert h.vwd.index_ids() == o.vwd.index_ids()


@pytest.mark.parametrize("kind", ["orb", "surf"])
@pytest.mark.parametrize("together", [True, False])
@pytest.mark.parametrize("device_frames", [True, False])
def test_incremental_stream(oracle, kind, together, device_frames):
    """device_frames: MemoryHip::update as ONE device call (lcd_frame_host: quantisation, the signature's references, update()'s append
    and the likelihood of Rtabmap.cpp:2117 -- the default since round 5) or the call-by-call path of rounds 1-4.  Frames that keep
    unquantised features take the call-by-call path in both modes, so the two are also interleaved here; forgetting removes words the
    device appended itself."""
    from rtabmap_amd.vwdictionary import MemoryHip
    frames = _frames(kind, 14, 160)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=together)
    h = MemoryHip(nndr=0.8, new_words_compared_together=together)
    h.set_device_frames(device_frames)
    W = 6                                            # working-memory size: older frames are forgotten (words get removed)
    for t, desc in enumerate(frames):
        nq = None if t % 5 else desc.shape[0] - 7   # some frames keep features that are not quantised (ids -1,-2,..)
        so, ido = o.update(desc, nq)
        sh, idh = h.update(desc, nq)
        assert so == sh and idh == ido, "frame %d" % t
        _same_state(o, h)
        ids = np.array(o.signature_ids(), np.int32)
        oi, Lo = o.compute_likelihood(np.array(ido, np.int32), ids)
        # Memory::computeLikelihood(signature, ids): answered from the frame's own device call when update() took the fast path
        fi, Lf = h.compute_likelihood_of(sh, ids)
        assert oi.tolist() == fi.tolist()
        np.testing.assert_allclose(Lf, Lo, rtol=RTOL, atol=ATOL)
        flat = h.compute_likelihood_flat(sh)
        assert (flat is not None) == (device_frames and nq is None)
        if flat is not None:
            assert flat[0].tolist() == sorted(flat[0].tolist()) and set(flat[0].tolist()) <= set(oi.tolist()) and sh in flat[0].tolist()
            lut = dict(zip(oi.tolist(), Lo.tolist()))
            np.testing.assert_allclose(flat[1], np.array([lut[i] for i in flat[0].tolist()], np.float32), rtol=RTOL, atol=ATOL)
        hi, Lh = h.compute_likelihood(np.array(idh, np.int32), ids)
        assert oi.tolist() == hi.tolist()
        np.testing.assert_allclose(Lh, Lo, rtol=RTOL, atol=ATOL)
        if Lo[:-1].size and Lo[:-1].max() > 0:
            assert int(np.argmax(Lh[:-1])) == int(np.argmax(Lo[:-1]))
            assert int(np.argmax(Lf[:-1])) == int(np.argmax(Lo[:-1]))
        if so > W:
            o.forget(so - W)
            h.forget(so - W)
            assert h.get_ni(so - W) == o.get_ni(so - W) == desc.shape[0]
    for w in o.vwd.word_ids()[::9]:
        assert h.vwd.word_refs(w) == o.vwd.word_refs(w)
    h.close()


@pytest.mark.parametrize("kind", ["orb", "surf"])
def test_find_nn_after_stream(oracle, kind):
    """enableWordsRef path: re-activated words are matched with findNN against indexed + not yet indexed words."""
    from rtabmap_amd.vwdictionary import MemoryHip
    frames = _frames(kind, 5, 120)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
    h = MemoryHip()
    for desc in frames[:4]:
        o.update(desc)
        h.update(desc)
    # after the last addNewWords the new words are not indexed yet: findNN must see them too
    assert h.vwd.not_indexed_words == o.vwd.not_indexed_words > 0
    q = frames[4].copy()
    q[:30] = frames[3][:30]                           # exact copies of descriptors that just became words
    assert h.vwd.find_nn(q) == o.vwd.find_nn(q)
    h.close()


def test_fixed_dictionary_from_text_file(oracle, tmp_path):
    """Kp/IncrementalDictionary=false + Kp/DictionaryPath (stand-in for data/Dictionary49k.txt, same text format)."""
    from rtabmap_amd.vwdictionary import MemoryHip
    vocab = synth.vocab_surf(3000)
    path = os.path.join(str(tmp_path), "Dictionary3k.txt")
    synth.write_dictionary_text(path, vocab)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce, incremental=False)
    assert o.vwd.load_fixed_text(path) == 3000
    h = MemoryHip(incremental=False, dictionary_path=path)
    assert h.vwd.visual_words == 3000 and h.vwd.indexed_words == 3000
    for t in range(3):
        q = synth.queries_surf(vocab, 200, seed=300 + t)
        so, ido = o.update(q)
        sh, idh = h.update(q)
        assert idh == ido and len(idh) == 200
    ids = np.array(o.signature_ids(), np.int32)
    oi, Lo = o.compute_likelihood(np.array(ido, np.int32), ids)
    hi, Lh = h.compute_likelihood(np.array(idh, np.int32), ids)
    np.testing.assert_allclose(Lh, Lo, rtol=RTOL, atol=ATOL)
    # round trip of the exported dictionary (exportDictionary writes %f: 6 decimals)
    out = os.path.join(str(tmp_path), "exported.txt")
    refs = os.path.join(str(tmp_path), "refs.txt")
    h.vwd.export_text(refs, out)
    o2 = os.path.join(str(tmp_path), "exported_oracle.txt")
    o.vwd.export_text(os.path.join(str(tmp_path), "refs_oracle.txt"), o2)
    assert open(out).read() == open(o2).read()
    assert open(refs).read() == open(os.path.join(str(tmp_path), "refs_oracle.txt")).read()
    h.close()


def test_error_paths_return_empty_like_the_reference(oracle):
    from rtabmap_amd.vwdictionary import VWDictionaryHip
    d = VWDictionaryHip()
    assert d.add_new_words(np.zeros((0, 64), np.float32), 1) == []            # "Descriptors size is null!"
    assert len(d.add_new_words(synth.vocab_surf(5), 1)) == 5
    assert d.add_new_words(np.zeros((3, 32), np.float32), 2) == []            # size mismatch with the dictionary
    assert d.add_new_words(np.zeros((3, 64), np.uint8), 2) == []              # type mismatch
    assert d.find_nn(np.zeros((2, 32), np.float32)) == [0, 0]
    f = VWDictionaryHip(incremental=False)
    assert f.add_new_words(synth.vocab_surf(4), 1) == []                      # fixed dictionary without words
    d.close(); f.close()


@pytest.mark.parametrize("full", [False, True])
def test_bayes_filter_hip_behind_the_reference_interface(oracle, full):
    """BayesFilterHip::computePosterior(memory, likelihood) -- the reference's call (Rtabmap.cpp:2131) -- frame after frame against the
    restated BayesFilter on the same adjusted likelihood: MemoryHip keeps the short-term / working memory split, the neighbour link
    of addSignatureToStm and the loop closures of addLink, answers getNeighborsId; the filter hands each new id's neighbourhood to
    the device (Bayes/FullPredictionUpdate = false, the reference's default) or every id's on every call (true).  Signatures are
    transferred out of the memory along the way.  Posterior entry by entry, the same highest hypothesis."""
    import collections
    from rtabmap_amd.vwdictionary import BayesFilterHip, MemoryHip
    from bayes_model import DEFAULT_LC
    from test_host_memory_graph import _search
    T, q, STM = 64, 80, 5
    frames = _frames("surf", T, q, base_n=900)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
    ob = oracle.OracleBayesFilter(DEFAULT_LC)
    h = MemoryHip(stm_size=STM)
    hb = BayesFilterHip(full_prediction_update=full)
    odom, loop = collections.defaultdict(set), collections.defaultdict(set)
    alive, stm = set(), []
    depth = DEFAULT_LC.shape[0] - 1
    n_updates = 0
    for t in range(T):
        so, wo = o.update(frames[t])
        sh, wh = h.update(frames[t])
        assert (sh, wh) == (so, wo)
        if stm:
            odom[stm[-1]].add(sh); odom[sh].add(stm[-1])
        stm.append(sh); alive.add(sh)
        while len(stm) > STM:
            stm.pop(0)
        assert h.st_mem() == stm
        wm = h.working_mem()
        assert wm == [-1] + sorted(alive - set(stm))
        if len(wm) < 3:
            continue
        ids = np.array(wm, np.int32)
        hi, Lh = h.compute_likelihood(np.array(wh, np.int32), ids)
        assert hi.tolist() == wm
        adj = oracle.adjust_likelihood(Lh)                       # Rtabmap::adjustLikelihood, restated: the same input for both filters
        for s in wm[1:]:                                         # what Memory::getNeighborsId answers right now
            d = _search(odom, loop, alive, s, depth)
            k = sorted(d)
            ob.set_neighbors(s, k, [d[x] for x in k])
        ob.set_stm(stm)
        exp = ob.compute_posterior(ids, adj, dense=True, incremental=not full)
        pid, post = hb.compute_posterior(h, ids, adj)
        assert pid.tolist() == wm, hb.last_error()
        pos = exp > 0
        r = float(np.median(post[pos].astype(np.float64) / exp[pos].astype(np.float64)))
        assert abs(r - 1.0) <= 2e-6, (t, r)
        np.testing.assert_allclose(post, exp.astype(np.float64) * r, rtol=2e-5, atol=1e-9, err_msg="frame %d" % t)
        hid, hval = oracle.OracleBayesFilter.hypothesis(ids, exp)
        gid, gval = hb.highest_hypothesis
        np.testing.assert_allclose(gval, hval, rtol=1e-5, atol=1e-6)
        top = np.sort(exp[1:].astype(np.float64))[::-1]
        if len(top) < 2 or top[0] - top[1] > 1e-4 * top[0]:
            assert gid == hid, t
        n_updates += 1
        # the graph changes: a loop closure between the newest signature of the working memory and an old one; the oldest leaves
        if t % 9 == 5 and len(wm) > 12:
            a, b = wm[-1], wm[1 + (t % 5)]
            assert h.add_link(a, b)
            loop[a].add(b); loop[b].add(a)
        if t % 11 == 7 and len(wm) > 20:
            x = wm[1]
            o.forget(x); h.forget(x)
            alive.discard(x)
    assert n_updates > 40
    # BayesFilter::reset: the next posterior starts from scratch
    hb.reset(); ob.reset()
    wm = h.working_mem()
    ids = np.array(wm, np.int32)
    adj = np.ones(len(wm), np.float32); adj[0] = 2.0; adj[len(wm) // 2] = 7.0
    for s in wm[1:]:
        d = _search(odom, loop, alive, s, depth)
        k = sorted(d)
        ob.set_neighbors(s, k, [d[x] for x in k])
    exp = ob.compute_posterior(ids, adj, dense=True, incremental=not full)
    pid, post = hb.compute_posterior(h, ids, adj)
    np.testing.assert_allclose(post, exp, rtol=3e-5, atol=1e-9)
    # a likelihood that leaves out a signature of the working memory is not what the device state describes: refused, loudly
    before = hb.compute_posterior(h, ids, adj)[1]
    pid2, post2 = hb.compute_posterior(h, np.delete(ids, 3), np.delete(adj, 3))
    assert "working memory" in hb.last_error() and pid2.tolist() == wm
    hb.close(); h.close()


def test_rtabmap_hip_process_detects_loop_closures_like_the_restated_block(oracle):
    """RtabmapHip::process -- the loop-closure detection block of Rtabmap::process (Rtabmap.cpp:2046-2222, :3129-3186) over the host
    mirrors -- on a route of 40 places followed by a second pass over places 5..34.  Stage by stage against the oracle: word ids
    identical; raw likelihood within 1e-4 of Memory::computeLikelihood; the adjusted vector against Rtabmap::adjustLikelihood
    restated on the same raw vector; the posterior against the restated BayesFilter on the same adjusted vector; the highest
    hypothesis, its acceptance (Rtabmap/LoopThr, single-hypothesis rule) and the loop-closure links that result, against the same
    rules stated here.  The second pass must close loops on the places it revisits."""
    import collections
    from rtabmap_amd.vwdictionary import RtabmapHip
    from bayes_model import DEFAULT_LC
    from test_host_memory_graph import _search
    STM, q, thr = 5, 100, 0.11
    route = list(range(40)) + list(range(5, 35))
    places = [synth.vocab_surf(q, seed=5000 + p) for p in range(40)]
    rng = np.random.default_rng(11)

    def view(p):
        d = places[p] + rng.standard_normal(places[p].shape).astype(np.float32) * np.float32(0.02)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        return np.ascontiguousarray(d)

    r = RtabmapHip(loop_thr=thr, stm_size=STM)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce)
    ob = oracle.OracleBayesFilter(DEFAULT_LC)
    odom, loop = collections.defaultdict(set), collections.defaultdict(set)
    alive, stm = set(), []
    depth = DEFAULT_LC.shape[0] - 1
    place_of, closures = {}, []
    for t, p in enumerate(route):
        desc = view(p)
        res = r.process(desc)
        so, wo = o.update(desc)
        assert res["ok"] and res["id"] == so and r.last_word_ids() == wo
        place_of[so] = p
        if stm:
            odom[stm[-1]].add(so); odom[so].add(stm[-1])
        stm.append(so); alive.add(so)
        while len(stm) > STM:
            stm.pop(0)
        wm = [-1] + sorted(alive - set(stm))
        assert r.memory.working_mem() == wm and r.memory.st_mem() == stm
        if len(wm) < 2:
            assert res["highest"] == (0, 0.0) and res["loop"][0] == 0
            continue
        ids = np.array(wm, np.int32)
        rid, raw = r.vector("raw")
        assert rid.tolist() == wm
        oi, Lo = o.compute_likelihood(np.array(wo, np.int32), ids)
        np.testing.assert_allclose(raw, Lo, rtol=RTOL, atol=ATOL)
        aid, adj = r.vector("likelihood")
        np.testing.assert_allclose(adj, oracle.adjust_likelihood(raw), rtol=2e-6, atol=1e-7)
        for s in wm[1:]:
            d = _search(odom, loop, alive, s, depth)
            k = sorted(d)
            ob.set_neighbors(s, k, [d[x] for x in k])
        ob.set_stm(stm)
        exp = ob.compute_posterior(ids, adj, dense=True, incremental=True)
        pid, post = r.vector("posterior")
        assert pid.tolist() == wm
        pos = exp > 0
        ratio = float(np.median(post[pos].astype(np.float64) / exp[pos].astype(np.float64)))
        assert abs(ratio - 1.0) <= 2e-6
        np.testing.assert_allclose(post, exp.astype(np.float64) * ratio, rtol=2e-5, atol=1e-9, err_msg="frame %d" % t)
        hid, hval = oracle.OracleBayesFilter.hypothesis(ids, exp)
        gid, gval = res["highest"]
        np.testing.assert_allclose(gval, hval, rtol=1e-5, atol=1e-6)
        top = np.sort(exp[1:].astype(np.float64))[::-1]
        if len(top) < 2 or top[0] - top[1] > 1e-4 * top[0]:
            assert gid == hid
        # acceptance (Rtabmap.cpp:2185-2210 with Rtabmap/LoopRatio = 0): over the threshold, and more than one hypothesis
        accept = gid > 0 and gval >= thr and len(wm) > 2 and abs(gval - thr) > 1e-5
        if abs(gval - thr) > 1e-5:
            assert (res["loop"][0] == gid) == bool(accept), (t, res, gval)
        if res["loop"][0] > 0:
            a, b = so, res["loop"][0]
            if b not in odom[a] and b not in loop[a]:
                loop[a].add(b); loop[b].add(a)
            closures.append((t, p, place_of[b]))
            # the link is in the graph of the mirror: the hypothesis is a neighbour of margin 0 now
            assert r.memory.get_neighbors_id(so, 2).get(b) == 0
    first_pass = [c for c in closures if c[0] < 40]
    second_pass = [c for c in closures if c[0] >= 40]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):                                               # kept for the record of the round
        with open(os.path.join(out_dir, "rtabmap_hip_closures.txt"), "w") as f:
            f.write("first pass (frame, place, place of the hypothesis): %r\nsecond pass: %r\n" % (first_pass, second_pass))
    # The second pass closes loops (29 of 30 frames on the MI355X run of this round, 15 of them at the revisited place +-2, the others
    # on signatures that carry the probability mass earlier closures left along the graph: the filter's dynamics, identical in the
    # restated block above).  With a handful of hypotheses in the working memory the best one passes Rtabmap/LoopThr = 0.11 easily --
    # in the reference as well -- so the first pass is recorded, not judged.
    assert len(second_pass) >= 10, closures
    right = [c for c in second_pass if abs(c[1] - c[2]) <= 2]
    assert len(right) >= 0.4 * len(second_pass), second_pass
    r.close()


@pytest.mark.parametrize("kind,rows", [("surf", 300), ("orb", 400), ("surf", 40)])
def test_temporary_dictionary_of_a_frame_pair(oracle, kind, rows):
    """The SECOND caller of the boundary: RegistrationVis.cpp:1482-1503 builds a temporary VWDictionary per frame pair --
    addNewWords(descriptorsFrom, 1) -> update() -> addNewWords(descriptorsTo, 2) -> clear() -- to match the two frames' features.
    Through VWDictionaryHip against the restated VWDictionary, id for id; the create / two calls / destroy cost is reported (it is what
    INTEGRATION.md's size threshold for strategy 5 rests on)."""
    import time
    from rtabmap_amd.vwdictionary import VWDictionaryHip
    rng = np.random.default_rng(rows)
    if kind == "orb":
        a = rng.integers(0, 256, (rows, 32), dtype=np.uint8)
        b = a[rng.permutation(rows)[: rows * 3 // 4]] ^ np.packbits(rng.random((rows * 3 // 4, 256)) < 0.02, axis=1)
        b = np.ascontiguousarray(np.concatenate([b, rng.integers(0, 256, (rows // 4, 32), dtype=np.uint8)]))
    else:
        a = synth.vocab_surf(rows, seed=rows)
        b = a[rng.permutation(rows)[: rows * 3 // 4]] + rng.standard_normal((rows * 3 // 4, 64)).astype(np.float32) * np.float32(0.02)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        b = np.ascontiguousarray(np.concatenate([b, synth.vocab_surf(rows // 4, seed=rows + 1)]).astype(np.float32))
    costs = []
    for rep in range(3):
        o = oracle.OracleVWDictionary(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
        t0 = time.perf_counter()
        h = VWDictionaryHip(nndr=0.8, new_words_compared_together=True)
        from_h = h.add_new_words(a, 1)
        h.update()
        to_h = h.add_new_words(b, 2)
        h.clear()
        h.close()
        costs.append(time.perf_counter() - t0)
        from_o = o.add_new_words(a, 1)
        o.update()
        to_o = o.add_new_words(b, 2)
        assert from_h == from_o and to_h == to_o, "pair %d" % rep
        assert len(set(from_h) & set(to_h)) > rows // 2          # most of frame B's features match words frame A created
        o.close()
    print("temporary dictionary, %s %d + %d descriptors: %.2f ms per pair (first pair %.2f ms)" % (kind, rows, b.shape[0], 1e3 * min(costs), 1e3 * costs[0]))


def test_rtabmap_hip_survives_a_featureless_frame(oracle):
    """A frame without features gives a signature without words: it never gets references, hence no slot on the device.  Once it
    leaves the short-term memory it is in the likelihood like any other place (Rtabmap.cpp:2046-2115); the reference's filter
    carries it with probability 0.  The filter must keep working -- and must not accept a stale hypothesis."""
    from rtabmap_amd.vwdictionary import RtabmapHip
    STM, q = 3, 80
    places = [synth.vocab_surf(q, seed=7000 + p) for p in range(10)]
    r = RtabmapHip(loop_thr=0.11, stm_size=STM)
    ids = []
    for t in range(22):
        desc = np.zeros((0, 64), np.float32) if t == 4 else places[t % 10]
        res = r.process(desc)
        assert res["ok"], "frame %d" % t
        ids.append(res["id"])
        if t > 4 + STM + 1:
            pi, pv = r.vector("posterior")
            assert pi.size > 1 and np.isfinite(pv).all() and abs(float(pv.sum()) - 1.0) < 1e-3, "frame %d: the filter stopped updating" % t
            assert ids[4] in pi.tolist() and pv[pi.tolist().index(ids[4])] == 0.0        # the wordless signature: probability 0
    # the second pass over the places closes loops although the bad signature sits in the working memory
    assert res["highest"][0] > 0
    r.close()


@pytest.mark.parametrize("kind", ["orb", "surf"])
def test_engine_rebuilt_from_the_host_mirror(oracle, kind):
    """Recovery: the device engine is a cache of VWDictionaryHip's maps.  Mid-stream the handle is thrown away and
    re-created from them -- indexed words in row order, every signature's references -- and the stream goes on as if nothing had
    happened: word ids, dictionary state and likelihood equal the oracle's before and after."""
    from rtabmap_amd.vwdictionary import MemoryHip
    frames = _frames(kind, 16, 140)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    h = MemoryHip(nndr=0.8, new_words_compared_together=True)
    W = 7
    for t, desc in enumerate(frames):
        if t in (6, 11):
            assert h.vwd.rebuild_engine(), h.vwd.last_error()
        so, ido = o.update(desc)
        sh, idh = h.update(desc)
        assert so == sh and idh == ido, "frame %d" % t
        _same_state(o, h)
        ids = np.array(o.signature_ids(), np.int32)
        oi, Lo = o.compute_likelihood(np.array(ido, np.int32), ids)
        hi, Lh = h.compute_likelihood(np.array(idh, np.int32), ids)
        assert oi.tolist() == hi.tolist()
        np.testing.assert_allclose(Lh, Lo, rtol=RTOL, atol=ATOL, err_msg="frame %d" % t)
        if so > W:
            o.forget(so - W)
            h.forget(so - W)
    h.close()


def test_statistics_carry_the_reference_names_and_roctx_ranges_do_not_change_results(oracle):
    """SURVEY.md section 5, tracing row: MemoryHip reports the stages the reference times under the reference's statistic names
    (Statistics.h:178,189-190,200-201,209-212; emitted at Memory.cpp:5931,6062, Rtabmap.cpp:4357,4367), and with lcd_set_option("roctx", 1)
    the engine and the mirror bracket their stages with roctx ranges -- word ids and likelihood stay the oracle's with the ranges on."""
    from rtabmap_amd.vwdictionary import MemoryHip
    frames = _frames("surf", 8, 120)
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    h = MemoryHip(nndr=0.8, new_words_compared_together=True)
    st0 = h.statistics()
    assert set(st0) == set(MemoryHip.STAT_NAMES) and all(v == 0.0 for v in st0.values())      # every key exists from the start (Statistics::_defaultData)
    for name in MemoryHip.STAT_NAMES:                                                       # "<Prefix>/<Name>/<unit>" as RTABMAP_STATS builds them
        assert name.count("/") == 2 and name.split("/")[0] in ("Timing", "TimingMem", "Keypoint")
    for t, desc in enumerate(frames):
        if t == 3:
            rc = h.set_engine_option("roctx", 1)        # the engine exists from the first update() on
            assert rc in (0, 5), rc                     # LCD_OK, or LCD_ERR_UNSUPPORTED where libroctx64.so is not installed
        so, ido = o.update(desc)
        sh, idh = h.update(desc)
        assert so == sh and idh == ido, "frame %d" % t
        ids = np.array(o.signature_ids(), np.int32)
        oi, Lo = o.compute_likelihood(np.array(ido, np.int32), ids)
        hi, Lh = h.compute_likelihood_of(sh, ids)
        assert oi.tolist() == hi.tolist()
        np.testing.assert_allclose(Lh, Lo, rtol=RTOL, atol=ATOL, err_msg="frame %d" % t)
        if so > 5:
            o.forget(so - 5)
            h.forget(so - 5)
    st = h.statistics(refresh=True)
    assert st["TimingMem/Add_new_words/ms"] > 0.0 and st["TimingMem/Pre_update/ms"] > 0.0 and st["Timing/Likelihood_computation/ms"] > 0.0
    assert st["Timing/Forgetting/ms"] > 0.0 and st["TimingMem/Joining_dictionary_update/ms"] == 0.0
    assert st["Keypoint/Current_frame/words"] == 120.0
    assert st["Keypoint/Dictionary_size/words"] == float(o.vwd.visual_words)
    assert st["Keypoint/Indexed_words/words"] == float(h.vwd.indexed_words)
    assert st["Keypoint/Index_memory_usage/KB"] > 0.0
    h.close()
