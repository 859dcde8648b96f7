"""GPU: Memory::loadDataFromDb through the host mirror (MemoryHip::loadDataFromDb, rtabmap_amd/host/DbLoaderHip.h): a database made from
the reference's table definitions (tests/test_db_loader.py) is loaded -- dictionary by one update(), references by one bulk registration --
and the memory then behaves like an oracle memory that got the same words and signatures one call at a time: dictionary state,
likelihood, and the word ids of frames processed afterwards.

Written in round 4 without a GPU; first run (and green) in round 5's first GPU call (profiles/r05_first_call.txt): part of `pytest -m gpu`."""
import numpy as np
import pytest

from rtabmap_amd import synth
from test_db_loader import _make_db
from test_gpu_frame_stream import RTOL, ATOL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["surf", "orb"])
def test_load_data_from_db_matches_a_memory_filled_call_by_call(oracle, tmp_path, kind):
    from rtabmap_amd.vwdictionary import MemoryHip
    path = str(tmp_path / "map.db")
    ids, rows, sigs = _make_db(path, kind, n_words=400, n_nodes=60, seed=11, reference_last_word=True)   # (the oracle has no setLastWordId)
    h = MemoryHip(nndr=0.8, new_words_compared_together=True)
    assert h.load_data_from_db(path, last_state_only=False) == len(sigs)
    # the oracle: an incremental dictionary keeps only the words the loaded signatures reference (Memory.cpp:394-424), in ascending id
    referenced = sorted({int(w) for ws in sigs.values() for w in ws if w > 0})
    row_of = {int(i): r for i, r in zip(ids, rows)}
    o = oracle.OracleMemory(strategy=oracle.kNNBruteForce, nndr=0.8, new_words_compared_together=True)
    for w in referenced:
        o.vwd.add_word(w, row_of[w])
    o.vwd.update()
    for node in sorted(sigs):
        o.add_signature_with_id(node, np.sort(sigs[node]))
    assert h.vwd.visual_words == o.vwd.visual_words == len(referenced)
    assert h.vwd.indexed_words == o.vwd.indexed_words and h.vwd.index_ids() == o.vwd.index_ids()
    assert h.vwd.total_active_references == o.vwd.total_active_references
    assert h.num_signatures() == o.num_signatures() == len(sigs)
    for node in sorted(sigs):
        assert h.get_ni(node) == o.get_ni(node) == len(sigs[node])
    all_ids = np.array(sorted(sigs), np.int32)
    for node in list(sorted(sigs))[::13]:
        if not len(sigs[node]):
            continue
        oi, Lo = o.compute_likelihood(sigs[node], all_ids)
        hi, Lh = h.compute_likelihood(sigs[node], all_ids)
        assert oi.tolist() == hi.tolist()
        np.testing.assert_allclose(Lh, Lo, rtol=RTOL, atol=ATOL)
        assert int(np.argmax(Lh)) == int(np.argmax(Lo))
    # the stream goes on from the loaded state: new frames quantise against the loaded dictionary, new word ids continue after the last one
    rng = np.random.default_rng(5)
    for t in range(4):
        pick = rng.choice(len(referenced), 60)
        base = np.stack([row_of[referenced[k]] for k in pick])
        if kind == "surf":
            desc = (base + 0.02 * rng.standard_normal(base.shape)).astype(np.float32)
            desc[-10:] = synth.vocab_surf(10, seed=900 + t)
        else:
            desc = base.copy()
            desc[:, 0] ^= np.uint8(1)
            desc[-10:] = synth.vocab_orb(10, seed=900 + t)
        so, ido = o.update(desc)
        sh, idh = h.update(desc)
        assert so == sh and ido == idh, "frame %d after the load" % t
    h.close()
