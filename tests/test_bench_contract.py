"""The bench line the driver reads: the committed run of the driver's command (profiles/r04_bench.json) carries every field of the
contract, quotes BASELINE.json's metric on BASELINE.json's headline configuration, and its numbers are consistent with each other."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_the_drivers_line_has_the_contracts_fields():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    d = _line("r04_bench.json")
    assert d["metric"] == base["metric"] and d["unit"] == "candidates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32"                                        # the arithmetic of the results: fp32 distances, fp32 likelihood
    c = d["config"]
    assert "49k words" in c["workload"] and "100000 signatures" in c["workload"] and "500 desc/frame" in c["workload"] and "model" not in c
    # value = signatures scored per second: frames/s x 100 000, from the step time
    assert abs(d["value"] - 100000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] in ("TFLOP/s", "GB/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] is not None and r["ms"] > 0
    b = d["cpu_baseline"]
    assert b["kind"] in ("reference", "port") and b["cores"] >= 1 and b["value"] > 0 and b["unit"] == "candidates/s" and b["sample"]
    assert d["value"] / b["value"] >= 50.0                            # the north star's bar, against the reference's own configuration
    best_cpu = max(v["value"] for v in b["variants"].values())
    assert d["value"] / best_cpu >= 50.0                              # ... and against the most generous CPU variant
    p = d["parity"]
    assert p["word_ids_equal"] and p["argmax_equal"] and p["likelihood_max_rel"] <= 1e-4
    assert p["timed_engine"]["likelihood_max_rel"] <= 1e-4 and p["timed_engine"]["argmax_equal"]


def test_the_secondary_lines_carry_roofline_and_parity():
    for name in ("r04_bench_orb.json", "r04_bench_replay_1m.json", "r04_bench_replay_30k.json"):
        d = _line(name)
        assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["parity"]["word_ids_equal"], name
        rel = d["parity"].get("likelihood_max_rel", d["parity"].get("tail", {}).get("likelihood_max_rel"))   # (the ORB line: of its last frames)
        assert rel is not None and rel <= 1e-4, name
    assert _line("r04_bench_replay_30k.json")["parity"]["adjust_likelihood_and_best_candidate_equal"]
    assert _line("r04_bench_replay_1m.json")["recall"]["recall"] == 1.0
    for name in ("r04_bench_1m.json", "r04_bench_125k_words.json", "r04_bench_200steps.json"):
        assert _line(name)["roofline"]["frac"] > 0, name
