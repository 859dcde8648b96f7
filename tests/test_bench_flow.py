"""CPU: the control flow of bench.py's headline run with the device layer replaced by stand-ins (no kernel runs, no number means anything):
rank 0 prints exactly ONE JSON line with the driver's keys at N = 1, and at N = 2 a secondary measurement that never completes -- a
collective that hangs -- does not cost the line: the watchdog prints it and ends the rank (exit code 0).  Runs bench.main() in a
subprocess."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STANDINS = textwrap.dedent('''
    import importlib.util, os, sys, time
    import numpy as np
    spec = importlib.util.spec_from_file_location("bench", os.path.join(%(root)r, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    import torch
    import torch.distributed as dist

    class FakeStream:
        cuda_stream = 0
        def synchronize(self): pass

    class FakeTensor:
        def cuda(self): return self
        def data_ptr(self): return 0

    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 1
    torch.cuda.set_device = lambda d: None
    torch.cuda.Stream = lambda *a, **k: FakeStream()
    torch.from_numpy = lambda a: FakeTensor()
    dist.init_process_group = lambda *a, **k: None
    dist.barrier = lambda *a, **k: None
    dist.get_backend = lambda *a, **k: "gloo"
    dist.all_reduce = lambda t, op=None: None
    dist.destroy_process_group = lambda: None
    import rtabmap_amd
    from rtabmap_amd import synth

    class FakeEngine:
        def __init__(self, *a, **k): pass
        def set_option(self, *a): pass
        def profile_read(self): return (0.02, 3, "frame_a_kernel (stand-in)")
        def profile_begin(self, n): pass
        def stats(self): return {"frame_host_ns": 29000 * 25, "frame_calls": 25}
        def close(self): pass

    steps, warmup, n_sig = 20, 5, 100000
    src = np.random.default_rng(7).integers(0, n_sig, min(64, max(8, steps)))          # make_frames(0) of bench.py
    top = int(src[(warmup + steps - 1) %% len(src)])

    class FakeLike:
        def __getitem__(self, k): return self
        def cpu(self): return self
        def numpy(self):
            a = np.zeros(n_sig + 64, np.float32)
            a[top] = 1.0
            return a

    class FakeStepper:
        def __init__(self, eng, torch_, d_frames, n_sig_, cap, log_frames=0, append=True):
            self.d_like, self.d_words_log, self.n_calls = FakeLike(), None, 0
        def __call__(self, i): pass

    rtabmap_amd.Engine = FakeEngine
    B.load_engine = lambda eng, vocab, words, owned=None: 0.04
    B.Stepper = FakeStepper
    B.timed_loop = lambda torch_, dist_, world, stream, step, steps_, warmup_, profile_eng=None, eng=None, per_step_events=True, prof_n=0: {
        "wall": 4.0e-5 * steps_, "dev_ms": 0.039 * steps_, "host_enqueue": 2.0e-5 * steps_, "per_step_ms": np.full(steps_ if per_step_events else 0, 0.033)}
    B.rooflines = lambda eng, nw, ns, shard, knn=None, knn_timed=None: ({"ms": 0.0209, "kernel": "frame_a_kernel (stand-in)", "frac": 0.06, "samples": 11},
                                                        {"ms": 0.0155, "kernel": "frame_b_kernel (stand-in)", "frac": 0.11})
    B.make_state = lambda n: (np.zeros((49000, 64), np.float32), np.ones((n, 500), np.int32))
    synth.frame_from_signature = lambda vocab, words, seed=0, **k: np.zeros((500, 64), np.float32)
''')


def _run(tail, env=None):
    code = STANDINS % {"root": ROOT} + textwrap.dedent(tail)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **(env or {})))


def test_one_json_line_with_the_drivers_keys():
    r = _run('''
        sys.argv = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline", "--no-pmc"]
        B.main()
    ''')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert abs(d["value"] - 20 * 100000 / (4.0e-5 * 20)) < 1.0 and abs(d["ms_per_step"] - 0.04) < 1e-9
    assert d["config"]["last_frame_top_candidate_ok"] is True and d["roofline"]["kernel"].startswith("frame_a_kernel")


def test_a_hanging_secondary_measurement_does_not_cost_the_line():
    r = _run('''
        import rtabmap_amd.sharded as S

        class Hang:
            def __init__(self, *a, **k): self.eng, self.lo, self.hi = None, 0, 1
            def load_vocabulary(self, *a, **k): time.sleep(3600)          # the collective that never completes
        S.ShardedLoopClosure = Hang
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-pmc"]
        B.main()
        print("main() returned although the secondary measurement hangs")
    ''', env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "LCD_BENCH_SECONDARY_TIMEOUT": "2", "LCD_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] == 2 * 20 * 100000 / (4.0e-5 * 20)
    assert "timed out" in d["config"]["secondary_parallelism_error"]


FAKE_SHARD = '''
    import rtabmap_amd.sharded as S

    class FakeShard:
        def __init__(self, *a, **k): self.eng, self.lo, self.hi = FakeEngine(), 0, 24500
        def load_vocabulary(self, rows, ids): pass
        def enable_device_append(self, *a, **k): pass
        def add_signatures_bulk(self, *a, **k): pass
        def frame(self, *a, **k): return None, None
        def retire(self, sig): pass
        def flush(self): return FakeLike()
        def close(self): pass
    S.ShardedLoopClosure = FakeShard
'''


def test_both_parallelisms_at_two_ranks():
    """N = 2 (stand-ins): replicas as the primary measurement with the sharded frame as the secondary key, and the other way round"""
    env = {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "LCD_BENCH_BACKEND": "gloo"}
    r = _run(textwrap.dedent(FAKE_SHARD) + textwrap.dedent('''
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-pmc"]
        B.main()
    '''), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "secondary_parallelism_error" not in d["config"], d["config"].get("secondary_parallelism_error")
    assert d["config"]["shard_ms_per_step"] > 0 and d["config"]["shard_value"] > 0 and "replicas" in d["config"]["parallelism"]
    r = _run(textwrap.dedent(FAKE_SHARD) + textwrap.dedent('''
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-pmc", "--parallelism", "shard"]
        B.main()
    '''), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "secondary_parallelism_error" not in d["config"], d["config"].get("secondary_parallelism_error")
    assert d["config"]["replicas_value"] > 0 and "sharded" in d["config"]["parallelism"]
    assert d["value"] == 20 * 100000 / (4.0e-5 * 20)                       # ONE frame stream over both GPUs
