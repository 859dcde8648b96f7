cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_append_dev.py -x -q -m gpu > $O/append.txt 2>&1; tail -3 $O/append.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
( time timeout 900 python bench.py --config orb_stream ) > $O/bench_orb.json 2> $O/bench_orb.err; tail -4 $O/bench_orb.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r3p/bench_default.json").read().strip().splitlines()[-1])
    print("default", d["ms_per_step"], d["value"], d["roofline"]["ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("roofline_score",{}).get("ms"))
    print(json.dumps(d["parity"])[:900]); print(json.dumps(d["cpu_baseline"])[:1500])
except Exception as e: print("default ERR", e)
try:
    d=json.loads(open("gpurun_out/r3p/bench_orb.json").read().strip().splitlines()[-1])
    print("orb", json.dumps(d)[:2500])
except Exception as e: print("orb ERR", e)
PY
